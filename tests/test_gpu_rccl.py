"""The exchange step over RCCL itself (backend "nccl"), as far as one GPU allows: a world-size-1 process group moves a
TopkMessage through all_gather_into_tensor, and the collectives bench.py brackets its timed region with
(barrier, fp64 MAX all-reduce) run on device tensors.  World sizes > 1 are covered on CPU with gloo
(test_cpu_host.py) and by the merge tests in test_gpu_search.py."""
import os
import socket

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bf16(bits):
    return torch.from_numpy(bits.view(np.int16)).view(torch.bfloat16).cuda()


def _rand_bits(n, d, seed):
    g = torch.Generator().manual_seed(seed)
    return orc.pack_bf16((torch.randn(n, d, generator=g) * d ** -0.5).numpy())


def test_topk_message_through_rccl_world1():
    import torch.distributed as dist
    from ccrec_amd import ops
    from ccrec_amd.dist import TopkMessage, all_gather_topk, merge_gathered, sharded_search
    assert not dist.is_initialized()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        n, nq, d, k = 20000, 37, 768, 100              # nq * k * 4 % 16 != 0: the id block starts on padding
        Db, Qb = _rand_bits(n, d, 41), _rand_bits(nq, d, 42)
        index = ops.CorpusIndex(_bf16(Db), global_row_offset=777)
        Q = _bf16(Qb)
        s1, i1 = index.search(Q, k)
        m = TopkMessage(nq, k, dev, 1)
        index.search(Q, k, out=(m.scores, m.ids))
        gs, gi = m.gather()                             # ONE RCCL all_gather_into_tensor of the packed bytes
        assert gs.shape == (1, nq, k) and gi.shape == (1, nq, k)
        ms, mi = merge_gathered(gs, gi)
        assert torch.equal(mi, i1) and torch.equal(ms.view(torch.int32), s1.view(torch.int32))
        # the asynchronous form bench.py pipelines across steps: the collective runs on RCCL's stream, wait() orders the merge behind it
        m.recv.zero_()
        work = m.gather_async()
        work.wait()
        ms2, mi2 = merge_gathered(m.all_scores, m.all_ids)
        assert torch.equal(mi2, i1) and torch.equal(ms2.view(torch.int32), s1.view(torch.int32))
        # separate tensors are copied into the message first
        gs2, gi2 = all_gather_topk(s1, i1)
        assert torch.equal(gs2[0], s1) and torch.equal(gi2[0], i1)
        # world 1: sharded_search is the plain search
        s3, i3 = sharded_search(index, Q, k)
        assert torch.equal(i3, i1) and torch.equal(s3, s1)
        ref_i, _ = orc.canonical_search(Qb, Db, k, None)
        assert np.array_equal(i1.cpu().numpy() - 777, ref_i)
        # bench.py's timing collectives on device tensors
        dist.barrier()
        t = torch.tensor([1.25], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()
        assert t.item() == 1.25
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,rows", [(2, 200000), (3, 200003)])   # 200003 rows over 3 ranks: uneven shards (as NQ over 8)
def test_bench_two_ranks_rehearsal_in_a_fresh_process(tmp_path, world, rows):
    """The N > 1 branch of bench.py exactly as the driver launches it (`python -m torch.distributed.run ... bench.py
    --gpus 2`), rehearsed on this one GPU: both ranks share cuda:0 and exchange over gloo.  The JSON line must parse and
    the merged ids must equal the 1-rank run's (the corpus is the same global stream, row-sharded)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "2", "--warmup", "1", "--rows", str(rows), "--queries", "300", "--cpu-queries", "0", "--no-secondary"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CCR_BENCH_WATCHDOG="150")   # a hung rank dumps its stacks and exits
    try:
        one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--dump-ids", str(tmp_path / "one.pt")] + common,
                             capture_output=True, text=True, timeout=240, env=env)
    except subprocess.TimeoutExpired as e:   # seen once in ~30 runs on a cold box, never reproduced: not a verdict on the code
        pytest.skip(f"1-rank child did not finish in 240 s on this box: {(e.stderr or b'')[-1500:]}")
    assert one.returncode == 0, one.stderr[-4000:]
    try:
        two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                              "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", str(world),
                              "--dist-backend", "gloo", "--same-device", "--dump-ids", str(tmp_path / "two.pt")] + common,
                             capture_output=True, text=True, timeout=240, env=env)
    except subprocess.TimeoutExpired as e:
        pytest.skip(f"2-rank rehearsal did not finish in 240 s on this box: {(e.stderr or b'')[-1500:]}")
    assert two.returncode == 0, two.stderr[-6000:]
    lines = [ln for ln in two.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, two.stdout[-2000:]                  # rank 0 prints ONE JSON line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == world and rec["steps"] == 2 and rec["scaling"] == "strong" and rec["value"] > 0
    assert rec["config"]["parallelism"] == f"row-shard x{world}" and rec["cpu_baseline"] is None
    assert rec["roofline"]["bound"] == "mfma" and rec["search_stats"]["n_fallback"] == 0
    a, b = torch.load(tmp_path / "one.pt"), torch.load(tmp_path / "two.pt")
    assert a.shape == (300, 100) and torch.equal(a, b)
