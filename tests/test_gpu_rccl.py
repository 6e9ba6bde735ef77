"""The exchange step over RCCL itself (backend "nccl"), as far as one GPU allows: a world-size-1 process group moves a
packed ShardMessage through all_gather_into_tensor (synchronous and asynchronous form, header round trip included), and
the collectives bench.py brackets its timed region with (barrier, fp64 MAX all-reduce) run on device tensors.  World sizes
> 1 are covered on CPU with gloo (test_cpu_host.py), by the merge tests in test_gpu_search.py and by the rehearsal below:
bench.py's N > 1 branch in child processes that share this GPU and exchange over gloo -- started both ways the driver may
start it (`python bench.py --gpus N` and `python -m torch.distributed.run ... bench.py --gpus N`)."""
import json
import os
import signal
import socket
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bf16(bits):
    return torch.from_numpy(bits.view(np.int16)).view(torch.bfloat16).cuda()


def _rand_bits(n, d, seed):
    g = torch.Generator().manual_seed(seed)
    return orc.pack_bf16((torch.randn(n, d, generator=g) * d ** -0.5).numpy())


def test_shard_message_through_rccl_world1():
    import torch.distributed as dist
    from ccrec_amd import ops
    from ccrec_amd.dist import ShardMessage, sharded_search, submit_sharded_search
    assert not dist.is_initialized()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        n, nq, d, k = 20000, 37, 768, 100              # nq * k * 4 % 16 != 0: the row block starts on padding
        Db, Qb = _rand_bits(n, d, 41), _rand_bits(nq, d, 42)
        index = ops.CorpusIndex(_bf16(Db), global_row_offset=777)
        Q = _bf16(Qb)
        s1, i1 = index.search(Q, k)
        # synchronous form: the search writes the message, ONE RCCL all_gather_into_tensor of the packed bytes, merge
        m = ShardMessage(nq, k, dev, 1)
        index.search_shard(Q, k, m.send)
        m.gather()
        h = ShardMessage.parse_headers(m.all_headers.cpu())[0]
        assert h["n_flagged"] == 0 and h["k_valid"] == k and h["row_offset"] == 777 and h["n_rows"] == n and h["n_covered"] == 0
        ms, mi = m.merge()
        assert torch.equal(mi, i1) and torch.equal(ms.view(torch.int32), s1.view(torch.int32))
        gs, gi = m.decoded()
        assert torch.equal(gi[0], i1) and torch.equal(gs[0], s1)
        # the asynchronous form bench.py pipelines across steps: no host synchronisation before the collective, the
        # headers come back through pinned memory from a side stream, finish() waits for the search's own event
        m2 = ShardMessage(nq, k, dev, 1)
        ex = submit_sharded_search(index, Q, k, message=m2)
        junk = torch.randn(4096, 4096, device=dev) @ torch.randn(4096, 4096, device=dev)   # later work on the compute stream
        ms2, mi2 = ex.result()
        assert torch.equal(mi2, i1) and torch.equal(ms2.view(torch.int32), s1.view(torch.int32))
        assert not ex.repeated and ex.headers[0]["n_flagged"] == 0 and ex.headers[0]["n_covered"] == 0
        assert index.last_stats()["path"] == 1 and index.last_stats()["ms_main"] > 0      # finish() filled the statistics
        del junk
        # world 1: sharded_search is the plain search
        s3, i3 = sharded_search(index, Q, k)
        assert torch.equal(i3, i1) and torch.equal(s3, s1)
        ref_i, _ = orc.canonical_search(Qb, Db, k, None)
        assert np.array_equal(i1.cpu().numpy() - 777, ref_i)
        # bench.py's timing collectives on device tensors
        dist.barrier()
        t = torch.tensor([1.25], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(t)]
        dist.all_gather(every, t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()
        assert t.item() == 1.25 and every[0].item() == 1.25
    finally:
        dist.destroy_process_group()


from helpers import run_child_with_evidence as _run_child   # noqa: E402  (a child that overruns FAILS the test, with evidence)


@pytest.mark.parametrize("world,rows,launcher", [(2, 200000, "plain"), (2, 200000, "torchrun"), (3, 200003, "plain"), (4, 200005, "torchrun")])
def test_bench_ranks_rehearsal_in_fresh_processes(tmp_path, world, rows, launcher):
    """The N > 1 branch of bench.py as the driver launches it -- `python bench.py --gpus N` (bench.py starts its own ranks as a
    child torch.distributed.run) and the explicit `python -m torch.distributed.run ... bench.py --gpus N` -- rehearsed on this
    one GPU: the ranks share cuda:0 and exchange over gloo.  The JSON line must parse and the merged ids must equal the 1-rank
    run's (the corpus is the same global stream, row-sharded; 200003 rows over 3 ranks: uneven shards, as NQ over 8)."""
    bench = os.path.join(ROOT, "bench.py")
    common = ["--steps", "2", "--warmup", "1", "--rows", str(rows), "--queries", "300", "--cpu-queries", "0", "--no-secondary"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CCR_BENCH_WATCHDOG="150")   # a hung rank dumps its stacks and exits
    torch.cuda.synchronize()
    torch.cuda.empty_cache()    # the children share this GPU: hand back what the earlier (full-size) tests left cached
    one_out, _ = _run_child([sys.executable, bench, "--gpus", "1", "--dump-ids", str(tmp_path / "one.pt")] + common, env, tmp_path, "one")
    rehearsal = ["--gpus", str(world), "--dist-backend", "gloo", "--same-device", "--dump-ids", str(tmp_path / "two.pt")] + common
    if launcher == "plain":
        cmd = [sys.executable, bench] + rehearsal
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), bench] + rehearsal
    two_out, two_err = _run_child(cmd, env, tmp_path, "ranks")
    assert len([ln for ln in one_out.splitlines() if ln.startswith("{")]) == 1
    lines = [ln for ln in two_out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, two_out[-2000:]                     # rank 0 prints ONE JSON line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == world and rec["steps"] == 2 and rec["scaling"] == "strong" and rec["value"] > 0
    assert rec["config"]["parallelism"] == f"row-shard x{world}" and rec["cpu_baseline"] is None
    assert rec["roofline"]["bound"] == "mfma" and rec["search_stats"]["n_fallback"] == 0
    assert rec["roofline"]["launches_per_step"] >= 1 and rec["roofline"]["flops_per_step"] > 0
    ex = rec["exchange"]
    assert ex["n_ranks_seen"] == world and ex["backend"] == "gloo" and ex["repeated_exchanges"] == 0
    from ccrec_amd.dist import short_list_length, short_lists_pay
    from ccrec_amd.ops import shard_message_bytes
    entries = short_list_length(100, world) if short_lists_pay(100, world) else 100      # 3 ranks: 70 of 100 entries; 2 ranks: full lists
    assert ex["entries_per_query_per_rank"] == entries and ex["message_bytes_per_rank"] == shard_message_bytes(300, entries)
    assert ex["rank_ms_per_step_min"] <= ex["rank_ms_per_step_max"] and ex["queries_repeated_with_full_lists"] == 0
    assert ex["search_ms_per_step"] > 0 and ex["host_wait_for_exchange_ms_per_step"] >= 0
    a, b = torch.load(tmp_path / "one.pt"), torch.load(tmp_path / "two.pt")
    assert a.shape == (300, 100) and torch.equal(a, b)


@pytest.mark.parametrize("world", [2, 3])
def test_bench_ranks_rehearsal_records_the_multi_gpu_side_runs(tmp_path, world):
    """One N > 1 invocation also measures the shapes the multi-GPU target is quoted on -- ranking()'s k = 1001 on the same corpus (the
    short-list exchange), configs[2] and configs[3] (here 1/40 of their rows and queries; configs[3] runs from 4 ranks up outside a
    rehearsal) -- each with its own `exchange` record: entries per query
    and rank, message bytes, repeats, per-rank min / max step, and the host-observed wait for the collective split from the search time."""
    bench = os.path.join(ROOT, "bench.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CCR_BENCH_WATCHDOG="200")
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    cmd = [sys.executable, bench, "--gpus", str(world), "--dist-backend", "gloo", "--same-device", "--steps", "2", "--warmup", "1", "--rows", "150000",
           "--queries", "200", "--cpu-queries", "0", "--rehearse-secondary", "40"]
    out, err = _run_child(cmd, env, tmp_path, "ranks")
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    rec = json.loads(lines[0])
    from ccrec_amd.dist import short_list_length
    from ccrec_amd.ops import shard_message_bytes
    sec = rec["secondary"]
    assert set(sec) == {"k1001", "msmarco", "config4"}
    k1001, ms, c4 = sec["k1001"], sec["msmarco"], sec["config4"]
    kl = short_list_length(1001, world)
    assert k1001["exchange"]["entries_per_query_per_rank"] == kl < 1001 and k1001["exchange"]["message_bytes_per_rank"] == shard_message_bytes(200, kl)
    assert k1001["exchange"]["full_list_message_bytes_per_rank"] == shard_message_bytes(200, 1001)
    assert "REHEARSAL" in ms["workload"] and ms["exchange"]["message_bytes_per_rank"] == shard_message_bytes(6980 // 40, ms["exchange"]["entries_per_query_per_rank"])
    from ccrec_amd.dist import short_lists_pay
    k4 = short_list_length(1000, world) if short_lists_pay(1000, world) else 1000      # configs[3]: 50 M x 1024 / 40, 250 queries, top-1000
    assert "REHEARSAL" in c4["workload"] and "50,000,000 x 1024" in c4["workload"]
    assert c4["exchange"]["entries_per_query_per_rank"] == k4 and c4["exchange"]["message_bytes_per_rank"] == shard_message_bytes(10000 // 40, k4)
    for side in (k1001, ms, c4):
        ex = side["exchange"]
        assert side["value"] > 0 and side["scaling"] == "strong" and side["n_fallback"] == 0 and side["roofline"]["bound"] == "mfma"
        assert ex["n_ranks_seen"] == world and ex["repeated_exchanges"] == 0 and ex["queries_repeated_with_full_lists"] == 0
        assert 0 < ex["rank_ms_per_step_min"] <= ex["rank_ms_per_step_max"] and ex["search_ms_per_step"] >= 0   # (0: a shard too small for the fused path has no phase events)
        assert ex["host_wait_for_exchange_ms_per_step"] >= 0 and ex["host_wait_for_exchange_ms_max"] >= ex["host_wait_for_exchange_ms_per_step"]


def test_bench_refuses_a_world_size_that_contradicts_gpus(tmp_path):
    """WORLD_SIZE set by a launcher but different from --gpus: a clear message and exit code 2 before any GPU call."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], capture_output=True, text=True,
                         timeout=120, env=env)
    assert run.returncode == 2 and "WORLD_SIZE=2" in run.stderr and "{" not in run.stdout


def _flagged_rank_worker(rank, world, port, out_dir):
    """Real kernels, two processes on this GPU over gloo: rank 1's shard holds 9 000 identical rows and the queries equal that row,
    so ITS asynchronous search flags every query (mass ties beyond the re-score cap) while rank 0 flags none; the headers tell both
    ranks, both repeat the all-gather after finish(), and the merged lists equal the single-index search."""
    import torch.distributed as dist
    sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]
    from ccrec_amd import ops
    from ccrec_amd.dist import shard_bounds, submit_sharded_search
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    n, d, k, nq = 140_000, 128, 64, 40
    Db = _rand_bits(n, d, 71)
    Db[n - 9000:] = Db[n - 1]                       # the tail (rank 1's shard) is one row 9 000 times
    Qb = np.tile(Db[n - 1], (nq, 1))
    lo, hi = shard_bounds(n, world, rank)
    index = ops.CorpusIndex(_bf16(Db[lo:hi]), global_row_offset=lo)
    ex = submit_sharded_search(index, _bf16(Qb), k)
    s, i = ex.result()
    flagged = [h["n_flagged"] for h in ex.headers]
    whole = ops.CorpusIndex(_bf16(Db))
    s1, i1 = whole.search(_bf16(Qb), k)
    ok = torch.equal(i, i1) and torch.equal(s.view(torch.int32), s1.view(torch.int32))
    ok = ok and ex.repeated and flagged[0] == 0 and flagged[1] == nq and index.last_stats()["n_fallback"] == (nq if rank == 1 else 0)
    open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write("ok" if ok else f"MISMATCH {flagged} {ex.repeated}")
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_flags_every_query_and_all_ranks_repeat_the_exchange(tmp_path):
    import torch.multiprocessing as mp
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    mp.spawn(_flagged_rank_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"


def _short_rank_worker(rank, world, port, out_dir):
    """Real kernels, `world` processes on this GPU over gloo, the SHORT-list exchange end to end: k = 300 -> every rank searches and sends
    its canonical top-210 (world 2) / top-157 (world 3); queries 0 and 4 have their whole top-k inside the last shard, so its list is
    consumed to its end for exactly those two, every rank derives the same two flags from the gathered bytes and all repeat them with
    full lists; the merged lists equal the single-index search bit for bit."""
    import torch.distributed as dist
    sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]
    from oracle import oracle as orc
    from ccrec_amd import ops
    from ccrec_amd.dist import shard_bounds, short_list_length, submit_sharded_search, sharded_search
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    n, d, k, nq = 120_000, 128, 300, 24
    rs = np.random.RandomState(17)
    Df = rs.randn(n, d).astype(np.float32) / 11
    Qf = np.linalg.qr(rs.randn(d, nq))[0].T.astype(np.float32)
    lo_last = shard_bounds(n, world, world - 1)[0]
    for j, qi in enumerate((0, 4)):
        rows = slice(lo_last + 100 + j * k, lo_last + 100 + (j + 1) * k)
        Df[rows] = 2.0 * Qf[qi] + 0.3 * Df[rows]
    Db, Qb = orc.pack_bf16(Df), orc.pack_bf16(Qf)
    Db[n // 2 + 1] = Db[1]
    lo, hi = shard_bounds(n, world, rank)
    index = ops.CorpusIndex(_bf16(Db[lo:hi]), global_row_offset=lo)
    Q = _bf16(Qb)
    ex = submit_sharded_search(index, Q, k, short_lists=True)
    s, i = ex.result()
    whole = ops.CorpusIndex(_bf16(Db))
    s1, i1 = whole.search(Q, k)
    ok = torch.equal(i, i1) and torch.equal(s.view(torch.int32), s1.view(torch.int32))
    ok = ok and ex.message.k == short_list_length(k, world) < k and ex.fallback_queries == 2 and not ex.repeated
    s2, i2 = sharded_search(index, Q, k)                       # the automatic choice (k = 300: short lists pay for 2 and 3 ranks)
    ok = ok and torch.equal(i2, i1) and torch.equal(s2.view(torch.int32), s1.view(torch.int32))
    s3, i3 = sharded_search(index, Q, k, short_lists=False)    # and the full-list exchange
    ok = ok and torch.equal(i3, i1) and torch.equal(s3.view(torch.int32), s1.view(torch.int32))
    open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write("ok" if ok else f"MISMATCH fallback={ex.fallback_queries} k_list={ex.message.k}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_short_list_exchange_with_real_kernels_and_a_skewed_shard(tmp_path, world):
    import torch.multiprocessing as mp
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    mp.spawn(_short_rank_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"


def test_bench_ranks_rehearsal_on_a_corpus_in_topical_order_suspends_the_short_lists(tmp_path):
    """ranking()'s k = 1001 over 3 ranks on a corpus whose clusters are CONTIGUOUS row ranges (--data sorted: passages of one topic lie
    next to each other, as in a corpus in source order): a query's top-1001 sits in one or two shards, their short lists (k / R + 6 sigma
    entries) are consumed to their ends, and the flagged queries are repeated with full lists -- still the single-GPU ids.  Once an
    exchange has repeated more than 5 % of its queries every rank suspends the shortcut at the same step and the later steps send full
    lists (no second search, no second collective): the record says at which step."""
    bench = os.path.join(ROOT, "bench.py")
    common = ["--steps", "3", "--warmup", "1", "--rows", "180000", "--queries", "200", "--k", "1001", "--data", "sorted", "--cpu-queries", "0",
              "--no-secondary"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CCR_BENCH_WATCHDOG="150")
    env.pop("CCREC_SHORT_LISTS", None)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    _run_child([sys.executable, bench, "--gpus", "1", "--dump-ids", str(tmp_path / "one.pt")] + common, env, tmp_path, "one")
    out, _ = _run_child([sys.executable, bench, "--gpus", "3", "--dist-backend", "gloo", "--same-device", "--dump-ids", str(tmp_path / "three.pt")] + common,
                        env, tmp_path, "ranks")
    rec = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][0])
    a, b = torch.load(tmp_path / "one.pt"), torch.load(tmp_path / "three.pt")
    assert a.shape == (200, 1001) and torch.equal(a, b)
    ex = rec["exchange"]
    from ccrec_amd.dist import short_list_length
    # the warm-up step ran with short lists and repeated most of its queries; the timed steps ran with full lists
    assert ex["short_lists_suspended_at_step"] == 1, ex
    assert ex["entries_per_query_per_rank"] == 1001 > short_list_length(1001, 3) and ex["queries_repeated_with_full_lists"] == 0
