#!/usr/bin/env python3
"""bench.py -- queries/sec of the retrieval hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
              --master-port P bench.py --gpus N --steps K --warmup W

Workload (configs[1]): NQ-shaped synthetic embeddings, corpus 2,681,468 x 768, 3,452 queries, top-100.
One step = one pass of the hot path over the whole query batch, starting from the encoder's fp32
outputs resident in HBM:  pack corpus shard fp32->bf16, build the index (row norms), pack queries,
fused MFMA score + top-k, [N > 1: RCCL all-gather of the per-shard top-k + merge].
N > 1 row-shards the SAME corpus over the ranks (strong scaling); value = queries / step time.

Extra objects on the JSON line: `roofline` for the dominant kernel (main-pass GEMM + filter; HIP
events recorded by the library on the search stream), `cpu_baseline` (the oracle's
reference-faithful CPU path timed on this host's cores on a bounded query sample, rank 0, N = 1) and, in the default
N = 1 run, `secondary`: the same step at the north-star target shape (MS-MARCO scale 8,841,823 x 768, 6,980 queries,
top-100, with its own CPU leg), at k = 1001 (what ranking() asks for) and the configs[4] in-batch-negative loss step -- short runs,
never part of `value`.
--data clustered: a non-iid corpus (1,024 Gaussian clusters, log-normal row norms, 3 % duplicate rows).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "crowd-coachable-recommendations_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

N_ROWS, DIM, N_Q, TOP_K = 2_681_468, 768, 3_452, 100
MSMARCO_ROWS, MSMARCO_Q = 8_841_823, 6_980
C4_ROWS, C4_Q, C4_DIM, C4_K = 50_000_000, 10_000, 1024, 1000    # BASELINE.json configs[3]
MFMA_PEAK_TFLOPS = 2500.0   # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r06_locality.json")   # tools/exp_locality.py: separate rocprofv3 --pmc passes per workload


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=N_ROWS, help="corpus rows (default: NQ)")
    ap.add_argument("--queries", type=int, default=N_Q)
    ap.add_argument("--dim", type=int, default=DIM)
    ap.add_argument("--k", type=int, default=TOP_K)
    ap.add_argument("--data", default="gaussian", choices=["gaussian", "clustered", "sorted", "outlier"],
                    help="gaussian: iid N(0, 1/dim) rows (BASELINE.md); clustered: 1,024 clusters, log-normal norms, 3 %% duplicates; "
                         "sorted: the same with every cluster's rows CONTIGUOUS (a corpus in topical order)")
    ap.add_argument("--sim", default="dot", choices=["dot", "cos"], help="cos: rows are L2-normalised by the pack kernel (CCREC_SIM_TYPE=cos)")
    ap.add_argument("--cpu-queries", type=int, default=64, help="query sample of the CPU baseline (0 = skip)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the MS-MARCO-scale and k = 1001 side runs")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (single-GPU rehearsal of N > 1)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--dump-ids", default=None, help="rank 0 saves the final [queries, k] id tensor here (tests)")
    ap.add_argument("--rehearse-secondary", type=int, default=0, metavar="D",
                    help="tests: run the N > 1 side runs on a custom shape too, the configs[2] one on 1/D of its rows and queries")
    return ap.parse_args()


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def host_threads():
    """Every core this process may use: the affinity mask, capped by the cgroup CPU quota (CCR_BENCH_CPU_THREADS overrides)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    forced = os.environ.get("CCR_BENCH_CPU_THREADS")
    return max(1, int(forced)) if forced else max(1, n)


def gen_rows(n, dim, seed, device, data="gaussian", chunk=262144, lo=0, hi=None):
    """Rows [lo, hi) of the n-row fp32 matrix `seed` names, generated on device chunk by chunk.  Chunk c (global rows
    [c chunk, (c + 1) chunk)) is drawn from its OWN generator state (seed, c), so a rank of an N-way row sharding generates
    only the chunks that overlap its shard and still holds exactly the rows the single-GPU run holds: the corpus is the
    same for every N, and no rank ever materialises more than its shard plus one chunk.
    gaussian: N(0,1) / sqrt(dim).
    clustered / sorted: 1,024 cluster centres (seeded, shared by corpus and queries), row = 0.8 centre + 0.6 noise (unit scale),
    times a log-normal norm (sigma 0.35); 3 % of the rows of a chunk are exact duplicates of other rows of that chunk.
    outlier: gaussian, with every 500,000th corpus row scaled by CCR_BENCH_OUTLIER (default 100): rows whose norm is far above the
    rest (the filter margins scale with the LARGEST row norm, DESIGN 4.4 'known limit')."""
    hi = n if hi is None else hi
    assert 0 <= lo <= hi <= n
    g = torch.Generator(device=device)
    out = torch.empty(hi - lo, dim, dtype=torch.float32, device=device)
    centres = None
    if data in ("clustered", "sorted"):
        gc = torch.Generator(device=device).manual_seed(777)
        centres = torch.randn(1024, dim, generator=gc, device=device) * dim ** -0.5
    for c in range(lo // chunk, (hi + chunk - 1) // chunk if hi > lo else 0):
        c_lo = c * chunk
        c_hi = min(n, c_lo + chunk)
        m = c_hi - c_lo
        g.manual_seed(seed * 1_000_003 + c)
        x = torch.randn(m, dim, generator=g, device=device) * dim ** -0.5
        if centres is not None:
            cid = torch.randint(0, 1024, (m,), generator=g, device=device)
            if data == "sorted" and seed == 1234:   # corpus rows in topical order: cluster c owns rows [c n / 1024, (c + 1) n / 1024)
                cid = (torch.arange(c_lo, c_hi, device=device, dtype=torch.int64) * 1024 // n).clamp_(max=1023)
            x = 0.8 * centres[cid] + 0.6 * x
            x *= torch.exp(0.35 * torch.randn(m, 1, generator=g, device=device))
            ndup = int(0.03 * m)
            if ndup and m > 1:
                dst = torch.randint(0, m, (ndup,), generator=g, device=device)
                src = torch.randint(0, m, (ndup,), generator=g, device=device)
                # a destination drawn twice keeps its FIRST source: an indexed store with repeated destinations leaves the winner to
                # the hardware, and the ranks of a sharded run would then hold different rows than the single-GPU run
                order = torch.arange(ndup, device=device)
                first = torch.full((m,), ndup, dtype=torch.int64, device=device).scatter_reduce_(0, dst, order, "amin")
                keep = first[dst] == order
                x[dst[keep]] = x[src[keep]]
        if data == "outlier" and seed == 1234:
            first = (c_lo + 499_999) // 500_000 * 500_000   # global rows 0, 500 000, ... inside this chunk
            if first < c_hi:
                x[first - c_lo::500_000] *= float(os.environ.get("CCR_BENCH_OUTLIER", "100"))
        a, b = max(lo, c_lo), min(hi, c_hi)
        out[a - lo:b - lo] = x[a - c_lo:b - c_lo]
    return out


def cpu_baseline(corpus_bf16, queries_bf16, nq_sample, k, gpu_ids):
    """Reference-faithful CPU path (oracle.reference_ranking == scripts/ms_marco_eval.py:203-235:
    chunked fp32 matmul into a host [Q,N] matrix, per-row full descending sort, keep 1001) on the
    same bf16-rounded values, on every core this process may use."""
    from oracle import oracle as orc
    Ed = corpus_bf16.float().cpu().numpy()
    Eq = queries_bf16[:nq_sample].float().cpu().numpy()
    torch.set_num_threads(host_threads())
    log(f"cpu baseline: {nq_sample} queries x {Ed.shape[0]} rows, {torch.get_num_threads()} threads")
    t0 = time.time()
    ids, _ = orc.reference_ranking(Eq, Ed, 2048, "dot")
    dt = time.time() - t0
    rec = orc.recall_at_k(gpu_ids[:nq_sample].cpu().numpy(), ids[:, :k])
    # SURVEY 8d variant (ii), the best the host can do with the same libraries: fp32 Q @ D^T + torch.topk(k) per query
    # block, no host score matrix, no full sort (reported beside the reference-faithful number, never instead of it)
    nq2 = min(4 * nq_sample, queries_bf16.shape[0])
    Eq2, Edt = queries_bf16[:nq2].float().cpu(), torch.from_numpy(Ed)
    t0 = time.time()
    for lo in range(0, nq2, 64):
        (Eq2[lo:lo + 64] @ Edt.T).topk(k, dim=1)
    dt2 = time.time() - t0
    return {"value": round(nq_sample / dt, 3), "unit": "queries/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{nq_sample} queries x full {Ed.shape[0]}-row corpus, fp32 matmul chunks of 2048 + per-row full sort, "
                      f"keep 1001 ({dt:.1f} s)",
            "recall_at_k_of_gpu_vs_cpu": round(rec, 5),
            "best_effort_value": round(nq2 / dt2, 3),
            "best_effort_sample": f"{nq2} queries, fp32 matmul + torch.topk({k}) in blocks of 64 queries ({dt2:.1f} s)"}


def completeness_all_queries(corpus_bf16, queries_bf16, scores, chunk=65536):
    """Untimed check over EVERY query of the step's result (the CPU leg re-computes a bounded sample only): an independent fp32 sweep of
    the whole packed corpus -- torch's own matmul, chunk by chunk on the device -- counts, per query, the rows above the list's k-th
    score + 1e-4 (+- 1e-5); a list that missed such a row would hold fewer of them than the sweep finds.  -> the record for the JSON line."""
    Qf = queries_bf16.float()
    kth = scores[:, -1].float()
    hi_thr, lo_thr = (kth + 1.1e-4)[None, :], (kth + 0.9e-4)[None, :]
    above_hi = torch.zeros(Qf.shape[0], dtype=torch.int64, device=Qf.device)
    above_lo = torch.zeros_like(above_hi)
    for lo in range(0, corpus_bf16.shape[0], chunk):
        a = corpus_bf16[lo:lo + chunk].float() @ Qf.T
        above_hi += (a > hi_thr).sum(0)
        above_lo += (a > lo_thr).sum(0)
    in_list = (scores > (kth + 1e-4)[:, None]).sum(1)
    bad = int(((above_hi > in_list) | (in_list > above_lo)).sum())
    return {"queries_checked": int(Qf.shape[0]), "queries_with_a_missing_row_above_the_cut": bad,
            "rule": "fp32 torch.matmul sweep of the whole corpus on the device: rows scoring above the list's k-th score + 1e-4 must all be in the list"}


class Workload:
    """One configuration of the hot path on this rank: resident fp32 inputs, packed buffers, the step closure.

    Steps are PIPELINED on the host: step i enqueues pack + index + asynchronous search (+ the all-gather behind it on RCCL's
    stream), then completes step i - 1 (ccr_search_finish waits for THAT search's own event; the exchange's headers arrive in
    pinned memory from a side stream; the merge is enqueued) -- so the GPU never waits for the host between two steps, and the
    exchange of step i - 1 overlaps the pack and search of step i.  Packed shard, query pack, norm bounds and exchange message
    are double-buffered: a step that had to re-do flagged queries after the fact still finds its operands."""

    def __init__(self, rows, queries, dim, k, data, dev, rank, world, backend, normalize=False):
        from ccrec_amd.dist import shard_bounds
        self.rows, self.queries, self.dim, self.k, self.world, self.dev, self.backend = rows, queries, dim, k, world, dev, backend
        self.normalize = normalize
        self.lo, self.hi = shard_bounds(rows, world, rank)
        # a rank generates only its own rows of the global corpus (per-chunk generator states): identical corpus for every N
        self.corpus_f32 = gen_rows(rows, dim, 1234, dev, data, lo=self.lo, hi=self.hi)
        self.queries_f32 = gen_rows(queries, dim, 4321, dev, data)
        n_local = self.hi - self.lo
        self.k_local = min(k, n_local)
        if world > 1:
            assert self.k_local == k, "shard smaller than k"
        from ccrec_amd.ops import padded_dim
        pdim = padded_dim(dim)    # packed rows are zero-padded to a multiple of 8 (--dim 300 -> 304)
        self.slots = [{"shard": torch.empty(n_local, pdim, dtype=torch.bfloat16, device=dev),
                       "qpack": torch.empty(queries, pdim, dtype=torch.bfloat16, device=dev),
                       "bounds": torch.empty(n_local, dtype=torch.float32, device=dev),   # norm bound per packed row (pack kernel)
                       "ws": None,   # search workspace, handed from the slot's previous index to its next one
                       "message": None} for _ in range(2)]
        self.shard, self.qpack = self.slots[0]["shard"], self.slots[0]["qpack"]
        self.prev = None        # the step whose search / exchange is still in flight
        # the pack of step i + 1 (HBM-bound, its own buffers) runs on a side stream from the moment the main pass of step i is done: beside
        # step i's select stage (gather- and latency-bound) instead of behind it
        self.side = torch.cuda.Stream(device=dev) if (str(dev).startswith("cuda") and os.environ.get("CCR_BENCH_SIDE_PACK", "1") != "0") else None
        self.nstep = 0
        self.suspended_at_step = None
        self.index = self.scores = self.ids = None
        self.set_k(k)

    def set_k(self, k):
        """(Re)size the exchange for top-k: every rank sends k_list entries per query -- the short-list exchange (ccrec_amd/dist.py:
        k / R + 6 sigma entries, the merge verifies the shortcut) where it pays, full lists otherwise."""
        from ccrec_amd.dist import ShardMessage, short_list_length, short_lists_pay
        self.drain()
        self.k = k
        self.k_local = min(k, self.hi - self.lo)
        if self.world > 1:
            assert self.k_local == k, "shard smaller than k"
        self.k_list = short_list_length(k, self.world) if (self.world > 1 and short_lists_pay(k, self.world)) else k
        for b in self.slots:
            b["message"] = ShardMessage(self.queries, self.k_list, self.dev, self.world) if self.world > 1 else None
        self.reset_counters()

    def reset_counters(self):
        self.done_stats = []    # last_stats() of every completed step
        self.repeats = 0        # exchanges that had to be repeated (some rank's search had flagged queries: completed by finish(), gathered again)
        self.fallback_queries = 0   # short lists: queries repeated with full lists (a list was consumed to its end)
        self.wait_ms = []       # host-observed wait for each step's collective, at the time the step is completed (one step later)
        self.host_syncs = 0     # times an exchange had to synchronise with the compute stream (0 on the pipelined path)

    def step(self):
        from ccrec_amd import ops
        from ccrec_amd.dist import submit_sharded_search, short_lists_pay
        if self.world > 1 and self.k_list < self.k and not short_lists_pay(self.k, self.world):
            # an exchange repeated too many queries (a corpus in topical order): every rank has seen the same flags and switches to
            # full lists at this same step
            self.suspended_at_step = self.nstep
            self.drain()
            counters = (self.done_stats, self.repeats, self.fallback_queries, self.wait_ms, self.host_syncs)
            self.set_k(self.k)
            self.done_stats, self.repeats, self.fallback_queries, self.wait_ms, self.host_syncs = counters
        b = self.slots[self.nstep % 2]
        self.nstep += 1
        if self.side is not None:
            main = torch.cuda.current_stream()
            if self.prev is not None:
                self.prev[0].stream_wait_main_pass(self.side)   # (the slot's buffers were last read by the search TWO steps ago, long complete by then)
            else:
                self.side.wait_stream(main)
            with torch.cuda.stream(self.side):
                ops.pack_bf16(self.corpus_f32, out=b["shard"], norm_bounds=b["bounds"], normalize=self.normalize)
            main.wait_stream(self.side)
        else:
            ops.pack_bf16(self.corpus_f32, out=b["shard"], norm_bounds=b["bounds"], normalize=self.normalize)   # pack + norm bound of every packed row in one pass
        index = ops.CorpusIndex(b["shard"], global_row_offset=self.lo, norm_bounds=b["bounds"], workspace=b["ws"])
        ops.pack_bf16(self.queries_f32, out=b["qpack"], normalize=self.normalize)
        if self.world > 1:
            # asynchronous search straight into the packed message, all-gather behind it on the communication stream
            cur = (index, submit_sharded_search(index, b["qpack"], self.k, message=b["message"]), None, None)
        else:
            s, i = index.search(b["qpack"], self.k_local, defer=True)
            cur = (index, None, s, i)
        b["ws"] = index.workspace     # the slot's previous index was completed a step ago (drain below completes the other slot's)
        self.drain()
        self.prev = cur
        self.shard, self.qpack = b["shard"], b["qpack"]

    def drain(self):
        """Complete the step still in flight: finish its search (its own event), merge the gathered per-shard lists."""
        if self.prev is None:
            return
        index, exchange, s, i = self.prev
        self.prev = None
        if exchange is not None:
            s, i = exchange.result()
            self.repeats += int(exchange.repeated)
            self.fallback_queries += exchange.fallback_queries
            self.wait_ms.append(exchange.wait_ms)
            self.host_syncs += exchange.host_syncs
        else:
            index.finish()
        self.index, self.scores, self.ids = index, s, i
        self.done_stats.append(index.last_stats())

    def fence(self):
        self.drain()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(self, steps, warmup, tag):
        for _ in range(warmup):
            self.step()
        self.fence()
        if self.done_stats:
            log(f"{tag}: warmup step", self.done_stats[-1])
        self.reset_counters()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.fence()
        elapsed = time.perf_counter() - t0
        stats = self.done_stats
        per_rank = [elapsed]
        if self.world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
            every = [torch.zeros_like(t) for _ in range(self.world)]
            dist.all_gather(every, t)
            per_rank = [float(x.item()) for x in every]
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        ms_per_step = elapsed / steps * 1e3
        log(f"{tag}: timed {steps} steps: {ms_per_step:.3f} ms/step")
        st = stats[-1]
        avg_main = sum(m["ms_main"] for m in stats) / len(stats)
        flops = 2.0 * self.queries * (self.hi - self.lo) * self.dim
        achieved = flops / (avg_main * 1e-3) / 1e12 if avg_main > 0 else 0.0
        return {"qps": self.queries * steps / elapsed, "ms_per_step": ms_per_step, "stats": st, "avg_main": avg_main, "flops": flops,
                "achieved_tflops": achieved, "n_fallback_max": max(m["n_fallback"] for m in stats),
                "rank_ms_per_step": [e / steps * 1e3 for e in per_rank], "exchange_repeats": self.repeats,
                "fallback_queries": self.fallback_queries, "search_ms": sum(m["ms_total"] for m in stats) / len(stats),
                "exchange_host_syncs": self.host_syncs, "short_lists_suspended_at_step": self.suspended_at_step,
                "exchange_wait_ms": (sum(self.wait_ms) / len(self.wait_ms)) if self.wait_ms else 0.0,
                "exchange_wait_ms_max": max(self.wait_ms) if self.wait_ms else 0.0}

    def release(self):
        self.drain()
        self.corpus_f32 = self.queries_f32 = self.shard = self.qpack = self.index = self.scores = self.ids = self.slots = None
        torch.cuda.empty_cache()


def inbatch_side_run(dev, B=1024, d=768, iters=200):
    """configs[4]: the in-batch-negative contrastive step (bbpr.py:205-212) at B = 1024, d = 768: forward + backward of the HIP loss
    (ccr_inbatch_ce_fwd/bwd: three kernel launches) next to the reference's torch formulation (mm, mm, cat, scale, CrossEntropyLoss +
    autograd) on this GPU, in fp32 and -- the comparator that matches this library's bf16 operands -- under autocast(bf16).
    `roofline`: the step is 3 x 2 x 2 B^2 d = 9.66 GFLOP (two logit blocks forward, x 3 with the backward) = microseconds of
    matrix work, so what bounds it is LAUNCH LATENCY: dependent kernels + the host's autograd round trip; `kernels_ms` (HIP events
    around the library calls alone, operands already bf16) against `value` shows the split."""
    from ccrec_amd import _lib, ops
    g = torch.Generator(device=dev).manual_seed(0)
    q, p, n = (torch.randn(B, d, device=dev, generator=g) * d ** -0.5 for _ in range(3))

    def ours():
        a, b, c = (t.clone().requires_grad_(True) for t in (q, p, n))
        loss = ops.inbatch_ce(a, b, c, 20.0)
        loss.backward()
        return loss

    def ref(dtype):
        a, b, c = (t.clone().requires_grad_(True) for t in (q, p, n))
        with torch.autocast("cuda", dtype=dtype, enabled=dtype != torch.float32):
            scores = torch.cat([a @ b.T, a @ c.T], 1) * 20.0
            loss = torch.nn.CrossEntropyLoss()(scores.float(), torch.arange(B, device=dev))
        loss.backward()
        return loss

    out = {"workload": f"configs[4]: in-batch-negative loss forward + backward, B = {B}, d = {d}, inv_temperature 20"}
    for name, fn in (("hip_ms", ours), ("torch_fp32_ms", lambda: ref(torch.float32)), ("torch_bf16_autocast_ms", lambda: ref(torch.bfloat16))):
        for _ in range(30):     # (the first steps of a variant pay its module loads and the allocator's first blocks)
            loss = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            loss = fn()
        torch.cuda.synchronize()
        out[name] = round((time.perf_counter() - t0) / iters * 1e3, 4)
        out[name.replace("_ms", "_loss")] = round(float(loss), 6)
    out["steps_per_s"] = round(1e3 / out["hip_ms"], 1)
    # the library calls alone (the calls the autograd function makes: fp32 blocks resident -> one-launch bf16 pack + forward, then the
    # backward), HIP events on the stream they are launched on
    lib = _lib.load()
    packed = torch.empty(3, B, d, dtype=torch.bfloat16, device=dev)
    loss_t, lse = torch.empty(1, device=dev), torch.empty(B, device=dev)
    grads = [torch.empty(B, d, device=dev) for _ in range(3)]
    ws = torch.empty(int(lib.ccr_inbatch_ce_workspace_bytes(B, d)), dtype=torch.uint8, device=dev)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def kernels():
        _lib.check(lib.ccr_inbatch_ce_fwd_f32(ptr(q), ptr(p), ptr(n), B, d, 20.0, ptr(packed), ptr(loss_t), ptr(lse), ptr(ws), ws.numel(), stream))
        _lib.check(lib.ccr_inbatch_ce_bwd(ptr(packed[0]), ptr(packed[1]), ptr(packed[2]), ptr(lse), B, d, 20.0, 1.0, ptr(grads[0]), ptr(grads[1]), ptr(grads[2]),
                                          ptr(ws), ws.numel(), stream))
    for _ in range(5):
        kernels()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        kernels()
    e1.record()
    torch.cuda.synchronize()
    kms = e0.elapsed_time(e1) / iters
    flops = 3 * 2 * 2.0 * B * B * d          # 2 x 2 B^2 d forward (two logit blocks), x 3 with the backward (dQ over 2B keys, dP, dN) = 9.66 GFLOP
    out["kernels_ms"] = round(kms, 4)
    out["roofline"] = {"bound": "launch latency (4 dependent kernels of microseconds of MFMA work each)", "kernel": "inbatch_pack3_kernel + inbatch_fwd_kernel<FRAG> + inbatch_prep_kernel + inbatch_gemm3_kernel",
                       "achieved": round(flops / (kms * 1e-3) / 1e12, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": round(flops / (kms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4), "flops_per_step": flops,
                       "launch_floor_ms": round(3 * 0.0019 + 0.0015, 4),
                       "note": "flops = the survey's algorithmic count; the backward's MFMA work is 3x its share of it (the gradient of the logits enters as "
                               "three bf16 parts so that fp32-accurate products run on the bf16 matrix cores: 16.1 GFLOP executed per step); floor = 3 kernel "
                               "boundaries at ~1.9 us + one memset node (MI355X_MICROARCH price list); hip_ms adds the one-launch bf16 pack and the host-side "
                               "autograd round trip (clone, Function.apply, backward)", "traffic": None}
    return out


def encode_side_run(dev, texts=16384):
    """SURVEY 8 f2 beside the search: `texts` synthetic passages (mean 136 tokens, pre-tokenised ids) through a random-init BERT-base
    under the reference's own `torch.autocast("cuda")` (fp16, scripts/al_0_rank.py:8,125) with the length-sorted encoder -- the encoder
    layers as torch modules, then on the library's attention /
    add + LayerNorm kernels (ccrec_amd/fused_bert.py).  GPU seconds from events around the batches; host tokenisation excluded
    (one chunk, prepared before the GPU starts).  Bounded: ~10 s with the model construction."""
    import numpy as np
    try:
        from transformers import BertConfig, BertModel
    except Exception as e:      # the encoder itself is the reference's dependency, not this library's
        return {"skipped": f"transformers not importable: {e}"}
    from ccrec_amd.encode import LengthSortedEncoder
    from ccrec_amd.item_tower import NaiveItemTower

    class IdTokenizer:          # texts are id strings: batching + GPU time only
        pad_token_id = 0

        def __call__(self, texts, truncation=True, padding=False, max_length=200, return_tensors="pt"):
            ids = [[101] + [int(w) for w in t.split()][: max_length - 2] + [102] for t in texts]
            return {"input_ids": ids, "attention_mask": [[1] * len(r) for r in ids]}

    rs = np.random.RandomState(0)
    lens = np.clip(rs.normal(135, 30, texts).astype(int), 20, 198)
    corpus = [" ".join(map(str, rs.randint(1000, 30000, n))) for n in lens]
    torch.manual_seed(0)
    tower = NaiveItemTower(BertModel(BertConfig(vocab_size=30522)).eval(), torch.nn.LayerNorm(768, elementwise_affine=False)).to(dev)
    out = {"workload": f"{texts} synthetic passages (mean {float(lens.mean()) + 2:.0f} tokens), random-init BERT-base, fp16 autocast (the reference's torch.cuda.amp.autocast()), "
                       "length-sorted batches of <= 65,536 tokens, masked mean pooling + bf16 pack into the shard", "unit": "passages/s (GPU time of the batches)"}
    rows = {}
    for name, fused in (("torch_modules", False), ("layer_kernels", True)):
        enc = LengthSortedEncoder(tower, IdTokenizer(), max_length=200, max_tokens=65536, max_batch=2048, chunk_texts=10 ** 9, fused=fused)
        with torch.autocast("cuda"):                  # the CUDA default: fp16 -- the layer kernels run in the caller's autocast type
            enc.encode(corpus[:2048])                 # GEMM shapes, library load: untimed
            f32 = torch.empty(texts, 768, dtype=torch.float32, device=dev)
            enc.encode(corpus, out_f32=f32)
        rows[name] = f32
        out[name] = {"value": round(texts / enc.stats["gpu_busy_s"], 1), "gpu_busy_s": round(enc.stats["gpu_busy_s"], 3),
                     "batches": enc.stats["batches"], "padded_tokens": enc.stats["padded_tokens"], "layer_dtype": str(enc.stats.get("layer_dtype"))}
    # forward flops of the REAL tokens (no padding row is computed on the layer-kernel path): per token and layer 2 x (4 d^2 + 2 d ffn) in the
    # projections + 4 d x (its sequence's length) in attention
    d_model, ffn, layers = 768, 3072, 12
    tok = (lens + 2).astype(np.float64)
    flops = layers * float((tok * (2.0 * (4 * d_model * d_model + 2 * d_model * ffn)) + 4.0 * d_model * tok * tok).sum())
    tf = flops / out["layer_kernels"]["gpu_busy_s"] / 1e12
    out["roofline"] = {"bound": "mfma", "kernel": "encoder forward: hipBLASLt projections (63 % of the GPU time) + attention / add + LayerNorm / GELU / embedding kernels",
                       "achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS, 4),
                       "flops_per_pass": flops, "traffic": None,
                       "note": "a fraction of the MFMA peak for the WHOLE forward; its HBM-bound passes (residual + LayerNorm, GELU, embeddings) carry no flops to speak of "
                               "and run at 4.7 - 5.9 TB/s (DESIGN 4.8)"}
    cos = torch.nn.functional.cosine_similarity(rows["torch_modules"], rows["layer_kernels"], dim=1)
    out["speedup"] = round(out["layer_kernels"]["value"] / out["torch_modules"]["value"], 3)
    out["cosine_of_pooled_rows_min"] = round(float(cos.min()), 6)
    del rows, tower
    torch.cuda.empty_cache()
    return out


def bm25_workload(docs=500_000, queries=2000, vocab=50_000, k1=1.2, b=0.75):
    """The synthetic BM25 workload of the bench line (also tools/one_bm25.py): postings of `docs` documents of 20-79 Zipf(1.07) words
    over `vocab` terms built with numpy, `queries` queries of 3-11 Zipf words -> (model, query term arrays, df, raw arrays)."""
    import numpy as np
    from ccrec_amd.bm25 import BM25
    rs = np.random.RandomState(0)
    p = 1.0 / np.arange(1, vocab + 1) ** 1.07
    p /= p.sum()
    lens = rs.randint(20, 80, docs)
    doc_of = np.repeat(np.arange(docs, dtype=np.int64), lens)
    term_of = rs.choice(vocab, int(lens.sum()), p=p).astype(np.int64)
    key, counts = np.unique(term_of * docs + doc_of, return_counts=True)        # term-major, documents ascending inside a term
    terms, rows = key // docs, (key % docs).astype(np.int32)
    indptr = np.zeros(vocab + 1, np.int64)
    np.cumsum(np.bincount(terms, minlength=vocab), out=indptr[1:])
    df = np.diff(indptr)
    idf = np.log(docs / np.maximum(df, 1).astype(np.float64))
    doc_k = k1 * (1 - b + b * lens / lens.mean())
    qs = [np.unique(rs.choice(vocab, rs.randint(3, 12), p=p)).astype(np.int32) for _ in range(queries)]
    qs = [q[df[q] > 0] for q in qs]
    model = BM25.from_postings(indptr, rows, counts.astype(np.float32), doc_k, idf, k1=k1, b=b)
    return model, qs, df, (indptr, rows, counts, doc_k, idf, k1)


def bm25_side_run(dev, docs=500_000, queries=2000, vocab=50_000, k=1001, cpu_queries=10):
    """SURVEY 8 f4 beside the search: the lexical leg of the candidate builder (scripts/bm_25.py, ranking_bm25) on the device --
    `queries` Zipf queries of 3-11 words against `docs` synthetic documents (20-79 words, Zipf 1.07 over `vocab` terms), top-1001 =
    ranking_bm25's KEEP.  The postings are built with numpy (the text analysis is host work and not what is measured); `value` is
    the library call (tables + kernels, ccr_bm25_search); the reference formulation (scipy column slice + dense divide + row sum +
    full sort per query, bm_25.py:31-52 + ms_marco_eval.py:177-185) runs on the same host for `cpu_queries` queries.  ~10 s."""
    import numpy as np
    model, qs, df, (indptr, rows, counts, doc_k, idf, k1) = bm25_workload(docs, queries, vocab)
    model.transform_terms_topk(qs[:64], k)
    best = 1e9
    for _ in range(3):
        s, i = model.transform_terms_topk(qs, k)
        best = min(best, model.last_search_seconds)
    postings = int(sum(int(df[q].sum()) for q in qs))
    stats = model.last_stats()
    nnz = len(rows)
    # ALGORITHMIC HBM bytes of one call: the index once (4 B id + 8 B finished contribution per posting; every later read of a list is
    # an on-chip re-read), the candidate records written and read once (8 B each way; ~2.6 k per query at k = 1001), the output
    cand = 2700 * queries
    algo_bytes = nnz * 12 + cand * 16 + queries * k * 12
    pmc = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r05_bm25_pmc.json")))
    except Exception:
        pass
    out = {"workload": f"BM25 (k1 1.2, b 0.75): {queries} queries of 3-11 Zipf words x {docs:,} synthetic documents ({nnz:,} postings, "
                       f"{postings:,} touched), top-{k} (ranking_bm25's KEEP)", "value": round(queries / best, 1), "unit": "queries/s (library call: tables + kernels)",
           "ms_per_call": round(best * 1e3, 3), "path": stats,
           "roofline": {"bound": "hbm", "kernel": "bm25_tile_kernel<1024,2,FILTER,TABLE> (+ sample / threshold / top-k sort)",
                        "achieved": round(algo_bytes / best / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(algo_bytes / best / 1e9 / HBM_PEAK_GBS, 4),
                        "bytes_per_unit": "12 per posting of the index (read from HBM once per call) + 16 per candidate record + 12 per output entry",
                        "algorithmic_bytes_per_call": algo_bytes,
                        "traffic": (pmc or {}).get("fabric_bytes_per_call"),
                        "traffic_source": ("offline rocprofv3 --pmc passes of tools/one_bm25.py (profiles/r05_bm25_pmc.json: FETCH_SIZE x 2 KiB + WRITE_SIZE x 1 KiB "
                                           "of the call's kernels); not measured in this run") if pmc else None,
                        "note": "far below the HBM roof by construction: the 2,000 queries re-read the index ~90x and those reads are served by the XCDs' L2 "
                                "(the kernel's on-chip request stream and instruction issue bound it, see postings_per_s); no [queries][documents] score row exists any more",
                        "postings_per_s": round(postings / best / 1e9, 1), "postings_unit": "G postings/s (touched postings / call time)"}}
    if cpu_queries > 0:
        import scipy.sparse as sp
        X = sp.csc_matrix((counts.astype(np.float64), rows, indptr), shape=(docs, vocab))
        t0 = time.perf_counter()
        agree = ties = other = 0
        gpu_ids = i[:cpu_queries].cpu().numpy()
        for qi in range(cpu_queries):
            t = qs[qi]
            Xq = X[:, t]
            denom = Xq + doc_k[:, None]
            numer = Xq.multiply(np.broadcast_to(idf[None, t], Xq.shape)) * (k1 + 1)
            sol = torch.Tensor(np.asarray((numer / denom).sum(1)).ravel())
            _, order = sol.sort(descending=True)
            # tie-aware agreement (torch.sort's order inside equal scores is arbitrary, and documents of equal length and counts tie
            # exactly): an id only ONE side keeps must score the CPU's k-th score -- to one fp32 ulp, the CPU sums a query's terms pairwise
            cpu_top, kth = set(order[:k].tolist()), float(sol[order[k - 1]])
            gpu_top = set(gpu_ids[qi].tolist())
            agree += len(cpu_top & gpu_top)
            for d_ in gpu_top ^ cpu_top:
                if abs(float(sol[d_]) - kth) <= abs(kth) * 2.0 ** -23:
                    ties += 1
                else:
                    other += 1
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(cpu_queries / cdt, 2), "unit": "queries/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{cpu_queries} queries, scipy column slice + dense divide + row sum + full sort",
                               "recall_of_gpu_ids": round((agree + ties / 2) / (cpu_queries * k), 5),
                               "recall_rule": "an id kept by one side only counts as agreeing when its CPU score equals the CPU's k-th score to one fp32 ulp (a tie at the cut)",
                               "ids_differing_only_by_ties": ties // 2, "ids_differing_otherwise": other}
        out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    del model
    torch.cuda.empty_cache()
    return out


def small_batches_side_run(index, qpack, n_rows, dim, k=TOP_K, sizes=(1, 16, 64, 128, 256, 384, 512), reps=20):
    """SURVEY 8d's small query batches (Q in {1, 16, 64, 512}; 256 = one full query tile) on the resident NQ index: search-only time
    (the index and the packed queries exist: an interactive ranking() call against an encoded corpus), `reps` searches back to back, each
    completed before the next starts.  Below a few hundred queries the main pass can at best stream the corpus once: `roofline` is the
    HBM one, bytes = 2 N d per search, the main pass's duration from the library's events on the search stream.  n_q <= 64
    runs the streaming main pass (csrc/ccr_narrow.hip), 65 .. 128 the same kernel as two query groups on paired workgroups (every row
    pulled twice, once of them from the L2 / the Infinity Cache), larger batches the tile kernels (257 .. 384 queries: ONE block of the
    256 x 384 form of the main pass instead of two 256-query blocks)."""
    out = {"workload": f"configs[1] corpus resident and indexed, top-{k}, batches of n_q queries (search only)", "unit": "ms per search", "batches": {}}
    pmc = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r06_small_batches_pmc.json")))   # tools/pmc_small_batches_json.sh
    except Exception:
        pass
    for nq in sizes:
        Q = qpack[:nq].contiguous()
        for _ in range(3):
            index.search(Q, k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mains = []
        for _ in range(reps):
            index.search(Q, k)
            mains.append(index.last_stats()["ms_main"])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        st = index.last_stats()
        main = sum(mains) / len(mains)
        bytes_ = 2.0 * n_rows * dim
        gbs = bytes_ / (main * 1e-3) / 1e9
        traffic = ((pmc or {}).get(str(nq)) or {}).get("fabric_bytes_main_pass")
        out["batches"][str(nq)] = {
            "ms_per_search": round(ms, 4), "queries_per_s": round(nq / ms * 1e3, 1), "phases_ms": phases_obj(st),
            "main_pass": ("narrow_filter_kernel (streaming: queries resident in LDS, corpus straight into the MFMA operand registers"
                          + (")" if nq <= 64 else "; two query groups on paired workgroups)")) if st.get("ranges") == 1 and st.get("sublists") == 2
                         else ("gemm_topk16w_kernel (384-query tiles)" if st.get("main_tile_queries") == 384 else "gemm_topk16_kernel<EPI_FILTER> (256-query tiles)"),
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                         "bytes_per_search": bytes_, "main_pass_ms": round(main, 4), "traffic": traffic,
                         "traffic_source": "offline rocprofv3 --pmc passes (profiles/r06_small_batches_pmc.json: FETCH_SIZE x 2 + WRITE_SIZE of one search's main-pass launches); not measured in this run" if traffic else None}}
    return out


def roofline_obj(r, traffic=None, traffic_source=None):
    """`achieved` = algorithmic flops of one step's main pass (2 n_q n_rows dim: every launch of the pass covers its share of the
    corpus, together exactly once) / the main pass's duration per step from the library's HIP events on the search stream."""
    st = r["stats"]
    return {"bound": "mfma", "kernel": ("gemm_topk16w_kernel (main pass, 256 x 384 tiles, v_mfma_f32_16x16x32_bf16)" if st.get("main_tile_queries") == 384
                                        else "gemm_topk16_kernel<EPI_FILTER> (main pass, v_mfma_f32_16x16x32_bf16)" if st.get("sublists") == 8
                                        else "gemm_topk_kernel<EPI_FILTER> (main pass, v_mfma_f32_32x32x16_bf16)"),
            "achieved": round(r["achieved_tflops"], 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(r["achieved_tflops"] / MFMA_PEAK_TFLOPS, 4),
            # the same algorithmic flops over the WHOLE timed step (pack + index + query pack + sample + thresholds + main pass + select):
            # what `value` itself is worth against the MFMA peak; `frac` above covers the main-pass launches only
            "step_frac": round(r["flops"] / (r["ms_per_step"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
            "flops_per_step": r["flops"],
            "main_pass_ms_per_step": round(r["avg_main"], 4), "launches_per_step": st.get("main_launches"),
            "traffic": traffic, "traffic_source": traffic_source}


def phases_obj(st):
    return {"sample_pass": round(st["ms_sample"], 3), "threshold": round(st["ms_threshold"], 3), "main_pass": round(st["ms_main"], 3),
            "select_rescore": round(st["ms_select"], 3), "fallback": round(st["ms_fallback"], 3), "search_total": round(st["ms_total"], 3)}


def exchange_obj(w, r):
    """What one step moves between the ranks and what the host saw of it (N > 1)."""
    from ccrec_amd.ops import shard_message_bytes
    msg = shard_message_bytes(w.queries, w.k_list)
    short = w.k_list < w.k
    return {"n_ranks_seen": dist.get_world_size(), "backend": dist.get_backend(),
            "collective": "one all_gather_into_tensor of the packed shard message per step (header + fp32 scores + u32 local rows)",
            "lists": (f"short: every rank sends its canonical top-{w.k_list} of top-{w.k} (k / R + 6 sigma + 8); the merge verifies the cuts, "
                      f"queries that fail are repeated with full lists" if short else f"full: every rank sends its canonical top-{w.k}"),
            "entries_per_query_per_rank": w.k_list, "message_bytes_per_rank": msg, "gathered_bytes_per_rank_per_step": msg * w.world,
            "full_list_message_bytes_per_rank": shard_message_bytes(w.queries, w.k),
            "repeated_exchanges": r["exchange_repeats"], "queries_repeated_with_full_lists": r["fallback_queries"],
            "compute_stream_syncs_in_exchange": r["exchange_host_syncs"], "short_lists_suspended_at_step": r["short_lists_suspended_at_step"],
            "rank_ms_per_step_min": round(min(r["rank_ms_per_step"]), 3), "rank_ms_per_step_max": round(max(r["rank_ms_per_step"]), 3),
            # the per-shard search on this rank's stream (library events) vs. what the host waited for the collective of a step when it
            # completed that step one step later (0 = it had long arrived behind the next step's pack and search)
            "search_ms_per_step": round(r["search_ms"], 3), "host_wait_for_exchange_ms_per_step": round(r["exchange_wait_ms"], 3),
            "host_wait_for_exchange_ms_max": round(r["exchange_wait_ms_max"], 3)}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD `python -m torch.distributed.run` (one
    process per GPU, RCCL rendezvous on 127.0.0.1), relay rank 0's JSON line and the ranks' stderr, return the child's exit
    code.  Called before this process has made any GPU call (importing torch makes none), and the child is a child process,
    not an exec."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    log("launching", " ".join(cmd))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for line in proc.stdout:   # rank 0 prints ONE JSON line; anything else on stdout goes to stderr
        if line.startswith("{"):
            sys.stdout.write(line)
            sys.stdout.flush()
        else:
            sys.stderr.write(line)
    return proc.wait()


def install_watchdog():
    """Tests (CCR_BENCH_WATCHDOG = seconds): a rank that is still running then dumps every thread's Python stack and exits;
    SIGUSR1 dumps the stacks at any time (the parent test sends it before it kills a child that overran its time limit).
    CCR_BENCH_WATCHDOG_DIR: the dumps go to <dir>/stacks.rank<R>.txt instead of stderr, so they survive a killed pipe."""
    secs = int(os.environ.get("CCR_BENCH_WATCHDOG", "0"))
    if secs <= 0:
        return
    import faulthandler
    import signal
    out = sys.stderr
    d = os.environ.get("CCR_BENCH_WATCHDOG_DIR")
    if d:
        out = open(os.path.join(d, f"stacks.rank{os.environ.get('RANK', '0')}.txt"), "w")
    faulthandler.enable(file=out, all_threads=True)
    faulthandler.register(signal.SIGUSR1, file=out, all_threads=True)
    faulthandler.dump_traceback_later(secs, exit=True, file=out)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))   # plain `python bench.py --gpus N`: this process never touches a GPU
    install_watchdog()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        log(f"--gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus {args.gpus}` (it launches its own ranks) "
            f"or through `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}`")
        sys.exit(2)
    if args.same_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

    from ccrec_amd import ops

    w = Workload(args.rows, args.queries, args.dim, args.k, args.data, dev, rank, world, args.dist_backend, normalize=args.sim == "cos")
    log(f"rank {rank}: inputs resident, rows [{w.lo},{w.hi}), data={args.data}")
    r = w.run(args.steps, args.warmup, "main")
    st = r["stats"]

    # untimed extras: pack-kernel HBM rate
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.pack_bf16(w.corpus_f32, out=w.slots[0]["shard"], norm_bounds=w.slots[0]["bounds"], normalize=w.normalize)
    e1.record()
    torch.cuda.synchronize()
    pack_ms = e0.elapsed_time(e1)
    pack_gbs = (w.hi - w.lo) * args.dim * 6 / (pack_ms * 1e-3) / 1e9

    default_shape = (args.rows, args.queries, args.dim, args.k, args.data, args.sim) == (N_ROWS, N_Q, DIM, TOP_K, "gaussian", "dot")
    # HBM traffic of the dominant kernel cannot be read from inside the process (it needs rocprofv3 --pmc passes): the
    # figure below is the offline PMC measurement of THIS command committed under profiles/, labelled as such
    def offline_traffic(tag):
        """(bytes per step, label) of the main pass from the committed PMC measurement of the same workload, or (None, None)."""
        try:
            rec = json.load(open(PMC_SUMMARY))[tag]
            return rec["traffic_bytes"], (f"offline rocprofv3 --pmc passes of this workload (profiles/r06_locality.json[{tag}]: FETCH_SIZE x 2 + "
                                          f"WRITE_SIZE of the main-pass launches of one step); not measured in this run")
        except Exception:
            return None, None

    traffic, traffic_source = offline_traffic("nq_default") if (default_shape and world == 1) else (None, None)

    workload = ("configs[1]: NQ corpus top-100, corpus row-sharded over n_gpus" if default_shape else
                f"custom shape (not the headline config): {args.rows:,} x {args.dim} corpus ({args.data}, {args.sim}), {args.queries:,} queries, "
                f"top-{args.k}, corpus row-sharded over n_gpus")
    out = {
        "metric": ("queries/sec, exhaustive inner-product top-100 retrieval (NQ-shaped 2,681,468 x 768 bf16 corpus)" if default_shape
                   else f"queries/sec, exhaustive inner-product top-{args.k} retrieval ({args.rows:,} x {args.dim} bf16 corpus)"),
        "value": round(r["qps"], 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(r["ms_per_step"], 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "bf16", "data": {"gaussian": "synthetic", "clustered": "synthetic (clustered, log-normal norms, 3% duplicates)",
                                   "sorted": "synthetic (clustered in topical row order, log-normal norms, 3% duplicates)",
                                   "outlier": "synthetic (gaussian, every 500,000th corpus row scaled by CCR_BENCH_OUTLIER)"}[args.data],
        "config": {"workload": workload, "corpus_rows": args.rows,
                   "dim": args.dim, "queries": args.queries, "k": args.k,
                   "step": "pack corpus shard fp32->bf16 + index build + pack queries + fused MFMA score/top-k (asynchronous: the host enqueues "
                           "step i + 1 before it completes step i; the corpus pack of step i + 1 runs on a side stream from the end of step i's main "
                           "pass, beside its select stage)"
                           + (" + RCCL all-gather of the packed shard message + merge (the exchange of step i overlaps the pack and search of step i + 1)" if world > 1 else ""),
                   "parallelism": f"row-shard x{world}"},
        "roofline": roofline_obj(r, traffic, traffic_source),
        "phases_ms": dict(phases_obj(st), corpus_pack=round(pack_ms, 3)),
        "pack_kernel": {"bound": "hbm", "achieved": round(pack_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(pack_gbs / HBM_PEAK_GBS, 4)},
        "search_stats": dict({k_: st[k_] for k_ in ("path", "n_fallback", "sample_tiles", "ranges", "sublists", "cap", "n_candidates", "opt_rank", "n_retried", "n_dense", "main_tile_queries")},
                             candidates_per_query=round(st["n_candidates"] / max(1, args.queries), 1), n_fallback_max=r["n_fallback_max"]),
    }
    if world > 1:
        out["exchange"] = exchange_obj(w, r)
    if args.dump_ids and rank == 0:
        torch.save(w.ids.cpu(), args.dump_ids)
    if rank == 0 and world == 1:
        try:     # every query of the last timed step against an independent sweep (beside the CPU leg's bounded sample)
            out["all_queries_check"] = completeness_all_queries(w.shard, w.qpack, w.scores)
        except Exception as e:
            out["all_queries_check"] = {"skipped": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and args.cpu_queries > 0:
        out["cpu_baseline"] = cpu_baseline(w.shard, w.qpack, min(args.cpu_queries, args.queries), args.k, w.ids)
    elif rank == 0:
        out["cpu_baseline"] = None

    # ---- side runs of the default invocation (short; never part of `value`)
    if world == 1 and default_shape and not args.no_secondary:
        sec = {}
        side_steps, side_warm = min(args.steps, 5), min(args.warmup, 2)
        # (0) SURVEY 8d's small batches on the index the last timed step built
        try:
            w.drain()
            sec["small_batches"] = small_batches_side_run(w.index, w.qpack, args.rows, args.dim)
        except Exception as e:      # a side run must never cost the line its headline
            sec["small_batches"] = {"skipped": f"{type(e).__name__}: {e}"}
        # (a) what ranking() asks for: the same corpus at k = 1001
        w.set_k(1001)
        r2 = w.run(side_steps, max(1, side_warm), "k1001")
        sec["k1001"] = {"workload": "configs[1] corpus, top-1001 (the k of ranking(), ms_marco_eval.py:230)", "value": round(r2["qps"], 1),
                        "unit": "queries/s", "ms_per_step": round(r2["ms_per_step"], 3), "roofline": roofline_obj(r2, *offline_traffic("nq_k1001")),
                        "phases_ms": phases_obj(r2["stats"]),
                        "candidates_per_query": round(r2["stats"]["n_candidates"] / args.queries, 1),
                        "n_fallback": r2["n_fallback_max"], "opt_rank": r2["stats"]["opt_rank"], "launches": r2["stats"]["main_launches"]}
        w.release()
        # (b) the north-star target shape: MS-MARCO scale on ONE GPU, with its own CPU leg on the same host
        m = Workload(MSMARCO_ROWS, MSMARCO_Q, DIM, TOP_K, "gaussian", dev, 0, 1, args.dist_backend)
        r3 = m.run(side_steps, max(1, side_warm), "msmarco")
        ms = {"workload": "configs[2] shape on one GPU: 8,841,823 x 768 corpus, 6,980 queries, top-100", "value": round(r3["qps"], 1),
              "unit": "queries/s", "ms_per_step": round(r3["ms_per_step"], 3), "roofline": roofline_obj(r3, *offline_traffic("msmarco_scale")),
              "phases_ms": phases_obj(r3["stats"]), "n_fallback": r3["n_fallback_max"]}
        if args.cpu_queries > 0:
            cb = cpu_baseline(m.shard, m.qpack, 16, TOP_K, m.ids)
            ms["cpu_baseline"] = cb
            ms["gpu_over_cpu"] = round(r3["qps"] / cb["value"], 1)
            ms["recall_at_100_vs_cpu"] = cb["recall_at_k_of_gpu_vs_cpu"]
        sec["msmarco_scale"] = ms
        m.release()
        sec["inbatch_b1024"] = inbatch_side_run(dev)
        try:
            sec["encode_passages"] = encode_side_run(dev)
        except Exception as e:      # a side run must never cost the line its headline
            sec["encode_passages"] = {"skipped": f"{type(e).__name__}: {e}"}
        try:
            sec["bm25"] = bm25_side_run(dev, cpu_queries=10 if args.cpu_queries > 0 else 0)
        except Exception as e:
            sec["bm25"] = {"skipped": f"{type(e).__name__}: {e}"}
        out["secondary"] = sec
    elif world > 1 and (default_shape or args.rehearse_secondary > 0) and not args.no_secondary:
        # N > 1: the shapes the multi-GPU target is quoted on, each sharded over the same ranks with its own exchange record
        sec = {}
        side_steps, side_warm = min(args.steps, 5), max(1, min(args.warmup, 2))
        # (a) ranking()'s own k on the NQ corpus (ms_marco_eval.py:230): the short-list exchange's shape
        w.set_k(1001)
        r2 = w.run(side_steps, side_warm, "k1001")
        sec["k1001"] = {"workload": "configs[1] corpus, top-1001 (the k of ranking(), ms_marco_eval.py:230), corpus row-sharded over n_gpus",
                        "value": round(r2["qps"], 1), "unit": "queries/s", "ms_per_step": round(r2["ms_per_step"], 3), "scaling": "strong",
                        "roofline": roofline_obj(r2), "phases_ms": phases_obj(r2["stats"]), "n_fallback": r2["n_fallback_max"],
                        "exchange": exchange_obj(w, r2)}
        w.release()
        # (b) configs[2]: MS-MARCO passages, 8,841,823 x 768, 6,980 queries, top-100 -- the shape ">= 6x at 8 GPUs" is quoted on
        div = max(1, args.rehearse_secondary)
        m = Workload(MSMARCO_ROWS // div, MSMARCO_Q // div, DIM, TOP_K, "gaussian", dev, rank, world, args.dist_backend)
        r3 = m.run(side_steps, side_warm, "msmarco")
        sec["msmarco"] = {"workload": "configs[2]: 8,841,823 x 768 corpus row-sharded over n_gpus, 6,980 queries, top-100" + (f" (REHEARSAL: 1/{div} of the rows and queries)" if div > 1 else ""),
                          "value": round(r3["qps"], 1), "unit": "queries/s", "ms_per_step": round(r3["ms_per_step"], 3), "scaling": "strong",
                          "roofline": roofline_obj(r3), "phases_ms": phases_obj(r3["stats"]), "n_fallback": r3["n_fallback_max"],
                          "exchange": exchange_obj(m, r3)}
        m.release()
        # (c) configs[3]: 50 M x 1024, 10 000 queries, top-1000 -- 12.8 GB of bf16 per GPU at 8 (fp32 inputs + two packed slots:
        # 51 GB per rank at 8, 102 GB at 4; not run below 4 ranks, where the resident fp32 inputs alone pass 100 GB per GPU)
        if world >= 4 or args.rehearse_secondary > 0:
            c = Workload(C4_ROWS // div, C4_Q // div, C4_DIM, min(C4_K, C4_ROWS // div // world), "gaussian", dev, rank, world, args.dist_backend)
            r4 = c.run(min(side_steps, 3), 1, "config4")
            sec["config4"] = {"workload": "configs[3]: 50,000,000 x 1024 corpus row-sharded over n_gpus, 10,000 queries, top-1000" + (f" (REHEARSAL: 1/{div} of the rows and queries)" if div > 1 else ""),
                              "value": round(r4["qps"], 1), "unit": "queries/s", "ms_per_step": round(r4["ms_per_step"], 3), "scaling": "strong",
                              "roofline": roofline_obj(r4), "phases_ms": phases_obj(r4["stats"]), "n_fallback": r4["n_fallback_max"],
                              "exchange": exchange_obj(c, r4)}
            c.release()
        out["secondary"] = sec
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
