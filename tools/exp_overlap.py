#!/usr/bin/env python3
"""Experiment: how much does a concurrent fp32->bf16 corpus pack (HBM stream on a side HIP stream) slow the fused search
down, and what does the pair cost when overlapped?  Decides whether a pipelined pack + search entry point pays.
  python tools/exp_overlap.py [rows] [queries]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]
import torch  # noqa: E402
from bench import gen_rows  # noqa: E402
from ccrec_amd import ops  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_681_468
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 3_452
dev = torch.device("cuda", 0)
src = gen_rows(rows, 768, 1234, dev)
q = ops.pack_bf16(gen_rows(nq, 768, 4321, dev))
shard = torch.empty(rows, 768, dtype=torch.bfloat16, device=dev)
other = torch.empty_like(shard)
mx = torch.empty(rows, device=dev)   # norm bound per packed row
ops.pack_bf16(src, out=shard, norm_bounds=mx)
index = ops.CorpusIndex(shard, norm_bounds=mx)
side = torch.cuda.Stream()


def timed(fn, n=8):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def search_only():
    index.search(q, 100)


def pack_only():
    ops.pack_bf16(src, out=other, norm_bounds=mx)


def both_serial():
    ops.pack_bf16(src, out=other, norm_bounds=mx)
    index.search(q, 100)


def both_overlapped():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.pack_bf16(src, out=other, norm_bounds=mx)
    index.search(q, 100, defer=True)
    torch.cuda.current_stream().wait_stream(side)
    index.finish()


def both_overlapped_search_first():
    index.search(q, 100, defer=True)
    side.wait_stream(torch.cuda.current_stream()) if False else None
    with torch.cuda.stream(side):
        ops.pack_bf16(src, out=other, norm_bounds=mx)
    torch.cuda.current_stream().wait_stream(side)
    index.finish()


a, b, c, d = timed(search_only), timed(pack_only), timed(both_serial), timed(both_overlapped)
e = timed(both_overlapped_search_first)
print(f"overlapped with the search launched first: {e:.3f} ms")
index.search(q, 100)
print(f"rows={rows} queries={nq}: search {a:.3f} ms, pack {b:.3f} ms, serial {c:.3f} ms, overlapped {d:.3f} ms "
      f"(ideal max = {max(a, b):.3f}); main pass alone {index.last_stats()['ms_main']:.3f} ms")
