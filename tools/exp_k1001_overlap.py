#!/usr/bin/env python3
"""Go / no-go for running the k = 1001 select of step i beside the main pass of step i + 1.  The select is inside
ccr_search, so the cheapest faithful experiment is two independent search pipelines over the same packed shard (their own indexes and
workspaces) on TWO HIP streams, offset by half a step: the select / re-score of one then runs while the other is in its main pass.
If the pair completes 2 n searches sooner than one stream does, moving the select to a side stream inside the library would pay.

  python tools/exp_k1001_overlap.py [rows] [queries] [k]"""
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
k = int(sys.argv[3]) if len(sys.argv) > 3 else 1001
dev = torch.device("cuda", 0)
nb = torch.empty(rows, device=dev)
shard = ops.pack_bf16(gen_rows(rows, 768, 1234, dev), norm_bounds=nb)
q = ops.pack_bf16(gen_rows(nq, 768, 4321, dev))
idx = [ops.CorpusIndex(shard, norm_bounds=nb) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
n = 10


def serial():
    with torch.cuda.stream(streams[0]):
        for j in range(2 * n):
            ix = idx[j % 2]
            ix.search(q, k, defer=True)
            if j:
                idx[(j - 1) % 2].finish()
        idx[(2 * n - 1) % 2].finish()


def two_streams():
    for j in range(2 * n):
        ix = idx[j % 2]
        if j >= 2:
            ix.finish()                      # this pipeline's previous search (its own event)
        with torch.cuda.stream(streams[j % 2]):
            ix.search(q, k, defer=True)
    for ix in idx:
        ix.finish()


for name, fn in (("one stream", serial), ("two streams", two_streams), ("one stream", serial), ("two streams", two_streams)):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (2 * n) * 1e3
    st = idx[0].last_stats()
    print(f"{name:12s}: {dt:7.3f} ms per search  (phases of the last: main {st['ms_main']:.2f}, select {st['ms_select']:.2f}, total {st['ms_total']:.2f})", flush=True)
