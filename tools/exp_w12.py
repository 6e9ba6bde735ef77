#!/usr/bin/env python3
"""Experiment: the twelve-wave (three waves per SIMD) main-pass kernel against the production eight-wave ping-pong.
  1. its raw MFMA scores must equal the production kernel's bit for bit (same K order);
  2. a filter pass that keeps nothing, timed against the production main pass on the same operands."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]
import torch  # noqa: E402
from bench import gen_rows  # noqa: E402
from ccrec_amd import ops, _lib  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.load()


def scores(index, q, mode, out):
    _lib.check(lib.ccr_scores(index._h, ctypes.c_void_p(q.data_ptr()), q.shape[0], mode, ctypes.c_void_p(out.data_ptr()),
                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "ccr_scores")


for (n, nq, d) in [(5000, 300, 768), (40000, 700, 128), (1537, 193, 64)]:
    D = ops.pack_bf16(gen_rows(n, d, 1, dev))
    Q = ops.pack_bf16(gen_rows(nq, d, 2, dev))
    ix = ops.CorpusIndex(D)
    a = ix.scores(Q, "mfma")
    b = torch.empty_like(a)
    scores(ix, Q, 2, b)
    torch.cuda.synchronize()
    print(f"n={n} nq={nq} d={d}: bit-equal {torch.equal(a.view(torch.int32), b.view(torch.int32))}, max diff {(a - b).abs().max().item():.3e}")

rows, nq = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2_681_468, 3_452)
D = ops.pack_bf16(gen_rows(rows, 768, 1234, dev))
Q = ops.pack_bf16(gen_rows(nq, 768, 4321, dev))
ix = ops.CorpusIndex(D)
scratch = torch.empty(64 << 20, dtype=torch.float32, device=dev)
for _ in range(2):
    scores(ix, Q, 3, scratch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    scores(ix, Q, 3, scratch)
torch.cuda.synchronize()
t12 = (time.perf_counter() - t0) / 5 * 1e3
ix.search(Q, 1)
ix.search(Q, 1)
t8 = ix.last_stats()["ms_main"]
fl = 2.0 * rows * nq * 768
print(f"rows={rows} queries={nq}: twelve-wave filter pass (no hits) {t12:.3f} ms = {fl / t12 / 1e9:.0f} TF; "
      f"production main pass at k=1 {t8:.3f} ms = {fl / t8 / 1e9:.0f} TF")
