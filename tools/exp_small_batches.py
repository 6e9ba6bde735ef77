"""Search-only time for small query batches on the NQ corpus (Q in {1, 16, 64, 512}): below ~256 queries the main pass
is an HBM stream of the corpus (2 N d bytes per pass)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "crowd-coachable-recommendations_amd"))
from bench import gen_rows  # noqa: E402
from ccrec_amd import ops  # noqa: E402

n, d = (int(sys.argv[1]) if len(sys.argv) > 1 else 2_681_468), 768   # argv[1]: corpus rows (100000 rows = 154 MB sit in the 256-MB Infinity Cache)
nb = torch.empty(n, device="cuda")
D = ops.pack_bf16(gen_rows(n, d, 1234, "cuda"), norm_bounds=nb)
Qall = ops.pack_bf16(gen_rows(1024, d, 4321, "cuda"))
index = ops.CorpusIndex(D, norm_bounds=nb)
for nq in (1, 16, 64, 65, 96, 128, 129, 256, 512, 1024):
    Q = Qall[:nq].contiguous()
    for _ in range(3):
        index.search(Q, 100)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20):
        index.search(Q, 100)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / 20 * 1e3
    st = index.last_stats()
    gbs = 2.0 * n * d / (st["ms_main"] * 1e-3) / 1e9
    tiles = (n + 255) // 256 * ((nq + 255) // 256)
    print(f"n_q {nq:5d}: [{st['ms_main'] * 1e3 / tiles * 256:.1f} us per 256x256 tile per workgroup; ranges {st['ranges']}] {ms:6.3f} ms per search ({nq / ms * 1e3:9.0f} queries/s); main pass {st['ms_main']:.3f} ms = {gbs:6.0f} GB/s of corpus bytes, "
          f"sample {st['ms_sample']:.3f}, thresholds {st['ms_threshold']:.3f}, select {st['ms_select']:.3f}", flush=True)
