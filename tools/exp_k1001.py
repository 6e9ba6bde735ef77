#!/usr/bin/env python3
"""k = 1001 on the NQ shape: conservative thresholds (CCR_OPTIMISTIC=0) against estimated ones at several ranks, same box,
same process: step phases from the library's HIP events.  python tools/exp_k1001.py [k]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]
from bench import gen_rows  # noqa: E402
from ccrec_amd import ops  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 1001
n, d, nq = 2_681_468, 768, 3_452
nb = torch.empty(n, device="cuda")
D = ops.pack_bf16(gen_rows(n, d, 1234, "cuda"), norm_bounds=nb)
Q = ops.pack_bf16(gen_rows(nq, d, 4321, "cuda"))
ref = None
for name, env in [("conservative", {"CCR_OPTIMISTIC": "0"}), ("estimated (planner)", {}), ("estimated rank 32", {"CCR_OPT_RANK": "32"}),
                  ("estimated rank 64", {"CCR_OPT_RANK": "64"}), ("estimated rank 96, 1/32 sample", {"CCR_OPT_RANK": "96", "CCR_SAMPLE_DIV": "32"}),
                  ("estimated rank 24, 1/128 sample", {"CCR_OPT_RANK": "24", "CCR_SAMPLE_DIV": "128"})]:
    for v in ("CCR_OPTIMISTIC", "CCR_OPT_RANK", "CCR_SAMPLE_DIV"):
        os.environ.pop(v, None)
    os.environ.update(env)
    index = ops.CorpusIndex(D, norm_bounds=nb)
    acc = {}
    for it in range(6):
        s, i = index.search(Q, k)
        st = index.last_stats()
        if it >= 2:
            for f in ("ms_sample", "ms_threshold", "ms_main", "ms_select", "ms_total", "ms_fallback"):
                acc[f] = acc.get(f, 0.0) + st[f] / 4
    if ref is None:
        ref = (s, i)
    same = torch.equal(i, ref[1]) and torch.equal(s.view(torch.int32), ref[0].view(torch.int32))
    print(f"{name:34s} rank {st['opt_rank']:3d} sample {st['sample_tiles']:4d} launches {st['main_launches']} cand/q {st['n_candidates'] / nq:7.0f} flagged {st['n_fallback']:3d} | "
          + " ".join(f"{f[3:]} {v:.3f}" for f, v in acc.items()) + f" | identical to conservative: {same}", flush=True)
