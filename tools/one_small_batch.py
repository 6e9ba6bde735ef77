#!/usr/bin/env python3
"""A few searches of n_q queries (argv[1], default 256) against the NQ-sized corpus: the target of rocprofv3 --pmc passes for the
single-query-block question (ridge-region batches)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]
from bench import gen_rows  # noqa: E402
from ccrec_amd import ops  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n, d = 2_681_468, 768
nb = torch.empty(n, device="cuda")
D = ops.pack_bf16(gen_rows(n, d, 1234, "cuda"), norm_bounds=nb)
Q = ops.pack_bf16(gen_rows(1024, d, 4321, "cuda"))[:nq].contiguous()
index = ops.CorpusIndex(D, norm_bounds=nb)
for _ in range(4):
    index.search(Q, 100)
torch.cuda.synchronize()
print("searches 4")
print(index.last_stats())
