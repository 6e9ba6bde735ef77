#!/usr/bin/env python3
"""A few launches of ccr_add_layernorm / ccr_embed_layernorm on one encoder batch shape: the target of a rocprofv3 --pmc pass."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]
from ccrec_amd import ops

T, d = 65536, 768
torch.manual_seed(0)
x = torch.randn(T, d, device="cuda").to(torch.bfloat16)
res = torch.randn(T, d, device="cuda")
g, b = torch.ones(d, device="cuda"), torch.zeros(d, device="cuda")
word, pos, typ = torch.randn(30522, d, device="cuda"), torch.randn(512, d, device="cuda"), torch.randn(2, d, device="cuda")
ids = torch.randint(0, 30522, (T,), device="cuda")
ps = torch.randint(0, 200, (T,), device="cuda")
for _ in range(5):
    ops.add_layernorm(x, res, g, b, 1e-12)
    ops.embed_layernorm(word, pos, typ, ids, ps, None, g, b, 1e-12)
torch.cuda.synchronize()
