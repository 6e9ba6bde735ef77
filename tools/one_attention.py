#!/usr/bin/env python3
"""One shape of ccr_attention_bf16, a few launches: the target of a rocprofv3 --pmc pass (tools/bench_attention.py times it)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]
from ccrec_amd import ops

B, L, lo, H = (int(v) for v in (sys.argv[1:5] + ["482", "136", "129", "12"][len(sys.argv) - 1:]))
DTYPE = torch.float16 if os.environ.get("ATT_DTYPE", "bf16") == "fp16" else torch.bfloat16   # fp16: the layer type under the reference's autocast
torch.manual_seed(0)
lens = torch.randint(lo, L + 1, (B,), dtype=torch.int32, device="cuda")
qkv = torch.randn(B * L, 3 * H * 64, device="cuda").to(DTYPE)
start = torch.arange(B, dtype=torch.int32, device="cuda") * L
out = torch.empty(B * L, H * 64, dtype=DTYPE, device="cuda")
for _ in range(5):
    ops.attention(qkv, start, lens, H, max_len=L, pad_len=L, out=out)
torch.cuda.synchronize()
