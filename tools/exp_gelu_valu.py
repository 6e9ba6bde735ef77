#!/usr/bin/env python3
"""Go / no-go input for fusing BERT's exact (erf) GELU into the FFN-in GEMM's epilogue: how much VALU time does the erf
cost by itself?  ccr_gelu_half on arrays that sit in the L2 / Infinity Cache (no HBM traffic: the VALU-bound rate) against the
FFN-in shape (58 K tokens x 3 072: HBM-bound), and the library GEMM of that shape with and without its bias epilogue."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]
from ccrec_amd import ops  # noqa: E402


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


T, H, F = 58_000, 768, 3072
for dt in (torch.bfloat16, torch.float16):
    small = (torch.randn(2048, 3072, device="cuda") * 2).to(dt)        # 12.6 MB: L2 / Infinity-Cache resident
    big = (torch.randn(T, F, device="cuda") * 2).to(dt)                # 356 MB: streams from HBM
    us_small = timed(lambda: ops.gelu_(small))
    us_big = timed(lambda: ops.gelu_(big))
    n_small, n_big = small.numel(), big.numel()
    print(f"{str(dt):15s} gelu on {n_small / 1e6:.1f} M cached elements: {us_small:7.1f} us = {n_small / us_small / 1e6:.2f} T erf/s (VALU-bound rate); "
          f"on the FFN-in shape ({n_big / 1e6:.0f} M): {us_big:7.1f} us = {n_big * 4 / us_big / 1e6:.2f} TB/s read + written; "
          f"erf VALU time alone at the cached rate for the FFN-in shape: {n_big / (n_small / us_small):6.1f} us")
    x = torch.randn(T, H, device="cuda").to(dt)
    w = torch.randn(F, H, device="cuda").to(dt) * 0.02
    b = torch.randn(F, device="cuda").to(dt)
    us_gemm = timed(lambda: torch.nn.functional.linear(x, w, b), 20)
    flops = 2.0 * T * H * F
    print(f"{'':15s} library GEMM + bias [{T} x {H}] x [{H} x {F}]: {us_gemm:7.1f} us = {flops / us_gemm / 1e6:.0f} TFLOP/s; GEMM + separate gelu pass: {us_gemm + us_big:7.1f} us")
