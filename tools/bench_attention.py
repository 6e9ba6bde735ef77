#!/usr/bin/env python3
"""Timing of the encoder layer's non-GEMM kernels (csrc/ccr_encoder.hip) against the torch ops they replace, on the batch shapes the
length-sorted encoder produces (token budget 65 536): ccr_attention_bf16 vs scaled_dot_product_attention with a key-padding mask,
ccr_add_layernorm vs add + layer_norm + cast.  Prints one line per shape."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]
from ccrec_amd import _lib, ops

if len(sys.argv) > 2 and sys.argv[1] == "--lib":   # a timing-only ablation build of the library (tools only; never the shipped one)
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    H = 12
    for B, L, lo in ((482, 136, 129), (1365, 48, 41), (327, 200, 193), (2730, 24, 17), (128, 512, 505)):
        torch.manual_seed(0)
        lens = torch.randint(lo, L + 1, (B,), dtype=torch.int32, device="cuda")
        qkv = torch.randn(B * L, 3 * H * 64, device="cuda").to(torch.bfloat16)
        start = torch.arange(B, dtype=torch.int32, device="cuda") * L
        out = torch.empty(B * L, H * 64, dtype=torch.bfloat16, device="cuda")
        t_ours = timed(lambda: ops.attention(qkv, start, lens, H, max_len=L, pad_len=L, out=out))
        q, k, v = (qkv.view(B, L, 3, H, 64)[:, :, i].transpose(1, 2) for i in range(3))
        mask = (torch.arange(L, device="cuda")[None, :] < lens[:, None])[:, None, None, :]
        t_sdpa = timed(lambda: F.scaled_dot_product_attention(q, k, v, attn_mask=mask))
        flops = 4.0 * float((lens.double() ** 2).sum()) * 64 * H
        x = torch.randn(B * L, 768, device="cuda").to(torch.bfloat16)
        res = torch.randn(B * L, 768, device="cuda")
        g, b = torch.ones(768, device="cuda"), torch.zeros(768, device="cuda")
        t_ln = timed(lambda: ops.add_layernorm(x, res, g, b, 1e-12))
        t_ln_torch = timed(lambda: F.layer_norm(x + res, (768,), g, b, 1e-12).to(torch.bfloat16))
        print(f"B {B:5d} L {L:4d}: attention {t_ours * 1e3:7.1f} us ({flops / t_ours / 1e9:6.1f} TFLOP/s)  torch sdpa+mask {t_sdpa * 1e3:7.1f} us"
              f"   add+LayerNorm(+bf16 copy) {t_ln * 1e3:6.1f} us ({B * L * 768 * 12 / t_ln / 1e9:5.2f} TB/s)  torch add, layer_norm, cast {t_ln_torch * 1e3:6.1f} us",
              flush=True)


if __name__ == "__main__":
    main()
