#!/usr/bin/env python3
"""Soak of ccr_attention_bf16 / ccr_add_layernorm / ccr_embed_layernorm over random shapes against fp32 torch (the per-seed body of
tests/test_gpu_encoder_kernels.py::test_attention_fuzz_against_fp32_reference, many more seeds, every max_len 1..512 reachable).
  python tools/soak_attention.py [--seeds 400]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]
from ccrec_amd import ops


def reference(qkv, starts, lens, H):
    T = qkv.shape[0]
    out = torch.zeros(T, H * 64, dtype=torch.float32, device=qkv.device)
    x = qkv.float().view(T, 3, H, 64)
    for s0, n in zip(starts, lens):
        q, k, v = (x[s0:s0 + n, i].transpose(0, 1) for i in range(3))
        out[s0:s0 + n] = (torch.softmax(q @ k.transpose(1, 2) * 0.125, dim=-1) @ v).transpose(0, 1).reshape(n, H * 64)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=400)
    args = ap.parse_args()
    worst = 0.0
    for seed in range(args.seeds):
        rs = np.random.RandomState(seed)
        H = int(rs.choice([1, 2, 3, 12, 16]))
        n = int(rs.randint(1, 30))
        top = int(rs.randint(1, 513))
        lens = [int(v) for v in rs.randint(1, top + 1, n)]
        if rs.rand() < 0.3:
            lens[int(rs.randint(0, n))] = top
        L = max(lens)
        padded = bool(rs.randint(0, 2))
        if padded:
            pad = min(512, L + int(rs.randint(0, 9)))
            starts, T = [i * pad for i in range(n)], pad * n
        else:
            pad, starts, T = 0, [int(v) for v in np.cumsum([0] + lens[:-1])], sum(lens)
        torch.manual_seed(seed)
        qkv = (torch.randn(T, 3 * H * 64, device="cuda") * float(rs.choice([0.3, 1.0, 2.5]))).to(torch.bfloat16)
        out = torch.full((T, H * 64), -3.0, dtype=torch.bfloat16, device="cuda")
        ops.attention(qkv, torch.tensor(starts, dtype=torch.int32, device="cuda"), torch.tensor(lens, dtype=torch.int32, device="cuda"),
                      H, max_len=L, pad_len=pad, out=out)
        ref = reference(qkv, starts, lens, H)
        got = out.float()
        for s0, m in zip(starts, lens):
            err = (got[s0:s0 + m] - ref[s0:s0 + m]).abs()
            tol = 1.5e-2 + 1.6e-2 * ref[s0:s0 + m].abs()
            assert (err <= tol).all(), (seed, H, lens, padded, float(err.max()))
            worst = max(worst, float(err.max()))
            if padded:
                assert (got[s0 + m:s0 + pad] == 0).all(), (seed, "padding rows")
        # LayerNorm kernels on a random row count of the same seed
        rows, dim = int(rs.randint(1, 3000)), int(rs.choice([256, 768, 1024]))
        x = torch.randn(rows, dim, device="cuda").to(torch.bfloat16)
        res = torch.randn(rows, dim, device="cuda")
        g, b = torch.rand(dim, device="cuda") + 0.5, torch.randn(dim, device="cuda")
        f32, b16 = ops.add_layernorm(x, res, g, b, 1e-12)
        torch.testing.assert_close(f32, torch.nn.functional.layer_norm(x.float() + res, (dim,), g, b, 1e-12), atol=3e-5, rtol=3e-5)
        assert torch.equal(b16, f32.to(torch.bfloat16))
        if seed % 50 == 49:
            print(f"{seed + 1} seeds ok, worst attention |error| {worst:.4f}", flush=True)
    print(f"soak_attention: {args.seeds} seeds ok, worst attention |error| {worst:.4f}")


if __name__ == "__main__":
    main()
