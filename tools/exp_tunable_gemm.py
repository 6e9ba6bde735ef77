#!/usr/bin/env python3
"""Go / no-go for tuning the encoder's library GEMMs: the four projections of a BERT-base layer at a packed batch of M tokens
(F.linear with bias, 16-bit), torch's default hipBLASLt heuristic against torch's TunableOp pick of the same libraries' solutions.

  python tools/exp_tunable_gemm.py [--m 65536] [--dtype fp16]"""
import argparse
import time

import torch
import torch.nn.functional as F


def bench(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=65536)
    ap.add_argument("--dtype", default="fp16")
    args = ap.parse_args()
    dt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    shapes = [("qkv", 2304, 768), ("attn_out", 768, 768), ("ffn_in", 3072, 768), ("ffn_out", 768, 3072)]
    g = torch.Generator(device="cuda").manual_seed(0)
    ops = []
    for name, n, k in shapes:
        x = torch.randn(args.m, k, device="cuda", generator=g).to(dt)
        w = (torch.randn(n, k, device="cuda", generator=g) * 0.03).to(dt)
        b = torch.randn(n, device="cuda", generator=g).to(dt)
        ops.append((name, n, k, x, w, b))
    base = {}
    for name, n, k, x, w, b in ops:
        t = bench(lambda: F.linear(x, w, b))
        base[name] = t
        print("default  %-9s M=%d N=%d K=%d: %7.1f us = %6.1f TFLOP/s" % (name, args.m, n, k, t * 1e6, 2 * args.m * n * k / t / 1e12), flush=True)
    import torch.cuda.tunable as tun
    tun.enable(True)
    tun.tuning_enable(True)
    tun.set_max_tuning_duration(30)
    tun.set_max_tuning_iterations(20)
    for name, n, k, x, w, b in ops:
        t0 = time.perf_counter()
        F.linear(x, w, b)
        torch.cuda.synchronize()
        tune_s = time.perf_counter() - t0
        t = bench(lambda: F.linear(x, w, b))
        print("tunable  %-9s: %7.1f us = %6.1f TFLOP/s (%.2fx, tuning took %.1f s)" % (name, t * 1e6, 2 * args.m * n * k / t / 1e12, base[name] / t, tune_s), flush=True)


if __name__ == "__main__":
    main()
