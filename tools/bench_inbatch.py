#!/usr/bin/env python3
"""In-batch-negative contrastive loss (B = 1024, d = 768): forward + backward time of
ccrec_amd.ops.inbatch_ce (ccr_inbatch_ce_fwd/bwd) next to the reference's torch formulation
(mm, mm, cat, scale, CrossEntropyLoss; src/ccrec/models/bbpr.py:205-212) on the same device.  One JSON line."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    from ccrec_amd import ops
    B, d, T = args.batch, args.dim, 20.0
    g = torch.Generator(device="cuda").manual_seed(0)
    q, p, n = (torch.randn(B, d, device="cuda", generator=g) * d ** -0.5 for _ in range(3))

    def ours():
        a, b, c = (t.clone().requires_grad_(True) for t in (q, p, n))
        loss = ops.inbatch_ce(a, b, c, T)
        loss.backward()
        return loss

    def ref(dtype):
        a, b, c = (t.clone().requires_grad_(True) for t in (q, p, n))
        with torch.autocast("cuda", dtype=dtype, enabled=dtype != torch.float32):
            scores = torch.cat([a @ b.T, a @ c.T], 1) * T
            loss = torch.nn.CrossEntropyLoss()(scores.float(), torch.arange(B, device="cuda"))
        loss.backward()
        return loss

    out = {"batch": B, "dim": d, "flops_fwd_bwd": 2 * 3 * 2 * B * B * d * 2}
    for name, fn in (("ccr_inbatch_ce", ours), ("torch_fp32", lambda: ref(torch.float32)), ("torch_bf16_autocast", lambda: ref(torch.bfloat16))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            loss = fn()
        torch.cuda.synchronize()
        out[name + "_ms"] = round((time.perf_counter() - t0) / args.iters * 1e3, 4)
        out[name + "_loss"] = float(loss)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
