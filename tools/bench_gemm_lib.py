#!/usr/bin/env python3
"""Reference point for the main pass: the vendor library's plain bf16 GEMM (torch.mm -> hipBLASLt / rocBLAS) on the same
operands, Q [3452 x 768] x D^T in corpus chunks, scores written to HBM as bf16 and NOT ranked -- i.e. only the first half of
what gemm_topk16_kernel does.  Random data (the clock the chip holds depends on it)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]


def main():
    n, nq, d = 2681468, 3452, 768
    g = torch.Generator(device="cuda").manual_seed(7)
    D = (torch.randn(n, d, device="cuda", generator=g) / d ** 0.5).to(torch.bfloat16)
    Q = (torch.randn(nq, d, device="cuda", generator=g) / d ** 0.5).to(torch.bfloat16)
    flops = 2.0 * nq * n * d
    for chunk in (65536, 262144, 1048576):
        outs = torch.empty(nq, chunk, dtype=torch.bfloat16, device="cuda")
        def one_pass():
            for lo in range(0, n, chunk):
                hi = min(n, lo + chunk)
                torch.mm(Q, D[lo:hi].T, out=outs[:, :hi - lo])
        for _ in range(2):
            one_pass()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            one_pass()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        print(f"torch.mm bf16, corpus chunks of {chunk}: {ms:.2f} ms per corpus pass = {flops / ms / 1e9:.0f} TFLOP/s "
              f"(+ {nq * n * 2 / 1e9:.1f} GB of scores written, unranked)", flush=True)

    # the obvious PyTorch-on-GPU formulation of the whole search: chunked mm (fp32 scores) + topk per chunk + topk of the partial lists
    k = 100
    for chunk in (262144, 1048576):
        scores = torch.empty(nq, chunk, dtype=torch.float32, device="cuda")
        Df = None
        def search():
            parts_s, parts_i = [], []
            for lo in range(0, n, chunk):
                hi = min(n, lo + chunk)
                sc = scores[:, :hi - lo]
                torch.mm(Q, D[lo:hi].T, out=sc) if sc.dtype == torch.bfloat16 else sc.copy_(torch.mm(Q, D[lo:hi].T))
                s_, i_ = sc.topk(k, dim=1)
                parts_s.append(s_)
                parts_i.append(i_ + lo)
            s_all, i_all = torch.cat(parts_s, 1), torch.cat(parts_i, 1)
            top, pos = s_all.topk(k, dim=1)
            return top, i_all.gather(1, pos)
        search()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            search()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        print(f"torch.mm + torch.topk({k}) in corpus chunks of {chunk}: {ms:.1f} ms per search of {nq} queries = {nq / ms * 1e3:.0f} queries/s "
              f"(bf16 GEMM scores: not the canonical order)", flush=True)


if __name__ == "__main__":
    main()
