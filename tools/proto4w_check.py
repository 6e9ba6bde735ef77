#!/usr/bin/env python3
"""Correctness check of the 256x384 four-wave experiment's data path (PROTO_MODE=3 build: `PROTO_MODE=3 bash tools/proto4w.sh`):
LDS-DMA piece mapping, swizzled fragment reads, asm MFMAs on asm-owned AGPR / compiler VGPR accumulators and the AGPR read-back
must reproduce the production kernel's raw MFMA scores bit for bit (same K order: two K halves per 32-element sub-stage)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]


def main():
    from ccrec_amd import ops
    lib = ctypes.CDLL(os.path.join(ROOT, "crowd-coachable-recommendations_amd", "lib", "libproto_gemm4w384_proto3.so"))
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    lib.ccr_proto4w_scores.argtypes = [vp, i64, i32, vp, i32, i32, vp]
    g = torch.Generator(device="cuda").manual_seed(11)
    bad = 0
    for n, nq, d, ranges in ((2048, 768, 128, 8), (5120, 384, 768, 8), (256 * 37, 1152, 256, 16)):
        D = ops.pack_bf16(torch.randn(n, d, device="cuda", generator=g) / d ** 0.5)
        Q = ops.pack_bf16(torch.randn(nq, d, device="cuda", generator=g) / d ** 0.5)
        out = torch.full((nq, n), float("nan"), device="cuda")
        rc = lib.ccr_proto4w_scores(D.data_ptr(), n, d, Q.data_ptr(), nq, ranges, out.data_ptr())
        ref = ops.CorpusIndex(D).debug_scores(Q, canonical=False)
        same = torch.equal(out.view(torch.int32), ref.view(torch.int32))
        close = torch.allclose(out, ref, rtol=0, atol=1e-5)
        print(f"n={n} nq={nq} d={d} ranges={ranges}: rc={rc} bit-identical={same} allclose={close} nan={int(torch.isnan(out).sum())} "
              f"max|diff|={float((out - ref).abs().nan_to_num(9).max()):.3g}", flush=True)
        bad += 0 if same else 1
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
