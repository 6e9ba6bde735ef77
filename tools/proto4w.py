#!/usr/bin/env python3
"""Times the four-wave main-pass experiments (tools/proto4w.sh builds them) on an NQ-shaped problem: the skeleton only
(LDS-DMA ring + operand reads + MFMAs, no filter, no output).
  python tools/proto4w.py                      256x256 tile, production work-item mapping (88 / 144 ranges, 2 query groups)
  PROTO=gemm4w384_proto python tools/proto4w.py   256x384 tile (rows a multiple of 256, 3 456 queries, 256 ranges)
  PROTO_MODE=1|2                                DMA-only builds of the 256x256 experiment"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]


def main():
    from ccrec_amd import ops
    name = os.environ.get("PROTO", "gemm4w_proto")
    lib = ctypes.CDLL(os.path.join(ROOT, "crowd-coachable-recommendations_amd", "lib",
                                   "libproto_" + name + os.environ.get("PROTO_MODE", "") + ".so"))
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    lib.ccr_proto4w_time.argtypes = [vp, i64, i32, vp, i32, i32, i32, i32, ctypes.POINTER(ctypes.c_float)]
    wide = name == "gemm4w384_proto"
    n, nq, d = (2681344, 3456, 768) if wide else (2681468, 3452, 768)
    g = torch.Generator(device="cuda").manual_seed(7)
    D = ops.pack_bf16(torch.randn(n, d, device="cuda", generator=g) / d ** 0.5)
    Q = ops.pack_bf16(torch.randn(nq, d, device="cuda", generator=g) / d ** 0.5)
    torch.cuda.synchronize()
    for ranges in ((256, 128) if wide else (88, 144)):
        ms = ctypes.c_float(0)
        rc = lib.ccr_proto4w_time(D.data_ptr(), n, d, Q.data_ptr(), nq, ranges, 2, 10, ctypes.byref(ms))
        flops = 2.0 * nq * n * d
        print(f"{name} ranges {ranges}: rc {rc}  {ms.value:.3f} ms per launch = {flops / ms.value / 1e9:.0f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
