#!/usr/bin/env python3
"""The encoder layer's four projections on the vendor library, with and without the bias epilogue, at three token counts
(go / no-go for a hand-written projection GEMM: the library already holds 1.1 - 1.4 PFLOP/s on these shapes).

  python tools/exp_linear_lib.py"""
import time, torch, torch.nn.functional as F
def bench(fn, iters=40):
    for _ in range(8): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/iters
g=torch.Generator(device="cuda").manual_seed(0)
for dt in (torch.float16, torch.bfloat16):
  for m in (65536, 61000, 32768):
    for name,n,k in [("qkv",2304,768),("attn_out",768,768),("ffn_in",3072,768),("ffn_out",768,3072)]:
        x=torch.randn(m,k,device="cuda",generator=g).to(dt); w=(torch.randn(n,k,device="cuda",generator=g)*0.03).to(dt); b=torch.randn(n,device="cuda",generator=g).to(dt)
        out=torch.empty(m,n,device="cuda",dtype=dt)
        t1=bench(lambda: F.linear(x,w,b)); t2=bench(lambda: F.linear(x,w)); t3=bench(lambda: torch.mm(x,w.t(),out=out))
        fl=2*m*n*k
        print(f"{str(dt)[6:]:9s} M={m} {name:9s}: bias {t1*1e6:7.1f} us {fl/t1/1e12:6.0f} TF | no bias {t2*1e6:7.1f} us {fl/t2/1e12:6.0f} TF | mm out= {t3*1e6:7.1f} us", flush=True)
