"""Exact dense path timing at the NQ shape: gaussian queries vs mass ties, forced dense vs fallback of the fused path."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "crowd-coachable-recommendations_amd"))
from bench import gen_rows  # noqa: E402
from ccrec_amd import ops  # noqa: E402

D = ops.pack_bf16(gen_rows(2681468, 768, 1234, "cuda"))
Qg = ops.pack_bf16(gen_rows(256, 768, 4321, "cuda"))
Dt = D.clone()
Dt[:9000] = Dt[0]
Qt = Dt[0:1].expand(256, 768).contiguous()


def timed(ix, Q, flag):
    ix.search(Q, 100, flag)
    torch.cuda.synchronize()
    t = time.time()
    ix.search(Q, 100, flag)
    torch.cuda.synchronize()
    return round((time.time() - t) * 1e3, 1), ix.last_stats()["n_dense"] if flag != 1 else None


ig, it = ops.CorpusIndex(D), ops.CorpusIndex(Dt)
for n in (16, 64, 200):
    print(n, "gaussian forced dense", timed(ig, Qg[:n], 1), "| tie corpus, gaussian queries, forced dense", timed(it, Qg[:n], 1),
          "| tie queries forced dense", timed(it, Qt[:n], 1), "| tie queries via fused fallback", timed(it, Qt[:n], 2), flush=True)
