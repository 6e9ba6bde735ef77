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

# small corpus (prime_pantry shape: 9,862 x 9,862, keep 1001): the default search is the margin path, flag 1 the fp64 path
Ds = ops.pack_bf16(gen_rows(9862, 768, 7, "cuda"))
ix = ops.CorpusIndex(Ds)
for flag, name in ((0, "default (MFMA rows + margin select)"), (1, "forced fp64 dense")):
    t, _ = timed(ix, Ds, flag) if False else (None, None)
    ix.search(Ds, 1001, flag)
    torch.cuda.synchronize()
    t0 = time.time()
    s, i = ix.search(Ds, 1001, flag)
    torch.cuda.synchronize()
    print("prime_pantry shape", name, round((time.time() - t0) * 1e3, 2), "ms", ix.last_stats()["n_dense"], flush=True)

# the realistic fallback: a handful of queries with a few hundred rows tied at their cut (more than rescore_cap = 256, fewer than 8,192)
Dm = D.clone()
Dm[1000:1400] = Dm[1000]
im = ops.CorpusIndex(Dm)
Qm = ops.pack_bf16(gen_rows(3452, 768, 4321, "cuda"))
Qm[10:15] = Dm[1000]
for _ in range(2):
    torch.cuda.synchronize()
    t0 = time.time()
    s, i = im.search(Qm, 100)
    torch.cuda.synchronize()
    st = im.last_stats()
    print("3,452 queries, 5 of them with 400 rows tied at the cut:", round((time.time() - t0) * 1e3, 2), "ms; fallback", round(st["ms_fallback"], 2),
          "ms; flagged", st["n_fallback"], "dense", st["n_dense"], flush=True)
s1, i1 = im.search(Qm[8:17], 100, 1)
assert torch.equal(i[8:17], i1) and torch.equal(s[8:17], s1)
