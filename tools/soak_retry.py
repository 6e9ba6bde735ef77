#!/usr/bin/env python3
"""Soak of the flagged-query paths (retry rounds, dense last resort): topically sorted corpora with cluster sizes from 300 to
60,000 rows, k from 1 to 4096, random shapes; the fused result of a query subset must equal the exact dense path bit for bit.
  python tools/soak_retry.py [cases]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]
from ccrec_amd import ops  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = "cuda"
tot = {"flagged": 0, "retried": 0, "dense": 0}
for c in range(cases):
    rs = np.random.RandomState(1000 + c)
    n = int(rs.randint(150_000, 900_000))
    d = int(rs.choice([64, 128, 768]))
    nq = int(rs.choice([40, 300, 1100]))
    k = int(rs.choice([1, 10, 100, 500, 1001, 2500, 4096]))
    ncl = int(rs.choice([8, 64, 512, 2048]))                 # clusters: 8 -> ~60,000-row clusters
    g = torch.Generator(device=dev).manual_seed(c)
    centres = torch.randn(ncl, d, generator=g, device=dev) * d ** -0.5
    cid = (torch.arange(n, device=dev) * ncl // n).clamp_(max=ncl - 1)     # contiguous clusters (topical order)
    spread = float(rs.choice([0.3, 0.6, 1.0]))
    D = 0.8 * centres[cid] + spread * torch.randn(n, d, generator=g, device=dev) * d ** -0.5
    D *= torch.exp(0.35 * torch.randn(n, 1, generator=g, device=dev))
    qc = torch.randint(0, ncl, (nq,), generator=g, device=dev)
    Q = 0.8 * centres[qc] + spread * torch.randn(nq, d, generator=g, device=dev) * d ** -0.5
    Db, Qb = ops.pack_bf16(D), ops.pack_bf16(Q)
    del D, Q
    index = ops.CorpusIndex(Db)
    s, i = index.search(Qb, k, 2)
    st = index.last_stats()
    pick = torch.arange(0, nq, max(1, nq // 24), device=dev)[:24]
    s1, i1 = index.search(Qb[pick], k, 1)
    ok = torch.equal(i[pick], i1) and torch.equal(s[pick].view(torch.int32), s1.view(torch.int32))
    tot["flagged"] += st["n_fallback"]
    tot["retried"] += st["n_retried"]
    tot["dense"] += st["n_dense"]
    print(f"case {c}: n={n} d={d} nq={nq} k={k} clusters={ncl} spread={spread} path={st['path']} flagged={st['n_fallback']} "
          f"retried={st['n_retried']} dense={st['n_dense']} ms_fallback={st['ms_fallback']:.1f} {'OK' if ok else 'MISMATCH'}", flush=True)
    assert ok
print("all cases equal the exact dense path;", tot)
