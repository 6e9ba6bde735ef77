#!/usr/bin/env python3
"""Soak of the 256 x 384 form of the main pass (round 6): random shapes -- 130 .. 4 000 queries, corpora of 20 k .. 1.5 M rows, dims that
are multiples of 32, k from 1 to 2 500 --, clustered corpora in random or topical order, duplicate rows (mass ties), norm outliers,
half of the cases with CCR_WIDE=1 pinned and half on the planner's choice; every case against the exact dense path of the same index,
ids and score bits.
  python tools/soak_wide.py [cases] [seed base]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]
from ccrec_amd import ops  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed_base = int(sys.argv[2]) if len(sys.argv) > 2 else 6000      # python tools/soak_wide.py 100 7000: another hundred cases
dev = "cuda"
used = mism = flagged = 0
for c in range(cases):
    rs = np.random.RandomState(seed_base + c)
    n = int(rs.randint(20_000, 1_500_000))
    d = int(rs.choice([32, 64, 96, 128, 256, 384, 768, 1024]))
    nq = int(rs.choice([rs.randint(130, 400), rs.randint(400, 1600), rs.randint(1600, 4000)]))
    if n * nq > 1.2e9:                                   # the dense check scores n x nq
        n = int(1.2e9 / nq)
    k = int(rs.choice([1, 10, 100, 500, 1001, 2500]))
    k = min(k, n)
    ncl = int(rs.choice([1, 16, 512]))
    g = torch.Generator(device=dev).manual_seed(seed_base + c)
    D = torch.randn(n, d, generator=g, device=dev) * d ** -0.5
    if ncl > 1:
        centres = torch.randn(ncl, d, generator=g, device=dev) * d ** -0.5
        cid = torch.randint(0, ncl, (n,), generator=g, device=dev) if rs.rand() < 0.5 else (torch.arange(n, device=dev) * ncl // n)
        D = 0.8 * centres[cid] + 0.6 * D
    D *= torch.exp(0.3 * torch.randn(n, 1, generator=g, device=dev))
    if rs.rand() < 0.4:                                   # duplicate rows: mass ties
        m = int(rs.randint(10, 3000))
        src = int(rs.randint(0, n))
        D[torch.randint(0, n, (m,), generator=g, device=dev)] = D[src].clone()
    if rs.rand() < 0.3:
        D[int(rs.randint(0, n))] *= 40.0                  # a norm outlier
    Q = torch.randn(nq, d, generator=g, device=dev) * d ** -0.5
    if ncl > 1 and rs.rand() < 0.5:
        Q = 0.8 * centres[torch.randint(0, ncl, (nq,), generator=g, device=dev)] + 0.6 * Q
    Db, Qb = ops.pack_bf16(D), ops.pack_bf16(Q)
    del D, Q
    pin = c % 2 == 0
    old = os.environ.pop("CCR_WIDE", None)
    if pin:
        os.environ["CCR_WIDE"] = "1"
    index = ops.CorpusIndex(Db)
    os.environ.pop("CCR_WIDE", None)
    if old is not None:
        os.environ["CCR_WIDE"] = old
    s, i = index.search(Qb, k, 2)
    st = index.last_stats()
    s1, i1 = index.search(Qb, k, 1)
    ok = torch.equal(i, i1) and torch.equal(s.view(torch.int32), s1.view(torch.int32))
    used += int(st["main_tile_queries"] == 384)
    flagged += int(st["n_fallback"] > 0)
    mism += int(not ok)
    print(f"wide {c}: n={n} d={d} nq={nq} k={k} clusters={ncl} pinned={int(pin)} path={st['path']} tile_q={st['main_tile_queries']} ranges={st['ranges']} "
          f"launches={st['main_launches']} rank={st['opt_rank']} flagged={st['n_fallback']} retried={st['n_retried']} dense={st['n_dense']} "
          f"{'OK' if ok else 'MISMATCH'}", flush=True)
    del index, Db, Qb, s, i, s1, i1
    torch.cuda.empty_cache()
print(f"soak_wide: {cases} cases, {mism} mismatches; 256 x 384 tiles used in {used}; cases with flagged queries {flagged}")
sys.exit(1 if mism else 0)
