#!/usr/bin/env python3
"""Soak of the estimated-threshold mode: random shapes with large k on iid, clustered and topically sorted corpora,
with the planner's rank and with ranks pinned low enough that many queries fail the select stage's check (or pass fewer than k
rows) and are retried; every result must equal the exact dense path bit for bit.   python tools/soak_estimated.py [cases]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]
from bench import gen_rows  # noqa: E402
from ccrec_amd import ops  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 36
tot = {"estimated": 0, "flagged": 0, "retried": 0, "dense": 0}
for c in range(cases):
    rs = np.random.RandomState(5000 + c)
    n = int(rs.randint(200_000, 1_200_000))
    d = int(rs.choice([64, 128, 200]))
    nq = int(rs.choice([40, 300, 700]))
    k = int(rs.choice([300, 600, 1001, 2500]))
    data = str(rs.choice(["gaussian", "clustered", "sorted"]))
    rank = int(rs.choice([0, 0, 4, 20]))
    for v in ("CCR_OPT_RANK", "CCR_OPTIMISTIC"):
        os.environ.pop(v, None)
    if rank:
        os.environ["CCR_OPT_RANK"] = str(rank)
        os.environ["CCR_OPTIMISTIC"] = "1"
    D = ops.pack_bf16(gen_rows(n, d, 1234 + c, "cuda", data))
    Q = ops.pack_bf16(gen_rows(nq, d, 4321 + c, "cuda", data))
    index = ops.CorpusIndex(D, global_row_offset=int(rs.choice([0, 1 << 35])))
    s, i = index.search(Q, k)
    st = index.last_stats()
    pick = torch.from_numpy(rs.permutation(nq)[:24]).cuda()
    s1, i1 = index.search(Q[pick], k, 1)
    ok = torch.equal(i[pick], i1) and torch.equal(s[pick].view(torch.int32), s1.view(torch.int32))
    tot["estimated"] += int(st["opt_rank"] > 0)
    tot["flagged"] += st["n_fallback"]
    tot["retried"] += st["n_retried"]
    tot["dense"] += st["n_dense"]
    print(f"case {c}: n={n} d={d} nq={nq} k={k} {data} pinned_rank={rank} -> opt_rank={st['opt_rank']} launches={st['main_launches']} "
          f"cand/q={st['n_candidates'] / nq:.0f} flagged={st['n_fallback']} retried={st['n_retried']} dense={st['n_dense']} "
          f"ms_fallback={st['ms_fallback']:.1f} {'OK' if ok else 'MISMATCH'}", flush=True)
    assert ok
print("all cases equal the exact dense path;", tot)
