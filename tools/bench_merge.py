#!/usr/bin/env python3
"""Shard-merge kernel timing (ccr_merge_topk) for R shards x 3 452 queries x top-k, checked against the oracle."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]
from ccrec_amd import ops
from oracle import oracle as orc
for R,k in ((2,100),(4,100),(8,100),(8,1001),(16,1001)):
    # distinct scores per query (small integers, exact in fp32): per-shard lists must already be in canonical order, and
    # random ids would put exact ties in the wrong order
    ranks=torch.rand(3452,R*k,device="cuda").argsort(dim=1).float()            # a permutation of 0..R*k-1 per query
    s=ranks.view(3452,R,k).permute(1,0,2).contiguous().sort(dim=2,descending=True).values
    i=torch.randint(0,2681468,(R,3452,k),device="cuda")
    a,b=ops.merge_topk(s,i)
    os_,oi=orc.merge_topk(s[:, :40].cpu().numpy(), i[:, :40].cpu().numpy())
    assert np.array_equal(a[:40].cpu().numpy(), os_) and np.array_equal(b[:40].cpu().numpy(), oi)
    for _ in range(3): ops.merge_topk(s,i)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(30): ops.merge_topk(s,i)
    torch.cuda.synchronize(); print("R",R,"k",k,"merge", round((time.perf_counter()-t)/30*1e6,1),"us")

# ---- the packed shard messages of the multi-GPU exchange (ccr_merge_shard_messages)
from ccrec_amd.dist import ShardMessage
for R, k, nq in ((8, 100, 3452), (8, 100, 6980), (8, 1001, 3452), (4, 100, 3452), (2, 100, 3452)):
    ranks = torch.rand(nq, R * k, device="cuda").argsort(dim=1).float()
    s = ranks.view(nq, R, k).permute(1, 0, 2).contiguous().sort(dim=2, descending=True).values
    rows = torch.randint(0, 335184, (R, nq, k), device="cuda")
    g = ShardMessage(nq, k, "cuda", R)
    for r in range(R):
        m = ShardMessage(nq, k, "cuda", 1)
        m.fill(s[r], rows[r] + r * 335184, r * 335184, 335184)
        g.recv.view(R, -1)[r].copy_(m.send)
    a, b = g.merge()
    os_, oi = orc.merge_topk(s[:, :40].cpu().numpy(), (rows + torch.arange(R, device="cuda").view(R, 1, 1) * 335184)[:, :40].cpu().numpy())
    assert np.array_equal(a[:40].cpu().numpy(), os_) and np.array_equal(b[:40].cpu().numpy(), oi)
    for _ in range(3):
        g.merge()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(50):
        g.merge()
    torch.cuda.synchronize()
    print("messages R", R, "k", k, "n_q", nq, "merge", round((time.perf_counter() - t) / 50 * 1e6, 1), "us")

# ---- the SHORT-list exchange (ccr_merge_short_lists): R ranks send k_list = short_list_length(k, R) entries per query, the merge keeps k
from ccrec_amd.dist import short_list_length
for R, k, nq in ((8, 1001, 3452), (4, 1001, 3452), (2, 1001, 3452), (8, 100, 3452), (8, 1000, 10000)):
    kl = short_list_length(k, R)
    # a global ranking per query dealt to the shards at random (the binomial shares the short lists are sized for); every shard sends its kl best
    owner = torch.randint(0, R, (nq, 4 * k), device="cuda")
    score = torch.arange(4 * k, 0, -1, device="cuda", dtype=torch.float32).repeat(nq, 1)          # rank g has score 4k - g
    s = torch.full((R, nq, kl), -1.0, device="cuda")
    rows = torch.zeros((R, nq, kl), dtype=torch.int64, device="cuda")
    for r in range(R):
        mine = owner == r
        pos = mine.cumsum(1) - 1
        take = mine & (pos < kl)
        qi, gi = take.nonzero(as_tuple=True)
        s[r, qi, pos[qi, gi]] = score[qi, gi]
        rows[r, qi, pos[qi, gi]] = gi                                                           # local row = global rank (distinct per shard)
    g = ShardMessage(nq, kl, "cuda", R)
    for r in range(R):
        m = ShardMessage(nq, kl, "cuda", 1)
        m.fill(s[r], rows[r] + r * 335184, r * 335184, 335184)
        g.recv.view(R, -1)[r].copy_(m.send)
    a, b, flags, count = ops.merge_short_lists(g.recv, R, nq, kl, k)
    assert int(count) == 0 and torch.equal(a, score[:, :k])
    for _ in range(3):
        ops.merge_short_lists(g.recv, R, nq, kl, k)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(50):
        ops.merge_short_lists(g.recv, R, nq, kl, k)
    torch.cuda.synchronize()
    print("short lists R", R, "k", k, "k_list", kl, "n_q", nq, "merge + verification", round((time.perf_counter() - t) / 50 * 1e6, 1), "us;",
          "message", g.nbytes, "B per rank (full lists:", ops.shard_message_bytes(nq, k), "B)")
