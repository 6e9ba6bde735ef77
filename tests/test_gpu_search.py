"""GPU parity proper: the HIP search path (through the C ABI) vs the CPU oracle and the golden vectors.
Integer/index work is compared bit-exact; scores are canonical fp32 values and compared bit-exact too."""
import os

import numpy as np
import pytest
import torch

from helpers import assert_rank_close, canonicalise
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

DENSE, FUSED = 1, 2


def _bf16(bits):
    return torch.from_numpy(bits.view(np.int16)).view(torch.bfloat16).cuda()


def _bits(t):
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


def _rand_bits(n, d, seed, scale=None):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, d, generator=g) * (scale if scale is not None else d ** -0.5)
    return orc.pack_bf16(x.numpy())


def _check_exact(index, Qb, Db, k, flags, block=None):
    s, i = index.search(_bf16(Qb), k, flags)
    ref_i, ref_s = orc.canonical_search(Qb, Db, k, block)
    got_i, got_s = i.cpu().numpy() - index.offset, s.cpu().numpy()
    assert np.array_equal(got_i, ref_i), f"ids differ in {np.sum(got_i != ref_i)} places"
    assert np.array_equal(got_s.view(np.uint32), ref_s.view(np.uint32))
    return index.last_stats()


# ----------------------------------------------------------------------------------------- golden, via ranking()
def _table_func(table):
    return lambda rows: table[torch.as_tensor(rows, dtype=torch.long)]


def _ranking_via_api(g, sim, block=None):
    from ccrec_amd.ms_marco_eval import ranking
    os.environ["CCREC_SIM_TYPE"] = sim
    Eq, Ed = torch.from_numpy(g["Eq"]), torch.from_numpy(g["Ed"])
    nq, nd = Eq.shape[0], Ed.shape[0]
    table = torch.cat([Eq, Ed], 0)
    queries = {f"q{i}": i for i in range(nq)}
    corpus = {f"p{j}": nq + j for j in range(nd)}
    block_dict = None if block is None else {f"q{i}": [f"p{j}" for j in block[i]] for i in range(nq)}
    prof = ranking(corpus, queries, _table_func(table), int(g["batch_size"]), block_dict)
    L = min(1001, nd)
    ids = np.array([[int(p[1:]) for p in prof[f"q{i}"]] for i in range(nq)], np.int64)
    sc = np.array([list(prof[f"q{i}"].values()) for i in range(nq)], np.float32)
    assert ids.shape == (nq, L)
    return ids, sc


@pytest.mark.parametrize("name,sim,trunc,tol", [
    ("g1_ranking_dot.npz", "dot", False, 2e-6), ("g4_ranking_trunc.npz", "dot", True, 2e-6),
    ("g2_ranking_cos.npz", "cos", False, 1e-3)])
def test_ranking_api_vs_reference_golden(golden_dir, name, sim, trunc, tol):
    g = np.load(os.path.join(golden_dir, name))
    ids, sc = _ranking_via_api(g, sim)
    assert_rank_close(ids, sc, g["ids"], g["scores"], tol=tol, truncated=trunc)     # vs the reference's output
    ref_i, ref_s = orc.canonical_ranking(g["Eq"], g["Ed"], sim)                      # vs the oracle: bit-exact
    assert np.array_equal(ids, ref_i) and np.array_equal(sc.view(np.uint32), ref_s.view(np.uint32))


def test_ranking_api_block_dict(golden_dir):
    g = np.load(os.path.join(golden_dir, "g3_ranking_block.npz"))
    ptr, idx = g["block_ptr"], g["block_idx"]
    block = [idx[ptr[i]:ptr[i + 1]].tolist() for i in range(len(ptr) - 1)]
    ids, sc = _ranking_via_api(g, "dot", block)
    ref_i, ref_s = orc.canonical_ranking(g["Eq"], g["Ed"], "dot", block=block)
    assert np.array_equal(ids, ref_i) and np.array_equal(sc.view(np.uint32), ref_s.view(np.uint32))
    ri, rs = canonicalise(g["ids"], g["scores"])
    assert_rank_close(ids, sc, ri, rs, tol=2e-6)
    from ccrec_amd.ms_marco_eval import ranking
    with pytest.raises(AssertionError, match="block id not found"):
        ranking({"p0": 0, "p1": 1}, {"q0": 0}, _table_func(torch.from_numpy(g["Ed"][:2])), 8, {"q0": ["nope"]})


def test_ranking_api_cos_block_dict_and_truncation_together(golden_dir):
    """Golden g17 through ranking(): cos + block lists long enough to reach into the kept 1001 of 1 100 passages (and one short, one
    empty list) -- bit-exact against the oracle, within the bf16 tolerance of the reference's own output."""
    g = np.load(os.path.join(golden_dir, "g17_ranking_cos_block_trunc.npz"))
    ptr, idx = g["block_ptr"], g["block_idx"]
    block = [idx[ptr[i]:ptr[i + 1]].tolist() for i in range(len(ptr) - 1)]
    ids, sc = _ranking_via_api(g, "cos", block)
    ref_i, ref_s = orc.canonical_ranking(g["Eq"], g["Ed"], "cos", block=block)
    assert np.array_equal(ids, ref_i) and np.array_equal(sc.view(np.uint32), ref_s.view(np.uint32))
    assert_rank_close(ids, sc, g["ids"], g["scores"], tol=1e-3, truncated=True)
    os.environ["CCREC_SIM_TYPE"] = "dot"


def test_exact_arithmetic_ties_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "g5_ranking_exact_ties.npz"))
    ids, sc = _ranking_via_api(g, "dot")
    ri, rs = canonicalise(g["ids"], g["scores"])
    assert np.array_equal(ids, ri) and np.array_equal(sc.view(np.uint32), rs.view(np.uint32))


def test_sim_type_keyerror():
    from ccrec_amd.ms_marco_eval import ranking
    old = os.environ.pop("CCREC_SIM_TYPE", None)
    try:
        with pytest.raises(KeyError):
            ranking({"p": 0}, {"q": 0}, _table_func(torch.zeros(1, 8)), 4)
    finally:
        if old is not None:
            os.environ["CCREC_SIM_TYPE"] = old


# ----------------------------------------------------------------------------------------- score kernels
def test_canonical_and_mfma_scores():
    from ccrec_amd import ops
    Db, Qb = _rand_bits(700, 768, 1), _rand_bits(37, 768, 2)
    index = ops.CorpusIndex(_bf16(Db))
    can = index.scores(_bf16(Qb), "canonical").cpu().numpy()
    ref = orc.canonical_scores(Qb, Db)
    assert np.array_equal(can.view(np.uint32), ref.view(np.uint32))
    mf = index.scores(_bf16(Qb), "mfma").cpu().numpy()
    # MFMA fp32 accumulation: bounded by gamma * ||q|| * ||d|| (the filter margin); report the measured error
    qn, dn = orc.row_norms_bf16(Qb), orc.row_norms_bf16(Db)
    bound = 768 * 2.0 ** -23 * qn[:, None] * dn[None, :]
    err = np.abs(mf.astype(np.float64) - ref)
    print("max |mfma - canonical| =", err.max(), " worst fraction of the margin =", (err / bound).max())
    assert (err <= bound).all()
    # exact-integer operands (asymmetric): the MFMA product must be exact -> catches any layout transposition
    rs = np.random.RandomState(3)
    Di = orc.pack_bf16(rs.randint(-8, 9, size=(515, 128)).astype(np.float32))
    Qi = orc.pack_bf16(rs.randint(-8, 9, size=(261, 128)).astype(np.float32))
    ix2 = ops.CorpusIndex(_bf16(Di))
    mf2 = ix2.scores(_bf16(Qi), "mfma").cpu().numpy()
    assert np.array_equal(mf2, orc.canonical_scores(Qi, Di))


# ----------------------------------------------------------------------------------------- search paths
@pytest.mark.parametrize("n,nq,d,k", [(300, 7, 768, 300), (1000, 3, 64, 1), (5000, 70, 768, 100), (2049, 257, 1024, 1001),
                                      (777, 1, 8, 10), (1500, 5, 776, 50), (9000, 3, 64, 4096)])
def test_dense_path_exact(n, nq, d, k):
    from ccrec_amd import ops
    Db, Qb = _rand_bits(n, d, n), _rand_bits(nq, d, nq + 1)
    st = _check_exact(ops.CorpusIndex(_bf16(Db)), Qb, Db, k, DENSE)
    assert st["path"] == 0


@pytest.mark.parametrize("n,nq,d,k", [(20000, 300, 768, 100), (16384, 64, 768, 10), (30001, 513, 1024, 257),
                                      (40000, 33, 768, 1001), (9000, 1, 128, 5), (70000, 20, 64, 10),
                                      (50000, 10, 2048, 20), (140000, 6, 64, 4096)])
def test_fused_path_exact(n, nq, d, k):
    from ccrec_amd import ops
    Db, Qb = _rand_bits(n, d, n), _rand_bits(nq, d, nq + 1)
    st = _check_exact(ops.CorpusIndex(_bf16(Db), global_row_offset=12345), Qb, Db, k, FUSED)
    assert st["path"] == 1, st
    print(st)


def test_planner_default_is_fused_at_scale_and_exact():
    from ccrec_amd import ops
    n, nq, d, k = 300_000, 64, 768, 100
    Db, Qb = _rand_bits(n, d, 77), _rand_bits(nq, d, 78)
    st = _check_exact(ops.CorpusIndex(_bf16(Db)), Qb, Db, k, 0)
    assert st["path"] == 1 and st["n_fallback"] == 0, st
    print(st)


def test_mass_ties_and_duplicates_fall_back_exactly():
    """Exact-arithmetic corpus where thousands of rows tie at the top: the fused path must flag the
    queries (candidate overflow / margin set too large) and the dense fallback must give the canonical order."""
    from ccrec_amd import ops
    rs = np.random.RandomState(5)
    n, d = 12000, 128
    D = rs.randint(-4, 5, size=(n, d)).astype(np.float32) / 4
    D[2000:9000] = D[17]                    # 7001 identical rows
    Q = rs.randint(-4, 5, size=(5, d)).astype(np.float32) / 4
    Q[0] = D[17]                             # its top-k is inside the tie
    Db, Qb = orc.pack_bf16(D), orc.pack_bf16(Q)
    st = _check_exact(ops.CorpusIndex(_bf16(Db)), Qb, Db, 100, FUSED)
    assert st["path"] == 1 and st["n_fallback"] >= 1, st


def test_adversarial_sorted_corpus():
    """Scores increase with the row index, so the sampled threshold is loose for late rows."""
    from ccrec_amd import ops
    n, d = 50000, 64
    g = torch.Generator().manual_seed(9)
    base = torch.randn(d, generator=g)
    D = (torch.linspace(-1, 1, n)[:, None] * base[None, :] + 0.01 * torch.randn(n, d, generator=g)).numpy()
    Q = (base[None, :] * torch.tensor([[1.0], [-1.0], [0.5]])).numpy()
    Db, Qb = orc.pack_bf16(D), orc.pack_bf16(Q)
    st = _check_exact(ops.CorpusIndex(_bf16(Db)), Qb, Db, 64, FUSED)
    print(st)


def test_empty_and_invalid():
    from ccrec_amd import ops, _lib
    Db = _rand_bits(100, 64, 1)
    index = ops.CorpusIndex(_bf16(Db))
    s, i = index.search(torch.empty(0, 64, dtype=torch.bfloat16, device="cuda"), 5)
    assert s.shape == (0, 5) and i.shape == (0, 5)
    with pytest.raises(_lib.CcrError):
        index.search(_bf16(_rand_bits(2, 64, 2)), 101)      # k > n_rows
    with pytest.raises(_lib.CcrError):
        ops.CorpusIndex(torch.zeros(4, 12, dtype=torch.bfloat16, device="cuda"))   # dim % 8 != 0


def test_merge_and_sharded_search_equal_single():
    from ccrec_amd import ops
    n, nq, d, k = 30000, 50, 768, 100
    Db, Qb = _rand_bits(n, d, 31), _rand_bits(nq, d, 32)
    D, Q = _bf16(Db), _bf16(Qb)
    s1, i1 = ops.CorpusIndex(D).search(Q, k)
    from ccrec_amd.dist import shard_bounds
    parts = []
    for r in range(3):
        lo, hi = shard_bounds(n, 3, r)
        parts.append(ops.CorpusIndex(D[lo:hi].contiguous(), global_row_offset=lo).search(Q, k))
    gs, gi = torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts])
    ms, mi = ops.merge_topk(gs, gi)
    assert torch.equal(mi, i1) and torch.equal(ms.view(torch.int32), s1.view(torch.int32))
    os_, oi = orc.merge_topk(gs.cpu().numpy(), gi.cpu().numpy())
    assert np.array_equal(mi.cpu().numpy(), oi) and np.array_equal(ms.cpu().numpy(), os_)


@pytest.mark.parametrize("n,nq,k,world", [(20000, 37, 100, 3), (9000, 5, 1001, 8), (3000, 3, 2500, 8), (700, 9, 300, 3)])
def test_shard_messages_merge_equals_single(n, nq, k, world):
    """The one-collective exchange layout: every shard's search writes a packed message (header | scores | u32 local rows,
    ccr_search_shard), the gathered buffer is emulated by copying the messages side by side, ccr_merge_shard_messages reads
    them in place.  (20000, 37, 100): n_q k 4 is not a multiple of 16 -- block padding; k = 1001 x 8 ranks: the largest
    lists the LDS merge takes; k = 2500 x 8: the global-memory merge; (700, 300, 3 ranks): shards SMALLER than k -- k_valid
    < k through ccr_shard_message_fill, padding slots rank last.)"""
    from ccrec_amd import ops
    from ccrec_amd.dist import shard_bounds, ShardMessage
    d = 64 if k > 1001 else 768
    Db, Qb = _rand_bits(n, d, 33), _rand_bits(nq, d, 34)
    Db[n // 2 + 1] = Db[1]                     # cross-shard exact tie: the lower global id wins
    D, Q = _bf16(Db), _bf16(Qb)
    s1, i1 = ops.CorpusIndex(D).search(Q, k)
    gathered = ShardMessage(nq, k, D.device, world)
    for r in range(world):
        lo, hi = shard_bounds(n, world, r)
        m = ShardMessage(nq, k, D.device, 1)
        ix = ops.CorpusIndex(D[lo:hi].contiguous(), global_row_offset=lo)
        if hi - lo >= k:
            ix.search_shard(Q, k, m.send, defer=(r % 2 == 1))      # both forms; finish() of the deferred one below
            if r % 2 == 1:
                ix.finish()
            so, io = ix.search(Q, k)
            assert torch.equal(m.scores, so) and torch.equal(m.rows.to(torch.int64).bitwise_and(0xFFFFFFFF) + lo, io)
        else:
            so, io = ix.search(Q, hi - lo)
            m.fill(so, io, lo, hi - lo)
        gathered.recv.view(world, -1)[r].copy_(m.send)
    hdrs = ShardMessage.parse_headers(gathered.all_headers.cpu())
    assert [h["row_offset"] for h in hdrs] == [shard_bounds(n, world, r)[0] for r in range(world)]
    assert all(h["n_flagged"] == 0 and h["k_valid"] == min(k, h["n_rows"]) for h in hdrs)
    ms, mi = gathered.merge()
    assert torch.equal(mi, i1) and torch.equal(ms.view(torch.int32), s1.view(torch.int32))
    gs, gi = gathered.decoded()                # the same through the oracle's merge on the decoded lists
    os_, oi = orc.merge_topk(gs.cpu().numpy(), gi.cpu().numpy())
    assert np.array_equal(mi.cpu().numpy(), oi) and np.array_equal(ms.cpu().numpy(), os_)


@pytest.mark.parametrize("n,nq,k,world,skewed", [(40000, 21, 1001, 8, 0), (40000, 21, 1001, 8, 3), (30000, 9, 1001, 2, 1), (12000, 33, 100, 8, 2),
                                                 (2400, 5, 300, 3, 0)])
def test_short_shard_messages_merge_equals_single(n, nq, k, world, skewed):
    """The SHORT-list exchange (ccr_merge_short_lists; ranking()'s k = 1001 over 8 shards sends 196 entries per query and rank instead of
    1001): every shard searches and packs its canonical top-k_list (ccr_search_shard), the gathered buffer is emulated, the merge keeps k
    of the R k_list entries and flags the queries for which a list was consumed to its end.  iid rows: nobody is flagged and ids and
    score bits equal the single-index search (a cross-shard exact tie included).  `skewed` queries have their whole top-k in ONE shard
    (a corpus in topical order): exactly those are flagged -- same flags as the oracle's restatement -- and after the full-list repeat
    of THOSE queries every list is the single-index search's, bit for bit.  (2400 rows over 3 shards at k = 300: shards of 800 rows,
    k_list = 157.)"""
    from ccrec_amd import ops
    from ccrec_amd.dist import shard_bounds, short_list_length, ShardMessage
    d = 256
    Db, Qb = _rand_bits(n, d, 133), _rand_bits(nq, d, 134)
    Db[n // 2 + 1] = Db[1]                     # cross-shard exact tie: the lower global id wins
    lo_last = shard_bounds(n, world, world - 1)[0]
    if skewed:   # rows that score far above everything else for ONE query each, all inside the last shard
        rs = np.random.RandomState(5)
        Df = rs.randn(n, d).astype(np.float32) / 16
        Qf = np.linalg.qr(rs.randn(d, nq))[0].T.astype(np.float32)
        for j in range(skewed):
            rows = slice(lo_last + 7 + j * k, lo_last + 7 + (j + 1) * k)
            Df[rows] = 2.0 * Qf[2 * j] + 0.3 * Df[rows]
        Db, Qb = orc.pack_bf16(Df), orc.pack_bf16(Qf)
        Db[n // 2 + 1] = Db[1]
    D, Q = _bf16(Db), _bf16(Qb)
    s1, i1 = ops.CorpusIndex(D).search(Q, k)
    kl = short_list_length(k, world)
    assert kl < k and world * kl >= k
    gathered = ShardMessage(nq, kl, D.device, world)
    shards = []
    for r in range(world):
        lo, hi = shard_bounds(n, world, r)
        m = ShardMessage(nq, kl, D.device, 1)
        ix = ops.CorpusIndex(D[lo:hi].contiguous(), global_row_offset=lo)
        ix.search_shard(Q, kl, m.send, defer=(r % 2 == 1))
        if r % 2 == 1:
            ix.finish()
        gathered.recv.view(world, -1)[r].copy_(m.send)
        shards.append(ix)
    ms, mi, flags, count = ops.merge_short_lists(gathered.recv, world, nq, kl, k)
    want = sorted(2 * j for j in range(skewed))
    assert flags.nonzero().squeeze(1).tolist() == want and int(count) == len(want)
    gs, gi = gathered.decoded()
    hdrs = ShardMessage.parse_headers(gathered.all_headers.cpu())
    os_, oi, oflags = orc.merge_short_lists(gs.cpu().numpy(), gi.cpu().numpy(), [h["n_rows"] > h["k_valid"] for h in hdrs], k)
    assert np.array_equal(flags.cpu().numpy(), oflags)
    assert np.array_equal(mi.cpu().numpy(), oi) and np.array_equal(ms.cpu().numpy().view(np.uint32), os_.view(np.uint32))
    good = (flags == 0)
    assert torch.equal(mi[good], i1[good]) and torch.equal(ms[good].view(torch.int32), s1[good].view(torch.int32))
    if want:   # the repeat: full lists for the flagged queries only
        which = flags.nonzero().squeeze(1)
        full = ShardMessage(len(want), k, D.device, world)
        for r, ix in enumerate(shards):
            m = ShardMessage(len(want), k, D.device, 1)
            ix.search_shard(Q[which].contiguous(), k, m.send)
            full.recv.view(world, -1)[r].copy_(m.send)
        fs, fi = full.merge()
        ms[which], mi[which] = fs, fi
    assert torch.equal(mi, i1) and torch.equal(ms.view(torch.int32), s1.view(torch.int32))


def test_short_list_merge_flags_a_consumed_list_only_when_its_shard_holds_more():
    """The verification rule at its edges: a list consumed to its end is harmless when its shard sent EVERY row it has (k_valid = n_rows:
    nothing is unsent); padding slots never count as kept entries; a list whose LAST entry is kept is flagged even when the merged
    result happens to be right (the rule is a sufficient condition, checked exactly)."""
    from ccrec_amd import ops
    from ccrec_amd.dist import ShardMessage
    nq, kl, k, world = 2, 4, 6, 2
    gathered = ShardMessage(nq, kl, "cuda", world)
    # rank 0: a 4-row shard that sent all 4 rows (scores 9 8 7 6); rank 1: a 100-row shard that sent its top 4
    for r, (sc, n_rows) in enumerate([([[9, 8, 7, 6], [9, 8, 7, 6]], 4), ([[5, 4, 3, 2], [10, 9.5, 8.5, 6.5]], 100)]):
        m = ShardMessage(nq, kl, "cuda", 1)
        m.fill(torch.tensor(sc, dtype=torch.float32, device="cuda"), torch.arange(kl, device="cuda").repeat(nq, 1) + r * 1000, r * 1000, n_rows)
        gathered.recv.view(world, -1)[r].copy_(m.send)
    ms, mi, flags, count = ops.merge_short_lists(gathered.recv, world, nq, kl, k)
    # query 0: rank 0's list is consumed (all 4 kept) but the shard has no more rows; rank 1 keeps 2 of 4 -> final
    # query 1: rank 1's four entries all rank within the top 6 (10 9.5 9 8.5 8 7 | 6.5 would be 7th: NOT kept) -> 3 kept of 4 -> final
    assert flags.tolist() == [0, 0] and int(count) == 0
    assert ms.tolist() == [[9, 8, 7, 6, 5, 4], [10, 9.5, 9, 8.5, 8, 7]]
    assert mi.tolist() == [[0, 1, 2, 3, 1000, 1001], [1000, 1001, 0, 1002, 1, 2]]
    ms, mi, flags, count = ops.merge_short_lists(gathered.recv, world, nq, kl, 7)
    assert flags.tolist() == [0, 1] and int(count) == 1          # k = 7: rank 1's last entry (6.5) is kept -> its 5th row might outrank 6
    with pytest.raises(Exception):
        ops.merge_short_lists(gathered.recv, world, nq, kl, 9)   # 2 x 4 entries cannot fill 9 ranks


def test_shard_messages_of_a_corpus_smaller_than_k_fill_every_slot():
    from ccrec_amd import ops
    from ccrec_amd.dist import shard_bounds, ShardMessage, PAD_ID
    n, nq, k, world, d = 11, 4, 16, 3, 64
    Db, Qb = _rand_bits(n, d, 35), _rand_bits(nq, d, 36)
    D, Q = _bf16(Db), _bf16(Qb)
    gathered = ShardMessage(nq, k, D.device, world)
    for r in range(world):
        lo, hi = shard_bounds(n, world, r)
        m = ShardMessage(nq, k, D.device, 1)
        so, io = ops.CorpusIndex(D[lo:hi].contiguous(), global_row_offset=lo).search(Q, hi - lo)
        m.fill(so, io, lo, hi - lo)
        gathered.recv.view(world, -1)[r].copy_(m.send)
    ms, mi = gathered.merge()
    ref_i, ref_s = orc.canonical_search(Qb, Db, n)
    assert np.array_equal(mi.cpu().numpy()[:, :n], ref_i) and np.array_equal(ms.cpu().numpy()[:, :n], ref_s)
    tail = mi.cpu().numpy()[:, n:]
    assert bool(torch.isinf(ms[:, n:]).all()) and (tail > 2 ** 62).all() and (tail <= PAD_ID).all()
    assert all(len(set(row.tolist())) == k - n for row in tail)


@pytest.mark.parametrize("na,nb", [(5, 300), (600, 9000)])
def test_cos_sim_matches_reference_formula(na, nb):
    """scripts/ms_marco_eval.py:155-162: normalize(a) @ normalize(b).T.  Small case: bit-identical to the oracle's
    canonical scores of the normalised bf16 rows; large case (MFMA path): within 1e-3 of the fp32 formula, the
    tolerance the bf16 rounding of the normalised rows sets."""
    from ccrec_amd import ms_marco_eval as mm
    g = torch.Generator().manual_seed(na)
    a, b = torch.randn(na, 768, generator=g), torch.randn(nb, 768, generator=g)
    got = mm.cos_sim(a, b)
    ref = torch.nn.functional.normalize(a, p=2, dim=1) @ torch.nn.functional.normalize(b, p=2, dim=1).T
    assert got.shape == (na, nb)
    torch.testing.assert_close(got.cpu(), ref, rtol=0, atol=1e-3)
    if na * nb * 768 <= mm.CANONICAL_COS_SIM_MACS:
        can = orc.canonical_scores(orc.normalize_pack_bf16(a.numpy()), orc.normalize_pack_bf16(b.numpy()))
        assert np.array_equal(got.cpu().numpy().view(np.uint32), can.view(np.uint32))
    assert mm.cos_sim(a[0], b[:3]).shape == (1, 3)          # 1-D inputs are promoted like the reference does


def test_assign_topk_golden(golden_dir):
    from ccrec_amd.rime_util import _assign_topk
    g = np.load(os.path.join(golden_dir, "g8_assign_topk.npz"))
    k = int(g["k"])
    csr = _assign_topk((g["U"], g["V"]), k)
    assert csr.shape == (50, 4000) and np.array_equal(csr.indptr, g["indptr"]) and np.all(csr.data == 1)
    ids = csr.indices.reshape(50, k)
    ref_i, ref_s = orc.canonical_search(orc.pack_bf16(g["U"]), orc.pack_bf16(g["V"]), k)
    assert np.array_equal(ids, ref_i)
    ref_sc = np.take_along_axis(g["U"].astype(np.float64) @ g["V"].astype(np.float64).T, g["indices"], 1).astype(np.float32)
    assert_rank_close(ids, ref_s, g["indices"], ref_sc, tol=2e-6, truncated=True)

    class FakeMatMul:  # MatMulExpression duck type: .left [U,d], .right [d,I]  (score_array.py:320-323)
        left, right = g["U"], g["V"].T
    assert np.array_equal(_assign_topk(FakeMatMul(), k).indices, csr.indices)


@pytest.mark.parametrize("d", [300, 50])
def test_assign_topk_odd_width_golden(golden_dir, d):
    """g15: the reference's own _assign_topk on 300- and 50-wide factors; here the factors are zero-padded by the pack (304 / 56) and
    searched by the fused / margin paths with a zero-filled K tail.  ids == oracle bit for bit, == the reference's wherever its fp32
    scores are separated."""
    from ccrec_amd.rime_util import _assign_topk
    g = np.load(os.path.join(golden_dir, "g15_assign_topk_odd_width.npz"))
    U, V, ref, k = g[f"U{d}"], g[f"V{d}"], g[f"indices{d}"], int(g[f"k{d}"])
    csr = _assign_topk((U, V), k)
    ids = csr.indices.reshape(U.shape[0], k)
    ref_i, ref_s = orc.canonical_search(orc.pack_bf16(U), orc.pack_bf16(V), k)
    assert np.array_equal(ids, ref_i)
    ref_sc = np.take_along_axis(U.astype(np.float64) @ V.astype(np.float64).T, ref, 1).astype(np.float32)
    assert_rank_close(ids, ref_s, ref, ref_sc, tol=2e-6, truncated=True)


def test_many_query_blocks_multi_item_workgroups():
    """3,452 queries = 14 query blocks: every workgroup walks several (range, query-block) items.
    All queries are compared against the exact dense GPU path, a subset against the CPU oracle."""
    from ccrec_amd import ops
    n, nq, d, k = 60_000, 3452, 768, 100
    Db, Qb = _rand_bits(n, d, 91), _rand_bits(nq, d, 92)
    index = ops.CorpusIndex(_bf16(Db))
    s, i = index.search(_bf16(Qb), k, FUSED)
    st = index.last_stats()
    print(st)
    assert st["path"] == 1 and st["n_fallback"] == 0, st
    s2, i2 = index.search(_bf16(Qb), k, DENSE)
    assert torch.equal(i, i2) and torch.equal(s.view(torch.int32), s2.view(torch.int32))
    sub = np.r_[0:8, 1700:1708, 3444:3452]
    ref_i, ref_s = orc.canonical_search(Qb[sub], Db, k)
    assert np.array_equal(i.cpu().numpy()[sub], ref_i) and np.array_equal(s.cpu().numpy()[sub], ref_s)


def test_bbpr_transform_low_rank_score():
    """BertBPR.transform twin (bbpr.py:526-550): lazy user x item score, top-k through the fused path."""
    from ccrec_amd import ops
    from ccrec_amd.bbpr_transform import transform
    from ccrec_amd.rime_util import _assign_topk
    allb = _rand_bits(3000, 768, 5)
    rs = np.random.RandomState(1)
    i_to_ptr, j_to_ptr = rs.permutation(3000)[:40], rs.permutation(3000)[:2500]
    S = transform(_bf16(allb), i_to_ptr, j_to_ptr)
    assert S.shape == (40, 2500) and S.T.shape == (2500, 40)
    ref = orc.canonical_scores(allb[i_to_ptr], allb[j_to_ptr])
    assert np.array_equal(S.numpy().view(np.uint32), ref.view(np.uint32))
    csr = _assign_topk(S, 20)
    ref_i, _ = orc.rank(ref, 20)
    assert np.array_equal(csr.indices.reshape(40, 20), ref_i)


def test_rank_metrics_and_profile_tensors():
    """MRR@k / Recall@k on device vs the oracle's restatement of BEIR's mrr; ranking_profile <-> tensors."""
    from ccrec_amd.evaluation import profile_to_tensors, rank_metrics, tensors_to_profile
    rs = np.random.RandomState(4)
    nq, k, n = 57, 100, 5000
    ids = np.stack([rs.permutation(n)[:k] for _ in range(nq)]).astype(np.int64)
    qrels = [set(rs.randint(0, n, size=rs.randint(0, 4)).tolist()) | ({int(ids[q, rs.randint(0, k)])} if q % 3 else set())
             for q in range(nq)]
    got = rank_metrics(torch.from_numpy(ids).cuda(), qrels, (1, 5, 10, 100))
    for kk in (1, 5, 10, 100):
        assert got[f"MRR@{kk}"] == orc.mrr(ids, qrels, kk)
        rec = [len(set(ids[q, :kk].tolist()) & qrels[q]) / len(qrels[q]) for q in range(nq) if qrels[q]]
        assert abs(got[f"Recall@{kk}"] - round(float(np.mean(rec)), 5)) < 2e-5
    corpus_ids = [f"p{j}" for j in range(n)]
    prof = {f"q{q}": {corpus_ids[j]: float(-r) for r, j in enumerate(ids[q])} for q in range(nq)}
    qids, ti, ts = profile_to_tensors(prof, corpus_ids)
    assert np.array_equal(ti.numpy(), ids)
    assert tensors_to_profile(qids, corpus_ids, ti, ts) == prof


def test_evaluate_item_rec_matches_reference(golden_dir):
    """src/rime_lite/metrics/__init__.py:87-89 (SURVEY 8b signature) on a low-rank score: fixture g13 is the reference's
    evaluate_item_rec on MatMulExpression(U @ V.T); U, V are bf16-exact so the fused top-k sees the same values."""
    import scipy.sparse as sps
    from ccrec_amd.rime_util import evaluate_item_rec, evaluate_assigned, _assign_topk
    from ccrec_amd.bbpr_transform import LowRankScore
    from ccrec_amd import ops
    g = np.load(os.path.join(golden_dir, "g13_item_rec.npz"))
    U, V, k = g["U"], g["V"], int(g["k"])
    target = sps.csr_matrix((np.ones(len(g["target_indices"])), g["target_indices"], g["target_indptr"]), shape=(U.shape[0], V.shape[0]))
    out = evaluate_item_rec(target, (U, V), k)
    for key in ("prec", "recs/user", "item_cov", "item_ppl", "user_cov", "user_ppl", "obj_mean", "recall"):
        ref = float(g["m_" + key.replace("/", "_")])
        assert abs(out[key] - ref) <= 1e-6 * max(1.0, abs(ref)), (key, out[key], ref)
    assert out["prec"] > 0.1
    # the lazy low-rank score object gives the same numbers
    S = LowRankScore(ops.pack_bf16(torch.from_numpy(U).cuda()), ops.pack_bf16(torch.from_numpy(V).cuda()))
    out2 = evaluate_assigned(target, _assign_topk(S, k), S, axis=1)
    assert all(abs(out2[key] - out[key]) < 1e-9 for key in out)


def test_max_k_overflowing_select_lds_stays_on_the_fused_path():
    """k = 4096 (MAX_K): far more candidates per query than the select stage's LDS holds; the iterative bound
    tightening must still deliver the exact canonical result without sending the queries to the dense path."""
    from ccrec_amd import ops
    n, nq, d, k = 400_000, 24, 768, 4096
    Db, Qb = _rand_bits(n, d, 71), _rand_bits(nq, d, 72)
    index = ops.CorpusIndex(_bf16(Db))
    s0, i0 = index.search(_bf16(Qb), k, 2)
    st = index.last_stats()
    assert st["path"] == 1 and st["n_fallback"] == 0, st
    s1, i1 = index.search(_bf16(Qb), k, 1)
    assert torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))


def test_rescore_is_canonical_under_wide_dynamic_range():
    """Canonical scores are fp64 sums in ascending element order; with products 2^50 apart inside one dot product the
    fp64 partial sums DO round, so any shortcut that re-orders the sum would show here.  Both data sets must reproduce
    the oracle bit for bit: (a) ordinary data, (b) elements scaled by 2^-30 / 2^+20 / 2^-25."""
    from ccrec_amd import ops
    n, nq, d, k = 40_000, 33, 768, 50
    for wide in (False, True):
        g = torch.Generator().manual_seed(77 + wide)
        D = torch.randn(n, d, generator=g) / d ** 0.5
        Q = torch.randn(nq, d, generator=g) / d ** 0.5
        if wide:
            D[:, 5::37] *= 2.0 ** -30
            D[:, 11::41] *= 2.0 ** 20
            Q[:, 7::29] *= 2.0 ** -25
        Db, Qb = orc.pack_bf16(D.numpy()), orc.pack_bf16(Q.numpy())
        s, i = ops.CorpusIndex(_bf16(Db)).search(_bf16(Qb), k, 2)
        ref_i, ref_s = orc.canonical_search(Qb, Db, k)
        assert np.array_equal(i.cpu().numpy(), ref_i), wide
        assert np.array_equal(s.cpu().numpy().view(np.uint32), ref_s.view(np.uint32)), wide


def test_global_row_offset_beyond_32_bits():
    """ids = local row + global_row_offset in int64 on every path (fused, dense, merge)."""
    from ccrec_amd import ops
    off = (1 << 40) + 123
    n, nq, d, k = 20000, 9, 128, 50
    Db, Qb = _rand_bits(n, d, 91), _rand_bits(nq, d, 92)
    ref_i, ref_s = orc.canonical_search(Qb, Db, k, None)
    index = ops.CorpusIndex(_bf16(Db), global_row_offset=off)
    for flags in (FUSED, DENSE):
        s, i = index.search(_bf16(Qb), k, flags)
        assert i.dtype == torch.int64 and np.array_equal(i.cpu().numpy() - off, ref_i)
        assert np.array_equal(s.cpu().numpy().view(np.uint32), ref_s.view(np.uint32))
    half = n // 2
    parts = [ops.CorpusIndex(_bf16(Db[lo:hi]), global_row_offset=off + lo).search(_bf16(Qb), k) for lo, hi in ((0, half), (half, n))]
    ms, mi = ops.merge_topk(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    assert np.array_equal(mi.cpu().numpy() - off, ref_i)


def test_non_finite_embeddings_do_not_fault_and_are_path_independent():
    """NaN / Inf rows and a NaN query: not a supported input (include/ccr_retrieval.h), but nothing faults or hangs, the
    infinite filter margins send every query to the exact dense path, and forced-fused == forced-dense bit for bit."""
    from ccrec_amd import ops
    g = torch.Generator().manual_seed(1)
    for n in (20000, 400):
        D = torch.randn(n, 128, generator=g) / 128 ** 0.5
        Q = torch.randn(6, 128, generator=g) / 128 ** 0.5
        D[5] = float("nan")
        D[7] = float("inf")
        D[9, 3] = float("-inf")
        Q[2] = float("nan")
        index = ops.CorpusIndex(ops.pack_bf16(D.cuda()))
        Qb = ops.pack_bf16(Q.cuda())
        s_f, i_f = index.search(Qb, 10, FUSED)
        st = index.last_stats()
        s_d, i_d = index.search(Qb, 10, DENSE)
        assert st["path"] == 1 and st["n_fallback"] == 6, st
        assert torch.equal(i_f, i_d) and torch.equal(s_f.view(torch.int32), s_d.view(torch.int32))
        finite = torch.isfinite(s_f[0])
        assert finite.sum() >= 7 and torch.all(s_f[0][finite][:-1] >= s_f[0][finite][1:])



def test_generate_embeddings_contract(tmp_path, capsys):
    """scripts/ms_marco_eval.py:123-152: ceil(num / bs) batches in index order, progress line when the step is a power of
    two, total line, optional torch.save(name); an empty index list gives a [0, embedding_size] tensor."""
    from ccrec_amd.ms_marco_eval import generate_embeddings
    g = torch.Generator().manual_seed(1)
    table = torch.randn(70, 768, generator=g)
    seen = []

    def embedding_func(rows):
        seen.append(list(rows))
        return table[torch.as_tensor(rows, dtype=torch.long)]

    ids = [f"d{j}" for j in range(70)]
    data = {ids[j]: j for j in range(70)}
    name = str(tmp_path / "emb.pt")
    out = generate_embeddings(ids, data, embedding_func, 16, name=name)
    assert [len(b) for b in seen] == [16, 16, 16, 16, 6] and seen[1][0] == 16        # ceil(70 / 16) batches, index order
    assert out.is_cuda and tuple(out.shape) == (70, 768) and torch.equal(out.cpu(), table)
    assert torch.equal(torch.load(name).cpu(), table)
    text = capsys.readouterr().out
    assert "Processed 16 | 70" in text and "Processed 32 | 70" in text and "Processed 64 | 70" in text   # steps 1, 2, 4
    assert "Processed 48" not in text and "Processed total 70" in text
    packed = generate_embeddings(ids, data, embedding_func, 16, pack="cos")
    assert packed.dtype == torch.bfloat16
    assert np.array_equal(_bits(packed), orc.normalize_pack_bf16(table.numpy()))
    empty = generate_embeddings([], data, embedding_func, 16, embedding_size=768)
    assert tuple(empty.shape) == (0, 768)


def test_ranking_lazy_profile_equals_the_nested_dict(golden_dir, tmp_path):
    """ranking(..., lazy=True) returns a Mapping over the search's tensors: same keys, order, values, equality and file round
    trip as the reference's nested dict (scripts/ms_marco_eval.py:231-235), with a block_dict and with odd-width embeddings."""
    from ccrec_amd import ranking_profile as rp
    from ccrec_amd.ms_marco_eval import ranking
    g = np.load(os.path.join(golden_dir, "g3_ranking_block.npz"))
    os.environ["CCREC_SIM_TYPE"] = "dot"
    Eq, Ed = g["Eq"], g["Ed"]
    ptr, idx = g["block_ptr"], g["block_idx"]
    corpus = {f"p{j}": j for j in range(Ed.shape[0])}
    queries = {f"q{i}": Ed.shape[0] + i for i in range(Eq.shape[0])}
    table = torch.from_numpy(np.concatenate([Ed, Eq]))
    block = {f"q{i}": [f"p{j}" for j in idx[ptr[i]:ptr[i + 1]]] for i in range(Eq.shape[0])}
    eager = ranking(corpus, queries, _table_func(table), 64, block)
    lazy = ranking(corpus, queries, _table_func(table), 64, block, lazy=True)
    assert isinstance(eager, dict) and isinstance(lazy, rp.RankingProfile)
    assert lazy == eager and list(lazy) == list(eager) and all(list(lazy[q]) == list(eager[q]) for q in eager)
    assert lazy.top("q1", 3) == list(eager["q1"])[:3]
    lazy.save(tmp_path / "p.pt")
    assert rp.load(tmp_path / "p.pt") == eager
    # a width that is no multiple of 8 through the whole API (zero-padded by the pack)
    odd = table[:, :301].contiguous()
    a = ranking(corpus, queries, _table_func(odd), 64)
    ref_i, ref_s = orc.canonical_search(orc.pack_bf16(odd[Ed.shape[0]:].numpy()), orc.pack_bf16(odd[:Ed.shape[0]].numpy()), min(1001, Ed.shape[0]))
    got_i = np.array([[int(p[1:]) for p in a[q]] for q in queries])
    got_s = np.array([list(a[q].values()) for q in queries], np.float32)
    assert np.array_equal(got_i, ref_i) and np.array_equal(got_s.view(np.uint32), ref_s.view(np.uint32))


def test_index_lifecycle_pending_search_workspace_adoption_and_low_rank_add_zero():
    """(1) An index destroyed while an asynchronous search is pending waits for that search (ccr_index_destroy) -- nothing faults,
    the next index works; (2) an index adopts the previous one's workspace tensor and returns the same bits; (3) a deferred search
    is completed before its workspace is replaced; (4) LowRankScore + 0 (rime_lite's initial prior_score) is the score itself."""
    from ccrec_amd import ops
    from ccrec_amd.bbpr_transform import LowRankScore
    Db, Qb = _rand_bits(120_000, 256, 61), _rand_bits(300, 256, 62)
    D, Q = _bf16(Db), _bf16(Qb)
    ref = ops.CorpusIndex(D)
    s0, i0 = ref.search(Q, 50)
    for _ in range(3):                                      # (1)
        ix = ops.CorpusIndex(D)
        s, i = ix.search(Q, 50, defer=True)
        del ix                                              # never finished: the destroy waits for the search's own event
    torch.cuda.synchronize()
    assert torch.equal(i, i0) and torch.equal(s.view(torch.int32), s0.view(torch.int32))
    ws = ref.workspace                                      # (2)
    assert ws is not None and ws.numel() > 0
    again = ops.CorpusIndex(D, workspace=ws)
    s1, i1 = again.search(Q, 50)
    assert again.workspace.data_ptr() == ws.data_ptr() and torch.equal(i1, i0) and torch.equal(s1.view(torch.int32), s0.view(torch.int32))
    s2, i2 = again.search(Q, 50, defer=True)                # (3) a larger search needs a larger workspace: the pending one is finished first
    s3, i3 = again.search(Q, 1000)
    assert again._deferred is None and torch.equal(i2, i0) and s3.shape == (300, 1000)
    assert torch.equal(i3[:, :50], i0)
    low = LowRankScore(Q[:40].contiguous(), D[:5000].contiguous())   # (4)
    assert (low + 0) is low and (0 + low) is low and (low + None) is low
    dense = low + torch.ones(40, 5000, device="cuda")
    assert torch.allclose(dense, low.as_tensor() + 1.0)


def test_short_list_exchange_result_never_waits_for_later_work_on_the_compute_stream():
    """The stream-ordered (RCCL) path of ShardExchange where no second GPU exists: the gathered buffer is emulated, submit() gets a
    stand-in work handle, and the merge + verification runs on the SIDE stream with its flag count travelling to pinned memory.  Then the
    compute stream is kept busy for ~0.3 s AFTER submit() (the next steps' pack and search, in the pipelined loop): result() must
    return the single-index lists without waiting for it (r4 read count.item() behind that work) -- host_syncs == 0."""
    import time
    from ccrec_amd import ops
    from ccrec_amd.dist import shard_bounds, short_list_length, ShardMessage, ShardExchange, resume_short_lists
    resume_short_lists()
    n, nq, k, world, d = 6000, 24, 300, 3, 256
    Db, Qb = _rand_bits(n, d, 211), _rand_bits(nq, d, 212)
    D, Q = _bf16(Db), _bf16(Qb)
    s1, i1 = ops.CorpusIndex(D).search(Q, k)
    kl = short_list_length(k, world)
    gathered = ShardMessage(nq, kl, D.device, world)
    for r in range(world):
        lo, hi = shard_bounds(n, world, r)
        m = ShardMessage(nq, kl, D.device, 1)
        ops.CorpusIndex(D[lo:hi].contiguous(), global_row_offset=lo).search_shard(Q, kl, m.send)
        gathered.recv.view(world, -1)[r].copy_(m.send)
    torch.cuda.synchronize()

    class Arrived:                      # the collective's work handle: already complete
        def wait(self):
            return True

    ex = ShardExchange(gathered, None, None, None, k_out=k, queries=Q).submit(_work=Arrived())
    a = torch.randn(4096, 4096, device=D.device)
    t0 = time.perf_counter()
    for _ in range(60):                 # ~0.3 s of compute-stream work enqueued behind submit()
        a = (a @ a).clamp_(-1, 1)
    t_enqueue = time.perf_counter() - t0
    t0 = time.perf_counter()
    s, i = ex.result()
    t_result = time.perf_counter() - t0
    busy = torch.cuda.Event()
    busy.record()
    still_running = not busy.query()    # the matmuls are still in flight when result() has returned
    torch.cuda.synchronize()
    assert ex.host_syncs == 0 and ex.fallback_queries == 0 and not ex.repeated
    assert torch.equal(i, i1) and torch.equal(s.view(torch.int32), s1.view(torch.int32))
    assert still_running and t_result < 0.05, (t_enqueue, t_result)


def test_stream_wait_main_pass_orders_a_side_stream_behind_the_dominant_kernel():
    """ccr_search_stream_wait_main_pass: a side stream made to wait for the main pass of a deferred search packs ANOTHER shard beside the
    search's select stage; the search's results are those of a plain search, the pack's bits those of a pack on the main stream; after a
    dense-path search (nothing recorded) the call is a no-op."""
    from ccrec_amd import ops
    n, nq, d, k = 300_000, 400, 256, 100
    Db, Qb = _rand_bits(n, d, 301), _rand_bits(nq, d, 302)
    index = ops.CorpusIndex(_bf16(Db))
    Q = _bf16(Qb)
    s0, i0 = index.search(Q, k, FUSED)
    x = torch.randn(200_000, d, device="cuda")
    want = ops.pack_bf16(x)
    side = torch.cuda.Stream()
    out = torch.empty_like(want)
    torch.cuda.synchronize()
    s, i = index.search(Q, k, FUSED, defer=True)
    index.stream_wait_main_pass(side)
    with torch.cuda.stream(side):
        ops.pack_bf16(x, out=out)
    torch.cuda.current_stream().wait_stream(side)
    index.finish()
    torch.cuda.synchronize()
    assert torch.equal(i, i0) and torch.equal(s.view(torch.int32), s0.view(torch.int32))
    assert torch.equal(out.view(torch.int16), want.view(torch.int16))
    assert index.last_stats()["path"] == 1
    index.search(Q[:3], k, DENSE)
    index.stream_wait_main_pass(side)       # dense path: no event recorded, returns at once
    side.synchronize()
