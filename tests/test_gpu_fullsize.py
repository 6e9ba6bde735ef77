"""Full-size (BASELINE.json configs[1]: NQ 2,681,468 x 768, 3,452 queries, top-100) property tests.
The CPU oracle cannot score 9.3e9 pairs in seconds, so parity is established through size-independent
properties: every returned score is the canonical score of its id (oracle, C), lists are in canonical
order, for EVERY query an independent fp32 sweep of the whole corpus (torch's matmul on the device) finds
no missing row that beats the k-th score by more than 1e-4, and on a query subsample an fp32 BLAS sweep
of the WHOLE corpus finds every row that could possibly belong to the top-k; those are re-scored
canonically by the oracle and must reproduce the GPU list bit for bit."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _gen(n, dim, seed, chunk=262144):
    g = torch.Generator(device="cuda").manual_seed(seed)
    out = torch.empty(n, dim, dtype=torch.bfloat16, device="cuda")
    from ccrec_amd import ops
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        ops.pack_bf16(torch.randn(hi - lo, dim, generator=g, device="cuda") * dim ** -0.5, out=out[lo:hi])
    return out


def _bits(t):
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


def test_nq_config_properties():
    from ccrec_amd import ops
    n, nq, d, k = 2_681_468, 3_452, 768, 100
    D, Q = _gen(n, d, 1234), _gen(nq, d, 4321)
    index = ops.CorpusIndex(D)
    s, i = index.search(Q, k)
    st = index.last_stats()
    print(st)
    assert st["path"] == 1 and st["n_fallback"] == 0
    s_np, i_np = s.cpu().numpy(), i.cpu().numpy()
    # (1) canonical order inside every list: score descending, id ascending on ties
    ds = np.diff(s_np.astype(np.float64), axis=1)
    assert (ds <= 0).all()
    assert ((ds < 0) | (np.diff(i_np, axis=1) > 0)).all()
    assert i_np.min() >= 0 and i_np.max() < n
    assert all(len(set(row.tolist())) == k for row in i_np[::97])
    # (2) every score is the canonical score of its id (oracle on the host copy of the bf16 data)
    Db, Qb = _bits(D), _bits(Q)
    ref = orc.canonical_scores_pairs(Qb, Db, i_np)
    assert np.array_equal(ref.view(np.uint32), s_np.view(np.uint32))
    # (3) completeness on a query subsample: fp32 BLAS sweep of the whole corpus (error << 1e-4) selects
    #     every row within 1e-4 of the k-th score; canonical re-score + canonical order must equal the GPU list
    sub = np.r_[0:6, 1726:1731, 3447:3452]
    Df = orc.unpack_bf16(Db)
    approx = Df @ orc.unpack_bf16(Qb[sub]).T                      # [n, 16]
    for c, q in enumerate(sub):
        cand = np.nonzero(approx[:, c] >= s_np[q, -1] - 1e-4)[0]
        assert len(cand) >= k
        cs = orc.canonical_scores_pairs(Qb[q:q + 1], Db, cand[None, :])[0]
        o = np.lexsort((cand, -cs.astype(np.float64)))[:k]
        assert np.array_equal(cand[o], i_np[q]) and np.array_equal(cs[o], s_np[q])
    # (3b) completeness for EVERY query up to the 1e-4 band (an independent fp32 sweep of the whole corpus on the device)
    _completeness_sweep_all_queries(D, Q, s)
    # (4) idempotence of the shard merge: merging the list with itself shifted by n keeps the original
    gi = torch.stack([i, i + n])
    gs = torch.stack([s, s - 1.0])
    ms, mi = ops.merge_topk(gs, gi)
    assert torch.equal(mi, i) and torch.equal(ms, s)


def _completeness_sweep_all_queries(D, Q, s, chunk=65536):
    """EVERY query (not a subsample): no row that beats the list's k-th score by more than 1.1e-4 is missing from the list.  torch's own
    fp32 matmul (independent of this library's kernels; error ~1e-6 on these O(1) scores) sweeps the whole shard chunk by chunk on the
    device and counts, per query, the rows above k-th + 1e-4 +- 1e-5; the list's own count of such entries must lie between the two.
    (Rows inside the 1e-4 band around the cut are settled exactly on the query subsample, by the oracle.)"""
    Qf = Q.float()
    kth = s[:, -1].float()
    hi_thr, lo_thr = (kth + 1.1e-4)[None, :], (kth + 0.9e-4)[None, :]
    above_hi = torch.zeros(Q.shape[0], dtype=torch.int64, device=Q.device)
    above_lo = torch.zeros_like(above_hi)
    for lo in range(0, D.shape[0], chunk):
        a = D[lo:lo + chunk].float() @ Qf.T                       # [chunk, n_q] fp32
        above_hi += (a > hi_thr).sum(0)
        above_lo += (a > lo_thr).sum(0)
    in_list = (s > (kth + 1e-4)[:, None]).sum(1)
    bad = ((above_hi > in_list) | (in_list > above_lo)).nonzero().flatten()
    assert bad.numel() == 0, f"queries {bad[:8].tolist()}: rows above the cut {above_hi[bad[:8]].tolist()} / {above_lo[bad[:8]].tolist()}, in the list {in_list[bad[:8]].tolist()}"


def _check_against_oracle(D, Q, s, i, k, sub, offset=0):
    """The size-independent property scheme of test_nq_config_properties for any shape: canonical order, every returned
    score re-computed by the oracle bit for bit, and completeness on the queries `sub` (an fp32 sweep of the WHOLE shard --
    torch's own matmul, chunked on the device -- pre-selects every row within 1e-4 of the k-th score; the oracle re-scores
    those canonically and its canonical order must reproduce the list)."""
    n = D.shape[0]
    _completeness_sweep_all_queries(D, Q, s)
    s_np, i_np = s.cpu().numpy(), i.cpu().numpy() - offset
    ds = np.diff(s_np.astype(np.float64), axis=1)
    assert (ds <= 0).all()
    assert ((ds < 0) | (np.diff(i_np, axis=1) > 0)).all()
    assert i_np.min() >= 0 and i_np.max() < n
    Db, Qb = _bits(D), _bits(Q)
    ref = orc.canonical_scores_pairs(Qb, Db, i_np)
    assert np.array_equal(ref.view(np.uint32), s_np.view(np.uint32))
    Qs = Q[torch.as_tensor(sub, device=Q.device)].float()
    kth = torch.as_tensor(s_np[sub, -1], device=Q.device)
    cands = [[] for _ in sub]
    for lo in range(0, n, 1 << 20):
        a = D[lo:lo + (1 << 20)].float() @ Qs.T                      # [chunk, len(sub)]
        hit = (a >= (kth - 1e-4)[None, :]).nonzero()
        for c in range(len(sub)):
            cands[c].append((hit[hit[:, 1] == c, 0] + lo).cpu().numpy())
    for c, q in enumerate(sub):
        cand = np.concatenate(cands[c])
        assert len(cand) >= k
        cs = orc.canonical_scores_pairs(Qb[q:q + 1], Db, cand[None, :])[0]
        o = np.lexsort((cand, -cs.astype(np.float64)))[:k]
        assert np.array_equal(cand[o], i_np[q]) and np.array_equal(cs[o], s_np[q])


def test_msmarco_config_shard_and_full_properties():
    """BASELINE.json configs[2]: MS-MARCO passages 8,841,823 x 768, 6,980 queries, top-100 -- the single-GPU shape and the
    per-rank shard of its 8-way row sharding (1,105,228 rows, global ids with the rank's offset), against the oracle."""
    from ccrec_amd import ops
    n, nq, d, k = 8_841_823, 6_980, 768, 100
    D, Q = _gen(n, d, 1234), _gen(nq, d, 4321)
    sub = np.r_[0:6, 3488:3493, 6975:6980]
    index = ops.CorpusIndex(D)
    s, i = index.search(Q, k)
    st = index.last_stats()
    print(st)
    assert st["path"] == 1 and st["n_fallback"] == 0
    _check_against_oracle(D, Q, s, i, k, sub)
    # small batches against the same index: the streaming main pass (one query group up to 64 queries, two up to 128) with its
    # thresholds from the per-query kernel (8,640 sampled groups here: its LDS image needs the dynamic-LDS opt-in) -- the full batch's bits
    for nq_small in (1, 16, 100):
        ss, si = index.search(Q[:nq_small].contiguous(), k)
        st_s = index.last_stats()
        assert st_s["path"] == 1 and (st_s["ranges"], st_s["sublists"]) == (1, 2) and st_s["n_fallback"] == 0, st_s
        assert torch.equal(si, i[:nq_small]) and torch.equal(ss.view(torch.int32), s[:nq_small].view(torch.int32))
    # rank 3 of 8: rows [lo, hi) of the same corpus; the shard's list must be the full list restricted to the shard
    from ccrec_amd.dist import shard_bounds
    lo, hi = shard_bounds(n, 8, 3)
    assert hi - lo in (1_105_227, 1_105_228)
    shard = D[lo:hi]
    ix = ops.CorpusIndex(shard, global_row_offset=lo)
    s8, i8 = ix.search(Q, k)
    st8 = ix.last_stats()
    print(st8)
    assert st8["path"] == 1 and st8["n_fallback"] == 0 and int(i8.min()) >= lo and int(i8.max()) < hi
    _check_against_oracle(shard, Q, s8, i8, k, sub, offset=lo)
    inside = (i >= lo) & (i < hi)                                   # full-corpus winners that live in this shard ...
    for q in sub:
        mine = i[q][inside[q]]
        assert torch.equal(mine, i8[q][: mine.numel()])              # ... lead the shard's own list, in the same order


def test_config4_shard_properties():
    """BASELINE.json configs[3]: one rank's shard of the synthetic 50 M x 1024 corpus (6,250,000 rows, 12.8 GB bf16),
    10,000 queries, top-1000, against the oracle."""
    from ccrec_amd import ops
    n, nq, d, k = 6_250_000, 10_000, 1024, 1000
    D, Q = _gen(n, d, 1234 + 5), _gen(nq, d, 4321)
    lo = 5 * n                                                       # rank 5's global row offset (beyond int32)
    index = ops.CorpusIndex(D, global_row_offset=lo)
    s, i = index.search(Q, k)
    st = index.last_stats()
    print(st)
    assert st["path"] == 1 and st["n_fallback"] == 0
    sub = np.r_[0:6, 4998:5003, 9995:10000]
    _check_against_oracle(D, Q, s, i, k, sub, offset=lo)
    assert all(len(set(row.tolist())) == k for row in i.cpu().numpy()[::997])


def test_top1000_fused_equals_dense_at_1m():
    """config-4-shaped k (top-1000) on a 1M-row shard: the fused path must equal the exact dense path."""
    from ccrec_amd import ops
    n, nq, d, k = 1_000_000, 200, 1024, 1000
    D, Q = _gen(n, d, 7), _gen(nq, d, 8)
    index = ops.CorpusIndex(D, global_row_offset=5_000_000_000)   # global ids beyond int32
    s, i = index.search(Q, k, 2)
    st = index.last_stats()
    print(st)
    assert st["path"] == 1
    s2, i2 = index.search(Q, k, 1)
    assert torch.equal(i, i2) and torch.equal(s.view(torch.int32), s2.view(torch.int32))
    assert int(i.min()) >= 5_000_000_000


def test_prime_pantry_shaped_ranking_with_brand_blocks():
    """configs[0] shape: queries == corpus (9,862 items), block_dict = all items of the query's brand
    (1,960 Zipf-sized brands, self included), ranking() keeps 1001 entries.  Through the drop-in API."""
    from ccrec_amd.ms_marco_eval import ranking
    import os
    os.environ["CCREC_SIM_TYPE"] = "dot"
    n, d = 9862, 768
    g = torch.Generator().manual_seed(3)
    E = torch.randn(n, d, generator=g) / d ** 0.5
    rs = np.random.RandomState(5)
    brand = np.minimum((rs.zipf(1.3, size=n) - 1), 1959)          # a few very large brands, many tiny ones
    members = {b: np.nonzero(brand == b)[0].tolist() for b in np.unique(brand)}
    ids = [f"B{j:05d}" for j in range(n)]
    corpus = {ids[j]: j for j in range(n)}
    nq = 600                                                        # the API path is per-query python: keep it short
    queries = {ids[j]: j for j in range(nq)}
    block_dict = {ids[j]: [ids[m] for m in members[brand[j]]] for j in range(nq)}
    longest = max(len(v) for v in block_dict.values())
    assert longest > 100
    prof = ranking(corpus, queries, lambda rows: E[torch.as_tensor(rows, dtype=torch.long)], 2048, block_dict)
    assert len(prof) == nq and all(len(v) == 1001 for v in prof.values())
    sub = list(range(0, nq, 37))
    Eb = orc.pack_bf16(E.numpy())
    ref_i, ref_s = orc.canonical_search(Eb[sub], Eb, 1001, block=[members[brand[j]] for j in sub])
    for c, j in enumerate(sub):
        got = prof[ids[j]]
        assert [int(p[1:]) for p in got] == ref_i[c].tolist()
        assert np.array_equal(np.array(list(got.values()), np.float32).view(np.uint32), ref_s[c].view(np.uint32))
        assert ids[j] not in list(got)[: 1001 - longest]            # the query's own item is blocked (self-block)


@pytest.mark.parametrize("k", [100, 1001])
def test_clustered_non_iid_corpus_vs_oracle(k):
    """bench.py --data clustered (1,024 Gaussian clusters, log-normal row norms, 3 % exact duplicate rows) at 400 k rows:
    the fused path must stay on its fast route (no query sent to the exact fallback, survivors near the iid count) and
    reproduce the oracle bit for bit on a query subsample -- duplicates are exact ties, ordered by id."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import gen_rows
    from ccrec_amd import ops
    n, nq, d = 400_000, 700, 768
    D = ops.pack_bf16(gen_rows(n, d, 1234, "cuda", "clustered"))
    Q = ops.pack_bf16(gen_rows(nq, d, 4321, "cuda", "clustered"))
    norms = D.float().norm(dim=1)
    assert float(norms.max() / norms.median()) > 2.5               # a real spread of row norms
    index = ops.CorpusIndex(D)
    s, i = index.search(Q, k, 2)
    st = index.last_stats()
    print(st)
    assert st["path"] == 1 and st["n_fallback"] == 0
    assert st["n_candidates"] / nq < 40 * k                       # iid data: ~17 k (k = 100) / ~8 k (k = 1001) per query
    sub = np.r_[0:8, 346:354, 692:700]
    _check_against_oracle(D, Q, s, i, k, sub)
    s_np = s.cpu().numpy()
    assert (np.diff(s_np, axis=1) == 0).any()                       # the duplicates do produce exact ties in the lists


@pytest.mark.parametrize("k,big_cluster", [(100, False), (1001, False), (100, True)])
def test_topically_sorted_corpus_retries_on_the_fused_path(k, big_cluster):
    """A corpus in topical order (bench.py --data sorted: every cluster's rows contiguous, like passages of one article):
    whole tiles pass for a query and the sample misses its cluster, so candidate sub-lists overflow.  Such queries must be
    RETRIED on the fused path (thresholds re-tightened from their truncated lists), not sent to the 1.8-ms-per-query dense
    path; results stay bit-exact.  big_cluster: 16 clusters of 25,000 rows -- several whole tiles of one range pass."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import gen_rows
    from ccrec_amd import ops
    n, nq, d = 400_000, 600, 768
    rows = gen_rows(n, d, 1234, "cuda", "sorted")
    if big_cluster:   # fold the 1,024 clusters into 16 big ones by re-using centres: rows of cluster c move towards centre c // 64
        g = torch.Generator(device="cuda").manual_seed(777)
        centres = torch.randn(1024, d, generator=g, device="cuda") * d ** -0.5
        cid = (torch.arange(n, device="cuda") * 1024 // n).clamp_(max=1023)
        rows = rows + 0.8 * (centres[cid // 64 * 64] - centres[cid]) * rows.norm(dim=1, keepdim=True)
    D = ops.pack_bf16(rows)
    Q = ops.pack_bf16(gen_rows(nq, d, 4321, "cuda", "sorted"))
    index = ops.CorpusIndex(D)
    s, i = index.search(Q, k, 2)
    st = index.last_stats()
    print(st)
    assert st["path"] == 1
    if st["n_fallback"]:
        assert st["n_retried"] >= st["n_fallback"] - st["n_dense"] and st["n_dense"] <= max(8, st["n_fallback"] // 20), st
        assert st["ms_fallback"] < 60.0, st                          # the dense path alone would need ~0.3 ms per flagged query here
    sub = np.r_[0:8, 296:304, 592:600]
    _check_against_oracle(D, Q, s, i, k, sub)
    s1, i1 = index.search(Q[:64], k, 1)                              # exact dense path on a query block
    assert torch.equal(i[:64], i1) and torch.equal(s[:64].view(torch.int32), s1.view(torch.int32))
