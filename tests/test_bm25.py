"""BM25 sparse scorer.  CPU: the oracle restatement against fixtures produced by the reference's own
bm_25.BM25 / ranking_bm25 (tools/make_golden.py g12).  GPU: ccr_bm25_search against the oracle, bit for bit."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc


def _golden(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "g12_bm25.json")))
    dense = np.load(os.path.join(golden_dir, "g12_bm25_scores.npz"))["dense"]
    return g, dense


def test_oracle_bm25_matches_reference_fixture(golden_dir):
    g, dense = _golden(golden_dir)
    model = orc.bm25_fit(list(g["corpus"].values()), b=g["b"], k1=g["k1"])
    assert model["vocab"] == g["vocabulary"]                       # same analyser, same alphabetical feature order
    assert abs(model["doc_k"].mean() - g["k1"]) < 1e-9              # mean of k1 * (1 - b + b * len / avdl) is k1
    qtexts = list(g["queries"].values())
    got = np.stack([orc.bm25_scores(model, t) for t in qtexts])
    # the reference sums a query's terms in numpy's pairwise order (>= 8 terms), the oracle in ascending order:
    # fp64 results agree to 1e-12, i.e. to the last bit of fp32 except on rare rounding boundaries
    np.testing.assert_allclose(got, dense.astype(np.float32), rtol=2e-7, atol=0)
    assert (got != dense.astype(np.float32)).mean() < 1e-3
    oov = list(g["queries"]).index("q_oov")
    assert not got[oov].any() and len(orc.bm25_query_terms(model, g["queries"]["q_rep"])) == 3
    # ranking_bm25 profile: 1001 -> min(1001, N) entries, rank order, scores as the reference printed them
    ids, sc = orc.bm25_ranking(model, qtexts, 1001)
    cids = list(g["corpus"])
    for r, qid in enumerate(g["queries"]):
        ref = g["profile"][qid]
        assert len(ref) == ids.shape[1] == len(cids)
        ref_s = np.array(list(ref.values()), np.float32)
        np.testing.assert_allclose(sc[r], ref_s, rtol=2e-7, atol=0)
        # ids agree wherever the score is not tied with a neighbour (torch.sort leaves tie order unspecified)
        ref_i = [cids.index(p) for p in ref]
        uniq = np.flatnonzero(np.concatenate(([True], sc[r][1:] != sc[r][:-1])) & np.concatenate((sc[r][:-1] != sc[r][1:], [True])))
        assert all(ids[r][j] == ref_i[j] for j in uniq)
    d3, d7 = cids.index("d3"), cids.index("d7")                     # duplicate documents: the lower index ranks first
    for r in range(ids.shape[0]):
        row = list(ids[r])
        a, b_ = row.index(d3), row.index(d7)
        assert a < b_ and sc[r][a] == sc[r][b_] and all(d3 < ids[r][j] < d7 for j in range(a + 1, b_))


@pytest.mark.gpu
def test_hip_bm25_bit_exact_vs_oracle_and_profile(golden_dir):
    from ccrec_amd.bm25 import BM25, ranking_bm25
    g, dense = _golden(golden_dir)
    texts, qtexts = list(g["corpus"].values()), list(g["queries"].values())
    model = orc.bm25_fit(texts, b=g["b"], k1=g["k1"])
    hip = BM25(b=g["b"], k1=g["k1"]).fit(texts)
    assert hip.vocabulary_ == g["vocabulary"] and abs(hip.avdl - g["avdl"]) < 1e-12
    ref_i, ref_s = orc.bm25_ranking(model, qtexts, 50)
    s, i = hip.transform_topk(qtexts, 50)
    assert np.array_equal(i.cpu().numpy(), ref_i) and np.array_equal(s.cpu().numpy().view(np.uint32), ref_s.view(np.uint32))
    np.testing.assert_allclose(hip.transform(qtexts[0]), dense[0].astype(np.float32), rtol=2e-7)
    prof = ranking_bm25(g["corpus"], g["queries"])
    assert list(prof) == list(g["queries"])
    for qid in g["queries"]:
        np.testing.assert_allclose(list(prof[qid].values()), list(g["profile"][qid].values()), rtol=2e-7)
        assert len(prof[qid]) == len(g["profile"][qid])
    lazy = ranking_bm25(g["corpus"], g["queries"], lazy=True)       # the tensor-backed form: the same mapping
    assert lazy == prof and list(lazy) == list(prof) and lazy.top(qid, 3) == list(prof[qid])[:3]


@pytest.mark.gpu
def test_hip_bm25_larger_corpus_many_batches():
    """40k documents, 700 queries (several accumulator batches, long posting lists): ids and score bits vs the oracle."""
    from ccrec_amd.bm25 import BM25
    rs = np.random.RandomState(5)
    words = np.array([f"t{i}" for i in range(3000)])
    p = 1.0 / np.arange(1, 3001) ** 1.1
    p /= p.sum()
    texts = [" ".join(rs.choice(words, rs.randint(5, 60), p=p)) for _ in range(40000)]
    qtexts = [" ".join(rs.choice(words, rs.randint(1, 12), p=p)) for _ in range(700)]
    model = orc.bm25_fit(texts, 0.75, 1.2)
    hip = BM25(0.75, 1.2).fit(texts)
    s, i = hip.transform_topk(qtexts, 100)
    sub = list(range(0, 700, 23))
    ref_i, ref_s = orc.bm25_ranking(model, [qtexts[j] for j in sub], 100)
    assert np.array_equal(i.cpu().numpy()[sub], ref_i)
    assert np.array_equal(s.cpu().numpy()[sub].view(np.uint32), ref_s.view(np.uint32))


@pytest.mark.gpu
def test_parallel_analysis_builds_the_same_index():
    """BM25.fit(processes=n): the text analysis in worker processes (chunk vocabularies + per-document runs merged through one
    CSR -> CSC conversion) gives the same vocabulary, postings, idf, length factors and scores as the single-interpreter fit."""
    from ccrec_amd.bm25 import BM25
    rs = np.random.RandomState(9)
    words = np.array([f"t{i}" for i in range(2000)] + ["a", "über", "x2"])
    texts = [" ".join(rs.choice(words, rs.randint(0, 50))) + (" Don't STOP-me, now!" if j % 5 == 0 else "") for j in range(9000)]
    qtexts = [" ".join(rs.choice(words, rs.randint(1, 10))) for _ in range(200)]
    a = BM25(0.75, 1.2).fit(texts, processes=0)
    b = BM25(0.75, 1.2).fit(texts, processes=3, chunk_docs=1000)
    assert a.vocabulary_ == b.vocabulary_ and np.array_equal(a.indptr, b.indptr) and a.avdl == b.avdl
    assert torch.equal(a._doc_ids, b._doc_ids) and torch.equal(a._tf, b._tf) and torch.equal(a._doc_k, b._doc_k)
    sa, ia = a.transform_topk(qtexts, 100)
    sb, ib = b.transform_topk(qtexts, 100)
    assert torch.equal(ia, ib) and torch.equal(sa.view(torch.int32), sb.view(torch.int32))
    model = orc.bm25_fit(texts, 0.75, 1.2)
    ref_i, ref_s = orc.bm25_ranking(model, qtexts[:40], 100)
    assert np.array_equal(ib.cpu().numpy()[:40], ref_i) and np.array_equal(sb.cpu().numpy()[:40].view(np.uint32), ref_s.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 100, 1001])
def test_hip_bm25_sampled_selection_equals_the_dense_selection(k, monkeypatch):
    """The fused estimate-and-verify selection (csrc/ccr_bm25.hip: sampled pieces -> tau per row -> every tile filtered against tau in
    the scorer's registers -> in-LDS sort; rows it cannot finish are scored again with their fp32 rows stored and ranked by the exact
    dense selection, CCR_BM25_REDO_ROWS at a time) against the stored-rows path (CCR_BM25_DENSE_SELECT=1: every row stored and ranked
    exactly) and, on a subsample, the oracle: 120 k documents, queries that match most of the corpus (common terms), queries that
    match FEWER than k documents (rare terms: their top-k continues with zero scores in document order), an out-of-vocabulary query
    (all zeros), and 3 000 duplicated documents (mass ties at the cut -> the list floods -> the redo path)."""
    from ccrec_amd.bm25 import BM25
    rs = np.random.RandomState(17)
    words = np.array([f"t{i}" for i in range(6000)])
    p = 1.0 / np.arange(1, 6001) ** 1.05
    p /= p.sum()
    texts = [" ".join(rs.choice(words, rs.randint(5, 50), p=p)) for _ in range(120_000)]
    for j in range(3000):
        texts[40_000 + j] = texts[7]                                   # identical documents: identical scores
    qtexts = [" ".join(rs.choice(words, rs.randint(1, 12), p=p)) for _ in range(300)]
    qtexts += [" ".join(rs.choice(words[4000:], 2)) for _ in range(30)]   # rare terms only: tens of matching documents
    qtexts += ["zzzunknown qqqunknown", texts[7]]
    hip = BM25(0.75, 1.2).fit(texts)
    monkeypatch.delenv("CCR_BM25_DENSE_SELECT", raising=False)
    s, i = hip.transform_topk(qtexts, k)
    assert hip.last_stats()["path"] == "tile+fused_filter"
    monkeypatch.setenv("CCR_BM25_DENSE_SELECT", "1")
    s2, i2 = hip.transform_topk(qtexts, k)
    assert torch.equal(i, i2) and torch.equal(s.view(torch.int32), s2.view(torch.int32))
    assert hip.last_stats()["path"] == "tile+stored_rows"
    assert (s[-2] == 0).all() and i[-2].tolist() == list(range(k))        # nothing matches: zeros in document order
    if k > 1:
        assert i[-1, 0].item() == 7 and s[-1, 0].item() == s[-1, 1].item()     # the duplicates tie, the lowest document id first
    model = orc.bm25_fit(texts, 0.75, 1.2)
    sub = [0, 5, 299, 300, 317, 329, 330, 331]
    ref_i, ref_s = orc.bm25_ranking(model, [qtexts[j] for j in sub], k)
    assert np.array_equal(i.cpu().numpy()[sub], ref_i) and np.array_equal(s.cpu().numpy()[sub].view(np.uint32), ref_s.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("k", [10, 1001])
def test_hip_bm25_rows_the_filter_cannot_finish_are_redone_exactly(k, monkeypatch):
    """Estimates that fail, on purpose.  The sampled pieces of a 150 k-document corpus are documents [p * 65536, p * 65536 + 1024):
    term A occurs in every document OUTSIDE the pieces (the sample holds zeros only -> tau <= 0 -> 147 k positive scores flood the
    16 K-entry list), term B ONLY inside the pieces (tau far above what the other documents reach -> fewer than k pass at tau > 0).
    Both kinds of row must come back through the redo path -- scored again with their fp32 rows stored, CCR_BM25_REDO_ROWS = 3 rows
    at a time, exact dense selection -- with the bits of the stored-rows path; rows between them finish in the filter."""
    from ccrec_amd.bm25 import BM25
    n_docs, n_terms = 150_000, 300
    rs = np.random.RandomState(k)
    indptr, rows, counts, doc_k, idf = _random_postings(rs, n_docs, n_terms, dense_terms=3)
    docs = np.arange(n_docs)
    in_piece = (docs % 65536) < 1024
    extra_rows = [docs[~in_piece], docs[in_piece]]
    extra = [(r.astype(np.int32), rs.randint(1, 6, len(r)).astype(np.float32)) for r in extra_rows]
    indptr = np.concatenate([indptr, indptr[-1] + np.cumsum([len(r) for r, _ in extra])])
    rows = np.concatenate([rows] + [r for r, _ in extra])
    counts = np.concatenate([counts] + [c for _, c in extra])
    idf = np.concatenate([idf, [0.7, 2.5]])
    A, B = n_terms, n_terms + 1
    queries = []
    for j in range(40):
        # A / B rows: rare companions only (<= 125 postings each), so that the sampled rank is decided by A / B alone
        q = list(rs.choice(np.arange(150 if j % 4 < 2 else 6, n_terms), rs.randint(1, 6), replace=False))
        if j % 4 == 0:
            q.append(A)
        if j % 4 == 1:
            q.append(B)
        queries.append(np.sort(np.asarray(q)).astype(np.int32))
    monkeypatch.setenv("CCR_BM25_REDO_ROWS", "3")
    model = BM25.from_postings(indptr, rows, counts, doc_k, idf, k1=1.2)
    s, i = model.transform_terms_topk(queries, k)
    st = model.last_stats()
    assert st["path"] == "tile+fused_filter" and 10 <= st["rows_redone"] < 40, st      # the A rows always; the B rows when k > their hits
    monkeypatch.setenv("CCR_BM25_DENSE_SELECT", "1")
    s2, i2 = model.transform_terms_topk(queries, k)
    assert model.last_stats()["path"] == "tile+stored_rows"
    assert torch.equal(i, i2) and torch.equal(s.view(torch.int32), s2.view(torch.int32))


def _random_postings(rs, n_docs, n_terms, dense_terms):
    """Term-major postings of a random count matrix: `dense_terms` terms in ~half of the documents, the others with Zipf-like
    document frequencies down to a single posting and a few empty terms."""
    indptr, rows, counts = [0], [], []
    for t in range(n_terms):
        if t < dense_terms:
            df = int(n_docs * rs.uniform(0.3, 0.98))
        elif t % 97 == 5:
            df = 0
        else:
            df = max(1, int(n_docs * 0.2 / (t - dense_terms + 1) ** 1.1))
        r = np.sort(rs.choice(n_docs, df, replace=False)) if df else np.zeros(0, np.int64)
        rows.append(r)
        counts.append(rs.randint(1, 6, df))
        indptr.append(indptr[-1] + df)
    idf = np.log(n_docs / np.maximum(np.diff(indptr), 1).astype(np.float64))
    doc_k = 1.2 * (0.25 + 0.75 * rs.uniform(0.3, 2.5, n_docs))
    return np.asarray(indptr, np.int64), np.concatenate(rows).astype(np.int32), np.concatenate(counts).astype(np.float32), doc_k, idf


@pytest.mark.gpu
@pytest.mark.parametrize("n_docs,k", [(70_001, 100), (1024, 1024), (777, 50), (300_000, 1001)])
def test_hip_bm25_tile_scorer_equals_the_round_kernels(n_docs, k, monkeypatch):
    """The document-tile scorer (a wave per run of tiles, fp64 accumulators in LDS, cursors per term) against the round kernels with
    their fp64 rows in HBM (CCR_BM25_TILE=-1), ids and score bits: corpora that are not a multiple of the tile, smaller than one tile,
    exactly one tile; queries of 0, 1, 64 (one cursor group), 65 / 200 (four groups) and 257 terms (round kernels); terms without postings; a single
    query (every run is one tile: a binary search per tile and term); the other tile shapes."""
    from ccrec_amd.bm25 import BM25
    rs = np.random.RandomState(n_docs % 1000)
    n_terms = 400
    indptr, rows, counts, doc_k, idf = _random_postings(rs, n_docs, n_terms, dense_terms=6)
    queries = [np.sort(rs.choice(n_terms, rs.randint(1, 12), replace=False)).astype(np.int32) for _ in range(150)]
    queries += [np.zeros(0, np.int32), np.asarray([3], np.int32), np.asarray([5 + 97], np.int32),      # empty, one dense term, one empty term
                np.sort(rs.choice(n_terms, 64, replace=False)).astype(np.int32), np.arange(0, 12, dtype=np.int32)]
    results = {}
    for cfg in ("-1", "0", "1", "2"):
        monkeypatch.setenv("CCR_BM25_TILE", cfg)
        model = BM25.from_postings(indptr, rows, counts, doc_k, idf, k1=1.2)
        s, i = model.transform_terms_topk(queries, k)
        s1, i1 = model.transform_terms_topk(queries[7:8], k)           # one query: one tile per ticket
        assert torch.equal(i1[0], i[7]) and torch.equal(s1.view(torch.int32)[0], s.view(torch.int32)[7])
        results[cfg] = (s.view(torch.int32).cpu(), i.cpu())
    for cfg in ("0", "1", "2"):
        assert torch.equal(results[cfg][1], results["-1"][1]) and torch.equal(results[cfg][0], results["-1"][0]), cfg
    # the tile scorers above streamed the table of finished contributions (ccr_bm25_index_set_idf); without it (CCR_BM25_TABLE=0: tf and
    # K_d per posting, the division in the kernel) and with query weights that are not the index's idf (generic path): the same bits
    assert model.last_stats()["contribution_table"]
    monkeypatch.setenv("CCR_BM25_TILE", "0")
    monkeypatch.setenv("CCR_BM25_TABLE", "0")
    plain = BM25.from_postings(indptr, rows, counts, doc_k, idf, k1=1.2)
    monkeypatch.delenv("CCR_BM25_TABLE")
    s, i = plain.transform_terms_topk(queries, k)
    assert not plain.last_stats()["contribution_table"]
    assert torch.equal(i.cpu(), results["-1"][1]) and torch.equal(s.view(torch.int32).cpu(), results["-1"][0])
    other = BM25.from_postings(indptr, rows, counts, doc_k, idf, k1=1.2)
    other.idf = idf * 1.25                                              # the table holds idf, the queries ask for 1.25 idf
    s, i = other.transform_terms_topk(queries[:30], k)
    assert not other.last_stats()["contribution_table"]
    plain.idf = idf * 1.25
    s2, i2 = plain.transform_terms_topk(queries[:30], k)
    assert torch.equal(i, i2) and torch.equal(s.view(torch.int32), s2.view(torch.int32))
    # 65 and 200 distinct terms in a query (product descriptions as queries: the reference's prime_pantry set-up): four cursor groups per
    # lane, still the tile scorer; 257 terms: the whole call takes the round kernels -- same results for the other queries either way
    monkeypatch.setenv("CCR_BM25_TILE", "0")
    model = BM25.from_postings(indptr, rows, counts, doc_k, idf, k1=1.2)
    long_q = np.sort(rs.choice(n_terms, 65, replace=False)).astype(np.int32)
    longer_q = np.sort(rs.choice(n_terms, 200, replace=False)).astype(np.int32)
    s, i = model.transform_terms_topk(queries[:20] + [long_q, longer_q], k)
    assert model.last_stats()["path"].startswith("tile+")
    assert torch.equal(i[:20].cpu(), results["-1"][1][:20]) and torch.equal(s.view(torch.int32)[:20].cpu(), results["-1"][0][:20])
    too_long = np.sort(rs.choice(n_terms, 257, replace=False)).astype(np.int32)
    s2, i2 = model.transform_terms_topk(queries[:20] + [long_q, longer_q, too_long], k)
    assert model.last_stats()["path"] == "rounds+stored_rows"
    assert torch.equal(i2[:22], i) and torch.equal(s2.view(torch.int32)[:22], s.view(torch.int32))
    # the oracle's arithmetic on one row: fp64 sums in ascending term order, rounded once
    q = queries[3]
    acc = np.zeros(n_docs)
    for t in q:
        d = rows[indptr[t]:indptr[t + 1]]
        f = counts[indptr[t]:indptr[t + 1]].astype(np.float64)
        acc[d] = acc[d] + (f * idf[t]) * (1.2 + 1.0) / (f + doc_k[d])
    sc = acc.astype(np.float32)
    order = np.lexsort((np.arange(n_docs), -sc.astype(np.float64)))[:k]
    assert np.array_equal(results["0"][1][3].numpy(), order) and np.array_equal(results["0"][0][3].numpy().view(np.float32), sc[order])


@pytest.mark.gpu
def test_hip_bm25_index_create_rejects_unsorted_or_out_of_range_postings():
    """ccr_bm25_index_create validates what the tile scorer relies on (documents strictly ascending inside a term, ids inside the
    corpus) in one pass over the postings: a violation is CCR_ERR_INVALID with the position, not a silently wrong ranking."""
    from ccrec_amd import _lib
    from ccrec_amd.bm25 import BM25
    from ccrec_amd.ops import require_gpu
    rs = np.random.RandomState(3)
    indptr, rows, counts, doc_k, idf = _random_postings(rs, 5000, 40, dense_terms=2)

    def create(r):
        m = BM25(k1=1.2)
        m._lib, m.indptr, m.idf, m.n_docs = require_gpu(), indptr, idf, len(doc_k)
        return m._upload(r, counts, doc_k, len(idf))

    create(rows).transform_terms_topk([np.asarray([0, 3], np.int32)], 10)          # the valid index works
    swapped = rows.copy()
    p = int(indptr[1]) + 7                                                          # inside term 1
    swapped[p], swapped[p + 1] = swapped[p + 1], swapped[p]
    with pytest.raises(_lib.CcrError, match="ascend strictly inside a term .*first at posting %d" % p):
        create(swapped)
    dup = rows.copy()
    dup[p + 1] = dup[p]
    with pytest.raises(_lib.CcrError, match="ascend strictly"):
        create(dup)
    far = rows.copy()
    far[3] = 5000
    with pytest.raises(_lib.CcrError, match="outside \\[0, 5000\\)"):
        create(far)
    # a descent ACROSS a term boundary is the normal case and must pass (every list starts again at a low document)
    assert rows[indptr[1]] < rows[indptr[1] - 1]


@pytest.mark.gpu
def test_hip_bm25_more_queries_than_one_fused_batch(monkeypatch):
    """4 500 queries: the fused path takes 4 096 rows per batch (candidate lists, sample rows and term tables are per batch), so the
    second batch re-uses the first one's areas -- ids and score bits equal the stored-rows path's, and the first and last queries'
    equal a single-query call's."""
    from ccrec_amd.bm25 import BM25
    rs = np.random.RandomState(21)
    indptr, rows, counts, doc_k, idf = _random_postings(rs, 40_000, 300, dense_terms=4)
    queries = [np.sort(rs.choice(300, rs.randint(1, 9), replace=False)).astype(np.int32) for _ in range(4500)]
    monkeypatch.delenv("CCR_BM25_DENSE_SELECT", raising=False)
    model = BM25.from_postings(indptr, rows, counts, doc_k, idf, k1=1.2)
    s, i = model.transform_terms_topk(queries, 10)
    st = model.last_stats()
    assert st["path"] == "tile+fused_filter" and st["batches"] == 2, st
    for q in (0, 4095, 4096, 4499):
        s1, i1 = model.transform_terms_topk(queries[q:q + 1], 10)
        assert torch.equal(i1[0], i[q]) and torch.equal(s1.view(torch.int32)[0], s.view(torch.int32)[q]), q
    monkeypatch.setenv("CCR_BM25_DENSE_SELECT", "1")
    s2, i2 = model.transform_terms_topk(queries, 10)
    assert torch.equal(i, i2) and torch.equal(s.view(torch.int32), s2.view(torch.int32))


@pytest.mark.gpu
def test_hip_bm25_workspace_sizes(monkeypatch):
    """ccr_bm25_search_workspace_bytes_k sizes the workspace for one k: without the [queries][documents] score rows where the filter is fused
    (~130 KiB per query + the redo area), with them otherwise; ccr_bm25_search_workspace_bytes is the k-agnostic upper bound and a
    workspace of that size serves any k; a workspace that is too small is refused with CCR_ERR_WORKSPACE, not overrun."""
    import ctypes
    from ccrec_amd import _lib
    from ccrec_amd.bm25 import BM25
    rs = np.random.RandomState(33)
    n_docs = 200_000
    indptr, rows, counts, doc_k, idf = _random_postings(rs, n_docs, 200, dense_terms=3)
    monkeypatch.delenv("CCR_BM25_DENSE_SELECT", raising=False)
    model = BM25.from_postings(indptr, rows, counts, doc_k, idf, k1=1.2)
    lib, h = model._lib, model._h
    n_big = 3000                                                                      # 2.4 GB of score rows
    fused = int(lib.ccr_bm25_search_workspace_bytes_k(h, n_big, 12, 100))
    stored = int(lib.ccr_bm25_search_workspace_bytes_k(h, n_big, 12, 10_000))         # k beyond the candidate lists: score rows
    anyk = int(lib.ccr_bm25_search_workspace_bytes(h, n_big, 12))
    assert stored >= n_big * n_docs * 4 and anyk >= max(fused, stored)
    assert fused <= n_big * 160 * 1024 + (1 << 30) + n_big * 12 * 64 + (1 << 20) < stored          # lists 128 KiB + sample 16 KiB per query, 1 GiB of redo rows
    n_q = 300
    queries = [np.sort(rs.choice(200, rs.randint(1, 12), replace=False)).astype(np.int32) for _ in range(n_q)]
    mt = max(len(t) for t in queries)
    fused = int(lib.ccr_bm25_search_workspace_bytes_k(h, n_q, mt, 100))
    anyk = int(lib.ccr_bm25_search_workspace_bytes(h, n_q, mt))
    s, i = model.transform_terms_topk(queries, 100)
    # the same search through the C ABI with the k-agnostic workspace, and with one that is 4 KiB short
    q_ptr = np.zeros(n_q + 1, np.int64)
    q_ptr[1:] = np.cumsum([len(t) for t in queries])
    q_terms = np.concatenate(queries).astype(np.int32)
    q_idf = np.ascontiguousarray(model.idf[q_terms], np.float64)
    out_s = torch.empty(n_q, 100, dtype=torch.float32, device="cuda")
    out_i = torch.empty(n_q, 100, dtype=torch.int64, device="cuda")
    vp = ctypes.c_void_p
    stream = vp(torch.cuda.current_stream().cuda_stream)

    def call(nbytes):
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        return lib.ccr_bm25_search(h, q_ptr.ctypes.data_as(vp), q_terms.ctypes.data_as(vp), q_idf.ctypes.data_as(vp), n_q, 100, out_s.data_ptr(),
                                   out_i.data_ptr(), ws.data_ptr(), ws.numel(), stream)

    assert call(anyk) == 0
    torch.cuda.synchronize()
    assert torch.equal(out_i, i) and torch.equal(out_s.view(torch.int32), s.view(torch.int32))
    rc = call(fused - 4096)
    assert rc != 0 and "workspace" in lib.ccr_last_error().decode()


@pytest.mark.gpu
def test_hip_bm25_largest_sample_the_threshold_kernel_holds_in_lds(monkeypatch):
    """1,048,576 documents = 1,024 pieces of 1,024 documents, one of every 64 sampled: 16 pieces x 1,024 = 16,384 sampled scores per
    row = BM25_SAMPLE_MAX, i.e. exactly the 64 KiB of dynamic LDS bm25_threshold_kernel opts in to, under its static histogram
    (round-5 advisor: the full-size case had no test; the opt-in now carries head room).  The fused selection equals the stored-rows
    selection bit for bit."""
    import bench
    model, qs, df, _ = bench.bm25_workload(docs=1_048_576, queries=48, vocab=30_000)
    monkeypatch.delenv("CCR_BM25_DENSE_SELECT", raising=False)
    s, i = model.transform_terms_topk(qs, 1001)
    st = model.last_stats()
    assert st["path"] == "tile+fused_filter"
    monkeypatch.setenv("CCR_BM25_DENSE_SELECT", "1")
    s2, i2 = model.transform_terms_topk(qs, 1001)
    assert model.last_stats()["path"] == "tile+stored_rows"
    assert torch.equal(i, i2) and torch.equal(s.view(torch.int32), s2.view(torch.int32))
