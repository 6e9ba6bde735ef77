"""GPU parity of the round-2 entry points (through the C ABI) against the oracle and the reference's golden g14:
ccr_search_blocked (block lists of any length), ccr_search_sparse_prior / score_op (low-rank + sparse prior),
ccr_scores, the asynchronous search, the pooling backward, the BertBPR-shaped transform(D)."""
import operator
import os

import numpy as np
import pytest
import scipy.sparse as sps
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _bf16(bits):
    return torch.from_numpy(bits.view(np.int16)).view(torch.bfloat16).cuda()


def _bits(t):
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


def _rand_bits(n, d, seed, scale=None):
    g = torch.Generator().manual_seed(seed)
    return orc.pack_bf16((torch.randn(n, d, generator=g) * (scale if scale is not None else d ** -0.5)).numpy())


# ----------------------------------------------------------------------------------------- blocked ids of any length
def _block_lists(rs, nq, n, lengths):
    return [sorted(rs.choice(n, L, replace=False).tolist()) if L else [] for L in lengths[:nq]]


@pytest.mark.parametrize("n,k", [(20_000, 1001), (3_000, 1001), (300, 300)])
def test_search_blocked_any_length(n, k):
    """ms_marco_eval.py:224-227 blocks any number of ids: lists beyond k + len <= 4096 take the masked dense route."""
    from ccrec_amd import ops
    from ccrec_amd.ms_marco_eval import block_csr
    rs = np.random.RandomState(n)
    nq, d = 24, 128
    Db, Qb = _rand_bits(n, d, 1), _rand_bits(nq, d, 2)
    lengths = [0, 1, 5, min(n, 3095), min(n, 3096), min(n, 3200), min(n, 7000), n - k, n - k + 1, n - 3, n, 17] + [rs.randint(0, min(n, 60)) for _ in range(12)]
    lists = _block_lists(rs, nq, n, lengths)
    index = ops.CorpusIndex(_bf16(Db))
    ptr, idx = block_csr(lists, n)
    s, i = index.search_blocked(_bf16(Qb), k, ptr, idx)
    ref_i, ref_s = orc.canonical_search(Qb, Db, k, block=lists)
    assert np.array_equal(i.cpu().numpy(), ref_i)
    assert np.array_equal(s.cpu().numpy().view(np.uint32), ref_s.view(np.uint32))


def test_search_blocked_shard_ignores_foreign_ids():
    """Row-sharded use: every rank passes the whole (global) lists; a shard applies the ids inside [offset, offset + n)."""
    from ccrec_amd import ops
    n_total, d, nq, k = 9_000, 64, 10, 400
    Db, Qb = _rand_bits(n_total, d, 3), _rand_bits(nq, d, 4)
    rs = np.random.RandomState(7)
    lists = _block_lists(rs, nq, n_total, [0, 3, 50, 900, 4000, 6000, 8999, 9000, 2, 300])
    ref_i, ref_s = orc.canonical_search(Qb, Db, k, block=lists)
    parts_s, parts_i = [], []
    ptr = np.zeros(nq + 1, np.int64)
    ptr[1:] = np.cumsum([len(b) for b in lists])
    idx = np.concatenate([np.asarray(b, np.int64) for b in lists])
    for lo, hi in ((0, 3000), (3000, 6500), (6500, 9000)):
        index = ops.CorpusIndex(_bf16(Db[lo:hi]), global_row_offset=lo)
        s, i = index.search_blocked(_bf16(Qb), k, ptr, idx)
        assert int(i.min()) >= lo and int(i.max()) < hi
        parts_s.append(s)
        parts_i.append(i)
    ms, mi = ops.merge_topk(torch.stack(parts_s), torch.stack(parts_i))
    assert np.array_equal(mi.cpu().numpy(), ref_i)
    assert np.array_equal(ms.cpu().numpy().view(np.uint32), ref_s.view(np.uint32))


def test_ranking_api_blocks_more_than_3100_ids():
    """The drop-in ranking() with a 'brand' of 3,400 items (k + len = 4,401 > the 4,096-row fetch limit)."""
    from ccrec_amd.ms_marco_eval import ranking
    os.environ["CCREC_SIM_TYPE"] = "dot"
    n, d, nq = 6_000, 64, 12
    g = torch.Generator().manual_seed(9)
    E = torch.randn(n, d, generator=g) / d ** 0.5
    ids = [f"B{j:05d}" for j in range(n)]
    corpus = {ids[j]: j for j in range(n)}
    queries = {ids[j]: j for j in range(nq)}
    big = list(range(0, 3400))
    lists = [big if q % 2 == 0 else [q, q + 100] for q in range(nq)]
    block_dict = {ids[q]: [ids[m] for m in lists[q]] for q in range(nq)}
    prof = ranking(corpus, queries, lambda rows: E[torch.as_tensor(rows, dtype=torch.long)], 1024, block_dict)
    Eb = orc.pack_bf16(E.numpy())
    ref_i, ref_s = orc.canonical_search(Eb[:nq], Eb, 1001, block=lists)
    for q in range(nq):
        got = prof[ids[q]]
        assert [int(p[1:]) for p in got] == ref_i[q].tolist()
        assert np.array_equal(np.array(list(got.values()), np.float32).view(np.uint32), ref_s[q].view(np.uint32))


# ----------------------------------------------------------------------------------------- low-rank + sparse prior
def _g14(golden_dir):
    g = np.load(os.path.join(golden_dir, "g14_sparse_prior.npz"))
    nu, ni = g["U"].shape[0], g["V"].shape[0]
    P = sps.csr_matrix((g["prior_data"], g["prior_indices"], g["prior_indptr"]), shape=(nu, ni))
    return g, P


def test_assign_topk_with_sparse_prior_golden(golden_dir):
    """g14 = the reference's `_assign_topk(U @ V.T + sparse, k)`, evaluate_item_rec and score_op (bbpr.py:592-595)."""
    from ccrec_amd import ops
    from ccrec_amd.bbpr_transform import LowRankScore, score_op
    from ccrec_amd.rime_util import _assign_topk, evaluate_item_rec
    g, P = _g14(golden_dir)
    k = int(g["k"])
    low = LowRankScore(ops.pack_bf16(torch.from_numpy(g["U"]).cuda()), ops.pack_bf16(torch.from_numpy(g["V"]).cuda()))
    S = low + P
    csr = _assign_topk(S, k)
    assert csr.shape == P.shape and np.array_equal(csr.indptr, np.arange(0, csr.indices.size + 1, k))
    assert np.array_equal(csr.indices.reshape(-1, k), g["indices"])
    fin, ids = S.topk(k)
    np.testing.assert_allclose(fin.cpu().numpy(), g["topk_scores"], rtol=0, atol=2e-6 + 1e-4)
    # bit-exact against the oracle (fp64 finals, ids)
    ref_i, ref_f = orc.sparse_prior_search(orc.pack_bf16(g["U"]), orc.pack_bf16(g["V"]), g["prior_indptr"], g["prior_indices"],
                                           g["prior_data"], k)
    assert np.array_equal(ids.cpu().numpy(), ref_i)
    assert np.array_equal(fin.cpu().numpy().view(np.uint64), ref_f.view(np.uint64))
    # evaluate_item_rec(target, S, 1): the reference's metrics
    target = sps.csr_matrix((np.ones(g["target_indices"].size), g["target_indices"], g["target_indptr"]), shape=P.shape)
    out = evaluate_item_rec(target, S, 1)
    for name in ("prec", "recs/user", "item_cov", "item_ppl", "user_cov", "user_ppl", "obj_mean", "recall"):
        np.testing.assert_allclose(out[name], float(g["m_" + name.replace("/", "_")]), rtol=2e-6, atol=1e-9, err_msg=name)
    # score_op on both lazy types
    for op in ("max", "min", "sum"):
        np.testing.assert_allclose(score_op(low, op), float(g["low_" + op]), rtol=1e-5, atol=2e-4, err_msg=op)
        np.testing.assert_allclose(score_op(S, op), float(g["sum_" + op]), rtol=1e-6, atol=2e-4, err_msg=op)
        # max / min are canonical scores themselves (bit-equal to the oracle's); the sum is the exact fp64 sum of the bf16
        # products, which the oracle's sum of fp32-ROUNDED canonical scores only approximates (72,000 roundings)
        np.testing.assert_allclose(score_op(low, op), orc.score_op(orc.pack_bf16(g["U"]), orc.pack_bf16(g["V"]), op),
                                   rtol=0, atol=1e-4 if op == "sum" else 0)
    exact = float((g["U"].astype(np.float64).sum(0) * g["V"].astype(np.float64).sum(0)).sum())
    np.testing.assert_allclose(score_op(low, "sum"), exact, rtol=1e-12, atol=1e-10)
    # as_tensor() of the lazy sum is the dense fp64 matrix the reference materialises
    dense = S.as_tensor().cpu().numpy()
    assert dense.dtype == np.float64
    np.testing.assert_allclose(np.take_along_axis(dense, g["indices"], 1), g["topk_scores"], rtol=0, atol=1e-4)


def test_sparse_prior_random_vs_oracle_both_routes():
    """Short rows (over-fetch + merge) and rows with more prior entries than any over-fetch holds (masked dense route),
    negative values, ties between prior and plain columns, ids in a shard with an offset."""
    from ccrec_amd import ops
    rs = np.random.RandomState(5)
    n, d, nq, k = 6_000, 64, 30, 40
    g = torch.Generator().manual_seed(5)
    D = torch.round(torch.randn(n, d, generator=g) * 4) / 8        # coarse grid: exact ties between columns
    Q = torch.round(torch.randn(nq, d, generator=g) * 4) / 8
    Db, Qb = orc.pack_bf16(D.numpy()), orc.pack_bf16(Q.numpy())
    lens = [0, 1, 3, 4096, 4057, 4056, 200, 4000] + [int(rs.randint(0, 30)) for _ in range(nq - 8)]
    indptr = np.zeros(nq + 1, np.int64)
    indptr[1:] = np.cumsum(lens)
    indices = np.concatenate([np.sort(rs.choice(n, L, replace=False)) for L in lens]).astype(np.int64)
    data = np.round(rs.uniform(-3, 3, indices.size) * 8) / 8          # multiples of 1/8: finals tie with plain scores
    data[rs.rand(indices.size) < 0.1] = 1e5
    ref_i, ref_f = orc.sparse_prior_search(Qb, Db, indptr, indices, data, k)
    index = ops.CorpusIndex(_bf16(Db), global_row_offset=1 << 33)
    fin, ids = index.search_sparse_prior(_bf16(Qb), k, indptr, indices + (1 << 33), data)
    assert np.array_equal(ids.cpu().numpy() - (1 << 33), ref_i)
    assert np.array_equal(fin.cpu().numpy().view(np.uint64), ref_f.view(np.uint64))
    with pytest.raises(Exception, match="4096"):
        bad = np.arange(nq + 1, dtype=np.int64) * 4097
        index.search_sparse_prior(_bf16(Qb), k, bad, np.tile(np.arange(4097), nq) + (1 << 33), np.zeros(4097 * nq))


def test_assign_topk_accepts_rime_lite_expression_objects():
    """Objects shaped like rime_lite's ElementWiseExpression(add, [MatMulExpression, LazySparseMatrix]) dispatch to the
    sparse-prior search (score_array.py:300-339); anything else still raises."""
    from ccrec_amd.rime_util import _assign_topk

    class Dense:
        def __init__(self, c):
            self.c, self.shape = np.asarray(c), np.asarray(c).shape

    class MatMul:
        def __init__(self, left, right):
            self.left, self.right, self.shape = left, right, (left.shape[0], right.shape[1])

    class Sparse:
        def __init__(self, c):
            self.c, self.shape = c, c.shape

    class Add:
        op = operator.add

        def __init__(self, a, b):
            self.children, self.shape = [a, b], a.shape

    g = torch.Generator().manual_seed(2)
    U = (torch.randn(9, 40, generator=g) / 6).to(torch.bfloat16).float().numpy()     # dim 40: padded to a multiple of 8
    V = (torch.randn(500, 40, generator=g) / 6).to(torch.bfloat16).float().numpy()
    P = sps.random(9, 500, density=0.01, random_state=3, format="csr") * 7.0
    S = Add(MatMul(Dense(U), Dense(V.T)), Sparse(P))
    csr = _assign_topk(S, 6)
    Pc = sps.csr_matrix(P)
    Pc.sort_indices()
    ref_i, _ = orc.sparse_prior_search(orc.pack_bf16(U), orc.pack_bf16(V), Pc.indptr, Pc.indices, Pc.data, 6)
    assert np.array_equal(csr.indices.reshape(9, 6), ref_i)
    with pytest.raises(NotImplementedError):
        _assign_topk(Sparse(P), 3)


# ----------------------------------------------------------------------------------------- ccr_scores, async search
def test_scores_public_entry_modes():
    from ccrec_amd import ops
    Db, Qb = _rand_bits(1500, 128, 11), _rand_bits(70, 128, 12)
    index = ops.CorpusIndex(_bf16(Db))
    can = index.scores(_bf16(Qb), "canonical").cpu().numpy()
    assert np.array_equal(can.view(np.uint32), orc.canonical_scores(Qb, Db).view(np.uint32))
    mf = index.scores(_bf16(Qb), "mfma").cpu().numpy()
    assert np.abs(mf - can).max() < 1e-6
    with pytest.raises(KeyError):
        index.scores(_bf16(Qb), "fp8")


def test_async_search_equals_sync_including_many_flagged_queries():
    """CCR_SEARCH_ASYNC: no host synchronisation inside the call; the lists of un-flagged queries are final on the stream,
    ccr_search_finish (which waits for the search's own event) re-does the flagged ones.  Mass ties (identical rows) flag every query."""
    from ccrec_amd import ops
    n, d, k = 70_000, 128, 64
    Db, Qb = _rand_bits(n, d, 21), _rand_bits(300, d, 22)
    index = ops.CorpusIndex(_bf16(Db))
    s0, i0 = index.search(_bf16(Qb), k, 2)
    assert index.last_stats()["path"] == 1
    s1, i1 = index.search(_bf16(Qb), k, 2, defer=True)
    assert index._deferred is not None
    s1b, i1b = index.search(_bf16(Qb), k, 2)      # a search on an index with a deferred one pending completes that one first
    assert index._deferred is None and torch.equal(i1b, i0)
    index.finish()                                # nothing pending any more: a no-op
    st = index.last_stats()
    assert torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32)) and st["path"] == 1
    # 9,000 identical rows in front: every query has > rescore_cap rows tied at its cut -> all flagged (>> 16)
    Dt = Db.copy()
    Dt[:9000] = Dt[0]
    Qt = np.tile(Dt[0], (40, 1))
    index = ops.CorpusIndex(_bf16(Dt))
    s2, i2 = index.search(_bf16(Qt), k, 2, defer=True)
    index.finish()
    st = index.last_stats()
    assert st["n_fallback"] > 16, st
    ref_i, ref_s = orc.canonical_search(Qt[:3], Dt, k)
    assert np.array_equal(i2.cpu().numpy()[:3], ref_i) and np.array_equal(s2.cpu().numpy()[:3].view(np.uint32), ref_s.view(np.uint32))
    assert torch.equal(i2[0].expand_as(i2), i2)
    # synchronous form, 100 flagged queries: the dense path scores them in chunks of >= 64 inside the (free) candidate area
    Qh = _bf16(np.concatenate([Qt, Qb[:30], Qt, Qt[:20]]))
    s4, i4 = index.search(Qh, k, 2)
    assert index.last_stats()["n_dense"] == 100
    s5, i5 = index.search(Qh, k, 1)
    assert torch.equal(i4, i5) and torch.equal(s4.view(torch.int32), s5.view(torch.int32))
    # a handful of flagged queries: every OTHER query's list is final on the stream, finish() completes the flagged ones
    Qm = np.concatenate([Qb[:100], Qt[:5]])
    s3, i3 = index.search(_bf16(Qm), k, 2, defer=True)
    torch.cuda.synchronize()
    before = i3.cpu().numpy().copy()
    index.finish()
    st = index.last_stats()
    assert 1 <= st["n_fallback"] <= 16 and st["n_retried"] + st["n_dense"] >= st["n_fallback"]
    ref_i, ref_s = orc.canonical_search(Qm, Dt, k)
    assert np.array_equal(before[:98], ref_i[:98])                      # un-flagged queries were final before finish()
    assert np.array_equal(i3.cpu().numpy(), ref_i) and np.array_equal(s3.cpu().numpy().view(np.uint32), ref_s.view(np.uint32))


def test_nan_row_with_norm_path_equals_dense():
    """A NaN corpus row must leave a NaN norm bound (and the index a NaN maximum: fmaxf would drop it): every path then agrees."""
    from ccrec_amd import ops
    n, d, k = 40_000, 64, 20
    g = torch.Generator().manual_seed(4)
    D = torch.randn(n, d, generator=g) / 8
    D[12345, 7] = float("nan")
    Q = (torch.randn(16, d, generator=g) / 8).cuda()
    nb = torch.empty(n, device="cuda")
    Db = ops.pack_bf16(D.cuda(), norm_bounds=nb)
    assert torch.isnan(nb[12345]).item() and int(torch.isnan(nb).sum()) == 1
    Qb = ops.pack_bf16(Q)
    s0, i0 = ops.CorpusIndex(Db, norm_bounds=nb).search(Qb, k, 2)
    s1, i1 = ops.CorpusIndex(Db).search(Qb, k, 1)
    assert torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))
    s2, i2 = ops.CorpusIndex(Db).search(Qb, k, 2)                      # the index's own norm pass must keep the NaN too
    assert torch.equal(i2, i1) and torch.equal(s2.view(torch.int32), s1.view(torch.int32))


def test_colsum_bf16():
    from ccrec_amd import ops
    x = _rand_bits(5003, 200, 31, scale=1.0)
    got = ops.colsum_bf16(_bf16(x)).cpu().numpy()
    ref = orc.unpack_bf16(x).astype(np.float64).sum(0)
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-9)


# ----------------------------------------------------------------------------------------- pooling backward, tower training
def test_meanpool_backward_matches_torch_formula():
    from ccrec_amd import ops
    B, L, d = 13, 37, 96
    g = torch.Generator().manual_seed(1)
    for dtype, tol in ((torch.float32, 1e-6), (torch.bfloat16, 1e-2), (torch.float16, 2e-3)):
        h = torch.randn(B, L, d, generator=g).to(dtype).cuda().requires_grad_(True)
        mask = (torch.rand(B, L, generator=g) < 0.7).long()
        mask[:, 0] = 1
        mask = mask.cuda()
        w = torch.randn(B, d, generator=g).cuda()
        out = ops.meanpool(h, mask)
        (out * w).sum().backward()
        h2 = h.detach().clone().requires_grad_(True)
        ref = h2.float().masked_fill(~mask[..., None].bool(), 0.0).sum(1) / mask.sum(1)[..., None]   # item_tower.py:141-146
        (ref * w).sum().backward()
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(h.grad.float().cpu().numpy(), h2.grad.float().cpu().numpy(), rtol=tol, atol=tol)


def test_multiple_nrl_loss_backprops_through_the_tower():
    """The reference's training forward calls the tower with gradients on (bbpr.py:130-141, 195-197): with
    output_step mean_pooling the encoder must receive gradients (the fused pooling used to detach)."""
    from transformers import BertConfig, BertModel
    from ccrec_amd.bbpr_loss import multiple_nrl_loss
    from ccrec_amd.item_tower import NaiveItemTower
    torch.manual_seed(0)
    cfg = BertConfig(vocab_size=300, hidden_size=64, num_hidden_layers=1, num_attention_heads=4, intermediate_size=128,
                     max_position_embeddings=32, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    tower = NaiveItemTower(BertModel(cfg), torch.nn.LayerNorm(64, elementwise_affine=False)).cuda().train()
    g = torch.Generator().manual_seed(3)
    B, L = 16, 12
    toks = [{"input_ids": torch.randint(1, 300, (B, L), generator=g).cuda(),
             "attention_mask": (torch.arange(L)[None, :] < torch.randint(3, L + 1, (B, 1), generator=g)).long().cuda()} for _ in range(3)]

    def run(pool):
        tower.zero_grad()
        embs = [pool(t) for t in toks]
        loss = multiple_nrl_loss(*embs, inv_temperature=2.0, sim_type="dot")
        loss.backward()
        return float(loss), {n: p.grad.detach().clone() for n, p in tower.named_parameters() if p.grad is not None}

    def fused(t):
        out = tower(**t, output_step="mean_pooling")
        assert out.requires_grad
        return out

    def torch_formula(t):
        h = tower.cls_model(**t).last_hidden_state
        m = t["attention_mask"]
        return h.masked_fill(~m[..., None].bool(), 0.0).sum(1) / m.sum(1)[..., None]

    l0, g0 = run(fused)
    l1, g1 = run(torch_formula)
    assert abs(l0 - l1) < 1e-3 * max(1.0, abs(l1))
    assert g0.keys() == g1.keys() and len(g0) > 10
    for name in g1:
        a, b = g0[name].cpu().numpy(), g1[name].cpu().numpy()
        assert np.abs(a - b).max() <= 2e-2 * max(1e-6, np.abs(b).max()), name
    with torch.no_grad():   # eval / no_grad keeps the fused kernel without a graph
        assert not tower(**toks[0], output_step="mean_pooling").requires_grad


# ----------------------------------------------------------------------------------------- BertBPR-shaped transform(D)
class _Tok:
    def __call__(self, texts, truncation=True, padding="max_length", max_length=16, return_tensors="pt"):
        ids = [[1] + [2 + (sum(map(ord, w)) % 200) for w in t.split()][: max_length - 2] + [3] for t in texts]
        L = max_length if padding == "max_length" else max(len(r) for r in ids)
        input_ids, mask = torch.zeros(len(ids), L, dtype=torch.long), torch.zeros(len(ids), L, dtype=torch.long)
        for r, row in enumerate(ids):
            input_ids[r, : len(row)] = torch.tensor(row)
            mask[r, : len(row)] = 1
        return {"input_ids": input_ids, "attention_mask": mask}


class _Dataset:
    """The three attributes of a rime Dataset that BertBPR.transform touches (bbpr.py:287-291, 592-595)."""

    def __init__(self, user_hist, item_ids, prior):
        import pandas as pd
        self.user_in_test = pd.DataFrame({"_hist_items": user_hist})
        self.item_in_test = pd.DataFrame(index=pd.Index(item_ids))
        self.prior_score = prior


def test_bertbpr_transform_dataset_and_get_all_embeddings(monkeypatch):
    import pandas as pd
    from transformers import BertConfig, BertModel
    from ccrec_amd.bbpr_transform import BertBPR, LowRankPlusSparse, LowRankScore
    from ccrec_amd.item_tower import NaiveItemTower
    from ccrec_amd.rime_util import evaluate_item_rec
    monkeypatch.setenv("CCREC_SIM_TYPE", "dot")
    monkeypatch.setenv("CCREC_EMBEDDING_TYPE", "mean_pooling")
    torch.manual_seed(0)
    cfg = BertConfig(vocab_size=256, hidden_size=64, num_hidden_layers=1, num_attention_heads=4, intermediate_size=128,
                     max_position_embeddings=32)

    class Recording(NaiveItemTower):
        record = []

        def forward(self, *a, **k):
            out = super().forward(*a, **k)
            self.record.append(out.detach().float().cpu().numpy())
            return out

    tower = Recording(BertModel(cfg).eval(), torch.nn.LayerNorm(64, elementwise_affine=False)).cuda()
    rs = np.random.RandomState(1)
    words = [f"w{i}" for i in range(80)]
    n_items = 150
    item_df = pd.DataFrame({"TITLE": [" ".join(rs.choice(words, rs.randint(2, 9))) for _ in range(n_items)]},
                           index=[f"i{j}" for j in range(n_items)])
    bbpr = BertBPR(item_df, tower, _Tok(), max_length=16, batch_size=64)
    # get_all_embeddings: batches of 64 -> [150, 64] packed rows == pack(recorded encoder outputs)
    all_emb = bbpr.get_all_embeddings(tower, 64)
    rec = np.concatenate(Recording.record, 0)
    assert all_emb.shape == (n_items, 64) and all_emb.dtype == torch.bfloat16 and len(Recording.record) == 3
    assert np.array_equal(_bits(all_emb), orc.pack_bf16(rec))
    # transform(D): users = first history item, items = item_in_test order; + prior -> evaluate_item_rec(..., 1)
    users = [[f"i{u}", f"i{(u * 7) % n_items}"] for u in (3, 77, 149, 0, 12)]
    items = [f"i{j}" for j in rs.permutation(n_items)[:120]]
    prior = sps.random(5, 120, density=0.03, random_state=2, format="csr") * 1e5
    D = _Dataset(users, items, prior)
    Recording.record = []
    S = bbpr.transform(D)
    assert isinstance(S, LowRankScore) and S.shape == (5, 120)
    rec = np.concatenate(Recording.record, 0)
    Eb = orc.pack_bf16(rec)
    row = {f"i{j}": j for j in range(n_items)}
    Ub, Vb = Eb[[row[u[0]] for u in users]], Eb[[row[i] for i in items]]
    assert np.array_equal(_bits(S.user), Ub) and np.array_equal(_bits(S.item), Vb)
    assert np.array_equal(S.as_tensor().cpu().numpy().view(np.uint32), orc.canonical_scores(Ub, Vb).view(np.uint32))
    R = S + D.prior_score
    assert isinstance(R, LowRankPlusSparse)
    Pc = sps.csr_matrix(prior)
    Pc.sort_indices()
    ref_i, _ = orc.sparse_prior_search(Ub, Vb, Pc.indptr, Pc.indices, Pc.data, 1)
    target = sps.csr_matrix((np.ones(5), (np.arange(5), ref_i[:, 0])), shape=(5, 120))
    out = evaluate_item_rec(target, R, 1)
    assert out["prec"] == 1.0 and out["recall"] > 0
    # the debug branches of the reference's transform (bbpr.py:510-521)
    bbpr.random = True
    assert tuple(bbpr.transform(D).shape) == (5, 120)


def test_lazy_scores_slice_like_rime_lite_and_odd_widths_take_the_fused_path():
    """LazyScoreBase supports row slicing + collate_fn for batching (score_array.py:98-100, 227-231, 332-338); factor widths
    that are not a multiple of 64 (here 100) are zero-padded so the fused MFMA search applies."""
    from ccrec_amd import ops
    from ccrec_amd.bbpr_transform import LowRankPlusSparse, LowRankScore
    from ccrec_amd.rime_util import _assign_topk
    g = torch.Generator().manual_seed(8)
    nu, ni, d, k = 37, 6000, 100, 9
    U = (torch.randn(nu, d, generator=g) / 10).to(torch.bfloat16).float()
    V = (torch.randn(ni, d, generator=g) / 10).to(torch.bfloat16).float()
    Ub, Vb = orc.pack_bf16(U.numpy()), orc.pack_bf16(V.numpy())
    ref_i, _ = orc.canonical_search(Ub, Vb, k)
    csr = _assign_topk((U, V), k)                                       # (U, V) means U @ V.T
    assert np.array_equal(csr.indices.reshape(nu, k), ref_i)
    pad = torch.nn.functional.pad
    S = LowRankScore(ops.pack_bf16(pad(U, (0, 28)).cuda()), ops.pack_bf16(pad(V, (0, 28)).cuda()))
    _, ids = S.topk(k)
    assert S.index().last_stats()["path"] == 1                           # fused: 128-wide padded factors
    assert np.array_equal(ids.cpu().numpy(), ref_i)
    # slicing: int, slice, index array (modulo the row count, as LazyDenseMatrix does), then collate_fn
    parts = [S[0], S[1:20], S[np.array([20, 21, 22 + nu])], S[23:]]
    assert [p.shape for p in parts] == [(1, ni), (19, ni), (3, ni), (nu - 23, ni)]
    back = LowRankScore.collate_fn(parts)
    assert back.shape == S.shape and torch.equal(back.user, S.user)
    _, ids2 = back.topk(k)
    assert np.array_equal(ids2.cpu().numpy(), ref_i)
    P = sps.random(nu, ni, density=0.002, random_state=4, format="csr") * 3.0
    R = S + P
    Rs = LowRankPlusSparse.collate_fn([R[:10], R[10:]])
    f1, i1 = R.topk(k)
    f2, i2 = Rs.topk(k)
    assert torch.equal(i1, i2) and torch.equal(f1, f2)
    Pc = sps.csr_matrix(P)
    Pc.sort_indices()
    ref_pi, _ = orc.sparse_prior_search(Ub, Vb, Pc.indptr, Pc.indices, Pc.data, k)
    assert np.array_equal(i1.cpu().numpy(), ref_pi)


def test_empty_and_degenerate_inputs():
    """Zero queries, one row, k == n_rows, empty block lists and an empty prior through every round-2 entry point."""
    import os
    from ccrec_amd import ops
    from ccrec_amd.ms_marco_eval import ranking
    Db, Qb = _rand_bits(500, 64, 1), _rand_bits(5, 64, 2)
    index = ops.CorpusIndex(_bf16(Db))
    empty_q = _bf16(Qb)[:0]
    s, i = index.search(empty_q, 7)
    assert tuple(s.shape) == (0, 7) and tuple(i.shape) == (0, 7)
    s, i = index.search_blocked(empty_q, 7, np.zeros(1, np.int64), np.zeros(0, np.int64))
    assert tuple(i.shape) == (0, 7)
    f, i = index.search_sparse_prior(empty_q, 7, np.zeros(1, np.int64), np.zeros(0, np.int64), np.zeros(0))
    assert tuple(f.shape) == (0, 7) and f.dtype == torch.float64
    assert tuple(index.scores(empty_q).shape) == (0, 500)
    # no blocked ids / no prior entries at all == the plain search (scores as fp64 for the prior form)
    s0, i0 = index.search(_bf16(Qb), 500)                                # k == n_rows: the whole corpus, ranked
    assert sorted(i0[0].tolist()) == list(range(500))
    s1, i1 = index.search_blocked(_bf16(Qb), 500, np.zeros(6, np.int64), np.zeros(0, np.int64))
    assert torch.equal(i0, i1) and torch.equal(s0, s1)
    f2, i2 = index.search_sparse_prior(_bf16(Qb), 500, np.zeros(6, np.int64), np.zeros(0, np.int64), np.zeros(0))
    assert torch.equal(i0, i2) and torch.equal(s0.double(), f2)
    # a one-row corpus
    one = ops.CorpusIndex(_bf16(Db[:1]))
    s, i = one.search(_bf16(Qb), 1)
    assert i.flatten().tolist() == [0] * 5
    with pytest.raises(Exception, match="k="):
        one.search(_bf16(Qb), 2)                                          # k > n_rows is an error, as documented
    # ranking() with no queries: an empty profile (the corpus is still encoded, as the reference does)
    os.environ["CCREC_SIM_TYPE"] = "dot"
    table = torch.randn(20, 768)
    prof = ranking({f"p{j}": j for j in range(20)}, {}, lambda rows: table[torch.as_tensor(rows, dtype=torch.long)], 8)
    assert prof == {}


@pytest.mark.parametrize("route", ["pack_bounds", "own_pass"])
def test_norm_outlier_rows_keep_the_fused_path(route):
    """The filter margins are per 256-row tile (cq * tile norm), not per shard: a few rows with norms 30x - 1000x the rest
    must neither flood the candidate lists nor send queries to the dense path (one global max norm did: 7.5 s per NQ step at
    100x), and the result stays the canonical one.  Both index builds: norm bounds from the pack kernel, the index's own pass."""
    from ccrec_amd import ops
    n, d, nq, k = 600_000, 768, 300, 100
    g = torch.Generator(device="cuda").manual_seed(77)
    D = torch.randn(n, d, generator=g, device="cuda") / d ** 0.5
    D[1000] *= 100.0
    D[300_000] *= 1000.0
    D[450_000:450_050] *= 30.0
    Q = torch.randn(nq, d, generator=g, device="cuda") / d ** 0.5
    Q[7] *= 50.0                                  # a query norm outlier must only widen its own margin
    nb = torch.empty(n, device="cuda") if route == "pack_bounds" else None
    Db = ops.pack_bf16(D, norm_bounds=nb)
    Qb = ops.pack_bf16(Q)
    index = ops.CorpusIndex(Db, norm_bounds=nb)
    s, i = index.search(Qb, k)
    st = index.last_stats()
    assert st["path"] == 1 and st["n_fallback"] == 0, st
    assert st["n_candidates"] / nq < 60 * k, st    # the 1 052 outlier rows pass for every query, little else beyond the usual
    s1, i1 = index.search(Qb, k, 1)
    assert torch.equal(i, i1) and torch.equal(s.view(torch.int32), s1.view(torch.int32))
    bits = lambda t: t.view(torch.int16).cpu().numpy().view(np.uint16)   # noqa: E731
    ref_i, ref_s = orc.canonical_search(bits(Qb[:4]), bits(Db), k)
    assert np.array_equal(i[:4].cpu().numpy(), ref_i) and np.array_equal(s[:4].cpu().numpy(), ref_s)


def test_cancelling_components_margins_are_rigorous():
    """Rows whose large components cancel exactly in the canonical (fp64) score but not in the MFMA's fp32 accumulation: +M at
    element 0 and -M at element 100 (different MFMA K blocks), every query with q[0] == q[100].  The MFMA score of such a row is
    off by ~ulp(M) -- far more than the spacing of the top scores -- so the MFMA order is wrong around them; the per-tile /
    per-row margins must still deliver the canonical result on the fused path (no query may need the dense path)."""
    from ccrec_amd import ops
    n, d, nq, k = 120_000, 128, 64, 10
    g = torch.Generator().manual_seed(9)
    D = torch.randn(n, d, generator=g) / d ** 0.5
    Q = torch.randn(nq, d, generator=g) / d ** 0.5
    Q[:, 0] = Q[:, 100] = 0.5
    rows = torch.randint(0, n, (6,), generator=g)      # (dozens of such rows flood the lists of a corpus this small: every query
    for j, r in enumerate(rows.tolist()):             #  is then retried and ends on the dense path -- exact as well, but not the point)
        m = float(2 ** (8 + 2 * j))                  # M = 256 ... 262144
        D[r, 0], D[r, 100] = m, -m
        D[r, 1:100] *= 3.0                            # lift some of them into the top-k region
    Db, Qb = ops.pack_bf16(D.cuda()), ops.pack_bf16(Q.cuda())
    bits = lambda t: t.view(torch.int16).cpu().numpy().view(np.uint16)   # noqa: E731
    # the premise: MFMA scores of the cancelling rows are visibly wrong
    index = ops.CorpusIndex(Db)
    mf = index.scores(Qb[:4], mode="mfma")[:, rows.cuda()].cpu().double()
    ex = index.scores(Qb[:4], mode="canonical")[:, rows.cuda()].cpu().double()
    assert float((mf - ex).abs().max()) > 1e-3
    for flag in (0, 2):
        s, i = index.search(Qb, k, flag)
        st = index.last_stats()
        assert st["path"] == 1 and st["n_dense"] == 0, st
        s1, i1 = index.search(Qb, k, 1)
        assert torch.equal(i, i1) and torch.equal(s.view(torch.int32), s1.view(torch.int32))
    ref_i, ref_s = orc.canonical_search(bits(Qb[:8]), bits(Db), k)
    assert np.array_equal(i[:8].cpu().numpy(), ref_i) and np.array_equal(s[:8].cpu().numpy(), ref_s)
    assert len(set(ref_i.ravel().tolist()) & set(rows.tolist())) > 0   # cancelling rows do appear in the top-k


def test_flooded_lists_are_retried_in_groups():
    """150 of the 2,344 tiles pass the filter completely for EVERY query (each holds a row whose norm is 10^5 x the rest, so
    the tile's margin is wider than any score; 38,400 survivors per query): sub-lists of the first attempt overflow for all
    1,024 queries, all are flagged.  The retry must not hand them to the dense path: it takes them in groups of a few query blocks, each with the whole
    candidate area, and finishes every query on the fused path -- bit-exact."""
    from ccrec_amd import ops
    n, d, nq, k = 600_000, 128, 1024, 50
    g = torch.Generator(device="cuda").manual_seed(31)
    D = torch.randn(n, d, generator=g, device="cuda") / d ** 0.5
    Q = torch.randn(nq, d, generator=g, device="cuda") / d ** 0.5
    Q[:, 0] = Q[:, 100] = 0.5
    rows = torch.arange(150, device="cuda") * 4_000 + 77
    D[rows, 0], D[rows, 100] = 65536.0, -65536.0        # exact cancellation in the canonical score, a huge norm
    Db, Qb = ops.pack_bf16(D), ops.pack_bf16(Q)
    index = ops.CorpusIndex(Db)
    s, i = index.search(Qb, k)
    st = index.last_stats()
    assert st["path"] == 1 and st["n_fallback"] > nq // 2 and st["n_retried"] >= st["n_fallback"] and st["n_dense"] == 0, st
    pick = torch.arange(0, nq, 37, device="cuda")
    s1, i1 = index.search(Qb[pick], k, 1)
    assert torch.equal(i[pick], i1) and torch.equal(s[pick].view(torch.int32), s1.view(torch.int32))
    bits = lambda t: t.view(torch.int16).cpu().numpy().view(np.uint16)   # noqa: E731
    ref_i, ref_s = orc.canonical_search(bits(Qb[:4]), bits(Db), k)
    assert np.array_equal(i[:4].cpu().numpy(), ref_i) and np.array_equal(s[:4].cpu().numpy(), ref_s)


def test_large_query_batches_are_searched_in_pieces(monkeypatch):
    """ops.CorpusIndex bounds the queries per ccr_search call (the workspace grows with the query count): a batch above the
    limit must give the same rows as one call, for the plain, blocked and sparse-prior searches."""
    from ccrec_amd import ops
    n, d, nq, k = 30_000, 64, 700, 12
    Db, Qb = _bf16(_rand_bits(n, d, 3)), _bf16(_rand_bits(nq, d, 4))
    index = ops.CorpusIndex(Db, global_row_offset=100)
    rs = np.random.RandomState(1)
    lens = rs.randint(0, 6, nq)
    ptr = np.concatenate([[0], np.cumsum(lens)])
    idx = np.concatenate([np.sort(rs.choice(n, m, replace=False)) + 100 for m in lens]).astype(np.int64)
    val = rs.randn(idx.size)
    whole = (index.search(Qb, k), index.search_blocked(Qb, k, ptr, idx), index.search_sparse_prior(Qb, k, ptr, idx, val))
    monkeypatch.setattr(ops, "MAX_QUERIES_PER_SEARCH", 256)
    parts = (index.search(Qb, k), index.search_blocked(Qb, k, ptr, idx), index.search_sparse_prior(Qb, k, ptr, idx, val))
    for (s0, i0), (s1, i1) in zip(whole, parts):
        assert torch.equal(i0, i1) and torch.equal(s0, s1)
    with pytest.raises(AssertionError):
        index.search(Qb, k, defer=True)


def test_margin_path_small_corpus_and_flagged_queries():
    """The margin path (MFMA score rows + margin select + canonical re-score) behind the default search of a corpus too small for
    the sampled thresholds and behind the flagged queries of a fused search: bit-identical to the fp64 path (flag 1), ties of
    hundreds of rows included; more than 8,192 rows inside the margin fall through to the fp64 path."""
    from ccrec_amd import ops
    rs = np.random.RandomState(5)
    # (a) small corpus, default search = margin path; 700 identical rows sit in every query's top-k region
    Db = _rand_bits(2_000, 64, 11)
    Db[100:800] = Db[100]
    Qb = _rand_bits(300, 64, 12)
    Qb[:40] = Db[100]
    index = ops.CorpusIndex(_bf16(Db), global_row_offset=1 << 35)
    for k in (1, 50, 900):
        s0, i0 = index.search(_bf16(Qb), k)
        st = index.last_stats()
        assert st["path"] == 0 and st["n_dense"] == 0, st
        s1, i1 = index.search(_bf16(Qb), k, 1)
        assert torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))
    ref_i, ref_s = orc.canonical_search(Qb[:5], Db, 50)
    s0, i0 = index.search(_bf16(Qb[:5]), 50)
    assert np.array_equal(i0.cpu().numpy() - (1 << 35), ref_i) and np.array_equal(s0.cpu().numpy().view(np.uint32), ref_s.view(np.uint32))
    # (b) every row identical: 20,000 rows inside the margin > 8,192 -> the fp64 path finishes the query
    Dc = np.tile(_rand_bits(1, 64, 13), (20_000, 1))
    index = ops.CorpusIndex(_bf16(Dc))
    s0, i0 = index.search(_bf16(Qb[:9]), 30)
    assert index.last_stats()["n_dense"] == 9
    assert torch.equal(i0, torch.arange(30, device="cuda").expand(9, 30))
    # (c) a fused search whose flagged queries (400 rows tied at the cut > rescore_cap) finish on the margin path
    n = 300_000
    g = torch.Generator(device="cuda").manual_seed(3)
    D = ops.pack_bf16(torch.randn(n, 128, generator=g, device="cuda") / 128 ** 0.5)
    D[5000:5400] = D[5000]
    Q = ops.pack_bf16(torch.randn(600, 128, generator=g, device="cuda") / 128 ** 0.5)
    Q[17:23] = D[5000]
    index = ops.CorpusIndex(D)
    s0, i0 = index.search(Q, 100)
    st = index.last_stats()
    assert st["path"] == 1 and st["n_fallback"] == 6 and st["n_dense"] == 6, st
    s1, i1 = index.search(Q[10:30], 100, 1)
    assert torch.equal(i0[10:30], i1) and torch.equal(s0[10:30].view(torch.int32), s1.view(torch.int32))


def test_two_host_threads_search_their_own_indices_concurrently():
    """one index must not be searched from two host threads at once, DISTINCT indices are independent.  Two
    threads, each with its own index, stream and workspace, run searches at the same time (ctypes releases the GIL inside the
    library: the per-device slab / block caches, the LDS opt-in table and the error text are what they share); every result
    must equal the serial one bit for bit, index creation and destruction included."""
    import threading
    from ccrec_amd import ops
    shapes = [(150_000, 700, 256, 100), (90_000, 300, 768, 1001)]
    data, serial = [], []
    for t, (n, nq, d, k) in enumerate(shapes):
        D, Q = _bf16(_rand_bits(n, d, 300 + t)), _bf16(_rand_bits(nq, d, 400 + t))
        data.append((D, Q, k))
        s, i = ops.CorpusIndex(D).search(Q, k)
        serial.append((s.clone(), i.clone()))
    torch.cuda.synchronize()
    errors = []

    def worker(t):
        try:
            D, Q, k = data[t]
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for rep in range(12):
                    index = ops.CorpusIndex(D, global_row_offset=0)       # a fresh index per repetition: slab / block cache traffic
                    s, i = index.search(Q, k, defer=(rep % 2 == 1))
                    if rep % 2 == 1:
                        index.finish()
                    stream.synchronize()
                    if not (torch.equal(i, serial[t][1]) and torch.equal(s.view(torch.int32), serial[t][0].view(torch.int32))):
                        errors.append((t, rep, "mismatch"))
                    del index
        except Exception as e:   # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not any(th.is_alive() for th in threads), "a search thread did not finish"
    assert not errors, errors


@pytest.mark.parametrize("sim", ["dot", "cos"])
def test_bertbpr_transform_against_the_references_own_methods_golden_g19(golden_dir, monkeypatch, sim):
    """Golden g19: the reference's BertBPR.get_all_embeddings and BertBPR.transform (src/ccrec/models/bbpr.py:466-550), bound to a
    __new__-built instance by tools/make_golden.py (toy tokenizer with the class's own padding="max_length" kw, a numpy-seeded 768-wide
    BertModel).  The product's BertBPR on the same titles / weights / index maps: the packed embeddings are the reference's fp32 rows
    rounded to bf16 (up to fp32 noise of another BLAS), and transform(D)'s scores are the reference's dense matrix within the bf16
    rounding of the rows (2^-7 ||u|| ||v||; cos: 2^-7), with the same arg-max item per user wherever the reference separates its two
    best items by more than twice that."""
    import os
    import pandas as pd
    from helpers import G19_CFG, GoldenTokenizer, numpy_seeded_bert
    from ccrec_amd.bbpr_transform import BertBPR, LowRankScore
    from ccrec_amd.item_tower import NaiveItemTower
    g = np.load(os.path.join(golden_dir, "g19_bertbpr_transform.npz"))
    monkeypatch.setenv("CCREC_SIM_TYPE", sim)
    monkeypatch.setenv("CCREC_EMBEDDING_TYPE", "mean_pooling")
    titles = [str(t) for t in g["titles"]]
    n_items = len(titles)
    item_df = pd.DataFrame({"TITLE": titles}, index=[f"i{j}" for j in range(n_items)])
    tower = NaiveItemTower(numpy_seeded_bert(G19_CFG, int(g["seed"])), torch.nn.LayerNorm(768, elementwise_affine=False)).cuda()
    bb = BertBPR(item_df, tower, GoldenTokenizer(64), max_length=int(g["max_length"]), batch_size=32)
    E = g["all_emb"]
    emb = bb.get_all_embeddings(tower, 32)
    assert emb.shape == (n_items, 768) and emb.dtype == torch.bfloat16
    want = E / np.maximum(np.linalg.norm(E, axis=1, keepdims=True), 1e-12) if sim == "cos" else E
    got = emb.float().cpu().numpy()
    assert np.all(np.abs(got - want) <= 2.0 ** -8 * np.abs(want) + 2e-5 * np.abs(want).max())
    users = [[f"i{int(p)}"] for p in g["i_to_ptr"]]
    items = [f"i{int(j)}" for j in g["j_to_ptr"]]
    S = bb.transform(_Dataset(users, items, None))
    assert isinstance(S, LowRankScore) and S.shape == (11, 70)
    ref = g[f"scores_{sim}"]
    ours = S.as_tensor().cpu().numpy()
    U, V = E[g["i_to_ptr"]], E[g["j_to_ptr"]]
    bound = 2.0 ** -7 * ((np.linalg.norm(U, axis=1)[:, None] * np.linalg.norm(V, axis=1)[None, :]) if sim == "dot" else np.ones_like(ref)) + 1e-6
    assert np.all(np.abs(ours - ref) <= bound)
    top_s, top_i = S.topk(1)
    srt = np.sort(ref, axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 2 * bound.max()
    assert np.array_equal(top_i.cpu().numpy()[clear, 0], ref.argmax(1)[clear]) and clear.sum() >= 3
