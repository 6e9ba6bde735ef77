"""Pin the CPU oracle (oracle/) against golden vectors produced by the reference's own Python
(tools/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from helpers import assert_rank_close, canonicalise
from oracle import oracle as orc


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _block_list(g):
    ptr, idx = g["block_ptr"], g["block_idx"]
    return [idx[ptr[i]:ptr[i + 1]].tolist() for i in range(len(ptr) - 1)]


@pytest.mark.parametrize("name,sim,trunc", [
    ("g1_ranking_dot.npz", "dot", False),
    ("g4_ranking_trunc.npz", "dot", True),
])
def test_canonical_ranking_matches_reference_dot(golden_dir, name, sim, trunc):
    g = _load(golden_dir, name)
    ids, sc = orc.canonical_ranking(g["Eq"], g["Ed"], sim)
    assert ids.shape[1] == min(1001, g["Ed"].shape[0])
    # inputs are bf16-exact, so the only difference is fp32 (MKL) vs fp64 accumulation
    assert_rank_close(ids, sc, g["ids"], g["scores"], tol=2e-6, truncated=trunc)


def test_reference_faithful_restatement_dot(golden_dir):
    g = _load(golden_dir, "g1_ranking_dot.npz")
    ids, sc = orc.reference_ranking(g["Eq"], g["Ed"], int(g["batch_size"]), "dot")
    ci, cs = canonicalise(ids, sc)
    assert_rank_close(ci, cs, g["ids"], g["scores"], tol=1e-6)


def test_cos_ranking(golden_dir):
    g = _load(golden_dir, "g2_ranking_cos.npz")
    # faithful fp32 restatement reproduces the reference to fp32 noise
    ids, sc = orc.reference_ranking(g["Eq"], g["Ed"], int(g["batch_size"]), "cos")
    ci, cs = canonicalise(ids, sc)
    assert_rank_close(ci, cs, g["ids"], g["scores"], tol=1e-6)
    # canonical path rounds the normalised rows to bf16: scores within the 1e-3 bf16 tolerance
    ids2, sc2 = orc.canonical_ranking(g["Eq"], g["Ed"], "cos")
    assert_rank_close(ids2, sc2, g["ids"], g["scores"], tol=1e-3)
    assert orc.recall_at_k(g["ids"][:, :12], ids2[:, :10]) > 0.95  # our top-10 inside the reference top-12


def test_block_dict(golden_dir):
    g = _load(golden_dir, "g3_ranking_block.npz")
    block = _block_list(g)
    ids, sc = orc.canonical_ranking(g["Eq"], g["Ed"], "dot", block=block)
    ref_i, ref_s = g["ids"], g["scores"]
    for q, b in enumerate(block):
        nb = len(b)
        # blocked ids are kept, scored -1e6, and sort last
        assert set(ids[q, -nb:].tolist()) == set(b)
        assert np.all(sc[q, -nb:] == np.float32(-1e6))
        assert set(ref_i[q, -nb:].tolist()) == set(b)
        assert q in b  # self-block as in prime_pantry
    ri, rs = canonicalise(ref_i, ref_s)
    assert_rank_close(ids, sc, ri, rs, tol=2e-6)
    with pytest.raises(AssertionError, match="block id not found"):
        orc.canonical_ranking(g["Eq"][:1], g["Ed"], "dot", block=[[10 ** 6]])


def test_cos_block_dict_and_truncation_together(golden_dir):
    """Golden g17 (the reference's ranking() with CCREC_SIM_TYPE=cos, block lists of 0 / 40 / 120-200 ids and an 1 100-passage corpus):
    blocked passages score -1e6 and are KEPT -- those that fit fill the tail of the 1001 entries; with 40 blocked ids (< N - 1001) none
    is kept; cos scores within the bf16 tolerance of the reference's fp32 values."""
    g = _load(golden_dir, "g17_ranking_cos_block_trunc.npz")
    block = _block_list(g)
    ids, sc = orc.canonical_ranking(g["Eq"], g["Ed"], "cos", block=block)
    ref_i, ref_s = g["ids"], g["scores"]
    assert ids.shape == ref_i.shape == (9, 1001)
    for q, b in enumerate(block):
        kept = max(0, 1001 - (1100 - len(b)))                      # blocked entries inside the kept list
        assert int((ref_s[q] == np.float32(-1e6)).sum()) == kept == int((sc[q] == np.float32(-1e6)).sum())
        assert set(ids[q, 1001 - kept:].tolist()) <= set(b) and set(ref_i[q, 1001 - kept:].tolist()) <= set(b)
        assert ids[q, 1001 - kept:].tolist() == sorted(b)[:kept]   # the canonical tie rule among the -1e6 entries: ascending id
        assert not set(ids[q, :1001 - kept].tolist()) & set(b)
    assert_rank_close(ids, sc, ref_i, ref_s, tol=1e-3, truncated=True)


def test_exact_arithmetic_ties(golden_dir):
    g = _load(golden_dir, "g5_ranking_exact_ties.npz")
    ids, sc = orc.canonical_ranking(g["Eq"], g["Ed"], "dot")
    ri, rs = canonicalise(g["ids"], g["scores"])
    # every partial sum is exactly representable: bit-for-bit agreement, ties by ascending id
    assert np.array_equal(sc.view(np.uint32), rs.view(np.uint32))
    assert np.array_equal(ids, ri)
    # the engineered 31-way tie is present and ordered by index
    for q in range(ids.shape[0]):
        pos = np.nonzero(np.isin(ids[q], np.r_[100, 200:230]))[0]
        assert len(pos) == 31 and np.all(np.diff(pos) == 1)
        assert np.all(np.diff(ids[q][pos]) > 0)


def test_item_tower_pooling(golden_dir):
    g = _load(golden_dir, "g6_item_tower.npz")
    mp = orc.meanpool(g["hidden"], g["mask"])
    np.testing.assert_allclose(mp, g["mean_pooling"], rtol=2e-6, atol=2e-6)
    np.testing.assert_array_equal(g["hidden"][:, 0], g["cls"])
    np.testing.assert_allclose(orc.layer_norm(g["hidden"][:, 0]), g["mean_layer_norm"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("tag,B,sim", [("b8_dot", 8, "dot"), ("b32_dot", 32, "dot"), ("b8_cos", 8, "cos"), ("b32_cos", 32, "cos")])
def test_contrastive_loss_and_grads(golden_dir, tag, B, sim):
    g = _load(golden_dir, "g7_contrastive.npz")
    E = g[f"{tag}_E"]
    loss, dQ, dP, dN = orc.inbatch_ce(E[:B], E[B:2 * B], E[2 * B:], 20.0, sim)
    assert abs(loss - float(g[f"{tag}_loss"])) < 2e-5 * max(1.0, abs(loss))
    grad = np.concatenate([dQ, dP, dN], 0)
    np.testing.assert_allclose(grad, g[f"{tag}_grad"], rtol=2e-4, atol=2e-6)


def test_assign_topk(golden_dir):
    g = _load(golden_dir, "g8_assign_topk.npz")
    k = int(g["k"])
    ids, sc = orc.canonical_search(orc.pack_bf16(g["U"]), orc.pack_bf16(g["V"]), k)
    ref = g["indices"]
    ref_sc = np.take_along_axis((g["U"].astype(np.float64) @ g["V"].astype(np.float64).T), ref, 1).astype(np.float32)
    assert_rank_close(ids, sc, ref, ref_sc, tol=2e-6, truncated=True)
    assert np.array_equal(g["indptr"], np.arange(0, ref.size + 1, k))


@pytest.mark.parametrize("d", [300, 50])
def test_assign_topk_odd_width(golden_dir, d):
    """g15 = the reference's _assign_topk on factors whose width is no multiple of 8 (score_array.py:320-339 takes any)."""
    g = _load(golden_dir, "g15_assign_topk_odd_width.npz")
    U, V, ref, k = g[f"U{d}"], g[f"V{d}"], g[f"indices{d}"], int(g[f"k{d}"])
    ids, sc = orc.canonical_search(orc.pack_bf16(U), orc.pack_bf16(V), k)
    ref_sc = np.take_along_axis(U.astype(np.float64) @ V.astype(np.float64).T, ref, 1).astype(np.float32)
    assert_rank_close(ids, sc, ref, ref_sc, tol=2e-6, truncated=True)


def test_sparse_prior_topk_and_score_op(golden_dir):
    """g14 = the reference's rime_lite on (U @ V.T + sparse prior): _assign_topk, score_op (bbpr.py:592-595)."""
    g = _load(golden_dir, "g14_sparse_prior.npz")
    k = int(g["k"])
    Ub, Vb = orc.pack_bf16(g["U"]), orc.pack_bf16(g["V"])
    ids, fin = orc.sparse_prior_search(Ub, Vb, g["prior_indptr"], g["prior_indices"], g["prior_data"], k)
    assert np.array_equal(ids, g["indices"])                      # gaps between ranked finals are >> the fp32 matmul noise
    np.testing.assert_allclose(fin, g["topk_scores"], rtol=0, atol=2e-6 + 1e-9 * 1e5)
    for op in ("max", "min", "sum"):
        low = orc.score_op(Ub, Vb, op)
        both = orc.score_op(Ub, Vb, op, g["prior_indptr"], g["prior_indices"], g["prior_data"])
        np.testing.assert_allclose(low, float(g["low_" + op]), rtol=1e-5, atol=2e-4)
        np.testing.assert_allclose(both, float(g["sum_" + op]), rtol=1e-6, atol=2e-4)


def test_pack_bf16_bits(golden_dir):
    g = _load(golden_dir, "g9_pack_bf16.npz")
    bits = orc.pack_bf16(g["x"])
    assert np.array_equal(bits, g["bits"])
    np.testing.assert_array_equal(orc.unpack_bf16(bits).view(np.uint32) >> 16, bits.astype(np.uint32))


def test_merge_topk_is_topk_of_union():
    rs = np.random.RandomState(0)
    R, nq, k = 3, 5, 16
    sc = rs.randint(-50, 50, size=(R, nq, k)).astype(np.float32)   # many ties
    ids = np.stack([rs.permutation(1000)[: nq * k].reshape(nq, k) + 1000 * r for r in range(R)]).astype(np.int64)
    for r in range(R):
        for q in range(nq):
            o = np.lexsort((ids[r, q], -sc[r, q]))
            sc[r, q], ids[r, q] = sc[r, q][o], ids[r, q][o]
    ms, mi = orc.merge_topk(sc, ids)
    for q in range(nq):
        alls, alli = sc[:, q].ravel(), ids[:, q].ravel()
        o = np.lexsort((alli, -alls))[:k]
        assert np.array_equal(mi[q], alli[o]) and np.array_equal(ms[q], alls[o])


def test_item_tower_on_a_real_encoder(golden_dir):
    """g16 = the reference's NaiveItemTower around a real (local, seeded) transformers BertModel, fp32: transformers' own forward on
    the fixture's weights here + the oracle's pooling / LayerNorm restatement reproduce the reference's three output steps -- the
    fixture pins tower + encoder together for the GPU tests of the module and kernel forwards."""
    import json
    import torch
    from transformers import BertConfig, BertModel
    g = _load(golden_dir, "g16_item_tower_bert.npz")
    cfg = json.loads(str(g["config"]))
    model = BertModel(BertConfig(**cfg)).eval()
    state = {k[2:]: torch.from_numpy(g[k].astype(np.int16)).view(torch.bfloat16).float() for k in g.files if k.startswith("w_")}
    missing, unexpected = model.load_state_dict(state, strict=False)
    assert not unexpected and all("position_ids" in m or "token_type_ids" in m for m in missing)
    with torch.no_grad():
        hidden = model(input_ids=torch.from_numpy(g["ids"]), attention_mask=torch.from_numpy(g["mask"])).last_hidden_state.numpy()
    np.testing.assert_allclose(orc.meanpool(hidden, g["mask"]), g["out_mean_pooling"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(hidden[:, 0], g["out_cls"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(orc.layer_norm(hidden[:, 0]), g["out_mean_layer_norm"], rtol=0, atol=5e-6)


def test_mrr_beir_known_answers():
    """oracle.mrr_beir restates BEIR's published custom_metrics.mrr (scripts/al_0_rank.py:130-133; beir is absent here and the
    reference holds no MRR vectors: parity unpinned).  Hand-computed cases pin the restatement's own reading: first
    relevant hit within k, relevance > 0 only, results re-sorted by score, the sum divided by len(qrels) (queries missing from
    `results` count), 5 decimals."""
    qrels = {"a": {"d1": 1, "d9": 0}, "b": {"d2": 2}, "c": {"d3": 1}, "d": {"d4": 1}}
    results = {
        "a": {"d9": 0.9, "d1": 0.8, "d5": 0.1},      # the relevance-0 document does not count: first hit at rank 2
        "b": {"d7": 0.1, "d2": 0.7, "d8": 0.3},      # insertion order is not rank order: re-sorted by score -> rank 1
        "c": {"d5": 0.5, "d6": 0.4, "d7": 0.3, "d3": 0.2},   # rank 4
    }                                                  # "d" has no results: contributes 0 but counts in the denominator
    got = orc.mrr_beir(qrels, results, [1, 3, 10])
    assert got == {"MRR@1": round(1 / 4, 5), "MRR@3": round((0.5 + 1) / 4, 5), "MRR@10": round((0.5 + 1 + 0.25) / 4, 5)}
    # the id-row form with the qrels' count agrees
    ids = [[9, 1, 5, 0], [2, 8, 7, 0], [5, 6, 7, 3]]
    rel = [{1}, {2}, {3}]
    assert [orc.mrr(ids, rel, k, n_qrels=4) for k in (1, 3, 10)] == [got["MRR@1"], got["MRR@3"], got["MRR@10"]]
    assert orc.mrr(ids, rel, 10) == round((0.5 + 1 + 0.25) / 3, 5)


@pytest.mark.parametrize("tag,sim", [("dot", "dot"), ("cos_block", "cos")])
def test_generate_ranking_profile_golden_g18(golden_dir, tag, sim):
    """Golden g18 = the reference's OWN generate_ranking_profile (scripts/al_oracle_agent.py:83-129, its source executed by
    tools/make_golden.py) on a local seeded encoder: the profile it returned is what the oracle's ranking gives on the embeddings its
    embedding_func produced (the function adds tokeniser + tower + DataParallel around ranking(), nothing numeric)."""
    g = _load(golden_dir, "g18_ranking_profile_fn.npz")
    Eq, Ed = g[f"{tag}_query_emb"], g[f"{tag}_corpus_emb"]
    assert Eq.shape == (7, 64) and Ed.shape == (260, 64) and g[f"{tag}_ids"].shape == (7, 260)
    block = [r.tolist() for r in g["block"]] if tag == "cos_block" else None
    ids, sc = orc.reference_ranking(Eq, Ed, 512, sim, block=block)
    ci, cs = canonicalise(ids, sc)
    ri, rs = canonicalise(g[f"{tag}_ids"], g[f"{tag}_scores"])
    assert_rank_close(ci, cs, ri, rs, tol=1e-6)
    # the canonical (bf16) path: within the bf16 tolerance of the reference's fp32 scores; blocked passages last at -1e6
    ids2, sc2 = orc.canonical_ranking(Eq, Ed, sim, block=block)
    scale = float(np.abs(rs[rs > -1e5]).max())
    assert_rank_close(ids2, sc2, ri, rs, tol=1e-2 * max(1.0, scale))
    if block is not None:
        for q, b in enumerate(block):
            assert set(ids2[q, -len(b):].tolist()) == set(b) and np.all(sc2[q, -len(b):] == np.float32(-1e6))


def test_bertbpr_transform_golden_g19(golden_dir):
    """Golden g19 = the reference's OWN BertBPR.get_all_embeddings / transform (src/ccrec/models/bbpr.py:466-550): the dense score matrix it
    returned is all_emb[i_to_ptr] @ all_emb[j_to_ptr].T (dot) / the cosine of the same rows -- the restatement the product's transform
    is compared with."""
    g = _load(golden_dir, "g19_bertbpr_transform.npz")
    E, i_ptr, j_ptr = g["all_emb"], g["i_to_ptr"], g["j_to_ptr"]
    assert E.shape == (90, 768) and g["scores_dot"].shape == (11, 70)
    U, V = E[i_ptr], E[j_ptr]
    np.testing.assert_allclose(U.astype(np.float64) @ V.astype(np.float64).T, g["scores_dot"], rtol=0, atol=2e-5 * np.abs(g["scores_dot"]).max())
    np.testing.assert_allclose(orc.reference_cos_sim(U, V), g["scores_cos"], rtol=0, atol=2e-6)
    # canonical bf16 path against the reference's fp32 values: bf16 rounding of both operands (2^-8 relative per row, norm-scaled)
    for sim in ("dot", "cos"):
        got = orc.canonical_scores(orc.pack(U, sim), orc.pack(V, sim))
        bound = (np.linalg.norm(U, axis=1)[:, None] * np.linalg.norm(V, axis=1)[None, :]) if sim == "dot" else 1.0
        assert np.all(np.abs(got - g[f"scores_{sim}"]) <= 2.0 ** -7 * bound + 1e-6)
