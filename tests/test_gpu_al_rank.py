"""End-to-end rank step (al_0_rank.py:69-127) with a local random-init BertModel (no network): text ->
tokens -> encoder -> fused mean-pool -> bf16 pack -> fused search; checked against the oracle on the
SAME encoder outputs, plus the ranking_profile.pt cache/resume behaviour."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


class ToyTokenizer:
    """Whitespace tokenizer with the HF call signature used by the reference (padding=True, truncation, max_length)."""

    def __call__(self, texts, truncation=True, padding=True, max_length=32, return_tensors="pt"):
        ids = [[1] + [2 + (hash(w) % 500) for w in t.split()][: max_length - 2] + [3] for t in texts]
        L = max(len(r) for r in ids)
        input_ids = torch.zeros(len(ids), L, dtype=torch.long)
        mask = torch.zeros(len(ids), L, dtype=torch.long)
        for r, row in enumerate(ids):
            input_ids[r, : len(row)] = torch.tensor(row)
            mask[r, : len(row)] = 1
        return {"input_ids": input_ids, "attention_mask": mask}


def _tower():
    from transformers import BertConfig, BertModel
    from ccrec_amd.item_tower import NaiveItemTower

    class RecordingTower(NaiveItemTower):
        """Keeps every batch of pooled fp32 rows it returned, so that the oracle sees EXACTLY the encoder outputs
        the product path packed (a second BERT forward with another batch shape is not bit-reproducible)."""
        record = []

        def forward(self, *a, **k):
            out = super().forward(*a, **k)
            self.record.append(out.detach().float().cpu().numpy())
            return out

    torch.manual_seed(0)
    cfg = BertConfig(vocab_size=512, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                     max_position_embeddings=64)
    RecordingTower.record = []
    return RecordingTower(BertModel(cfg).eval(), torch.nn.LayerNorm(64, elementwise_affine=False))


def test_generate_ranking_profile_end_to_end(tmp_path):
    from ccrec_amd.al_rank import cached_ranking_profile, generate_ranking_profile
    os.environ["CCREC_SIM_TYPE"] = "dot"
    os.environ["CCREC_EMBEDDING_TYPE"] = "mean_pooling"
    os.environ["CCREC_MAX_LENGTH"] = "32"
    rs = np.random.RandomState(0)
    words = [f"w{i}" for i in range(200)]
    corpus = {f"p{j}": " ".join(rs.choice(words, rs.randint(3, 20))) for j in range(700)}
    queries = {f"q{i}": " ".join(rs.choice(words, rs.randint(2, 8))) for i in range(9)}
    tower, tok = _tower(), ToyTokenizer()
    prof = generate_ranking_profile(tower, "unused", corpus, queries, tokenizer=tok)
    assert list(prof) == list(queries) and all(len(v) == 700 for v in prof.values())

    # oracle on the recorded encoder outputs: ranking() encodes the queries first (1 batch), then the corpus
    rec = type(tower).record
    Eq, Ed = rec[0], np.concatenate(rec[1:], 0)
    assert Eq.shape == (9, 64) and Ed.shape == (700, 64)
    ref_i, ref_s = orc.canonical_ranking(Eq, Ed, "dot")
    got_i = np.array([[int(p[1:]) for p in prof[q]] for q in queries])
    got_s = np.array([list(prof[q].values()) for q in queries], np.float32)
    assert np.array_equal(got_i, ref_i) and np.array_equal(got_s.view(np.uint32), ref_s.view(np.uint32))

    # cache / resume: second call must not rebuild the model (al_0_rank.py:115-118)
    path = str(tmp_path / "ranking_profile.pt")
    calls = []

    class Wrap:
        item_tower = tower

    def make():
        calls.append(1)
        return Wrap()

    p1 = cached_ranking_profile(path, make, "unused", corpus, queries, tokenizer=tok)
    p2 = cached_ranking_profile(path, make, "unused", corpus, queries, tokenizer=tok)
    assert len(calls) == 1 and list(p1["q0"]) == list(p2["q0"]) and os.path.isfile(path)
    # under autocast the encoder runs in reduced precision: same shape, scores close to the fp32 run
    for q in queries:
        assert len(p1[q]) == 700
        s32, s16 = np.array(list(prof[q].values())[:5]), np.array(list(p1[q].values())[:5])
        np.testing.assert_allclose(s16, s32, rtol=0.05, atol=0.05)


def test_generate_ranking_profile_length_sorted_route_gives_the_same_ranking():
    """length_sorted=True (CCREC_LENGTH_SORTED=1) encodes through the length-sorted pipeline instead of the script's padded
    corpus-order batches: same contract (rank-ordered dicts of all 700 passages), same neighbours -- the two routes' scores differ only
    by the encoder's summation-order noise over different batch shapes, so the score vectors agree to 1e-4 and either route's top
    passage is within that noise of the other's top score (a near-tie may swap two passages)."""
    from ccrec_amd.al_rank import generate_ranking_profile
    os.environ["CCREC_SIM_TYPE"] = "dot"
    os.environ["CCREC_EMBEDDING_TYPE"] = "mean_pooling"
    os.environ["CCREC_MAX_LENGTH"] = "32"
    rs = np.random.RandomState(1)
    words = [f"w{i}" for i in range(200)]
    corpus = {f"p{j}": " ".join(rs.choice(words, rs.randint(3, 20))) for j in range(700)}
    queries = {f"q{i}": " ".join(rs.choice(words, rs.randint(2, 8))) for i in range(9)}
    block = {q: [f"p{int(j)}" for j in rs.choice(700, 3, replace=False)] for q in queries}

    class UnpaddedToy(ToyTokenizer):      # the length-sorted encoder asks for padding=False
        def __call__(self, texts, truncation=True, padding=True, max_length=32, return_tensors="pt", **kw):
            if padding is False:
                ids = [[1] + [2 + (hash(w) % 500) for w in t.split()][: max_length - 2] + [3] for t in texts]
                return {"input_ids": ids, "attention_mask": [[1] * len(r) for r in ids]}
            return super().__call__(texts, truncation, padding, max_length, return_tensors)

    tower, tok = _tower(), UnpaddedToy()
    a = generate_ranking_profile(tower, "unused", corpus, queries, block_dict=block, tokenizer=tok)
    b = generate_ranking_profile(tower, "unused", corpus, queries, block_dict=block, tokenizer=tok, length_sorted=True)
    assert list(a) == list(b) == list(queries)
    for q in queries:
        assert len(b[q]) == 700 and set(a[q]) == set(b[q])
        sa = np.array([a[q][p] for p in a[q]], np.float32)
        sb = np.array([b[q][p] for p in a[q]], np.float32)
        tol = 1e-4 * max(1.0, np.abs(sa[sa > -1e5]).max())
        assert np.abs(sa - sb).max() < tol
        # the other route's best passage is (within that noise) this route's best: a near-tie may swap the two, nothing more
        assert max(b[q].values()) - b[q][next(iter(a[q]))] < 2 * tol
        assert all(b[q][p] == -1e6 for p in block[q])
    os.environ["CCREC_LENGTH_SORTED"] = "1"
    try:
        c = generate_ranking_profile(tower, "unused", corpus, queries, block_dict=block, tokenizer=tok)
    finally:
        del os.environ["CCREC_LENGTH_SORTED"]
    assert all(list(c[q]) == list(b[q]) for q in queries)


@pytest.mark.parametrize("tag,sim", [("dot", "dot"), ("cos_block", "cos")])
def test_generate_ranking_profile_against_the_references_own_function_golden_g18(golden_dir, monkeypatch, tag, sim):
    """Golden g18: the reference's generate_ranking_profile (scripts/al_oracle_agent.py:83-129), run by tools/make_golden.py on a local
    numpy-seeded encoder.  The product's function on the same texts, tokenizer and weights:
      * feeds the search the embeddings the reference's embedding_func produced (queries first, then the corpus in dict order; fp32
        noise of another BLAS only: 2e-5 of the largest value);
      * returns, bit for bit, the oracle's canonical ranking of ITS embeddings (ids and score bits);
      * and that profile is the reference's: scores within the bf16 rounding of the rows (2^-7 ||q|| ||d||), ids equal at every rank
        the reference separates by more than twice that, blocked passages last at -1e6 (cos + block_dict case)."""
    from helpers import G18_CFG, GoldenTokenizer, assert_rank_close, canonicalise, numpy_seeded_bert
    from ccrec_amd.al_rank import generate_ranking_profile
    from ccrec_amd.item_tower import NaiveItemTower
    g = np.load(os.path.join(golden_dir, "g18_ranking_profile_fn.npz"))
    monkeypatch.setenv("CCREC_SIM_TYPE", sim)
    monkeypatch.setenv("CCREC_EMBEDDING_TYPE", "mean_pooling")
    monkeypatch.setenv("CCREC_MAX_LENGTH", str(int(g["max_length"])))
    corpus = {f"p{j}": str(t) for j, t in enumerate(g["corpus_texts"])}
    queries = {f"q{i}": str(t) for i, t in enumerate(g["query_texts"])}
    block = {q: [f"p{int(j)}" for j in g["block"][i]] for i, q in enumerate(queries)} if tag == "cos_block" else None

    class Recording(NaiveItemTower):
        record = []

        def forward(self, *a, **k):
            out = super().forward(*a, **k)
            self.record.append(out.detach().float().cpu().numpy())
            return out

    Recording.record = []
    tower = Recording(numpy_seeded_bert(G18_CFG, int(g["seed"])), torch.nn.LayerNorm(64, elementwise_affine=False))
    prof = generate_ranking_profile(tower, "unused", corpus, queries, block_dict=block, tokenizer=GoldenTokenizer(64))
    rec = Recording.record
    Eq, Ed = rec[0], np.concatenate(rec[1:], 0)
    ref_q, ref_d = g[f"{tag}_query_emb"], g[f"{tag}_corpus_emb"]
    assert Eq.shape == ref_q.shape and Ed.shape == ref_d.shape
    np.testing.assert_allclose(Eq, ref_q, rtol=0, atol=2e-5 * np.abs(ref_q).max())
    np.testing.assert_allclose(Ed, ref_d, rtol=0, atol=2e-5 * np.abs(ref_d).max())
    blk = [r.tolist() for r in g["block"]] if block is not None else None
    can_i, can_s = orc.canonical_ranking(Eq, Ed, sim, block=blk)
    got_i = np.array([[int(p[1:]) for p in prof[q]] for q in queries])
    got_s = np.array([list(prof[q].values()) for q in queries], np.float32)
    assert list(prof) == list(queries)
    assert np.array_equal(got_i, can_i) and np.array_equal(got_s.view(np.uint32), can_s.view(np.uint32))
    ri, rs = canonicalise(g[f"{tag}_ids"], g[f"{tag}_scores"])
    bound = 2.0 ** -7 * (float((np.linalg.norm(ref_q, axis=1)[:, None] * np.linalg.norm(ref_d, axis=1)[None, :]).max()) if sim == "dot" else 1.0)
    assert_rank_close(got_i, got_s, ri, rs, tol=bound)
    if block is not None:
        for i, q in enumerate(queries):
            assert [p for p in prof[q]][-4:] == sorted(block[q], key=lambda p: int(p[1:])) and all(prof[q][p] == -1e6 for p in block[q])


def test_replica_cache_on_a_real_device():
    """src/ccrec/util/data_parallel.py:8-20 on the one device this box has (the >1-device path needs a multi-GPU node: only the CPU test with
    a faked broadcast covers it): cache_replicas() really broadcasts the tower to cuda:0 -- a detached copy, not the module itself --,
    replicate() hands that stored copy back, the forward through the wrapper equals the module's, and the wrapper keeps working
    inside generate_ranking_profile's embedding_func (scripts/al_0_rank.py:92, 98-101)."""
    from ccrec_amd.replica_cache import DataParallel
    tower = _tower().cuda()
    dp = DataParallel(tower, device_ids=[0])
    assert dp.cache_replicas() is dp and list(dp._by_device) == [0]
    copy0 = dp._by_device[0]
    assert copy0 is not tower and dp.replicate(tower, [0])[0] is copy0
    p_src, p_cpy = next(tower.cls_model.parameters()), next(copy0.cls_model.parameters())
    assert p_cpy.device.type == "cuda" and p_cpy.is_leaf and torch.equal(p_src, p_cpy)      # detach=True: a leaf, not a view into an autograd graph
    tok = ToyTokenizer()(["w1 w2 w3", "w4"], max_length=16)
    with torch.no_grad():
        a = dp(**{k: v.cuda() for k, v in tok.items()}, output_step="mean_pooling")
        b = tower(**{k: v.cuda() for k, v in tok.items()}, output_step="mean_pooling")
    assert torch.equal(a, b)
