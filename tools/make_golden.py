#!/usr/bin/env python3
"""Generate golden input/output vectors by RUNNING the reference's own Python.

Dev-container only: needs /root/reference (never present on the GPU box).  The
fixtures written to tests/golden/*.npz contain arrays only (inputs + the outputs
the reference produced); no reference source travels.

Recipe:
  * inert stub modules for the four third-party packages the reference imports
    but this image lacks (pytorch_lightning, tensorboard, shap, beir);
  * /root/reference/{src,scripts} on sys.path, CCREC_* env set before import;
  * Tensor.cuda / cuda.synchronize no-op'd (that *is* the reference CPU path).

Fixtures:
  G1 ranking() dot                G2 ranking() cos
  G3 ranking() with block_dict    G4 ranking() N>1001 (truncation to 1001)
  G5 exact-arithmetic corpus with engineered ties
  G6 NaiveItemTower.forward (mean_pooling / cls / mean_layer_norm)
  G7 multiple_nrl in-batch-negative loss + autograd grads
  G8 _assign_topk on a MatMulExpression
  G9 fp32->bf16 RNE bit patterns (torch, not the reference)
  G10 request builder (scripts/al_0_rank.py '## creation' block, executed from the reference file in a prepared
      namespace: the script itself is not importable -- it parses argv and loads datasets at import time)
  G11 generate_train_data (scripts/al_oracle_agent.py, the function's own source executed the same way)
  G12 BM25: the reference's bm_25.BM25 (imported) and ranking_bm25 (function source executed) on a toy corpus
  G13 rime_lite.metrics.evaluate_item_rec on a MatMulExpression score (the §8b signature that consumes _assign_topk)
  G14 low-rank + sparse prior: _assign_topk / evaluate_item_rec / score_op on
      ElementWiseExpression(add, [U @ V.T, sparse]) -- the post-fit expression of bbpr.py:592-595 (reranking_prior 1e5)
  G15 _assign_topk on MatMulExpressions of width 300 and 50 (any factor width: score_array.py:320-339)
  G16 NaiveItemTower around a real BertModel; G17 ranking() with cos + block lists + truncation
  G18 generate_ranking_profile (scripts/al_oracle_agent.py:83-129, the function's own source executed) -- SURVEY 8a row a1
  G19 BertBPR.get_all_embeddings / BertBPR.transform (src/ccrec/models/bbpr.py:466-550, bound to a __new__-built instance) -- row a9
"""
import contextlib
import importlib.abc
import importlib.machinery
import io
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
STUB_ROOTS = {"pytorch_lightning", "tensorboard", "shap", "beir"}


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return None


class _LightningModule(torch.nn.Module):
    hparams = {}

    def save_hyperparameters(self, *a, **k):
        pass

    def log(self, *a, **k):
        pass


class _StubModule(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name == "LightningModule":
            return _LightningModule
        if name == "LightningDataModule":
            return type("LightningDataModule", (object,), {})
        return type(name, (_Dummy,), {})


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        pass


def _import_reference(sim_type="dot"):
    os.environ["CCREC_SIM_TYPE"] = sim_type
    os.environ.setdefault("CCREC_EMBEDDING_TYPE", "mean_pooling")
    os.environ.setdefault("CCREC_MAX_LENGTH", "256")
    os.environ.setdefault("CCREC_BBPR_INV_TEMPERATURE", "20")
    if not any(isinstance(f, _StubFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _StubFinder())
    for p in (f"{REF}/src", f"{REF}/scripts"):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.synchronize = lambda *a, **k: None
    with contextlib.redirect_stdout(io.StringIO()):
        import ms_marco_eval  # noqa: F401
        import ccrec.models.item_tower  # noqa: F401
        import ccrec.models.bbpr  # noqa: F401
        import rime_lite.util  # noqa: F401
        import rime_lite.util.score_array  # noqa: F401
    return sys.modules["ms_marco_eval"]


def bf16_exact(x: torch.Tensor) -> torch.Tensor:
    """fp32 tensor whose values are exactly representable in bf16."""
    return x.to(torch.bfloat16).to(torch.float32)


def _run_ranking(mod, Eq, Ed, batch_size, sim, block=None):
    """Call the reference ranking() with integer 'texts' that index embedding tables."""
    os.environ["CCREC_SIM_TYPE"] = sim
    nq, nd = Eq.shape[0], Ed.shape[0]
    table = torch.cat([Eq, Ed], 0)
    queries = {f"q{i}": i for i in range(nq)}
    corpus = {f"p{j}": nq + j for j in range(nd)}

    def embedding_func(rows):
        return table[torch.as_tensor(rows, dtype=torch.long)]

    block_dict = None
    if block is not None:
        block_dict = {f"q{i}": [f"p{j}" for j in block[i]] for i in range(nq)}
    with contextlib.redirect_stdout(io.StringIO()):
        prof = mod.ranking(corpus, queries, embedding_func, batch_size, block_dict)
    L = min(1001, nd)
    ids = np.zeros((nq, L), np.int64)
    sc = np.zeros((nq, L), np.float32)
    for i in range(nq):
        items = list(prof[f"q{i}"].items())
        assert len(items) == L, (len(items), L)
        ids[i] = [int(p[1:]) for p, _ in items]
        sc[i] = [s for _, s in items]
    return ids, sc


def g_ranking(mod):
    g = torch.Generator().manual_seed(1234)
    d = 768
    Ed = bf16_exact(torch.randn(300, d, generator=g) / d ** 0.5)
    Eq = bf16_exact(torch.randn(7, d, generator=g) / d ** 0.5)
    ids, sc = _run_ranking(mod, Eq, Ed, 64, "dot")
    np.savez_compressed(f"{OUT}/g1_ranking_dot.npz", Eq=Eq.numpy(), Ed=Ed.numpy(), ids=ids, scores=sc, batch_size=64)

    # G2: cos on raw (non-normalised, scaled) embeddings; reference normalises in fp32
    Ed2 = bf16_exact(Ed * (0.5 + torch.rand(300, 1, generator=g) * 3))
    Eq2 = bf16_exact(Eq * (0.5 + torch.rand(7, 1, generator=g) * 3))
    ids, sc = _run_ranking(mod, Eq2, Ed2, 64, "cos")
    np.savez_compressed(f"{OUT}/g2_ranking_cos.npz", Eq=Eq2.numpy(), Ed=Ed2.numpy(), ids=ids, scores=sc, batch_size=64)

    # G3: block_dict (queries == corpus as in prime_pantry; self-block + ragged lists)
    E = bf16_exact(torch.randn(200, d, generator=g) / d ** 0.5)
    rs = np.random.RandomState(7)
    groups = rs.randint(0, 23, size=200)
    block = [np.nonzero(groups == groups[i])[0].tolist() for i in range(200)]
    Eq3 = E[:40]
    ids, sc = _run_ranking(mod, Eq3, E, 64, "dot", block=block[:40])
    flat = np.concatenate([np.asarray(b, np.int64) for b in block[:40]])
    ptr = np.cumsum([0] + [len(b) for b in block[:40]]).astype(np.int64)
    np.savez_compressed(f"{OUT}/g3_ranking_block.npz", Eq=Eq3.numpy(), Ed=E.numpy(), ids=ids, scores=sc,
                        block_ptr=ptr, block_idx=flat, batch_size=64)

    # G4: N > 1001 pins the truncation to 1001 entries
    Ed4 = bf16_exact(torch.randn(1500, d, generator=g) / d ** 0.5)
    Eq4 = bf16_exact(torch.randn(5, d, generator=g) / d ** 0.5)
    ids, sc = _run_ranking(mod, Eq4, Ed4, 512, "dot")
    np.savez_compressed(f"{OUT}/g4_ranking_trunc.npz", Eq=Eq4.numpy(), Ed=Ed4.numpy(), ids=ids, scores=sc, batch_size=512)

    # G5: exact arithmetic (multiples of 1/8 in [-4,4]; every partial sum exact in fp32)
    # with engineered ties: duplicate corpus rows, incl. duplicates of top scorers.
    rs = np.random.RandomState(11)
    Ed5 = rs.randint(-32, 33, size=(400, d)).astype(np.float32) / 8
    Eq5 = rs.randint(-32, 33, size=(6, d)).astype(np.float32) / 8
    Ed5[50] = Ed5[10]
    Ed5[350] = Ed5[10]
    Ed5[399] = Ed5[0]
    Ed5[200:230] = Ed5[100]       # a 31-way tie (rows 100, 200..229)
    ids, sc = _run_ranking(mod, torch.from_numpy(Eq5), torch.from_numpy(Ed5), 128, "dot")
    np.savez_compressed(f"{OUT}/g5_ranking_exact_ties.npz", Eq=Eq5, Ed=Ed5, ids=ids, scores=sc, batch_size=128)


def g_ranking_cos_block(mod):
    """G17: the three switches of ranking() together -- CCREC_SIM_TYPE=cos on raw (un-normalised) embeddings, a block_dict whose lists
    are long enough that blocked passages (score -1e6) sit INSIDE the kept 1001 of an 1 100-passage corpus, and the 1001 truncation."""
    g = torch.Generator().manual_seed(1717)
    d = 768
    Ed = bf16_exact(torch.randn(1100, d, generator=g) / d ** 0.5 * (0.5 + torch.rand(1100, 1, generator=g) * 3))
    Eq = bf16_exact(torch.randn(9, d, generator=g) / d ** 0.5 * (0.5 + torch.rand(9, 1, generator=g) * 3))
    rs = np.random.RandomState(17)
    block = [sorted(rs.choice(1100, int(rs.randint(120, 201)), replace=False).tolist()) for _ in range(9)]
    block[3] = []                      # a query without blocked passages
    block[5] = list(range(40))         # fewer than N - 1001 = 99 blocked: none of them is kept
    ids, sc = _run_ranking(mod, Eq, Ed, 256, "cos", block=block)
    flat = np.concatenate([np.asarray(b, np.int64) for b in block])
    ptr = np.cumsum([0] + [len(b) for b in block]).astype(np.int64)
    np.savez_compressed(f"{OUT}/g17_ranking_cos_block_trunc.npz", Eq=Eq.numpy(), Ed=Ed.numpy(), ids=ids, scores=sc, block_ptr=ptr, block_idx=flat,
                        batch_size=256)
    os.environ["CCREC_SIM_TYPE"] = "dot"


def g_item_tower():
    from ccrec.models.item_tower import NaiveItemTower

    g = torch.Generator().manual_seed(99)
    B, L, d = 3, 16, 768
    hidden = torch.randn(B, L, d, generator=g)
    mask = torch.zeros(B, L, dtype=torch.long)
    for b, n in enumerate([16, 5, 1]):
        mask[b, :n] = 1

    class FakeCls(torch.nn.Module):
        device = torch.device("cpu")

        def forward(self, **inputs):
            return types.SimpleNamespace(last_hidden_state=hidden)

    ln = torch.nn.LayerNorm(d, elementwise_affine=False)
    tower = NaiveItemTower(FakeCls(), ln)
    inputs = {"input_ids": torch.ones(B, L, dtype=torch.long), "attention_mask": mask}
    out = {}
    with torch.no_grad():
        for step in ["mean_pooling", "cls", "mean_layer_norm"]:
            out[step] = tower(**inputs, output_step=step).numpy()
    np.savez_compressed(f"{OUT}/g6_item_tower.npz", hidden=hidden.numpy(), mask=mask.numpy(), **out)


def g_item_tower_bert():
    """g16: the reference's NaiveItemTower around a REAL (local, seeded, random-init) transformers BertModel, fp32 on the CPU: the
    three output steps on ragged right-padded token batches.  Pins tower + encoder together (g6 pins the pooling on a given hidden
    state).  Weights are rounded to bf16-exact values and stored as their 16-bit patterns (half the bytes; and the kernel forward's bf16
    weight copies are then exact, so only activation rounding separates it from these outputs)."""
    from transformers import BertConfig, BertModel
    from ccrec.models.item_tower import NaiveItemTower
    torch.manual_seed(1234)
    cfg = dict(vocab_size=64, hidden_size=256, num_hidden_layers=1, num_attention_heads=4, intermediate_size=256, max_position_embeddings=40)
    model = BertModel(BertConfig(**cfg)).eval()
    with torch.no_grad():
        for name, prm in model.named_parameters():
            if "attention.self.query.weight" in name or "attention.self.key.weight" in name:
                prm.mul_(12.0)                      # the default init leaves every softmax uniform
            if "LayerNorm.weight" in name:
                prm.uniform_(0.6, 1.4)
            if "bias" in name:
                prm.normal_(0.0, 0.05)
            prm.copy_(bf16_exact(prm))
    lens = [32, 1, 17, 5, 24, 9]
    g = torch.Generator().manual_seed(7)
    ids = torch.zeros(len(lens), 32, dtype=torch.long)
    mask = torch.zeros(len(lens), 32, dtype=torch.long)
    for b, n in enumerate(lens):
        ids[b, :n] = torch.randint(1, 64, (n,), generator=g)
        mask[b, :n] = 1
    tower = NaiveItemTower(model, torch.nn.LayerNorm(256, elementwise_affine=False))
    out = {}
    with torch.no_grad():
        for step in ["mean_pooling", "cls", "mean_layer_norm"]:
            out["out_" + step] = tower(input_ids=ids, attention_mask=mask, output_step=step).numpy()
    state = {"w_" + k: v.to(torch.bfloat16).view(torch.int16).numpy() for k, v in model.state_dict().items() if v.dtype == torch.float32}
    np.savez_compressed(f"{OUT}/g16_item_tower_bert.npz", ids=ids.numpy(), mask=mask.numpy(), config=np.array(json.dumps(cfg)), **state, **out)


def g_contrastive():
    from ccrec.models.bbpr import _BertBPR

    res = {}
    for tag, B, sim in [("b8_dot", 8, "dot"), ("b32_dot", 32, "dot"), ("b8_cos", 8, "cos"), ("b32_cos", 32, "cos")]:
        os.environ["CCREC_SIM_TYPE"] = sim
        os.environ["CCREC_BBPR_INV_TEMPERATURE"] = "20"
        g = torch.Generator().manual_seed(5 + B)
        d = 768
        scale = 1 / d ** 0.5 if sim == "dot" else 1.0
        E = (torch.randn(3 * B, d, generator=g) * scale).requires_grad_(True)
        m = _BertBPR.__new__(_BertBPR)
        torch.nn.Module.__init__(m)
        m.objective = "multiple_nrl"
        m.i_to_ptr = torch.arange(0, B)
        m.j_to_ptr = torch.arange(B, 3 * B)          # j-space: [pos(0..B-1), neg(B..2B-1)]
        m.user_to_negs = {u: [B + u] for u in range(B)}
        m.forward = lambda ptr: E[ptr]
        batch = torch.stack([torch.arange(B), torch.arange(B), torch.ones(B, dtype=torch.long)], 1)
        loss = m.training_and_validation_step(batch, 0)
        loss.backward()
        res[f"{tag}_E"] = E.detach().numpy()
        res[f"{tag}_loss"] = np.float32(loss.item())
        res[f"{tag}_grad"] = E.grad.numpy()
    np.savez_compressed(f"{OUT}/g7_contrastive.npz", **res)


def g_assign_topk():
    from rime_lite.util import _assign_topk
    from rime_lite.util.score_array import auto_cast_lazy_score

    g = torch.Generator().manual_seed(21)
    d = 768
    U = bf16_exact(torch.randn(50, d, generator=g) / d ** 0.5).numpy()
    V = bf16_exact(torch.randn(4000, d, generator=g) / d ** 0.5).numpy()
    S = auto_cast_lazy_score(U) @ auto_cast_lazy_score(V).T
    with contextlib.redirect_stdout(io.StringIO()):
        csr = _assign_topk(S, 10)
    np.savez_compressed(f"{OUT}/g8_assign_topk.npz", U=U, V=V, indices=csr.indices.reshape(50, 10).astype(np.int64),
                        indptr=csr.indptr.astype(np.int64), k=10)


def g_assign_topk_odd_width():
    """G15: the reference's rime_lite takes factors of ANY width (score_array.py:320-339): _assign_topk on 300- and 50-wide
    MatMulExpressions (the widths the fused kernels' zero-filled K tail and the zero-padding pack exist for)."""
    _import_reference("dot")
    from rime_lite.util import _assign_topk
    from rime_lite.util.score_array import auto_cast_lazy_score

    out = {}
    for d, nu, nv, k in ((300, 40, 3000, 10), (50, 30, 2500, 7)):
        g = torch.Generator().manual_seed(1000 + d)
        U = bf16_exact(torch.randn(nu, d, generator=g) / d ** 0.5).numpy()
        V = bf16_exact(torch.randn(nv, d, generator=g) / d ** 0.5).numpy()
        S = auto_cast_lazy_score(U) @ auto_cast_lazy_score(V).T
        with contextlib.redirect_stdout(io.StringIO()):
            csr = _assign_topk(S, k)
        out.update({f"U{d}": U, f"V{d}": V, f"indices{d}": csr.indices.reshape(nu, k).astype(np.int64), f"k{d}": k})
    np.savez_compressed(f"{OUT}/g15_assign_topk_odd_width.npz", **out)


def g_pack():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(64, 768, generator=g)
    x[0, :8] = torch.tensor([0.0, -0.0, float("inf"), -float("inf"), 1e-40, -1e-40, 3.3895314e38, 1.0])
    # exact half-way cases: bf16 ulp at 1.0 is 2^-7; 1 + 2^-8 ties to even (1.0), 1 + 3*2^-8 ties to 1+2^-6
    x[1, :4] = torch.tensor([1 + 2.0 ** -8, 1 + 3 * 2.0 ** -8, -(1 + 2.0 ** -8), 1 + 2.0 ** -8 + 2.0 ** -20])
    bits = x.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    np.savez_compressed(f"{OUT}/g9_pack_bf16.npz", x=x.numpy(), bits=bits)


def _reference_lines(rel_path, first_marker, last_marker=None):
    """Source text of a reference script between two marker lines (inclusive of the first; to EOF if no last)."""
    lines = open(os.path.join(REF, rel_path)).read().split("\n")
    lo = next(i for i, l in enumerate(lines) if l.startswith(first_marker))
    hi = len(lines) if last_marker is None else next(i for i, l in enumerate(lines) if i > lo and l.startswith(last_marker))
    return "\n".join(lines[lo:hi])


def _toy_profiles(seed, n_docs=40, n_queries=9, depth=12):
    rs = np.random.RandomState(seed)
    alphabet = list("abcdefghij KLMNOP,:.;?$!()&[]#@%^*~\u00e9\u4e2d")
    corpus = {str(100 + j): "".join(rs.choice(alphabet, rs.randint(5, 60))) for j in range(n_docs)}
    queries = {str(j): "query " + "".join(rs.choice(alphabet, rs.randint(3, 20))) for j in range(n_queries)}
    keys = list(corpus)

    def profile(overlap_with=None):
        prof = {}
        for qid in queries:
            order = [keys[i] for i in rs.permutation(n_docs)[:depth]]
            if overlap_with is not None and rs.rand() < 0.7:     # BM25 often repeats the dense top-2
                order[:2] = list(overlap_with[qid])[:2][::-1]
            prof[qid] = {pid: float(depth - r) for r, pid in enumerate(order)}
        return prof
    dense = profile()
    return corpus, queries, dense, profile(dense)


def g_requests():
    """G10: run the reference's own request-creation block on toy inputs, with and without landingImage."""
    import json
    import tempfile
    import pandas as pd
    import re as _re
    src = _reference_lines("scripts/al_0_rank.py", "## creation")
    cases = {}
    for name, seed, step, with_img in (("plain", 0, 1, False), ("images", 1, 2, True)):
        corpus, queries, dense, bm25 = _toy_profiles(seed)
        qids = list(queries)
        splits = [qids[0::3], qids[1::3], qids[2::3]]
        with tempfile.TemporaryDirectory() as tmp:
            os.environ["CCREC_DISPLAY_LENGTH"] = "25"
            ns = dict(np=np, pd=pd, re=_re, os=os, torch=torch, corpus=corpus, queries=queries, ranking_profile=dense,
                      ranking_profile_bm25=bm25, qids_split=splits, STEP=step, number_of_qid_split_batch=3,
                      landingImage=({**{q: f"img_q{q}.jpg" for q in queries}, **{p: f"img_{p}.jpg" for p in corpus}}
                                    if with_img else None),
                      current_working_dir=tmp, N_REPEATS=2, REPEAT_SEED=7)
            with contextlib.redirect_stdout(io.StringIO()):
                exec(compile(src, "al_0_rank.py[creation]", "exec"), ns)
            cases[name] = {
                "inputs": {"corpus": corpus, "queries": queries, "ranking_profile": dense, "ranking_profile_bm25": bm25,
                           "qids_split": splits, "STEP": step, "number_of_qid_split_batch": 3, "landingImage": ns["landingImage"],
                           "N_REPEATS": 2, "REPEAT_SEED": 7, "CCREC_DISPLAY_LENGTH": 25},
                "request_orig_csv": open(os.path.join(tmp, "request_orig.csv")).read(),
                "request_perm_csv": open(os.path.join(tmp, "request_perm.csv")).read(),
                "id_track": torch.load(os.path.join(tmp, "id_track.pt")),
            }
    json.dump(cases, open(os.path.join(OUT, "g10_requests.json"), "w"), ensure_ascii=True, indent=0)
    print("g10_requests.json", {k: len(v["request_perm_csv"]) for k, v in cases.items()})


def g_train_data():
    """G11: the reference's generate_train_data (al_oracle_agent.py) on toy inputs; random.shuffle seeded."""
    import json
    import random
    src = _reference_lines("scripts/al_oracle_agent.py", "def generate_train_data(", "def combine_train_data(")
    ns = dict(np=np, random=random)
    exec(compile(src, "al_oracle_agent.py[generate_train_data]", "exec"), ns)
    cases = {}
    for name, seed, with_keys in (("no_random_pad", 3, False), ("attention_check", 4, True)):
        corpus, queries, dense, bm25 = _toy_profiles(seed)
        rs = np.random.RandomState(seed)
        qrels = {q: {list(dense[q])[int(rs.randint(0, 6))]: 1} for q in queries}
        qids = list(queries)[::2] + list(queries)[1::2]
        random.seed(1234 + seed)
        out = ns["generate_train_data"](qids, qrels, dense, bm25, list(corpus) if with_keys else [], rng_seed=seed)
        cases[name] = {"inputs": {"qids": qids, "qrels": qrels, "ranking_profile": dense, "ranking_profile_2": bm25,
                                  "corpus_key_list": list(corpus) if with_keys else [], "rng_seed": seed,
                                  "random_seed": 1234 + seed},
                       "train_data": out}
    json.dump(cases, open(os.path.join(OUT, "g11_train_data.json"), "w"), indent=0)
    print("g11_train_data.json", {k: len(v["train_data"]) for k, v in cases.items()})


def g_bm25():
    """G12: dense BM25 score vectors from the reference's BM25.transform and the ranking_bm25 profile."""
    import json
    sys.path.insert(0, os.path.join(REF, "scripts"))
    import bm_25
    rs = np.random.RandomState(12)
    words = [f"w{i}" for i in range(120)] + ["Alpha", "beta-gamma", "x", "42", "caf\u00e9", "a1", "THE", "the"]
    zipf = 1.0 / np.arange(1, len(words) + 1)
    zipf /= zipf.sum()
    corpus = {f"d{j}": " ".join(rs.choice(words, rs.randint(1, 40), p=zipf)) for j in range(400)}
    corpus["d7"] = corpus["d3"]                                  # duplicate document: exact score tie
    queries = {f"q{i}": " ".join(rs.choice(words, rs.randint(1, 14), p=zipf)) for i in range(12)}
    queries["q_oov"] = "zzzz qqqq"                               # no vocabulary term: all scores 0
    queries["q_rep"] = "w0 w0 w0 w1 the THE"                     # repeated terms count once
    model = bm_25.BM25(b=0.75, k1=1.2).fit(list(corpus.values()))
    dense = np.stack([model.transform(q) for q in queries.values()])          # fp64 [nq, n_docs]
    src = _reference_lines("scripts/ms_marco_eval.py", "def ranking_bm25(", "def ranking(")
    ns = dict(BM25=bm_25.BM25, torch=torch)
    exec(compile(src, "ms_marco_eval.py[ranking_bm25]", "exec"), ns)
    with contextlib.redirect_stdout(io.StringIO()):
        prof = ns["ranking_bm25"](corpus, queries)
    json.dump({"corpus": corpus, "queries": queries, "b": 0.75, "k1": 1.2,
               "vocabulary": {k: int(v) for k, v in model.vectorizer.vocabulary_.items()},
               "avdl": float(model.avdl), "profile": prof}, open(os.path.join(OUT, "g12_bm25.json"), "w"), indent=0)
    np.savez_compressed(os.path.join(OUT, "g12_bm25_scores.npz"), dense=dense)
    print("g12_bm25", dense.shape, len(model.vectorizer.vocabulary_))


def g_item_rec():
    """G13: evaluate_item_rec(target, MatMulExpression(U @ V.T), k) from the reference's rime_lite."""
    _import_reference("dot")
    import scipy.sparse as sps
    from rime_lite.metrics import evaluate_item_rec
    from rime_lite.util import auto_cast_lazy_score
    g = torch.Generator().manual_seed(13)
    U = bf16_exact(torch.randn(40, 64, generator=g) / 8).numpy()
    V = bf16_exact(torch.randn(900, 64, generator=g) / 8).numpy()
    rs = np.random.RandomState(13)
    true = U.astype(np.float64) @ V.astype(np.float64).T
    rows, cols = [], []
    for u in range(40):                       # relevant items: a random third of each user's 30 best + 5 random ones
        best = np.argsort(-true[u])[:30]
        pick = set(best[rs.rand(30) < 0.33].tolist()) | set(rs.randint(0, 900, 5).tolist())
        rows += [u] * len(pick)
        cols += sorted(pick)
    target = sps.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(40, 900))
    S = auto_cast_lazy_score(U) @ auto_cast_lazy_score(V).T
    out = evaluate_item_rec(target, S, 7, tie_breaker=0)
    np.savez_compressed(os.path.join(OUT, "g13_item_rec.npz"), U=U, V=V, target_indptr=target.indptr, target_indices=target.indices,
                        k=7, **{"m_" + k.replace("/", "_"): np.float64(v) for k, v in out.items()})
    print("g13_item_rec", out)


def g_sparse_prior():
    """G14: the reference's rime_lite on `transform(D) + D.prior_score`-shaped scores (dense low-rank + sparse prior)."""
    _import_reference("dot")
    import scipy.sparse as sps
    from rime_lite.metrics import evaluate_item_rec
    from rime_lite.util import _assign_topk, auto_cast_lazy_score
    from rime_lite.util.score_array import score_op
    g = torch.Generator().manual_seed(14)
    nu, ni, d, k = 60, 1200, 64, 5
    U = bf16_exact(torch.randn(nu, d, generator=g) / 8).numpy()
    V = bf16_exact(torch.randn(ni, d, generator=g) / 8).numpy()
    rs = np.random.RandomState(14)
    rows, cols, vals = [], [], []
    for u in range(nu):
        if u == 7:      # almost every column carries a prior: more entries than any over-fetch can hold
            c = np.sort(rs.choice(ni, ni - 2, replace=False))
            v = rs.uniform(-0.5, 0.5, c.size)
        elif u == 11:   # no prior at all
            c, v = np.zeros(0, np.int64), np.zeros(0)
        else:
            c = np.sort(rs.choice(ni, rs.randint(1, 7), replace=False))
            v = np.where(rs.rand(c.size) < 0.5, 1e5, rs.uniform(-2.0, 2.0, c.size))   # reranking_prior=1e5 + small +- values
        rows += [u] * c.size
        cols += c.tolist()
        vals += v.tolist()
    P = sps.csr_matrix((np.asarray(vals, np.float64), (rows, cols)), shape=(nu, ni))
    low = auto_cast_lazy_score(U) @ auto_cast_lazy_score(V).T
    S = low + auto_cast_lazy_score(P)
    with contextlib.redirect_stdout(io.StringIO()):
        csr = _assign_topk(S, k, tie_breaker=0)
        true = (U.astype(np.float64) @ V.astype(np.float64).T) + P.toarray()
        t_rows, t_cols = [], []
        for u in range(nu):                   # relevant items: half of each user's 4 best under the combined score
            best = np.argsort(-true[u])[:4]
            pick = sorted(set(best[rs.rand(4) < 0.5].tolist()) | {int(rs.randint(0, ni))})
            t_rows += [u] * len(pick)
            t_cols += pick
        target = sps.csr_matrix((np.ones(len(t_rows)), (t_rows, t_cols)), shape=(nu, ni))
        metrics = evaluate_item_rec(target, S, 1, tie_breaker=0)
        ops_low = {op: float(score_op(low, op)) for op in ("max", "min", "sum")}
        ops_sum = {op: float(score_op(S, op)) for op in ("max", "min", "sum")}
    full = S.as_tensor("cpu").numpy()
    np.savez_compressed(os.path.join(OUT, "g14_sparse_prior.npz"), U=U, V=V, prior_indptr=P.indptr.astype(np.int64),
                        prior_indices=P.indices.astype(np.int64), prior_data=P.data.astype(np.float64), k=k,
                        indices=csr.indices.reshape(nu, k).astype(np.int64),
                        topk_scores=np.take_along_axis(full, csr.indices.reshape(nu, k), 1).astype(np.float64),
                        target_indptr=target.indptr, target_indices=target.indices,
                        **{"m_" + kk.replace("/", "_"): np.float64(v) for kk, v in metrics.items()},
                        **{"low_" + kk: np.float64(v) for kk, v in ops_low.items()},
                        **{"sum_" + kk: np.float64(v) for kk, v in ops_sum.items()})
    print("g14_sparse_prior", metrics, ops_low, ops_sum)


def g_ranking_profile_fn():
    """g18: the reference's OWN generate_ranking_profile (scripts/al_oracle_agent.py:83-129; twin of al_0_rank.py:69-105),
    its source executed in a prepared namespace the g10 / g11 way (the script is not importable: it parses argv at import):
    AutoTokenizer.from_pretrained -> the deterministic toy tokenizer of tests/helpers.py, the model a `_BertMT`-shaped object whose
    .item_tower is the reference's NaiveItemTower around a local numpy-seeded BertModel, DataParallel = the reference's cached-replica
    class (no devices here: it calls the module), ranking = the reference's, EvaluateRetrieval -> an inert stub (beir is absent: the MRR
    print is not part of the fixture).  Cases: dot without blocks, cos with a block_dict.  Stored: texts, the seed / config of the
    encoder, the fp32 embeddings the reference's embedding_func produced, and the returned rank-ordered profile (ids + scores)."""
    import types
    import warnings
    mod = _import_reference("dot")
    sys.path.insert(0, os.path.join(os.path.dirname(OUT)))
    from helpers import G18_CFG, GoldenTokenizer, golden_texts, numpy_seeded_bert
    from ccrec.models.item_tower import NaiveItemTower
    from ccrec.util.data_parallel import DataParallel
    torch.nn.Module.cuda = lambda self, *a, **k: self
    src = _reference_lines("scripts/al_oracle_agent.py", "def generate_ranking_profile(", "# %%")
    res = {"config": np.array(json.dumps(G18_CFG)), "seed": 18, "max_length": 24}
    corpus = {f"p{j}": t for j, t in enumerate(golden_texts(260, 181))}
    queries = {f"q{i}": t for i, t in enumerate(golden_texts(7, 182, longest=8))}
    rs = np.random.RandomState(183)
    block = {q: [f"p{int(j)}" for j in rs.choice(260, 4, replace=False)] for q in queries}
    res["corpus_texts"], res["query_texts"] = np.array(list(corpus.values())), np.array(list(queries.values()))
    res["block"] = np.array([[int(p[1:]) for p in block[q]] for q in queries])
    for tag, sim, blk in (("dot", "dot", None), ("cos_block", "cos", block)):
        os.environ["CCREC_SIM_TYPE"] = sim
        os.environ["CCREC_EMBEDDING_TYPE"] = "mean_pooling"
        os.environ["CCREC_MAX_LENGTH"] = "24"
        tower = NaiveItemTower(numpy_seeded_bert(G18_CFG, 18), torch.nn.LayerNorm(64, elementwise_affine=False))
        recorded = []
        fwd = tower.forward

        def recording(*a, _f=fwd, **k):
            out = _f(*a, **k)
            recorded.append(out.detach().float().numpy().copy())
            return out
        tower.forward = recording
        auto_tok = types.SimpleNamespace(from_pretrained=lambda name: GoldenTokenizer(64))
        evaluator = type("EvaluateRetrieval", (), {"__init__": lambda self, *a: None, "evaluate_custom": lambda self, *a, **k: {}})
        save_to = os.path.join(OUT, "_g18_tmp.pt")
        ns = dict(torch=torch, os=os, warnings=warnings, AutoTokenizer=auto_tok, DataParallel=DataParallel, ranking=mod.ranking,
                  EvaluateRetrieval=evaluator)
        exec(compile(src, "al_oracle_agent.py[generate_ranking_profile]", "exec"), ns)
        with contextlib.redirect_stdout(io.StringIO()):
            prof = ns["generate_ranking_profile"](types.SimpleNamespace(item_tower=tower), "unused", corpus, queries, {}, save_to, blk)
        assert torch.load(save_to) == prof
        os.remove(save_to)
        emb = np.concatenate(recorded, 0)           # ranking() encodes the queries first, then the corpus
        res[f"{tag}_query_emb"], res[f"{tag}_corpus_emb"] = emb[:len(queries)], emb[len(queries):]
        res[f"{tag}_ids"] = np.array([[int(p[1:]) for p in prof[q]] for q in queries], np.int64)
        res[f"{tag}_scores"] = np.array([list(prof[q].values()) for q in queries], np.float32)
        assert list(prof) == list(queries)
    np.savez_compressed(os.path.join(OUT, "g18_ranking_profile_fn.npz"), **res)
    print("g18_ranking_profile_fn", res["dot_ids"].shape, res["cos_block_scores"][0, :3])


def g_bertbpr_transform():
    """g19: the reference's OWN BertBPR.get_all_embeddings and BertBPR.transform (src/ccrec/models/bbpr.py:466-550), bound
    methods of a `BertBPR.__new__` instance (the constructor downloads a tokenizer and builds Lightning loggers): item_titles, the toy
    tokenizer with the class's own tokenizer_kw (padding="max_length", bbpr.py:359-364), model.item_tower = the reference's
    NaiveItemTower around a local numpy-seeded 768-wide BertModel (get_all_embeddings allocates [n, 768]), `_get_data_module` -> an
    object with the i_to_ptr / j_to_ptr index maps of bbpr.py:287-293.  Stored: titles, index maps, the [n_items, 768] fp32 embeddings
    and the dense score matrices (dot and cos) the reference returned."""
    import types
    import pandas as pd
    _import_reference("dot")
    sys.path.insert(0, os.path.join(os.path.dirname(OUT)))
    from helpers import G19_CFG, GoldenTokenizer, golden_texts, numpy_seeded_bert
    from ccrec.models.bbpr import BertBPR
    from ccrec.models.item_tower import NaiveItemTower
    n_items = 90
    titles = golden_texts(n_items, 191, longest=12)
    rs = np.random.RandomState(192)
    i_to_ptr = [int(x) for x in rs.choice(n_items, 11, replace=False)]
    j_to_ptr = [int(x) for x in rs.permutation(n_items)[:70]]
    res = {"config": np.array(json.dumps(G19_CFG)), "seed": 19, "max_length": 16, "titles": np.array(titles),
           "i_to_ptr": np.array(i_to_ptr), "j_to_ptr": np.array(j_to_ptr)}
    os.environ["CCREC_EMBEDDING_TYPE"] = "mean_pooling"
    for sim in ("dot", "cos"):
        os.environ["CCREC_SIM_TYPE"] = sim
        bb = BertBPR.__new__(BertBPR)
        bb.item_titles = pd.Series(titles, index=[f"i{j}" for j in range(n_items)])
        bb.tokenizer = GoldenTokenizer(64)
        bb.tokenizer_kw = dict(padding="max_length", return_tensors="pt", max_length=16, truncation=True)
        bb.model = types.SimpleNamespace(item_tower=NaiveItemTower(numpy_seeded_bert(G19_CFG, 19), torch.nn.LayerNorm(768, elementwise_affine=False)))
        bb._get_data_module = lambda D: types.SimpleNamespace(i_to_ptr=i_to_ptr, j_to_ptr=j_to_ptr)
        with contextlib.redirect_stdout(io.StringIO()):
            if sim == "dot":
                res["all_emb"] = bb.get_all_embeddings(bb.model.item_tower, 32).numpy()
            S = bb.transform(None)
        res[f"scores_{sim}"] = np.asarray(S.as_tensor("cpu").numpy(), np.float32)
        assert res[f"scores_{sim}"].shape == (11, 70)
    np.savez_compressed(os.path.join(OUT, "g19_bertbpr_transform.npz"), **res)
    print("g19_bertbpr_transform", res["all_emb"].shape, res["scores_dot"][0, :3], res["scores_cos"][0, :3])


def main():
    """No arguments: every fixture.  `make_golden.py g10 g11`: only the named groups (g1 = all ranking fixtures)."""
    os.makedirs(OUT, exist_ok=True)
    want = set(sys.argv[1:])
    groups = [("g1", lambda: g_ranking(_import_reference("dot"))), ("g6", g_item_tower), ("g7", g_contrastive),
              ("g8", g_assign_topk), ("g9", g_pack), ("g10", g_requests), ("g11", g_train_data), ("g12", g_bm25), ("g13", g_item_rec), ("g14", g_sparse_prior),
              ("g15", g_assign_topk_odd_width), ("g16", lambda: (_import_reference("dot"), g_item_tower_bert())[1]),
              ("g17", lambda: g_ranking_cos_block(_import_reference("dot"))), ("g18", g_ranking_profile_fn), ("g19", g_bertbpr_transform)]
    for name, fn in groups:
        if not want or name in want:
            fn()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
