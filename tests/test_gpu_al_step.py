"""One active-learning step end to end (al_0_rank.py:107-218) on the device path: encode -> fused search -> MRR ->
BM25 -> request files, then the resume behaviour (a second call reuses ranking_profile.pt)."""
import os

import numpy as np
import pandas as pd
import pytest
import torch

from test_gpu_encode import ToyTokenizer, _tower, _texts

pytestmark = pytest.mark.gpu


def test_run_rank_step_writes_the_reference_files(tmp_path):
    from ccrec_amd.al_step import run_rank_step
    os.environ["CCREC_SIM_TYPE"] = "dot"
    os.environ["CCREC_DISPLAY_LENGTH"] = "40"
    corpus = {f"{j}": t for j, t in enumerate(_texts(1500, 11))}
    queries = {f"{i}": t for i, t in enumerate(_texts(24, 12, 2, 9))}
    rs = np.random.RandomState(0)
    qrels = {q: {str(int(rs.randint(0, 1500))): 1} for q in queries}
    step_qids = list(queries)[::2]
    tower, tok = _tower(), ToyTokenizer()
    out = run_rank_step(tower, tok, corpus, queries, qrels, step_qids, step=0, results_dir=str(tmp_path),
                        encoder_kw={"max_length": 32, "max_tokens": 4096}, autocast=False)
    work = tmp_path / "data_iteration_0"
    for f in ("ranking_profile.pt", "request_orig.csv", "request_perm.csv", "id_track.pt"):
        assert (work / f).is_file()
    prof = out["ranking_profile"]
    assert list(prof) == list(queries) and all(len(v) == 1001 for v in prof.values())
    assert set(out["mrr"]) == {f"{m}@{k}" for m in ("MRR", "Recall") for k in (1, 5, 10, 100)}
    orig = pd.read_csv(work / "request_orig.csv", dtype=str)
    perm = pd.read_csv(work / "request_perm.csv", dtype=str)
    assert len(orig) == len(step_qids) and len(perm) == 3 * len(orig)
    for _, row in orig.iterrows():
        qid = row["qid"][2:]
        pids = [row[f"pid-{i}"][2:] for i in range(1, 5)]
        assert pids[:2] == list(prof[qid])[:2] and len(set(pids)) == 4          # top-2 dense first, four distinct
    assert sorted(perm["qid"].tolist()) == sorted(orig["qid"].tolist() * 3)
    # resume: the profile on disk is reused (no encoder needed)
    again = run_rank_step(None, None, corpus, queries, qrels, step_qids, step=0, results_dir=str(tmp_path),
                          ranking_profile_bm25={q: prof[q] for q in queries})
    assert again["ranking_profile"] == prof and again["mrr"] == out["mrr"]


def test_ranking_profile_pt_is_the_references_nested_dict_by_default(tmp_path):
    """al_0_rank.py:127 writes torch.save(ranking_profile) of the nested {qid: {pid: score}} dict and :118 reads it back with a plain
    torch.load: a RESULTS_DIR shared with the reference's scripts must resume either way.  Default = that form (rank-ordered inner
    dicts); compat_profile=False = the tensor form, which this package's loader (and only it) turns back into a Mapping."""
    from ccrec_amd import ranking_profile
    from ccrec_amd.al_step import run_rank_step
    os.environ["CCREC_SIM_TYPE"] = "dot"
    os.environ["CCREC_DISPLAY_LENGTH"] = "40"
    corpus = {f"{j}": t for j, t in enumerate(_texts(1200, 21))}
    queries = {f"{i}": t for i, t in enumerate(_texts(6, 22, 2, 9))}
    qrels = {q: {"3": 1} for q in queries}
    bm25 = {q: {"5": 2.0, "6": 1.0, "7": 0.5} for q in queries}
    tower, tok = _tower(), ToyTokenizer()
    kw = dict(encoder_kw={"max_length": 32, "max_tokens": 4096}, autocast=False, ranking_profile_bm25=bm25)
    out = run_rank_step(tower, tok, corpus, queries, qrels, list(queries)[:2], step=0, results_dir=str(tmp_path / "a"), **kw)
    raw = torch.load(tmp_path / "a" / "data_iteration_0" / "ranking_profile.pt", weights_only=False)
    assert type(raw) is dict and list(raw) == list(queries)
    for q in queries:
        assert type(raw[q]) is dict and list(raw[q].items()) == list(out["ranking_profile"][q].items())
        scores = list(raw[q].values())
        assert all(type(v) is float for v in scores) and scores == sorted(scores, reverse=True) and len(scores) == 1001
    out2 = run_rank_step(tower, tok, corpus, queries, qrels, list(queries)[:2], step=0, results_dir=str(tmp_path / "b"),
                         compat_profile=False, **kw)
    raw2 = torch.load(tmp_path / "b" / "data_iteration_0" / "ranking_profile.pt")          # weights_only default: tensors + strings
    assert isinstance(raw2, dict) and "format" in raw2
    assert ranking_profile.load(tmp_path / "b" / "data_iteration_0" / "ranking_profile.pt") == out2["ranking_profile"] == raw
    assert "save_s" in out["timings"] and out["timings"]["corpus_encoder"]["texts"] == 1200


def test_mrr_divides_by_the_number_of_queries_in_the_qrels(tmp_path):
    """BEIR's mrr sums over the queries of `results` and divides by len(qrels) -- the reference passes its FULL qrels
    (scripts/al_0_rank.py:130-133) -- and counts only documents with relevance > 0.  A profile over 8 of the qrels' 20 queries,
    with zero-relevance entries among the qrels: rank_metrics(n_qrels=20) == oracle.mrr_beir on the same dicts."""
    from oracle import oracle as orc
    from ccrec_amd.al_step import run_rank_step
    os.environ["CCREC_SIM_TYPE"] = "dot"
    os.environ["CCREC_DISPLAY_LENGTH"] = "40"
    corpus = {f"{j}": t for j, t in enumerate(_texts(1100, 31))}
    queries = {f"{i}": t for i, t in enumerate(_texts(8, 32, 2, 9))}
    tower, tok = _tower(), ToyTokenizer()
    bm25 = {q: {"5": 2.0, "6": 1.0, "7": 0.5} for q in queries}
    first = run_rank_step(tower, tok, corpus, queries, {q: {} for q in queries}, list(queries)[:2], step=0, results_dir=str(tmp_path / "probe"),
                          encoder_kw={"max_length": 32, "max_tokens": 4096}, autocast=False, ranking_profile_bm25=bm25)
    prof = first["ranking_profile"]
    rs = np.random.RandomState(3)
    qrels = {}
    for i in range(20):                                  # 20 queries in the qrels, only "0" .. "7" are ranked
        q = str(i)
        if q in queries:
            ranked = list(prof[q])
            r = int(rs.choice([0, 1, 3, 7, 40, 500]))
            qrels[q] = {ranked[r]: 1, ranked[0 if r else 2]: 0, "1099": int(rs.randint(0, 2))}      # a relevance-0 document ranked first
        else:
            qrels[q] = {"1": 1}
    out = run_rank_step(None, None, corpus, queries, qrels, list(queries)[:2], step=0, results_dir=str(tmp_path / "probe"),
                        ranking_profile_bm25=bm25)
    want = orc.mrr_beir(qrels, {q: dict(prof[q]) for q in queries}, [1, 5, 10, 100])
    got = {k: v for k, v in out["mrr"].items() if k.startswith("MRR")}
    assert got == want, (got, want)
    assert 0 < want["MRR@100"] < 8 / 20 + 1e-9             # the absent 12 queries count in the denominator
