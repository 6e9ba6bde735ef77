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
