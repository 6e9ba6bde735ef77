"""the request builder / simulated annotator that consume a ranking_profile.  Fixtures g10/g11 were
produced by executing the reference's own source (tools/make_golden.py g10 g11); the mirror must reproduce the CSV
text, the id_track mapping and the training dictionaries exactly."""
import io
import json
import os
import random

import pytest
import torch


def _load(golden_dir, name):
    return json.load(open(os.path.join(golden_dir, name)))


@pytest.mark.parametrize("case", ["plain", "images"])
def test_build_requests_reproduces_reference_files(golden_dir, tmp_path, case):
    from ccrec_amd.al_request import build_requests
    g = _load(golden_dir, "g10_requests.json")[case]
    i = g["inputs"]
    step = i["STEP"]
    os.environ["CCREC_DISPLAY_LENGTH"] = str(i["CCREC_DISPLAY_LENGTH"])
    out = build_requests(i["ranking_profile"], i["ranking_profile_bm25"], i["qids_split"][step % i["number_of_qid_split_batch"]],
                         i["corpus"], i["queries"], step, n_repeats=i["N_REPEATS"], repeat_seed=i["REPEAT_SEED"],
                         landing_image=i["landingImage"], out_dir=str(tmp_path))
    assert open(tmp_path / "request_orig.csv").read() == g["request_orig_csv"]
    assert open(tmp_path / "request_perm.csv").read() == g["request_perm_csv"]
    assert torch.load(tmp_path / "id_track.pt") == g["id_track"] == out["id_track"]
    buf = io.StringIO()
    out["request_perm"].to_csv(buf, index=False)
    assert buf.getvalue() == g["request_perm_csv"]
    assert len(out["request_perm"]) == i["N_REPEATS"] * len(out["request_orig"])


@pytest.mark.parametrize("case", ["no_random_pad", "attention_check"])
def test_generate_train_data_matches_reference(golden_dir, case):
    from ccrec_amd.al_request import generate_train_data, combine_train_data
    g = _load(golden_dir, "g11_train_data.json")[case]
    i = g["inputs"]
    random.seed(i["random_seed"])
    got = generate_train_data(i["qids"], i["qrels"], i["ranking_profile"], i["ranking_profile_2"], i["corpus_key_list"],
                              rng_seed=i["rng_seed"])
    assert got == g["train_data"] and list(got) == list(g["train_data"])
    assert combine_train_data({"x": 1}, got)["x"] == 1


def test_filter_string_and_candidate_rules():
    import numpy as np
    from ccrec_amd.al_request import filter_string, pick_candidates
    assert filter_string("a#bé [c]{d}~$!", 100) == "ab [c]d$!"
    assert filter_string("abcdef", 3) == "abc"
    keys = [f"p{i}" for i in range(10)]
    rng = np.random.RandomState(0)
    c = pick_candidates(["p1", "p2", "p3"], ["p2", "p1", "p7", "p8"], keys, rng)
    assert c[:3] == ["p1", "p2", "p7"] and len(c) == 4 and len(set(c)) == 4
    # BM25 list exhausted without a new passage: two random pads
    c = pick_candidates(["p1", "p2"], ["p2", "p1"], keys, np.random.RandomState(1))
    assert c[:2] == ["p1", "p2"] and len(set(c)) == 4
