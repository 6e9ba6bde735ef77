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


def _lazy(profile):
    """A nested-dict fixture as the tensor-backed RankingProfile the device path returns (corpus order = sorted pids)."""
    import numpy as np
    from ccrec_amd.ranking_profile import RankingProfile
    corpus_ids = sorted({p for inner in profile.values() for p in inner})
    pos = {p: i for i, p in enumerate(corpus_ids)}
    rows = np.array([[pos[p] for p in inner] for inner in profile.values()], dtype=np.int64)
    scores = np.array([list(inner.values()) for inner in profile.values()], dtype=np.float32)
    return RankingProfile(list(profile), corpus_ids, rows, scores), corpus_ids


@pytest.mark.parametrize("case", ["plain", "images"])
def test_build_requests_from_a_lazy_profile_reproduces_reference_files(golden_dir, tmp_path, case):
    """The same fixture through the lazy profile: identical files, and only the step's queries were materialised."""
    from ccrec_amd.al_request import build_requests
    g = _load(golden_dir, "g10_requests.json")[case]
    i = g["inputs"]
    step = i["STEP"]
    os.environ["CCREC_DISPLAY_LENGTH"] = str(i["CCREC_DISPLAY_LENGTH"])
    lens = {len(v) for v in i["ranking_profile"].values()}
    if len(lens) != 1:
        pytest.skip("ragged fixture lists have no tensor form")
    prof, _ = _lazy(i["ranking_profile"])
    step_qids = i["qids_split"][step % i["number_of_qid_split_batch"]]
    bm25 = i["ranking_profile_bm25"]
    if len({len(v) for v in bm25.values()}) == 1:     # the BM25 leg as a lazy profile too (the rule reads its first three passages)
        bm25, _ = _lazy(bm25)
    build_requests(prof, bm25, step_qids, i["corpus"], i["queries"], step, n_repeats=i["N_REPEATS"],
                   repeat_seed=i["REPEAT_SEED"], landing_image=i["landingImage"], out_dir=str(tmp_path))
    assert open(tmp_path / "request_orig.csv").read() == g["request_orig_csv"]
    assert open(tmp_path / "request_perm.csv").read() == g["request_perm_csv"]
    assert torch.load(tmp_path / "id_track.pt") == g["id_track"]
    assert not prof._cache        # the request rule reads two ranks per query: no inner dict was built at all


def test_ranking_profile_mapping_and_file_forms(tmp_path):
    import pickle
    import numpy as np
    from ccrec_amd import ranking_profile as rp
    rs = np.random.RandomState(0)
    corpus_ids = [f"doc{j}" for j in range(500)] + [("tuple", 7)]          # ids of any hashable type
    qids = [f"q{i}" for i in range(40)]
    rows = np.stack([rs.permutation(501)[:30] for _ in qids]).astype(np.int64)
    scores = -np.sort(-rs.rand(40, 30).astype(np.float32), axis=1)
    prof = rp.RankingProfile(qids, corpus_ids, torch.from_numpy(rows), torch.from_numpy(scores))
    ref = {q: {corpus_ids[j]: float(s) for j, s in zip(rows[i], scores[i])} for i, q in enumerate(qids)}
    assert len(prof) == 40 and list(prof) == qids and "q3" in prof and "nope" not in prof and prof.get("nope") is None
    assert prof == ref and dict(prof) == ref and prof.to_dict() == ref
    assert list(prof["q7"]) == list(ref["q7"]) and list(prof["q7"].values()) == list(ref["q7"].values())   # rank order kept
    assert prof.top("q7", 2) == list(ref["q7"])[:2] and [k for k, _ in prof.items()] == qids
    with pytest.raises(KeyError):
        prof["nope"]
    assert set(prof._cache) == set(qids)                                    # every query was read above
    # the tensor form loads under torch.load's default (weights_only), the compat form is the reference's nested dict
    prof.save(tmp_path / "p.pt")
    raw = torch.load(tmp_path / "p.pt")
    assert raw["format"] == rp.FORMAT and len(rp._pids(raw["pids"])) == len(np.unique(rows)) and raw["rows"].shape == (40, 30)
    strs = rp.RankingProfile(qids, [f"d{j}" for j in range(501)], rows, scores)       # all-string ids are stored as ONE joined string
    assert strs.state()["pids"]["count"] == len(np.unique(rows)) and rp.from_state(strs.state()) == strs.to_dict()
    back = rp.load(tmp_path / "p.pt")
    assert isinstance(back, rp.RankingProfile) and back == ref and list(back["q0"]) == list(ref["q0"])
    prof.save(tmp_path / "c.pt", compat=True)
    assert torch.load(tmp_path / "c.pt") == ref and rp.load(tmp_path / "c.pt") == ref
    assert pickle.loads(pickle.dumps(prof)) == ref
    # back to row tensors in corpus order: directly, from the file form (rows index the named passages only), from a nested dict
    for p in (prof, back, ref):
        q2, r2, s2 = rp.as_tensors(p, corpus_ids)
        assert q2 == qids and np.array_equal(r2.numpy(), rows) and np.array_equal(s2.numpy(), scores)
