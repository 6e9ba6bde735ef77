"""The step right after the rank path: turn a ranking_profile into the labelling request of one
active-learning step, and into simulated-oracle training data.

Reference behaviour kept bit for bit (the parity tests replay fixtures produced by the reference's own code):
  * candidates per query (scripts/al_0_rank.py:166-181): the two best dense passages, then the best BM25 passage
    not already present, then corpus passages drawn with RandomState(STEP).choice(len(corpus)) until there are four
    (a draw that repeats a candidate is discarded, exactly as the reference's loop consumes the stream);
  * display text (:139-141): characters outside [a-zA-Z0-9 ,:.;?$!()&[]] removed, cut to CCREC_DISPLAY_LENGTH;
  * request_orig.csv columns (:144-158), `q_` / `p_` id prefixes, optional landingImage columns, id_track.pt mapping
    every displayed text to its prefixed id (:183-190);
  * request_perm.csv (:198-216): N_REPEATS passes over the rows, one RandomState(REPEAT_SEED).permutation(4) per row,
    applied to passages, pids and images alike;
  * generate_train_data (scripts/al_oracle_agent.py:134-181): top-2 dense + BM25 fill-up to four, optional
    attention-check passage, module-level random.shuffle, qrels decide positives.
Inputs are the {qid: {pid: score}} profiles that ccrec_amd.ms_marco_eval.ranking / encode.ranking_sharded return.
"""
import os
import random
import re

import numpy as np

_DISPLAY_DROP = re.compile(r"[^a-zA-Z0-9 ,:.;?$!()&\[\]]")
BASE_COLUMNS = ["query"] + [f"passage-{i}" for i in range(1, 5)] + ["qid"] + [f"pid-{i}" for i in range(1, 5)]
IMAGE_COLUMNS = ["img-q"] + [f"img-{i}" for i in range(1, 5)]


def filter_string(text, display_length=None):
    limit = int(os.environ["CCREC_DISPLAY_LENGTH"]) if display_length is None else int(display_length)
    return _DISPLAY_DROP.sub("", text)[:limit]


def pick_candidates(dense_order, bm25_order, corpus_keys, rng):
    """[pid] * 4 for one query; `rng` is the step-wide RandomState shared by all queries, consumed in query order."""
    chosen = list(dense_order[:2])
    for pid in bm25_order:
        if len(chosen) == 3:
            break
        if pid not in chosen:
            chosen.append(pid)
    while len(chosen) < 4:
        pid = corpus_keys[rng.choice(len(corpus_keys))]
        if pid not in chosen:
            chosen.append(pid)
    return chosen


def _shuffled(row, order, with_images):
    out = [row[0]] + [row[1 + i] for i in order] + [row[5]] + [row[6 + i] for i in order]
    if with_images:
        out += [row[10]] + [row[11 + i] for i in order]
    return out


def build_requests(ranking_profile, ranking_profile_bm25, step_qids, corpus, queries, step, n_repeats=3, repeat_seed=42,
                   landing_image=None, out_dir=None, display_length=None):
    """-> {"request_orig": DataFrame, "request_perm": DataFrame, "id_track": dict}; with out_dir also writes
    request_orig.csv, request_perm.csv and id_track.pt there (the reference's file names)."""
    import pandas as pd
    step_qids = set(step_qids)
    corpus_keys = list(corpus.keys())
    draw = np.random.RandomState(step)
    columns = BASE_COLUMNS + (IMAGE_COLUMNS if landing_image is not None else [])
    rows, id_track = [], {}
    top = getattr(ranking_profile, "top", None)   # a lazy RankingProfile names a query's best passages without building its dict
    bm25_top = getattr(ranking_profile_bm25, "top", None)
    for qid in ranking_profile:                   # (only the step's queries are read)
        if qid not in step_qids:
            continue
        dense_order = top(qid, 2) if top is not None else list(ranking_profile[qid].keys())   # the rule reads ranks[0:2] only
        # (and the first BM25 passage that is not one of those two: it is among the first three)
        bm25_order = bm25_top(qid, 3) if bm25_top is not None else list(ranking_profile_bm25[qid].keys())
        cands = pick_candidates(dense_order, bm25_order, corpus_keys, draw)
        shown = [filter_string(corpus[pid], display_length) for pid in cands]
        row = [queries[qid], *shown, f"q_{qid}", *(f"p_{pid}" for pid in cands)]
        if landing_image is not None:
            row += [landing_image[qid], *(landing_image[pid] for pid in cands)]
        rows.append(row)
        id_track[queries[qid]] = f"q_{qid}"
        id_track.update({text: f"p_{pid}" for pid, text in zip(cands, shown)})
    request_orig = pd.DataFrame(rows, columns=columns)
    order_rng = np.random.RandomState(repeat_seed)
    request_perm = pd.DataFrame([_shuffled(row, order_rng.permutation(4), landing_image is not None)
                                 for _ in range(n_repeats) for row in rows], columns=columns)
    if out_dir is not None:
        import torch
        os.makedirs(out_dir, exist_ok=True)
        torch.save(id_track, os.path.join(out_dir, "id_track.pt"))
        request_orig.to_csv(os.path.join(out_dir, "request_orig.csv"), index=False)
        request_perm.to_csv(os.path.join(out_dir, "request_perm.csv"), index=False)
    return {"request_orig": request_orig, "request_perm": request_perm, "id_track": id_track}


def generate_train_data(qids, qrels, ranking_profile, ranking_profile_2, corpus_key_list=(), rng_seed=None):
    """Simulated annotator (al_oracle_agent.py:134-181): {qid: {"pos_pid": [..], "neg_pid": [..]}}.
    Uses the module-level `random.shuffle` like the reference (seed `random` for reproducibility)."""
    draw = np.random.RandomState(rng_seed)
    corpus_key_list = list(corpus_key_list)
    out = {}
    for qid in qids:
        pids = list(ranking_profile[qid].keys())[:2]
        for pid in ranking_profile_2[qid].keys():
            if len(pids) == 4:
                break
            if pid not in pids:
                pids.append(pid)
        if corpus_key_list:       # one random passage as an attention check
            pids = pids[:3]
            while len(pids) < 4:
                pid = corpus_key_list[draw.choice(len(corpus_key_list))]
                if pid not in pids:
                    pids.append(pid)
        random.shuffle(pids)
        relevant = set(qrels[qid].keys())
        hits = [pid for pid in pids if pid in relevant]
        if hits:
            out[qid] = {"pos_pid": [hits[-1]], "neg_pid": [pid for pid in pids if pid not in relevant]}
        elif not corpus_key_list:   # the attention-check variant skips queries without a labelled passage
            out[qid] = {"pos_pid": pids[:1], "neg_pid": pids[1:]}
    return out


def combine_train_data(previous, new):
    previous.update(new)
    return previous
