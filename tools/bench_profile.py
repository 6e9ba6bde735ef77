#!/usr/bin/env python3
"""Host time around the search at the NQ shape: from the [Q, 1001] result tensors to the labelling request
of one active-learning step, with the lazy tensor-backed profile and with the reference's nested dict.

  python tools/bench_profile.py [--rows 2681468] [--queries 3452] [--step-queries 300]
Prints one JSON line: seconds of search, profile construction, MRR, request building and saving for both forms."""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2_681_468)
    ap.add_argument("--queries", type=int, default=3_452)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--step-queries", type=int, default=300)
    args = ap.parse_args()
    from bench import gen_rows
    from ccrec_amd import evaluation, ops
    from ccrec_amd.al_request import build_requests
    from ccrec_amd.ms_marco_eval import Retriever
    os.environ.setdefault("CCREC_DISPLAY_LENGTH", "200")
    dev = torch.device("cuda", 0)
    bounds = torch.empty(args.rows, device=dev)
    shard = ops.pack_bf16(gen_rows(args.rows, args.dim, 1234, dev), norm_bounds=bounds)
    qpack = ops.pack_bf16(gen_rows(args.queries, args.dim, 4321, dev))
    corpus_ids = [f"doc{j}" for j in range(args.rows)]
    corpus = dict.fromkeys(corpus_ids, "some passage text, with punctuation; and (brackets) [too]")
    qids = [f"q{i}" for i in range(args.queries)]
    queries = dict.fromkeys(qids, "what is the question?")
    rs = np.random.RandomState(0)
    qrels = {q: {corpus_ids[int(rs.randint(0, args.rows))]: 1} for q in qids}
    step_qids = qids[:args.step_queries]
    bm25 = {q: {corpus_ids[int(j)]: 1.0 for j in rs.randint(0, args.rows, 5)} for q in step_qids}
    retr = Retriever(corpus_ids, shard, norm_bounds=bounds)
    retr.corpus_id_array()
    wanted = {p for r in qrels.values() for p in r}
    pos = {pid: i for i, pid in enumerate(corpus_ids) if pid in wanted}     # prepared before the search (al_step.run_rank_step)
    rel = [[pos[p] for p in qrels[q]] for q in qids]
    tmp = tempfile.mkdtemp()
    out = {"rows": args.rows, "queries": args.queries, "keep": 1001, "step_queries": len(step_qids)}
    # warm every stage once (library load, pandas import, allocator): the numbers below are steady-state host time
    wp, wi, _ = retr.ranking_profile(qids[:8], qpack[:8], lazy=True, with_tensors=True)
    evaluation.rank_metrics(wi, rel[:8], (1, 5, 10, 100))
    build_requests(wp, bm25, step_qids[:4], corpus, queries, 0, out_dir=os.path.join(tmp, "warm"))
    for form in ("lazy", "nested_dict"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s, i = retr.search(qpack, 1001)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        profile, ids_t, _ = retr.ranking_profile(qids, qpack, lazy=(form == "lazy"), with_tensors=True)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        mrr = evaluation.rank_metrics(ids_t, rel, (1, 5, 10, 100))
        t3 = time.perf_counter()
        build_requests(profile, bm25, step_qids, corpus, queries, 0, out_dir=os.path.join(tmp, form))
        t4 = time.perf_counter()
        if form == "lazy":
            profile.save(os.path.join(tmp, "p.pt"))
        else:
            torch.save(profile, os.path.join(tmp, "d.pt"))
        t5 = time.perf_counter()
        out[form] = {"search_s": round(t1 - t0, 4), "search_plus_profile_s": round(t2 - t1, 4),
                     "profile_host_s": round((t2 - t1) - (t1 - t0), 4), "mrr_s": round(t3 - t2, 4),
                     "build_requests_s": round(t4 - t3, 4), "save_s": round(t5 - t4, 4),
                     "search_to_requests_done_s": round((t4 - t1) - (t1 - t0), 4), "mrr@100": mrr["MRR@100"]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
