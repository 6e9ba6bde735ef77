#!/usr/bin/env python3
"""BM25 measurement: queries/s of the device scorer (ccrec_amd.bm25) vs the reference-faithful host path
(scipy CSC column slice + dense divide + row sum + full sort per query, scripts/bm_25.py:31-52 and
scripts/ms_marco_eval.py:165-186) on a synthetic Zipf corpus.  Prints one JSON line.

  python tools/bench_bm25.py [--docs 500000] [--queries 2000] [--cpu-queries 20]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=500000)
    ap.add_argument("--queries", type=int, default=2000)
    ap.add_argument("--vocab", type=int, default=50000)
    ap.add_argument("--cpu-queries", type=int, default=20)
    ap.add_argument("--k", type=int, default=1001)
    args = ap.parse_args()
    from ccrec_amd.bm25 import BM25
    rs = np.random.RandomState(0)
    words = np.array([f"t{i}" for i in range(args.vocab)])
    p = 1.0 / np.arange(1, args.vocab + 1) ** 1.07
    p /= p.sum()
    lens = rs.randint(20, 80, args.docs)
    flat = rs.choice(args.vocab, int(lens.sum()), p=p)
    texts, o = [], 0
    for n in lens:
        texts.append(" ".join(words[flat[o:o + n]]))
        o += n
    qtexts = [" ".join(words[rs.choice(args.vocab, rs.randint(3, 12), p=p)]) for _ in range(args.queries)]
    t0 = time.perf_counter()
    model = BM25(0.75, 1.2).fit(texts)
    fit_s = time.perf_counter() - t0
    model.transform_topk(qtexts[:64], args.k)          # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s, i = model.transform_topk(qtexts, args.k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    terms = [model.query_terms(q) for q in qtexts]
    postings = int(sum(int(model.indptr[t + 1] - model.indptr[t]) for ts in terms for t in ts))
    # the document-tile scorer: per posting doc id 4 + tf 4 (the K_d gather and the accumulators stay on chip); per (query, doc) cell the
    # fp32 score written once 4 + read once by the collect pass 4.  (The round kernels' formula, 32 per posting + 40 per cell, is kept
    # beside it: rounds 1-3 and the first half of round 4 quote it.)
    search_s = model.last_search_seconds
    alg_bytes = postings * 8 + args.queries * args.docs * 8
    old_bytes = postings * 32 + args.queries * args.docs * 40
    out = {"metric": "BM25 queries/s (top-%d of %d documents)" % (args.k, args.docs), "value": round(args.queries / dt, 1),
           "unit": "queries/s", "seconds": round(dt, 4), "library_call_seconds": round(search_s, 4),
           "library_call_queries_per_s": round(args.queries / search_s, 1), "fit_seconds_host": round(fit_s, 1), "postings_touched": postings,
           "hbm_GBps_algorithmic": round(alg_bytes / search_s / 1e9, 1), "hbm_GBps_round_kernel_formula": round(old_bytes / search_s / 1e9, 1),
           "nnz": int(model.indptr[-1]), "vocab": len(model.vocabulary_), "scorer": os.environ.get("CCR_BM25_TILE", "0")}
    if args.cpu_queries:
        import scipy.sparse as sp
        rows = model._doc_ids.cpu().numpy()
        X = sp.csc_matrix((model._tf.cpu().numpy().astype(np.float64), rows, model.indptr), shape=(args.docs, len(model.vocabulary_)))
        doc_k = model._doc_k.cpu().numpy()
        t0 = time.perf_counter()
        rec = 0.0
        for qi in range(args.cpu_queries):
            t = terms[qi]
            Xq = X[:, t]
            denom = Xq + doc_k[:, None]
            numer = Xq.multiply(np.broadcast_to(model.idf[None, t], Xq.shape)) * (model.k1 + 1)
            sol = torch.Tensor(np.asarray((numer / denom).sum(1)).ravel())
            osc, order = sol.sort(descending=True)
            rec += len(set(order[:args.k].tolist()) & set(i[qi].tolist())) / args.k
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(args.cpu_queries / cdt, 2), "unit": "queries/s", "cores": torch.get_num_threads(),
                               "kind": "port", "sample": f"{args.cpu_queries} queries, scipy column slice + dense divide + full sort",
                               "recall_of_gpu_ids": round(rec / args.cpu_queries, 4)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
