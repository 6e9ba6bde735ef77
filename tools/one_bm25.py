#!/usr/bin/env python3
"""The bench line's BM25 workload (bench.bm25_workload: 500 k documents x 2 000 queries x top-1001) run a few times: the library
call's time, the path it took, and -- under rocprofv3 -- its kernels.

  python tools/one_bm25.py [--docs 500000] [--queries 2000] [--k 1001] [--reps 5] [--check]
--check: the same call with CCR_BM25_DENSE_SELECT=1 (every row stored and ranked exactly) must return the same ids and score bits."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=500000)
    ap.add_argument("--queries", type=int, default=2000)
    ap.add_argument("--vocab", type=int, default=50000)
    ap.add_argument("--k", type=int, default=1001)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    import bench
    model, qs, df, _ = bench.bm25_workload(args.docs, args.queries, args.vocab)
    model.transform_terms_topk(qs[:64], args.k)
    times = []
    for _ in range(args.reps):
        s, i = model.transform_terms_topk(qs, args.k)
        times.append(model.last_search_seconds * 1e3)
    print("library call ms:", " ".join("%.3f" % t for t in times), "| best %.3f ms = %.1f k queries/s |" % (min(times), args.queries / min(times)),
          model.last_stats(), "| workspace %.2f GB" % (torch.cuda.max_memory_allocated() / 1e9), flush=True)
    if args.check:
        os.environ["CCR_BM25_DENSE_SELECT"] = "1"
        s2, i2 = model.transform_terms_topk(qs, args.k)
        print("stored-rows path:", model.last_stats(), "%.3f ms" % (model.last_search_seconds * 1e3), "| ids equal", bool((i == i2).all()),
              "| score bits equal", bool((s.view(torch.int32) == s2.view(torch.int32)).all()), flush=True)


if __name__ == "__main__":
    main()
