#!/usr/bin/env python3
"""A/B of the BM25 scorers on one fitted index: CCR_BM25_TILE = -1 (round kernels + fp64 rows) against the document-tile
scorer's shapes; every shape must return the round kernels' ids and score bits.

  python tools/exp_bm25_tile.py [--docs 500000] [--queries 2000] [--cfgs=-1,0,1,2]"""
import argparse
import ctypes
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
    ap.add_argument("--k", type=int, default=1001)
    ap.add_argument("--cfgs", default="-1,0,1,2")
    args = ap.parse_args()
    from ccrec_amd import _lib
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
    model = BM25(0.75, 1.2).fit(texts)
    ref = None
    for cfg in [int(c) for c in args.cfgs.split(",")]:
        os.environ["CCR_BM25_TILE"] = str(cfg)
        model._lib.ccr_bm25_index_destroy(model._h)
        model._h = ctypes.c_void_p()
        _lib.check(model._lib.ccr_bm25_index_create(model.indptr.ctypes.data_as(ctypes.c_void_p), model._doc_ids.data_ptr(),
                                                    model._tf.data_ptr(), model._doc_k.data_ptr(), len(model.vocabulary_), model.n_docs,
                                                    model.k1, ctypes.byref(model._h)), "ccr_bm25_index_create")
        model.transform_topk(qtexts, args.k)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            s, i = model.transform_topk(qtexts, args.k)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        if ref is None:
            ref = (s.clone(), i.clone())
            same = "reference"
        else:
            same = "ids %s, score bits %s" % (bool((i == ref[1]).all()), bool((s.view(torch.int32) == ref[0].view(torch.int32)).all()))
        print("CCR_BM25_TILE=%d: %.2f ms = %.1f k queries/s (%s)" % (cfg, best * 1e3, args.queries / best / 1e3, same), flush=True)


if __name__ == "__main__":
    main()
