#!/usr/bin/env python3
"""One whole active-learning rank step (scripts/al_0_rank.py:107-218) at scale, end to end on one MI355X:
corpus + queries as TEXT -> tokenise (Rust word-level tokenizer, worker processes) -> random-init BERT-base (bf16 autocast,
length-sorted batches, fused pool + pack into the shard) -> fused search (top-1001) -> MRR on device -> request files.
`ranking_profile_bm25` is an input of the reference's step (al_commons.py:55-58, loaded from disk): a synthetic one is passed in.

  python tools/bench_al_step.py [--passages 1000000] [--queries 3452] [--step-queries 300]
Prints one JSON line with the stage timings.  Reference point: the corpus encode alone takes 1 190-1 205 s for the
2.68 M NQ passages on 4 x A10G, i.e. ~445 s per million passages."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd"), os.path.join(ROOT, "tools")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--passages", type=int, default=1_000_000)
    ap.add_argument("--queries", type=int, default=3_452)
    ap.add_argument("--step-queries", type=int, default=300)
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--bm25", action="store_true", help="compute the BM25 leg inside the step (ranking_bm25 beside the encode) instead of passing it in")
    args = ap.parse_args()
    from transformers import BertConfig, BertModel
    from bench_encode import fast_tokenizer
    from ccrec_amd.al_step import run_rank_step
    from ccrec_amd.item_tower import NaiveItemTower
    os.environ["CCREC_SIM_TYPE"] = "dot"
    os.environ.setdefault("CCREC_DISPLAY_LENGTH", "250")
    rs = np.random.RandomState(0)
    words = np.array([f"w{i}" for i in range(30000)], dtype=object)
    t0 = time.perf_counter()
    plen = np.clip(rs.normal(135, 30, args.passages).astype(int), 20, 198)
    corpus = {f"doc{j}": " ".join(words[rs.randint(0, 30000, n)]) for j, n in enumerate(plen)}
    qlen = np.clip(rs.normal(12, 4, args.queries).astype(int), 3, 40)
    queries = {f"q{i}": " ".join(words[rs.randint(0, 30000, n)]) for i, n in enumerate(qlen)}
    qrels = {q: {f"doc{int(rs.randint(0, args.passages))}": 1} for q in queries}
    step_qids = list(queries)[:args.step_queries]
    bm25 = {q: {f"doc{int(j)}": 1.0 for j in rs.randint(0, args.passages, 5)} for q in queries}
    gen_s = time.perf_counter() - t0
    torch.manual_seed(0)
    tower = NaiveItemTower(BertModel(BertConfig(num_hidden_layers=args.layers, vocab_size=30522)).eval(),
                           torch.nn.LayerNorm(768, elementwise_affine=False)).cuda()
    tok = fast_tokenizer()
    kw = {"max_length": 200, "max_tokens": 65536, "max_batch": 2048, "chunk_texts": 32768, "host_threads": 4, "host_processes": 4}
    warm = {k: corpus[k] for k in list(corpus)[:4096]}                 # GEMM shapes, library load, worker start-up: untimed
    warm_bm25 = {q: {k: 1.0 for k in list(warm)[:5]} for q in queries}
    run_rank_step(tower, tok, warm, dict(list(queries.items())[:64]), qrels, step_qids[:8], 0, tempfile.mkdtemp(), ranking_profile_bm25=warm_bm25,
                  encoder_kw=kw)
    out_dir = tempfile.mkdtemp()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run_rank_step(tower, tok, corpus, queries, qrels, step_qids, 0, out_dir, ranking_profile_bm25=None if args.bm25 else bm25,
                        encoder_kw=kw)
    wall = time.perf_counter() - t0
    tm = res["timings"]
    enc = tm.pop("corpus_encoder")
    line = {"passages": args.passages, "queries": args.queries, "keep": 1001, "step_queries": len(step_qids), "wall_s": round(wall, 2),
            "timings_s": {k: round(v, 3) for k, v in tm.items()},
            "corpus_encode": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in enc.items()},
            "bm25_inside_the_step": bool(args.bm25), "passages_per_s_end_to_end": round(args.passages / wall, 1), "text_generation_s": round(gen_s, 1),
            "files": sorted(os.listdir(os.path.join(out_dir, "data_iteration_0")))}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
