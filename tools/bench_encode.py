#!/usr/bin/env python3
"""Encode-side measurement for SURVEY 8 f2: texts/s of a random-init BERT-base item tower under the reference's
padding policies vs the length-sorted encoder, on synthetic token-length distributions.

  fixed      : every text padded to CCREC_MAX_LENGTH (item_tower.py:27-33), fp32 pooled rows copied to the host
               and stacked (ms_marco_eval.py:141-149) -- the reference path
  batch_max  : corpus-order batches padded to their longest text (al_0_rank.py:73-84, padding=True), same host copy
  sorted     : ccrec_amd.encode.LengthSortedEncoder (token budget, pooled + packed bf16 rows scattered into the shard)

Usage: python tools/bench_encode.py [--texts 20000] [--dist titles|passages] [--layers 12]
Prints one JSON object per mode."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]


class IdTokenizer:
    """Texts are pre-tokenised id strings ("17 4 99 ..."): isolates batching/padding policy from tokenizer speed."""
    pad_token_id = 0

    def __call__(self, texts, truncation=True, padding=True, max_length=200, return_tensors="pt"):
        ids = [[101] + [int(w) for w in t.split()][: max_length - 2] + [102] for t in texts]
        if padding is False:
            return {"input_ids": ids, "attention_mask": [[1] * len(r) for r in ids]}
        L = max_length if padding == "max_length" else max(len(r) for r in ids)
        input_ids = torch.zeros(len(ids), L, dtype=torch.long)
        mask = torch.zeros(len(ids), L, dtype=torch.long)
        for r, row in enumerate(ids):
            input_ids[r, : len(row)] = torch.as_tensor(row)
            mask[r, : len(row)] = 1
        return {"input_ids": input_ids, "attention_mask": mask}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--texts", type=int, default=20000)
    ap.add_argument("--dist", default="titles", choices=["titles", "passages"])
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--max-length", type=int, default=200)
    ap.add_argument("--batch", type=int, default=512)
    args = ap.parse_args()
    from transformers import BertConfig, BertModel
    from ccrec_amd.item_tower import NaiveItemTower
    from ccrec_amd.encode import LengthSortedEncoder

    rs = np.random.RandomState(0)
    if args.dist == "titles":     # product titles (prime_pantry-like): short, long tail
        lens = np.clip(rs.lognormal(np.log(18), 0.5, args.texts).astype(int), 3, args.max_length - 2)
    else:                         # 100-word passages (NQ / MS MARCO-like)
        lens = np.clip(rs.normal(135, 30, args.texts).astype(int), 20, args.max_length - 2)
    texts = [" ".join(map(str, rs.randint(1000, 30000, n))) for n in lens]
    torch.manual_seed(0)
    cfg = BertConfig(num_hidden_layers=args.layers)   # BERT-base geometry, random weights
    tower = NaiveItemTower(BertModel(cfg).eval(), torch.nn.LayerNorm(768, elementwise_affine=False)).cuda()
    tok = IdTokenizer()

    def reference_style(padding):
        nonlocal texts
        out = []
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            for lo in range(0, len(texts), args.batch):
                toks = tok(texts[lo:lo + args.batch], padding=padding, max_length=args.max_length)
                emb = tower(**{k: v.cuda() for k, v in toks.items()}, output_step="mean_pooling")
                out.append(emb.float().cpu())
        return torch.vstack(out)

    def sorted_style():
        nonlocal texts
        enc = LengthSortedEncoder(tower, tok, max_length=args.max_length, max_tokens=args.batch * 128, max_batch=4 * args.batch)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            shard = enc.encode(texts, sim="dot")
        return shard, enc.stats

    results = {}
    all_texts = texts
    for name, fn in (("fixed", lambda: reference_style("max_length")), ("batch_max", lambda: reference_style(True)),
                     ("sorted", sorted_style)):
        texts = all_texts[:2048]      # untimed warm-up of this mode's GEMM shapes
        fn()
        texts = all_texts
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        results[name] = r
        line = {"mode": name, "dist": args.dist, "texts": args.texts, "mean_tokens": float(lens.mean() + 2),
                "max_length": args.max_length, "seconds": round(dt, 3), "texts_per_s": round(args.texts / dt, 1)}
        if name == "sorted":
            line.update(r[1])
        print(json.dumps(line), flush=True)
    ref = results["fixed"].cuda()
    got = results["sorted"][0].float()
    cos = torch.nn.functional.cosine_similarity(ref, got, dim=1)
    print(json.dumps({"check": "cosine(sorted bf16 rows, fixed-padding fp32 rows)", "min": float(cos.min()), "mean": float(cos.mean())}))


if __name__ == "__main__":
    main()
