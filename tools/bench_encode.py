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


def fast_tokenizer(vocab_words=30000):
    """A Rust (HF `tokenizers`) word-level tokenizer over a synthetic vocabulary, built in memory (no hub access): the
    host cost of a real fast tokenizer -- its batch encoder runs in Rust threads and releases the GIL."""
    from tokenizers import Tokenizer
    from tokenizers.models import WordLevel
    from tokenizers.pre_tokenizers import Whitespace
    from tokenizers.processors import TemplateProcessing
    from transformers import PreTrainedTokenizerFast
    vocab = {"[PAD]": 0, "[UNK]": 1, "[CLS]": 2, "[SEP]": 3}
    vocab.update({f"w{i}": 4 + i for i in range(vocab_words)})
    tk = Tokenizer(WordLevel(vocab, unk_token="[UNK]"))
    tk.pre_tokenizer = Whitespace()
    tk.post_processor = TemplateProcessing(single="[CLS] $A [SEP]", special_tokens=[("[CLS]", 2), ("[SEP]", 3)])
    return PreTrainedTokenizerFast(tokenizer_object=tk, pad_token="[PAD]", unk_token="[UNK]", cls_token="[CLS]", sep_token="[SEP]")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokenizer", default="ids", choices=["ids", "fast"],
                    help="ids: pre-tokenised id strings (isolates the batching policy); fast: a Rust word-level tokenizer (real host cost)")
    ap.add_argument("--modes", default="fixed,batch_max,sorted",
                    help="comma list of fixed, batch_max, sorted (chunked pipeline), serial (sorted, but tokenise everything first)")
    ap.add_argument("--chunk-texts", type=int, default=65536)
    ap.add_argument("--host-threads", type=int, default=4)
    ap.add_argument("--host-processes", type=int, default=0, help="tokenizer worker processes (0: host threads only)")
    ap.add_argument("--sweep", default="", help='extra "sorted" runs over the same texts: comma list of chunk_texts:host_threads, e.g. "32768:4,32768:8"')
    ap.add_argument("--texts", type=int, default=20000)
    ap.add_argument("--dist", default="titles", choices=["titles", "passages"])
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--max-length", type=int, default=200)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--autocast", default="bf16", choices=["bf16", "fp16"],
                    help="the autocast type around the encode: fp16 is the reference's torch.cuda.amp.autocast() (scripts/al_0_rank.py:8,125)")
    ap.add_argument("--fused", default="auto", choices=["auto", "off"],
                    help="sorted modes: encoder layers on the library's attention / add + LayerNorm kernels (auto) or as torch modules (off); "
                         "the extra mode name sorted_modules always runs the torch modules")
    args = ap.parse_args()
    ac_dtype = torch.float16 if args.autocast == "fp16" else torch.bfloat16
    from transformers import BertConfig, BertModel
    from ccrec_amd.item_tower import NaiveItemTower
    from ccrec_amd.encode import LengthSortedEncoder

    rs = np.random.RandomState(0)
    if args.dist == "titles":     # product titles (prime_pantry-like): short, long tail
        lens = np.clip(rs.lognormal(np.log(18), 0.5, args.texts).astype(int), 3, args.max_length - 2)
    else:                         # 100-word passages (NQ / MS MARCO-like)
        lens = np.clip(rs.normal(135, 30, args.texts).astype(int), 20, args.max_length - 2)
    if args.tokenizer == "fast":
        words = np.array([f"w{i}" for i in range(30000)], dtype=object)
        texts = [" ".join(words[rs.randint(0, 30000, n)]) for n in lens]
        tok = fast_tokenizer()
    else:
        texts = [" ".join(map(str, rs.randint(1000, 30000, n))) for n in lens]
        tok = IdTokenizer()
    torch.manual_seed(0)
    cfg = BertConfig(num_hidden_layers=args.layers, vocab_size=30522)   # BERT-base geometry, random weights
    tower = NaiveItemTower(BertModel(cfg).eval(), torch.nn.LayerNorm(768, elementwise_affine=False)).cuda()

    def reference_style(padding):
        nonlocal texts
        out = []
        with torch.no_grad(), torch.autocast("cuda", dtype=ac_dtype):
            for lo in range(0, len(texts), args.batch):
                toks = tok(texts[lo:lo + args.batch], truncation=True, padding=padding, max_length=args.max_length, return_tensors="pt")
                emb = tower(**{k: v.cuda() for k, v in toks.items()}, output_step="mean_pooling")
                out.append(emb.float().cpu())
        return torch.vstack(out)

    def sorted_style(chunk=None, threads=None, procs=None, fused=None):
        nonlocal texts
        enc = LengthSortedEncoder(tower, tok, max_length=args.max_length, max_tokens=args.batch * 128, max_batch=4 * args.batch,
                                  chunk_texts=chunk or args.chunk_texts, host_threads=threads or args.host_threads,
                                  host_processes=args.host_processes if procs is None else procs,
                                  fused=(args.fused == "auto") if fused is None else fused)
        with torch.autocast("cuda", dtype=ac_dtype):
            shard = enc.encode(texts, sim="dot")
        enc.close()
        st = dict(enc.stats)
        st["gpu_idle_frac"] = round(1.0 - st["gpu_busy_s"] / max(st["wall_s"], 1e-9), 4)
        return shard, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}

    results = {}
    all_texts = texts
    table = {"fixed": lambda: reference_style("max_length"), "batch_max": lambda: reference_style(True), "sorted": sorted_style,
             "serial": lambda: sorted_style(chunk=10 ** 9), "sorted_modules": lambda: sorted_style(fused=False)}
    names = args.modes.split(",")
    for spec in [x for x in args.sweep.split(",") if x]:
        c, t, pr = (list(int(v) for v in spec.split(":")) + [None])[:3]    # chunk_texts:host_threads[:host_processes]
        table[f"sorted[{spec}]"] = (lambda c=c, t=t, pr=pr: sorted_style(chunk=c, threads=t, procs=pr))
        names.append(f"sorted[{spec}]")
    for name in names:
        fn = table[name]
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
        if name.startswith("sorted") or name == "serial":   # (sorted, sorted_modules, sorted[...])
            line.update(r[1])
            line["tokenizer"] = args.tokenizer
        print(json.dumps(line), flush=True)
        if name not in ("fixed", "sorted", "sorted_modules"):
            results[name] = None     # keep only what the final checks need
    if results.get("fixed") is not None and results.get("sorted") is not None:
        ref = results["fixed"].cuda()
        got = results["sorted"][0].float()
        cos = torch.nn.functional.cosine_similarity(ref, got, dim=1)
        print(json.dumps({"check": "cosine(sorted bf16 rows, fixed-padding fp32 rows)", "min": float(cos.min()), "mean": float(cos.mean())}))
    if results.get("sorted") is not None and results.get("sorted_modules") is not None:
        cos = torch.nn.functional.cosine_similarity(results["sorted_modules"][0].float(), results["sorted"][0].float(), dim=1)
        print(json.dumps({"check": "cosine(rows through the library's layer kernels, rows through the torch modules)",
                          "min": float(cos.min()), "mean": float(cos.mean())}))


if __name__ == "__main__":
    main()
