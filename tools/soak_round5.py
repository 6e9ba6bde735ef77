#!/usr/bin/env python3
"""Soak of what round 5 added or rewrote, every case against an independent path:
  (a) BM25: the tile scorer with the FUSED top-k filter (sampled tau, candidate records, redo of the rows it cannot finish, a few
      rows at a time) and the contribution table -- on or off, or bypassed by query weights that are not the index's idf -- against the
      round kernels with every fp32 row stored and the exact dense selection: ids and score bits;
  (b) the in-batch loss (one-launch pack, fragment-major forward where the shape allows, prep + three-part GEMM backward) against an
      fp64 torch formulation on the same bf16-rounded operands: loss 2e-5, gradients 2e-4 (the tolerances of tests/test_gpu_inbatch.py),
      and bit-identical on a second run;
  (c) the streaming main pass of small batches (its thresholds now come from threshold_small_kernel) against the exact dense path
      (tools/soak_round4.py's cases), and the same for batches of 65 .. 128 queries (six resident query tiles / two query groups).
  python tools/soak_round5.py [cases]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd"), os.path.join(ROOT, "tools")]
from ccrec_amd import ops  # noqa: E402
import soak_round4 as r4  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = "cuda"


def soak_bm25(c):
    from ccrec_amd.bm25 import BM25
    rs = np.random.RandomState(7000 + c)
    n_docs = int(rs.choice([5000, 40_000, 70_001, 262_144, 500_000]))
    n_terms = int(rs.choice([50, 400, 3000]))
    dense_terms = int(rs.randint(0, 8))
    indptr, rows, counts = [0], [], []
    for t in range(n_terms):
        if t < dense_terms:
            df = int(n_docs * rs.uniform(0.2, 1.0))
        elif rs.rand() < 0.03:
            df = 0
        else:
            df = max(1, int(n_docs * rs.uniform(0.05, 0.3) / (t - dense_terms + 1) ** rs.uniform(0.8, 1.3)))
        if df and rs.rand() < 0.15:                         # a term that lives in ONE stretch of the corpus (topical order): the sample misses or overrates it
            lo = int(rs.randint(0, max(1, n_docs - df)))
            r = np.arange(lo, lo + df)
        else:
            r = np.sort(rs.choice(n_docs, df, replace=False)) if df else np.zeros(0, np.int64)
        rows.append(r)
        counts.append(rs.randint(1, 9, df))
        indptr.append(indptr[-1] + df)
    idf = np.log(n_docs / np.maximum(np.diff(indptr), 1).astype(np.float64))
    if rs.rand() < 0.2:
        idf[rs.randint(0, n_terms, 3)] *= -1.0              # negative weights: rows with negative scores go through the redo path
    doc_k = 1.2 * (0.25 + 0.75 * rs.uniform(0.2, 3.0, n_docs))
    indptr = np.asarray(indptr, np.int64)
    rows, counts = np.concatenate(rows).astype(np.int32), np.concatenate(counts).astype(np.float32)
    nq = int(rs.choice([1, 7, 300]))
    queries = [np.sort(rs.choice(n_terms, rs.randint(0, min(n_terms, 20)), replace=False)).astype(np.int32) for _ in range(nq)]
    k = min(int(rs.choice([1, 100, 1001, 4000])), n_docs)
    table = str(int(rs.rand() < 0.7))
    redo = str(int(rs.choice([0, 2, 5])))
    other_idf = rs.rand() < 0.2
    out, stats = {}, None
    for cfg, dense in (("-1", "1"), (str(int(rs.choice([0, 1, 2, 3, 4]))), None)):
        os.environ["CCR_BM25_TILE"] = cfg
        os.environ["CCR_BM25_TABLE"] = table
        os.environ["CCR_BM25_REDO_ROWS"] = redo
        if dense:
            os.environ["CCR_BM25_DENSE_SELECT"] = dense
        else:
            os.environ.pop("CCR_BM25_DENSE_SELECT", None)
        m = BM25.from_postings(indptr, rows, counts, doc_k, idf, k1=1.2)
        if other_idf:
            m.idf = idf * 0.75                              # the queries do not use the table's idf: the generic scorer
        s, i = m.transform_terms_topk(queries, k)
        out[cfg] = (s.view(torch.int32).clone(), i.clone(), cfg)
        stats = m.last_stats()
    (a, b) = out.values()
    ok = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    print(f"bm25 {c}: docs={n_docs} terms={n_terms} dense={dense_terms} nnz={len(rows)} nq={nq} k={k} shape={b[2]} table={table} other_idf={other_idf} "
          f"redo_rows={redo} {stats} {'OK' if ok else 'MISMATCH'}", flush=True)
    assert ok
    return stats["rows_redone"]


def soak_inbatch(c):
    rs = np.random.RandomState(8000 + c)
    B = int(rs.choice([1, 7, 32, 33, 64, 100, 256, 1000, 1024, 1200]))
    d = int(rs.choice([16, 48, 64, 128, 256, 768, 1024]))
    inv_t = float(rs.choice([1.0, 20.0, 50.0]))
    g = torch.Generator(device=dev).manual_seed(c)
    scale = d ** -0.5 * float(rs.choice([0.5, 1.0, 2.0]))
    q, p, n = (torch.randn(B, d, generator=g, device=dev) * scale for _ in range(3))
    if rs.rand() < 0.3:
        p = (q + 0.1 * p).contiguous()                      # positives near their queries: peaked softmax rows
    go = float(rs.choice([1.0, 3.0, 0.01]))
    res = []
    for _ in range(2):
        a, b, cc = (t.clone().requires_grad_(True) for t in (q, p, n))
        loss = ops.inbatch_ce(a, b, cc, inv_t)
        (loss * go).backward()
        res.append((loss.detach().clone(), torch.stack([a.grad, b.grad, cc.grad])))
    same = torch.equal(res[0][0].view(torch.int32), res[1][0].view(torch.int32)) and torch.equal(res[0][1].view(torch.int32), res[1][1].view(torch.int32))
    a, b, cc = (t.to(torch.bfloat16).double().requires_grad_(True) for t in (q, p, n))
    ref = torch.nn.functional.cross_entropy(torch.cat([a @ b.T, a @ cc.T], 1) * inv_t, torch.arange(B, device=dev))
    (ref * go).backward()
    rg = torch.stack([a.grad, b.grad, cc.grad])
    el = abs(float(res[0][0]) - float(ref)) / max(1.0, abs(float(ref)))
    # gradients: 2e-4 of the largest entry -- or, for softmax rows so peaked that the whole gradient is a cancellation residue (p_ii = 1 - 1e-9:
    # any fp32 evaluation of exp(s - lse) - 1 is noise there, torch's own included), the absolute noise floor of 2B fp32 exponentials
    diff = float((res[0][1].double() - rg).abs().max())
    xmax = float(max(q.abs().max(), p.abs().max(), n.abs().max()))
    eg = diff / max(1e-30, float(rg.abs().max()))
    ok = same and el < 2e-5 and (eg < 2e-4 or diff < 1e-6 * go * inv_t * xmax)
    print(f"inbatch {c}: B={B} d={d} inv_t={inv_t} grad_out={go} loss err {el:.2e} grad err {eg:.2e} (abs {diff:.1e}) deterministic={same} {'OK' if ok else 'MISMATCH'}", flush=True)
    assert ok


redone = narrow = narrow_wide = 0
for c in range(cases):
    redone += soak_bm25(c)
    soak_inbatch(c)
    narrow += int(r4.soak_streaming(c))
    narrow_wide += int(r4.soak_streaming(c, 65, 129))        # six resident query tiles (<= 96 queries) / two query groups (<= 128)
for v in ("CCR_BM25_TILE", "CCR_BM25_TABLE", "CCR_BM25_REDO_ROWS", "CCR_BM25_DENSE_SELECT"):
    os.environ.pop(v, None)
print(f"all {cases} cases of each kind agree with their independent paths; BM25 rows through the redo path: {redone}; streaming kernel used in {narrow} of {cases} small and {narrow_wide} of {cases} 65 .. 128-query batches")
