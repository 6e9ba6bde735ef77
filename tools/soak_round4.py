#!/usr/bin/env python3
"""Soak of what round 4 added, every case against an independent exact path, bit for bit:
  (a) the streaming main pass of small batches (n_q <= 64) against the exact dense path -- random shapes, clustered corpora,
      duplicate rows (mass ties), norm outliers, k from 1 to 2 500;
  (b) the BM25 document-tile scorer + sampled selection against the round kernels + exact dense selection -- random postings;
  (c) the short-list shard merge against the merge of full lists -- random skew and ties between shards.
  python tools/soak_round4.py [cases]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crowd-coachable-recommendations_amd")]
from ccrec_amd import ops  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = "cuda"


def soak_streaming(c, nq_lo=1, nq_hi=65):
    rs = np.random.RandomState(4000 + c + 1000 * nq_lo)
    n = int(rs.randint(20_000, 1_500_000))
    d = int(rs.choice([64, 128, 256, 384, 768, 1024]))
    nq = int(rs.randint(nq_lo, nq_hi))
    k = int(rs.choice([1, 10, 100, 500, 1001, 2500]))
    k = min(k, n)
    ncl = int(rs.choice([1, 16, 512]))
    g = torch.Generator(device=dev).manual_seed(c)
    D = torch.randn(n, d, generator=g, device=dev) * d ** -0.5
    if ncl > 1:
        centres = torch.randn(ncl, d, generator=g, device=dev) * d ** -0.5
        cid = torch.randint(0, ncl, (n,), generator=g, device=dev) if rs.rand() < 0.5 else (torch.arange(n, device=dev) * ncl // n)
        D = 0.8 * centres[cid] + 0.6 * D
    D *= torch.exp(0.3 * torch.randn(n, 1, generator=g, device=dev))
    if rs.rand() < 0.4:                                   # duplicate rows: mass ties
        m = int(rs.randint(10, 3000))
        src = int(rs.randint(0, n))
        D[torch.randint(0, n, (m,), generator=g, device=dev)] = D[src].clone()
    if rs.rand() < 0.3:
        D[int(rs.randint(0, n))] *= 40.0                  # a norm outlier
    Q = torch.randn(nq, d, generator=g, device=dev) * d ** -0.5
    if rs.rand() < 0.3:
        Q[0] = D[int(rs.randint(0, n))]
    Db, Qb = ops.pack_bf16(D), ops.pack_bf16(Q)
    del D, Q
    index = ops.CorpusIndex(Db)
    s, i = index.search(Qb, k, 2)
    st = index.last_stats()
    s1, i1 = index.search(Qb, k, 1)
    ok = torch.equal(i, i1) and torch.equal(s.view(torch.int32), s1.view(torch.int32))
    print(f"streaming {c}: n={n} d={d} nq={nq} k={k} clusters={ncl} path={st['path']} ranges={st['ranges']} sublists={st['sublists']} "
          f"flagged={st['n_fallback']} retried={st['n_retried']} dense={st['n_dense']} {'OK' if ok else 'MISMATCH'}", flush=True)
    assert ok
    return st["ranges"] == 1 and st["path"] == 1


def soak_bm25(c):
    from ccrec_amd.bm25 import BM25
    rs = np.random.RandomState(5000 + c)
    n_docs = int(rs.choice([900, 5000, 70_001, 262_144, 400_000]))
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
        r = np.sort(rs.choice(n_docs, df, replace=False)) if df else np.zeros(0, np.int64)
        rows.append(r)
        counts.append(rs.randint(1, 9, df))
        indptr.append(indptr[-1] + df)
    idf = np.log(n_docs / np.maximum(np.diff(indptr), 1).astype(np.float64))
    doc_k = 1.2 * (0.25 + 0.75 * rs.uniform(0.2, 3.0, n_docs))
    indptr = np.asarray(indptr, np.int64)
    rows, counts = np.concatenate(rows).astype(np.int32), np.concatenate(counts).astype(np.float32)
    nq = int(rs.choice([1, 7, 300]))
    queries = [np.sort(rs.choice(n_terms, rs.randint(0, min(n_terms, 20)) , replace=False)).astype(np.int32) for _ in range(nq)]
    k = min(int(rs.choice([1, 100, 1001])), n_docs)
    out = {}
    for cfg, dense in (("-1", "1"), (str(int(rs.choice([0, 1, 2]))), None)):
        os.environ["CCR_BM25_TILE"] = cfg
        if dense:
            os.environ["CCR_BM25_DENSE_SELECT"] = dense
        else:
            os.environ.pop("CCR_BM25_DENSE_SELECT", None)
        m = BM25.from_postings(indptr, rows, counts, doc_k, idf, k1=1.2)
        s, i = m.transform_terms_topk(queries, k)
        out[cfg] = (s.view(torch.int32).clone(), i.clone(), cfg)
    (a, b) = out.values()
    ok = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    print(f"bm25 {c}: docs={n_docs} terms={n_terms} dense={dense_terms} nnz={len(rows)} nq={nq} k={k} shape={b[2]} {'OK' if ok else 'MISMATCH'}", flush=True)
    assert ok


def soak_short_lists(c):
    rs = np.random.RandomState(6000 + c)
    R = int(rs.choice([2, 3, 4, 8]))
    nq = int(rs.randint(1, 200))
    k = int(rs.choice([10, 100, 1001]))
    per = int(rs.randint(k, 3 * k + 50))                       # rows per shard that matter
    skew = rs.rand() < 0.4
    # true per-shard sorted lists (scores with ties), full length = min(k, per)
    full = min(k, per)
    from ccrec_amd.dist import short_list_length
    kl = min(full, short_list_length(k, R))
    sc = np.round(rs.randn(R, nq, per).astype(np.float32) * (4 if rs.rand() < 0.5 else 64)) / (4 if rs.rand() < 0.5 else 1)
    if skew:
        sc[0] += 3.0                                           # one shard holds most of the top
    ids = np.tile(np.arange(per, dtype=np.int64)[None, None], (R, nq, 1)) + (np.arange(R, dtype=np.int64) * per)[:, None, None]
    order = np.lexsort((ids, -sc.astype(np.float64)), axis=-1)
    sc_s, id_s = np.take_along_axis(sc, order, -1), np.take_along_axis(ids, order, -1)
    # exact answer: merge of the full lists
    cat_s, cat_i = sc_s[:, :, :full].transpose(1, 0, 2).reshape(nq, -1), id_s[:, :, :full].transpose(1, 0, 2).reshape(nq, -1)
    o = np.lexsort((cat_i, -cat_s.astype(np.float64)), axis=-1)[:, :min(k, R * full)]
    ref_s, ref_i = np.take_along_axis(cat_s, o, -1), np.take_along_axis(cat_i, o, -1)
    # device: messages with short lists
    msg_bytes = ops.shard_message_bytes(nq, kl)
    gathered = torch.zeros(R, msg_bytes, dtype=torch.uint8, device=dev)
    for r in range(R):
        ops.shard_message_fill(gathered[r], nq, kl, torch.from_numpy(sc_s[r, :, :kl].copy()).to(dev), torch.from_numpy(id_s[r, :, :kl].copy()).to(dev),
                               row_offset=r * per, n_rows=per)
    k_out = ref_s.shape[1]
    s, i, flags, count = ops.merge_short_lists(gathered, R, nq, kl, k_out)
    flagged = set(np.flatnonzero(flags.cpu().numpy()).tolist())
    assert len(flagged) == int(count)
    good = [q for q in range(nq) if q not in flagged]
    ok = np.array_equal(s.cpu().numpy()[good].view(np.uint32), ref_s[good].view(np.uint32)) and np.array_equal(i.cpu().numpy()[good], ref_i[good])
    # a flagged query must be one whose exact answer needs an entry beyond some rank's short list
    for q in flagged:
        need = [(ref_i[q] // per == r).sum() for r in range(R)]
        assert max(need) >= kl or kl == full, (q, need, kl)
    print(f"short lists {c}: R={R} nq={nq} k={k} k_list={kl} per={per} skew={skew} flagged={len(flagged)} {'OK' if ok else 'MISMATCH'}", flush=True)
    assert ok


if __name__ == "__main__":
    narrow = 0
    for c in range(cases):
        narrow += int(soak_streaming(c))
        soak_bm25(c)
        soak_short_lists(c)
    print(f"all {cases} cases of each kind equal their exact paths; streaming kernel used in {narrow} of {cases}")
