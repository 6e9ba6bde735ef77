"""
oracle.py -- numpy/ctypes face of the CPU restatement (oracle/ccr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.

Parity status: PINNED -- every function here is checked against golden vectors
produced by the reference's own Python (tools/make_golden.py, tests/golden/,
tests/test_oracle_golden.py).

Two families:
  * canonical_*  : the bit-exact definition the HIP path must reproduce
                   (bf16 inputs, fp64-ordered accumulate, (score desc, idx asc)).
  * reference_*  : a faithful restatement of what the reference does on CPU
                   (fp32 chunked matmul into a host [Q,N] matrix, per-row full
                   descending sort, keep 1001) -- used for the golden check and
                   timed as the CPU baseline ("port").
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libccr_oracle.so")
_SRC = os.path.join(_HERE, "ccr_oracle.c")


def build(force=False):
    """gcc -O2 -fopenmp the C restatement into oracle/_build/ (seconds)."""
    if not force and os.path.isfile(_SO) and os.path.getmtime(_SO) >= os.path.getmtime(_SRC):
        return _SO
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    cmd = ["gcc", "-O2", "-fopenmp", "-fPIC", "-shared", "-ffp-contract=off", "-o", _SO, _SRC, "-lm"]
    subprocess.check_call(cmd)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


_i64 = ctypes.c_int64
_int = ctypes.c_int


# ----------------------------------------------------------------------------- pack
def pack_bf16(x):
    """fp32 -> bf16 bits (uint16), round-to-nearest-even (== torch .to(bfloat16))."""
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(x.shape, np.uint16)
    lib().orc_pack_bf16(_p(x), _p(out), _i64(x.size))
    return out


def unpack_bf16(b):
    b = np.ascontiguousarray(b, np.uint16)
    out = np.empty(b.shape, np.float32)
    lib().orc_unpack_bf16(_p(b), _p(out), _i64(b.size))
    return out


def normalize_pack_bf16(x):
    """rows x / max(||x||, 1e-12) -> bf16 bits, canonical reduction order (see .c)."""
    x = np.ascontiguousarray(x, np.float32)
    assert x.ndim == 2   # any width: a partial last chunk of the canonical sum counts as zero-padded
    out = np.empty(x.shape, np.uint16)
    lib().orc_normalize_pack_bf16(_p(x), _p(out), _i64(x.shape[0]), _int(x.shape[1]))
    return out


def row_norms(x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(x.shape[0], np.float32)
    lib().orc_row_norms(_p(x), _p(out), _i64(x.shape[0]), _int(x.shape[1]))
    return out


def row_norms_bf16(b):
    b = np.ascontiguousarray(b, np.uint16)
    out = np.empty(b.shape[0], np.float32)
    lib().orc_row_norms_bf16(_p(b), _p(out), _i64(b.shape[0]), _int(b.shape[1]))
    return out


def pack(x, sim="dot"):
    """What the product's index/query pack does: dot -> RNE pack, cos -> normalise + pack."""
    return normalize_pack_bf16(x) if sim == "cos" else pack_bf16(x)


# ----------------------------------------------------------------------------- canonical scoring / ranking
def canonical_scores(Qb, Db):
    """[nq,nd] fp32 canonical scores of bf16-bit matrices."""
    Qb = np.ascontiguousarray(Qb, np.uint16)
    Db = np.ascontiguousarray(Db, np.uint16)
    assert Qb.shape[1] == Db.shape[1]
    out = np.empty((Qb.shape[0], Db.shape[0]), np.float32)
    lib().orc_scores(_p(Qb), _p(Db), _i64(Qb.shape[0]), _i64(Db.shape[0]), _int(Qb.shape[1]), _p(out))
    return out


def canonical_scores_pairs(Qb, Db, ids):
    """scores[i,c] = canonical score(Q[i], D[ids[i,c]])."""
    Qb = np.ascontiguousarray(Qb, np.uint16)
    Db = np.ascontiguousarray(Db, np.uint16)
    ids = np.ascontiguousarray(ids, np.int64)
    out = np.empty(ids.shape, np.float32)
    lib().orc_scores_pairs(_p(Qb), _p(Db), _p(ids), _i64(ids.shape[0]), _i64(ids.shape[1]), _int(Qb.shape[1]), _p(out))
    return out


def _csr(block, nq):
    if block is None:
        return None, None
    ptr = np.zeros(nq + 1, np.int64)
    ptr[1:] = np.cumsum([len(b) for b in block])
    idx = np.concatenate([np.asarray(b, np.int64) for b in block]) if ptr[-1] else np.zeros(0, np.int64)
    return ptr, np.ascontiguousarray(idx)


def rank(scores, k, block=None):
    """Order rows of a score matrix by (score desc, idx asc); blocked columns -> -1e6 first.

    block: list (len nq) of lists of blocked corpus indices, or (ptr, idx) CSR.
    Raises AssertionError("block id not found") like ms_marco_eval.py:226.
    """
    scores = np.array(scores, np.float32, order="C", copy=True)
    nq, nd = scores.shape
    if isinstance(block, tuple):
        ptr, idx = np.ascontiguousarray(block[0], np.int64), np.ascontiguousarray(block[1], np.int64)
    else:
        ptr, idx = _csr(block, nq)
    ids = np.empty((nq, k), np.int64)
    sc = np.empty((nq, k), np.float32)
    rc = lib().orc_rank(_p(scores), _i64(nq), _i64(nd), _i64(k), _p(ptr) if ptr is not None else None,
                        _p(idx) if idx is not None else None, _p(ids), _p(sc))
    assert rc != -2, "block id not found"
    assert rc == 0, f"orc_rank failed {rc}"
    return ids, sc


def canonical_search(Qb, Db, k, block=None):
    """Exhaustive canonical top-k: (ids[nq,k] int64, scores[nq,k] fp32)."""
    return rank(canonical_scores(Qb, Db), k, block)


def canonical_ranking(Eq, Ed, sim="dot", block=None):
    """Canonical counterpart of ranking(): pack -> score -> rank, keep min(1001, N)."""
    Qb, Db = pack(Eq, sim), pack(Ed, sim)
    return canonical_search(Qb, Db, min(1001, Db.shape[0]), block)


def merge_topk(scores, ids):
    """[R,nq,k] per-shard lists -> global top-k with the canonical order."""
    scores = np.ascontiguousarray(scores, np.float32)
    ids = np.ascontiguousarray(ids, np.int64)
    R, nq, k = scores.shape
    os_ = np.empty((nq, k), np.float32)
    oi = np.empty((nq, k), np.int64)
    lib().orc_merge_topk(_p(scores), _p(ids), _int(R), _i64(nq), _i64(k), _p(os_), _p(oi))
    return os_, oi


def merge_short_lists(scores, ids, truncated, k_out):
    """Test-side restatement of the short-list merge (csrc/ccr_merge.hip: merge_short_lists_kernel): [R, nq, kl] per-shard lists, each a
    shard's exact canonical top-kl (padding slots: -inf with ids above 2^62) -> the k_out best of the R kl entries per query in the
    canonical order (score desc, id asc), and flags[q] = 1 where every REAL entry of some list r with truncated[r] (its shard holds rows
    it did not send) is among the kept k_out -- then an unsent row of that shard may belong there too and the query must be repeated
    with full lists; otherwise the kept k_out are the global top-k_out.  -> (scores [nq, k_out], ids [nq, k_out], flags [nq] int32)."""
    scores = np.ascontiguousarray(scores, np.float32)
    ids = np.ascontiguousarray(ids, np.int64)
    R, nq, kl = scores.shape
    assert R * kl >= k_out
    os_ = np.empty((nq, k_out), np.float32)
    oi = np.empty((nq, k_out), np.int64)
    flags = np.zeros(nq, np.int32)
    src = np.repeat(np.arange(R), kl)
    for q in range(nq):
        s, i = scores[:, q].reshape(-1), ids[:, q].reshape(-1)
        order = np.lexsort((i, -s.astype(np.float64)))[:k_out]
        os_[q], oi[q] = s[order], i[order]
        kept = np.bincount(src[order], minlength=R)
        real = (ids[:, q] < (1 << 62)).sum(axis=1)
        flags[q] = int(any(truncated[r] and kept[r] >= real[r] for r in range(R)))
    return os_, oi, flags


def sparse_prior_search(Qb, Db, indptr, indices, data, k):
    """Top-k of (canonical low-rank score + sparse prior), the restatement of `_assign_topk(transform(D) + D.prior_score, k)`
    (src/rime_lite/util/__init__.py:117-155 over ElementWiseExpression(add, [dense, sparse]), score_array.py:300-318;
    call site src/ccrec/models/bbpr.py:592-595): final = (double) canonical_score + prior (torch promotes fp32 + fp64 to
    fp64), order (final desc, column asc).  -> (ids [nq, k] int64, finals [nq, k] float64)."""
    L = canonical_scores(Qb, Db).astype(np.float64)
    indptr, indices, data = np.asarray(indptr, np.int64), np.asarray(indices, np.int64), np.asarray(data, np.float64)
    for q in range(L.shape[0]):
        sl = slice(indptr[q], indptr[q + 1])
        np.add.at(L[q], indices[sl], data[sl])
    L = L + 0.0
    ids = np.empty((L.shape[0], k), np.int64)
    sc = np.empty((L.shape[0], k), np.float64)
    cols = np.arange(L.shape[1])
    for q in range(L.shape[0]):
        o = np.lexsort((cols, -L[q]))[:k]
        ids[q], sc[q] = o, L[q][o]
    return ids, sc


def score_op(Qb, Db, op, indptr=None, indices=None, data=None):
    """score_array.py:460-474: max / min / sum over the whole (low-rank [+ sparse prior]) score matrix, from the canonical
    scores in fp64 (the reference reduces fp32 / fp64 batches with torch.max / min / sum)."""
    L = canonical_scores(Qb, Db).astype(np.float64)
    if indptr is not None:
        for q in range(L.shape[0]):
            sl = slice(indptr[q], indptr[q + 1])
            np.add.at(L[q], np.asarray(indices[sl], np.int64), np.asarray(data[sl], np.float64))
    return float({"max": np.max, "min": np.min, "sum": np.sum}[op](L))


# ----------------------------------------------------------------------------- reference-faithful CPU path
def reference_cos_sim(a, b):
    """ms_marco_eval.py:155-162 in fp32."""
    a = np.atleast_2d(np.asarray(a, np.float32))
    b = np.atleast_2d(np.asarray(b, np.float32))
    an = a / np.maximum(np.linalg.norm(a, axis=1, keepdims=True), 1e-12).astype(np.float32)
    bn = b / np.maximum(np.linalg.norm(b, axis=1, keepdims=True), 1e-12).astype(np.float32)
    return an @ bn.T


def reference_ranking(Eq, Ed, batch_size, sim="dot", block=None, keep=1001):
    """ms_marco_eval.py:203-235 semantics with torch CPU ops (what the reference itself
    executes on a CPU-only host): fp32 chunked matmul into a host [Q,N] matrix, blocked
    ids -> -1e6, per-row full descending sort, keep the first 1001.
    Tie order inside equal scores is whatever torch.sort gives (unspecified)."""
    import torch

    Eq = torch.as_tensor(np.asarray(Eq, np.float32))
    Ed = torch.as_tensor(np.asarray(Ed, np.float32))
    nq, nd = Eq.shape[0], Ed.shape[0]
    M = torch.zeros(nq, nd)
    for lo in range(0, nd, batch_size):
        chunk = Ed[lo:lo + batch_size]
        if sim == "cos":
            s = torch.nn.functional.normalize(Eq, p=2, dim=1) @ torch.nn.functional.normalize(chunk, p=2, dim=1).T
        else:
            s = Eq @ chunk.T
        M[:, lo:lo + chunk.shape[0]] = s
    L = min(keep, nd)
    ids = np.empty((nq, L), np.int64)
    sc = np.empty((nq, L), np.float32)
    for q in range(nq):
        row = M[q]
        if block is not None:
            bi = np.asarray(block[q], np.int64)
            assert bi.size == 0 or (bi.min() >= 0 and bi.max() < nd), "block id not found"
            row[torch.as_tensor(bi)] = -1e6
        s, o = row.sort(descending=True)
        ids[q] = o[:L].numpy()
        sc[q] = s[:L].numpy()
    return ids, sc


# ----------------------------------------------------------------------------- encoder-side pieces
def meanpool(hidden, mask):
    """item_tower.py:141-146: masked mean over the token axis, fp32."""
    hidden = np.ascontiguousarray(hidden, np.float32)
    mask = np.ascontiguousarray(mask, np.int64)
    B, L, d = hidden.shape
    out = np.empty((B, d), np.float32)
    lib().orc_meanpool(_p(hidden), _p(mask), _i64(B), _i64(L), _int(d), _p(out))
    return out


def layer_norm(x, eps=1e-5):
    """torch.nn.LayerNorm(d, elementwise_affine=False) on the CLS row (item_tower.py:135-136)."""
    x = np.asarray(x, np.float64)
    mu = x.mean(-1, keepdims=True)
    var = x.var(-1, keepdims=True)
    return ((x - mu) / np.sqrt(var + eps)).astype(np.float32)


# ----------------------------------------------------------------------------- in-batch-negative contrastive loss
def inbatch_ce(Q, P, Ng, inv_temperature, sim="dot"):
    """bbpr.py:195-212 ('multiple_nrl'): logits = cat(Q P^T, Q N^T) * inv_T, CE(labels=arange(B)), mean.

    Returns (loss, dQ, dP, dN) in fp64 (reference accumulates in fp32; tests use tolerances).
    With sim == 'cos' the three blocks are L2-normalised first (bbpr.py:199-202) and the
    gradients are w.r.t. the un-normalised inputs.
    """
    Q, P, Ng = (np.asarray(a, np.float64) for a in (Q, P, Ng))
    B = Q.shape[0]
    raw = (Q, P, Ng)
    if sim == "cos":
        nrm = [np.maximum(np.linalg.norm(a, axis=1, keepdims=True), 1e-12) for a in raw]
        Q, P, Ng = (a / n for a, n in zip(raw, nrm))
    logits = np.concatenate([Q @ P.T, Q @ Ng.T], 1) * inv_temperature
    m = logits.max(1, keepdims=True)
    e = np.exp(logits - m)
    Z = e.sum(1, keepdims=True)
    lse = np.log(Z) + m
    loss = float((lse[:, 0] - logits[np.arange(B), np.arange(B)]).mean())
    G = e / Z
    G[np.arange(B), np.arange(B)] -= 1.0
    G *= inv_temperature / B
    dQ = G[:, :B] @ P + G[:, B:] @ Ng
    dP = G[:, :B].T @ Q
    dN = G[:, B:].T @ Q
    if sim == "cos":
        out = []
        for g, y, n in zip((dQ, dP, dN), (Q, P, Ng), nrm):
            out.append((g - y * (g * y).sum(1, keepdims=True)) / n)
        dQ, dP, dN = out
    return loss, dQ, dP, dN


# ----------------------------------------------------------------------------- metrics on id tensors
def recall_at_k(ids, ref_ids):
    """mean fraction of ref_ids rows recovered in ids rows (set semantics)."""
    hit = [len(set(a.tolist()) & set(b.tolist())) / max(1, len(b)) for a, b in zip(ids, ref_ids)]
    return float(np.mean(hit))


def mrr(ids, qrels, kmax, n_qrels=None):
    """BEIR custom_metrics.mrr restated on id rows (parity unpinned: beir absent, the reference holds no MRR vectors):
    first relevant hit within kmax -> 1/rank, summed over the result rows, divided by the number of queries in the qrels
    (n_qrels; None = one qrel entry per row), rounded to 5 dp."""
    tot = 0.0
    for row, rel in zip(ids, qrels):
        for r, j in enumerate(row[:kmax]):
            if int(j) in rel:
                tot += 1.0 / (r + 1)
                break
    return round(tot / max(1, len(qrels) if n_qrels is None else n_qrels), 5)


def mrr_beir(qrels, results, k_values):
    """The dict-level form the reference calls: EvaluateRetrieval.evaluate_custom(qrels, results, k_values, metric="mrr")
    (scripts/al_0_rank.py:130-133) -> beir.retrieval.custom_metrics.mrr, restated from BEIR's published source (pinned nowhere in the
    reference: setup.py lists no version; parity unpinned): per query of `results`, the k_max best documents by score (Python's stable
    sort, descending); relevant = documents of qrels[query] with relevance > 0; MRR@k += 1 / rank of the first relevant hit within k;
    every sum is divided by len(qrels) -- ALL queries of the qrels, also those `results` does not hold -- and rounded to 5 decimals."""
    out = {f"MRR@{k}": 0.0 for k in k_values}
    k_max = max(k_values)
    for qid, doc_scores in results.items():
        top = sorted(doc_scores.items(), key=lambda item: item[1], reverse=True)[:k_max]
        relevant = {d for d, r in qrels[qid].items() if r > 0}
        for k in k_values:
            for rank, (doc, _) in enumerate(top[:k]):
                if doc in relevant:
                    out[f"MRR@{k}"] += 1.0 / (rank + 1)
                    break
    return {name: round(v / len(qrels), 5) for name, v in out.items()}


# ----------------------------------------------------------------------------- BM25 (lexical leg of the candidate builder)
_TOKEN = None


def bm25_tokens(text):
    """The analyser the reference's TfidfVectorizer() uses with default settings (scripts/bm_25.py:13; scikit-learn
    1.x CountVectorizer: lowercase, token_pattern r"(?u)\\b\\w\\w+\\b", no stop words, unigrams)."""
    global _TOKEN
    if _TOKEN is None:
        import re
        _TOKEN = re.compile(r"(?u)\b\w\w+\b")
    return _TOKEN.findall(text.lower())


def bm25_fit(texts, b=0.75, k1=1.2):
    """BM25.fit (scripts/bm_25.py:17-28): vocabulary = sorted distinct tokens (scikit-learn orders features
    alphabetically), count matrix in term-major (CSC) form, idf_t - 1 = ln(n / df_t) (smooth_idf=False, bm_25.py:47),
    len_d = counted tokens of d, avdl = mean, doc_k = k1 * (1 - b + b * len_d / avdl) in fp64 (bm_25.py:45)."""
    docs = [bm25_tokens(t) for t in texts]
    vocab = {w: i for i, w in enumerate(sorted({w for d in docs for w in d}))}
    n, V = len(docs), len(vocab)
    rows, cols, vals = [], [], []
    for d, toks in enumerate(docs):
        cnt = {}
        for w in toks:
            cnt[vocab[w]] = cnt.get(vocab[w], 0) + 1
        for t, c in cnt.items():
            rows.append(d)
            cols.append(t)
            vals.append(c)
    rows, cols, vals = np.asarray(rows, np.int64), np.asarray(cols, np.int64), np.asarray(vals, np.float64)
    order = np.lexsort((rows, cols))                       # term-major, documents ascending inside a term
    rows, cols, vals = rows[order], cols[order], vals[order]
    indptr = np.zeros(V + 1, np.int64)
    np.add.at(indptr, cols + 1, 1)
    indptr = np.cumsum(indptr)
    df = np.diff(indptr).astype(np.float64)
    length = np.zeros(n, np.float64)
    np.add.at(length, rows, vals)
    avdl = length.mean()
    return {"vocab": vocab, "indptr": indptr, "doc_ids": rows.astype(np.int32), "tf": vals.astype(np.float32),
            "idf": np.log(n / df), "doc_k": k1 * (1 - b + b * length / avdl), "n_docs": n, "k1": float(k1), "b": float(b)}


def bm25_query_terms(model, text):
    """Distinct in-vocabulary terms of the query, ascending term id (CountVectorizer.transform([q]).indices)."""
    return np.asarray(sorted({model["vocab"][w] for w in bm25_tokens(text) if w in model["vocab"]}), np.int32)


def bm25_scores(model, text):
    """[n_docs] fp32 canonical BM25 scores of one query."""
    terms = bm25_query_terms(model, text)
    idf = np.ascontiguousarray(model["idf"][terms], np.float64)
    out = np.empty(model["n_docs"], np.float32)
    lib().orc_bm25_scores(_p(model["indptr"]), _p(model["doc_ids"]), _p(model["tf"]), _p(np.ascontiguousarray(model["doc_k"])),
                          _i64(model["n_docs"]), _p(terms), _p(idf), _i64(len(terms)), ctypes.c_double(model["k1"]), _p(out))
    return out


def bm25_ranking(model, query_texts, k):
    """ranking_bm25 (scripts/ms_marco_eval.py:165-186) with the canonical tie rule: (ids [nq,k], scores [nq,k])."""
    sc = np.stack([bm25_scores(model, t) for t in query_texts])
    return rank(sc, min(k, model["n_docs"]))
