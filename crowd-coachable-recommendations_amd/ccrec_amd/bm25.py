"""BM25 as a device-side sparse scorer: mirror of scripts/bm_25.py (class BM25) and of
ranking_bm25 (scripts/ms_marco_eval.py:165-186), whose output -- {qid: {pid: score}} with 1001 entries -- is the
`ranking_profile_bm25` the request builder consumes (al_request.build_requests).

The reference fits a TfidfVectorizer on the host and scores one query at a time with scipy against all N documents,
then sorts all N scores.  Here fit() builds the same vocabulary / counts / idf / length factors on the host (text
analysis is host work), keeps the postings on the device, and transform_topk() scores whole query batches with
ccr_bm25_search: fp64 accumulation in ascending term order, one rounding to fp32, exact top-k in the order
(score desc, document index asc).
"""
import ctypes
import re
import time

import numpy as np
import torch

from . import _lib
from .ops import require_gpu

_TOKEN = re.compile(r"(?u)\b\w\w+\b")   # scikit-learn's default token_pattern (what TfidfVectorizer() analyses with)
KEEP = 1001                             # ms_marco_eval.py:183


def analyse(text):
    return _TOKEN.findall(text.lower())


class BM25:
    """fit(X) / transform(q) like scripts/bm_25.py; transform_topk(queries, k) is the batched device path."""

    def __init__(self, b=0.75, k1=1.6):   # bm_25.py:12 (ranking_bm25 passes b=0.75, k1=1.2)
        self.b, self.k1 = float(b), float(k1)
        self._h = None

    def fit(self, X, processes=None, chunk_docs=50_000):
        """X: iterable of document strings.  processes: text-analysis worker processes (ccrec_amd/_bm25_worker.py; None: 8 from
        200 000 documents up, else in this interpreter; 0: never).  The analysis is pure-Python work -- 37 s per million 135-word
        passages in one interpreter -- and embarrassingly parallel over documents; the workers return per-chunk vocabularies and
        (document, term, count) runs, merged here into the same vocabulary / postings / idf / length factors."""
        self._lib = require_gpu()
        X = X if isinstance(X, list) else list(X)
        n = len(X)
        if processes is None:
            processes = 8 if n >= 200_000 else 0
        parts = self._analyse_parallel(X, int(processes), int(chunk_docs)) if processes > 0 and n > chunk_docs else [self._analyse(X)]
        vocab = sorted(set().union(*(p[0] for p in parts)))
        self.vocabulary_ = {w: i for i, w in enumerate(vocab)}
        V = len(vocab)
        assert n >= 1 and V >= 1, "empty corpus or empty vocabulary"
        voc = self.vocabulary_
        # chunk CSRs (documents x chunk vocabulary) -> one CSR over the global vocabulary -> CSC = term-major postings, documents
        # ascending inside a term (scipy's conversion is O(nnz) C code: no 270-M-element sort of (term, document) keys at the NQ size)
        import scipy.sparse as sps
        lengths = np.concatenate([p[1] for p in parts]).astype(np.float64)
        indptr_parts, terms_parts, counts_parts, base = [np.zeros(1, np.int64)], [], [], 0
        for pv, _, pin, pterms, pcounts in parts:
            remap = np.fromiter((voc[w] for w in pv), dtype=np.int64, count=len(pv))
            terms_parts.append(remap[pterms] if len(pv) else pterms.astype(np.int64))
            counts_parts.append(pcounts)
            indptr_parts.append(pin[1:] + base)
            base += int(pin[-1])
        csr = sps.csr_matrix((np.concatenate(counts_parts).astype(np.float32), np.concatenate(terms_parts), np.concatenate(indptr_parts)),
                             shape=(n, V))
        csc = csr.tocsc()
        csc.sort_indices()
        self.indptr = csc.indptr.astype(np.int64)
        rows, counts = csc.indices, csc.data
        self.avdl = float(lengths.mean())
        self.idf = np.log(n / np.diff(self.indptr).astype(np.float64))          # idf_ - 1 with smooth_idf=False
        doc_k = self.k1 * (1 - self.b + self.b * lengths / self.avdl)
        self.n_docs = n
        return self._upload(rows, counts, doc_k, V)

    def _upload(self, rows, counts, doc_k, n_terms):
        self._doc_ids = torch.from_numpy(np.ascontiguousarray(rows, np.int32)).cuda()
        self._tf = torch.from_numpy(np.ascontiguousarray(counts, np.float32)).cuda()
        self._doc_k = torch.from_numpy(np.ascontiguousarray(doc_k, np.float64)).cuda()
        self._h = ctypes.c_void_p()
        _lib.check(self._lib.ccr_bm25_index_create(self.indptr.ctypes.data_as(ctypes.c_void_p), self._doc_ids.data_ptr(),
                                                   self._tf.data_ptr(), self._doc_k.data_ptr(), n_terms, self.n_docs, self.k1,
                                                   ctypes.byref(self._h)), "ccr_bm25_index_create")
        # every query this class builds weighs a term with self.idf: the postings' finished contributions, once (8 B per posting)
        idf = np.ascontiguousarray(self.idf, np.float64)
        self._contrib = torch.empty(max(1, len(rows)), dtype=torch.float64, device="cuda")
        _lib.check(self._lib.ccr_bm25_index_set_idf(self._h, idf.ctypes.data_as(ctypes.c_void_p), self._contrib.data_ptr(),
                                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "ccr_bm25_index_set_idf")
        return self

    @classmethod
    def from_postings(cls, indptr, rows, counts, doc_k, idf, k1=1.6, b=0.75):
        """An index over ready-made term-major postings (CSC of the count matrix: documents strictly ascending inside a term) --
        what fit() builds from text; search it with transform_terms_topk()."""
        self = cls(b=b, k1=k1)
        self._lib = require_gpu()
        self.indptr = np.ascontiguousarray(indptr, np.int64)
        self.idf = np.ascontiguousarray(idf, np.float64)
        self.n_docs = len(doc_k)
        rows = np.asarray(rows)
        inner = np.ones(len(rows), bool)
        inner[self.indptr[:-1][self.indptr[:-1] < len(rows)]] = False        # first posting of every term
        assert len(self.indptr) == len(self.idf) + 1 and self.indptr[-1] == len(rows) == len(counts)
        assert not len(rows) or (0 <= rows.min() and rows.max() < self.n_docs), "document id out of range"
        assert (np.diff(rows.astype(np.int64))[inner[1:]] > 0).all(), "documents must ascend strictly inside a term"
        return self._upload(rows, counts, doc_k, len(self.idf))

    @staticmethod
    def _analyse(texts):
        from ._bm25_worker import analyse_chunk
        return analyse_chunk(texts)

    @staticmethod
    def _analyse_parallel(X, processes, chunk_docs):
        """Chunks of documents through worker processes (children over pipes: pickled texts out, pickled arrays back); a thread per
        worker keeps its pipe busy, the pipe I/O releases the GIL."""
        import os
        import pickle
        import subprocess
        import sys
        from concurrent.futures import ThreadPoolExecutor
        here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bm25_worker.py")
        chunks = [(lo, min(len(X), lo + chunk_docs)) for lo in range(0, len(X), chunk_docs)]
        processes = max(1, min(processes, len(chunks)))
        procs = [subprocess.Popen([sys.executable, "-u", here], stdin=subprocess.PIPE, stdout=subprocess.PIPE) for _ in range(processes)]
        out = [None] * len(chunks)

        def drive(w):
            p = procs[w]
            for ci in range(w, len(chunks), processes):
                lo, hi = chunks[ci]
                blob = pickle.dumps(X[lo:hi], protocol=pickle.HIGHEST_PROTOCOL)
                p.stdin.write(len(blob).to_bytes(8, "little"))
                p.stdin.write(blob)
                p.stdin.flush()
                head = p.stdout.read(8)
                if len(head) != 8:
                    raise RuntimeError("BM25 analysis worker ended unexpectedly")
                size = int.from_bytes(head, "little")
                data = p.stdout.read(size)
                if len(data) != size:
                    raise RuntimeError("BM25 analysis worker ended unexpectedly")
                out[ci] = pickle.loads(data)

        try:
            with ThreadPoolExecutor(max_workers=processes) as pool:
                list(pool.map(drive, range(processes)))
        finally:
            for p in procs:
                try:
                    p.stdin.write((0).to_bytes(8, "little"))
                    p.stdin.flush()
                    p.stdin.close()
                except Exception:
                    pass
            for p in procs:
                try:
                    p.wait(timeout=10)
                except Exception:
                    p.kill()
        return out

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.ccr_bm25_index_destroy(h)

    def query_terms(self, q):
        """Distinct in-vocabulary term ids of q, ascending (CountVectorizer.transform([q]).indices)."""
        voc = self.vocabulary_
        return np.asarray(sorted({voc[w] for w in analyse(q) if w in voc}), np.int32)

    def transform_topk(self, queries, k):
        """queries: list of strings -> (scores [n_q, k] fp32, document rows [n_q, k] int64) on the device."""
        assert self._h, "fit() first"
        return self.transform_terms_topk([self.query_terms(q) for q in queries], k)

    def transform_terms_topk(self, terms, k):
        """terms: per query, its distinct term ids in ascending order -> (scores [n_q, k] fp32, document rows [n_q, k] int64)."""
        assert self._h, "fit() first"
        k = min(int(k), self.n_docs)
        n_q = len(terms)
        q_ptr = np.zeros(n_q + 1, np.int64)
        q_ptr[1:] = np.cumsum([len(t) for t in terms])
        q_terms = np.concatenate(terms).astype(np.int32) if q_ptr[-1] else np.zeros(1, np.int32)
        q_idf = np.ascontiguousarray(self.idf[q_terms[:q_ptr[-1]]], np.float64) if q_ptr[-1] else np.zeros(1, np.float64)
        scores = torch.empty(n_q, k, dtype=torch.float32, device="cuda")
        ids = torch.empty(n_q, k, dtype=torch.int64, device="cuda")
        if n_q == 0:
            return scores, ids
        need = int(self._lib.ccr_bm25_search_workspace_bytes_k(self._h, n_q, int(max(len(t) for t in terms)), k))
        ws = torch.empty(need, dtype=torch.uint8, device="cuda")
        vp = ctypes.c_void_p
        t0 = time.perf_counter()
        _lib.check(self._lib.ccr_bm25_search(self._h, q_ptr.ctypes.data_as(vp), q_terms.ctypes.data_as(vp), q_idf.ctypes.data_as(vp),
                                             n_q, k, scores.data_ptr(), ids.data_ptr(), ws.data_ptr(), ws.numel(),
                                             vp(torch.cuda.current_stream().cuda_stream)), "ccr_bm25_search")
        torch.cuda.current_stream().synchronize()   # the host arrays above must outlive the stream work
        self.last_search_seconds = time.perf_counter() - t0   # the library call alone (tables + kernels), without the text analysis
        return scores, ids

    def last_stats(self):
        """Of the last search: which path ran, how many rows the fused filter could not finish, batches, the sampled threshold's rank."""
        out = np.zeros(4, np.int64)
        _lib.check(self._lib.ccr_bm25_search_last_stats(self._h, out.ctypes.data_as(ctypes.c_void_p)), "ccr_bm25_search_last_stats")
        return {"path": ("rounds+stored_rows", "tile+stored_rows", "tile+fused_filter")[int(out[0]) & 3], "contribution_table": bool(int(out[0]) & 4),
                "rows_redone": int(out[1]), "batches": int(out[2]), "sample_rank": int(out[3])}

    def transform(self, q, X=None):
        """bm_25.py:31-52: dense [n_docs] score vector of one query (small corpora / tests: reads back k = n_docs)."""
        assert X is None, "re-caching another corpus: fit() a new BM25 instead"
        s, i = self.transform_topk([q], self.n_docs)
        out = np.zeros(self.n_docs, np.float32)
        out[i[0].cpu().numpy()] = s[0].cpu().numpy()
        return out


def ranking_bm25(corpus, queries, b=0.75, k1=1.2, keep=KEEP, batch=4096, lazy=False):
    """scripts/ms_marco_eval.py:165-186: {qid: {pid: score}} in rank order, min(1001, N) entries per query.
    lazy=True: the same mapping backed by the result tensors (ranking_profile.RankingProfile) -- the request builder reads the first
    three passages of a few hundred queries, not 1001 x Q Python pairs."""
    from .ranking_profile import RankingProfile
    print("Fitting BM-25 model")
    model = BM25(b=b, k1=k1).fit(list(corpus.values()))
    print("Retrieval with BM-25 model")
    queries_ids, corpus_ids = list(queries.keys()), list(corpus.keys())
    rows, scores = [], []
    for lo in range(0, len(queries_ids), batch):
        qids = queries_ids[lo:lo + batch]
        print("processing query: {} | {}".format(lo, len(queries_ids)))
        s, i = model.transform_topk([queries[q] for q in qids], keep)
        rows.append(i.cpu())
        scores.append(s.cpu())
    k = min(int(keep), model.n_docs)
    rows = torch.cat(rows) if rows else torch.zeros(0, k, dtype=torch.int64)
    scores = torch.cat(scores) if scores else torch.zeros(0, k, dtype=torch.float32)
    profile = RankingProfile(queries_ids, corpus_ids, rows, scores)
    return profile if lazy else profile.to_dict()
