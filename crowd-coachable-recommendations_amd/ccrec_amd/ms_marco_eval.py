"""Drop-in for the retrieval driver of the reference (scripts/ms_marco_eval.py): same names, argument
meaning and error behaviour for generate_embeddings / cos_sim / ranking, with the
encode-output -> score -> top-k path on the MI355X library instead of host fp32 matrices.

What changes under the hood (reference line -> here):
  :141-149  per-batch D2H copy + vstack of fp32 embeddings -> batches are packed to bf16 on device
            into one resident shard (pack= keyword), fp32 [N,768] never materialises;
  :204-218  host [Q,N] score matrix filled by chunked matmuls     -> CorpusIndex.search (fused MFMA top-k);
  :224-227  scores[block_ind] = -1e6                               -> ccr_search_blocked (block lists of any length);
  :228-230  per-row H2D + full sort, keep 1001                     -> the same search call (k = min(1001, N)).
Order rule: score descending, corpus index ascending on equal scores (the reference's sort leaves
tie order unspecified).
"""
import math
import os
import time

import numpy as np
import torch

from . import ops
from .ranking_profile import RankingProfile

KEEP = 1001  # ms_marco_eval.py:230
CANONICAL_COS_SIM_MACS = 4e9   # above this many multiply-adds cos_sim() switches from the fp64 to the MFMA score kernel


def generate_embeddings(data_indices, data_dic, embedding_func, batch_size, embedding_size=768, name=None, pack=None,
                        device=None, norm_bounds=None):
    """scripts/ms_marco_eval.py:123-152.  Same batching, progress lines and return value (a [num, dim]
    tensor), but the result stays on the GPU.  pack=None -> fp32; pack="dot"/"cos" -> bf16 shard packed
    batch by batch (cos = L2-normalised first)."""
    ops.require_gpu()
    num = len(data_indices)
    num_batches = math.ceil(num / batch_size)
    device = torch.device(device if device is not None else "cuda")
    out = None
    tic = time.time()
    with torch.no_grad():
        for step in range(num_batches):
            if step != 0 and step & (step - 1) == 0:  # power of 2
                print(f"Processed {step * batch_size} | {num}", f"t={time.time() - tic:.1f}s",
                      f"/ {(time.time() - tic) * num / (step * batch_size):.1f}s")
            indices = data_indices[step * batch_size:(step + 1) * batch_size]
            text_batch = [data_dic[index] for index in indices]
            emb = torch.as_tensor(embedding_func(text_batch)).to(
                device, non_blocking=bool(int(os.environ.get("CCREC_NON_BLOCKING", "1"))))
            if out is None:
                dim = emb.shape[1] if pack is None else ops.padded_dim(emb.shape[1])   # packed rows: zero-padded to a multiple of 8
                out = torch.empty(num, dim, device=device, dtype=torch.float32 if pack is None else torch.bfloat16)
            lo = step * batch_size
            if pack is None:
                out[lo:lo + emb.shape[0]] = emb
            else:
                ops.pack_bf16(emb, normalize=(pack == "cos"), out=out[lo:lo + emb.shape[0]],
                              norm_bounds=None if norm_bounds is None else norm_bounds[lo:lo + emb.shape[0]])
    torch.cuda.synchronize()
    print(f"Processed total {num} t={time.time() - tic:.1f}s")
    if out is None:
        out = torch.empty(0, embedding_size if pack is None else ops.padded_dim(embedding_size), device=device,
                          dtype=torch.float32 if pack is None else torch.bfloat16)
    if name is not None:
        torch.save(out, name)
    return out


def cos_sim(a: torch.Tensor, b: torch.Tensor):
    """scripts/ms_marco_eval.py:155-162: cosine similarity matrix [len(a), len(b)].
    Rows are L2-normalised and rounded to bf16 by the pack kernel; the products are the canonical
    fp64-ordered scores of those bf16 rows (within 1e-3 of the fp32 reference)."""
    ops.require_gpu()
    if len(a.shape) == 1:
        a = a.unsqueeze(0)
    if len(b.shape) == 1:
        b = b.unsqueeze(0)
    a_n = ops.pack_bf16(a.cuda().float(), normalize=True)
    b_n = ops.pack_bf16(b.cuda().float(), normalize=True)
    # small matrices: the canonical fp64-ordered scores (bit-identical to the oracle); large ones: the MFMA tile kernel
    # (fp32 accumulation of the same bf16 rows, within 1e-6 of the canonical value) -- the fp64 path is VALU-bound
    big = a_n.shape[0] * b_n.shape[0] * a_n.shape[1] > CANONICAL_COS_SIM_MACS
    return ops.CorpusIndex(b_n).scores(a_n, "mfma" if big else "canonical")


def block_csr(block_lists, n_rows):
    """Per-query lists of blocked corpus rows -> CSR (ptr [Q+1], idx) with the rows of a query ascending and unique.
    AssertionError("block id not found") on a row outside [0, n_rows), as ms_marco_eval.py:226."""
    flat = [sorted(set(int(j) for j in b)) for b in block_lists]
    ptr = torch.zeros(len(flat) + 1, dtype=torch.int64)
    ptr[1:] = torch.cumsum(torch.tensor([len(f) for f in flat], dtype=torch.int64), 0)
    idx = torch.tensor([j for f in flat for j in f], dtype=torch.int64)
    assert idx.numel() == 0 or (int(idx.min()) >= 0 and int(idx.max()) < n_rows), "block id not found"
    return ptr, idx


class Retriever:
    """Build the index once per active-learning step, query many times.

    corpus_ids: list of passage ids in corpus row order; corpus_bf16: packed shard [N, dim]."""

    def __init__(self, corpus_ids, corpus_bf16, global_row_offset=0, norm_bounds=None):
        self.corpus_ids = list(corpus_ids)
        self.index = ops.CorpusIndex(corpus_bf16, global_row_offset, norm_bounds=norm_bounds)
        self._pos = None
        self._cid_arr = None   # corpus ids as a numpy object array (built on first use)

    def _positions(self):
        if self._pos is None:
            self._pos = {pid: i for i, pid in enumerate(self.corpus_ids)}
        return self._pos

    def search(self, queries_bf16, k, block_lists=None):
        """-> (scores [Q,k], ids [Q,k]) on device.  block_lists: per query list of blocked corpus rows (any length)."""
        n = self.index.n_rows
        k = min(k, n)
        if block_lists is None:
            return self.index.search(queries_bf16, k)
        ptr, idx = block_csr(block_lists, n)
        return self.index.search_blocked(queries_bf16, k, ptr, idx + self.index.offset)

    def corpus_id_array(self):
        """Corpus ids as a numpy object array (ids of any hashable type): a row of search results becomes its pids by ONE take."""
        if self._cid_arr is None:
            self._cid_arr = np.fromiter(self.corpus_ids, dtype=object, count=len(self.corpus_ids))
        return self._cid_arr

    def ranking_profile(self, queries_ids, queries_bf16, block_dict=None, keep=KEEP, with_tensors=False, lazy=False):
        """{qid: {pid: score}} in rank order.  lazy=True: a ranking_profile.RankingProfile over the search's tensors (same
        Mapping behaviour, inner dicts built when a query is read) instead of Q x keep Python pairs.
        with_tensors: also return the device tensors it was built from, (row ids [Q, keep] int64 into corpus_ids,
        scores [Q, keep]) -- evaluation.rank_metrics takes them as they are."""
        block_lists = None
        if block_dict is not None:
            print("using block_dict")
            pos = self._positions()
            block_lists = []
            for qid in queries_ids:
                rows = [pos.get(pid, -1) for pid in block_dict[qid]]
                assert -1 not in rows, "block id not found"
                block_lists.append(rows)
        scores_t, ids_t = self.search(queries_bf16, keep, block_lists)
        ids_t = ids_t - self.index.offset
        profile = RankingProfile(queries_ids, self.corpus_id_array(), ids_t, scores_t)
        if not lazy:
            # the reference's nested dict: 3.5 M (pid, score) pairs at the NQ shape -- one numpy take on the object array for the
            # pids, dict(zip()) from whole rows (1.3 s per NQ-sized profile: CPython's floor for that many inserts)
            profile = profile.to_dict()
        return (profile, ids_t, scores_t) if with_tensors else profile


def ranking(corpus, queries, embedding_func, batch_size, block_dict=None, lazy=False):
    """scripts/ms_marco_eval.py:189-235: {qid: {pid: score}} ordered by rank, min(1001, N) entries per
    query; similarity from os.environ["CCREC_SIM_TYPE"] (KeyError if unset, as in the reference);
    blocked ids are scored -1e6, not removed; AssertionError("block id not found") on unknown ids.
    lazy=True (not in the reference): the same Mapping backed by the result tensors (ranking_profile.RankingProfile)."""
    ops.require_gpu()
    queries_ids, corpus_ids = list(queries.keys()), list(corpus.keys())
    sim = "cos" if os.environ["CCREC_SIM_TYPE"] == "cos" else "dot"
    queries_embeddings = generate_embeddings(queries_ids, queries, embedding_func, batch_size, pack=sim)
    # the pack kernel leaves a norm bound per packed row, batch by batch: the index build then needs no pass over the shard
    bounds = torch.empty(len(corpus_ids), dtype=torch.float32, device="cuda") if corpus_ids else None
    passage_embeddings = generate_embeddings(corpus_ids, corpus, embedding_func, batch_size, pack=sim, norm_bounds=bounds)
    retriever = Retriever(corpus_ids, passage_embeddings, norm_bounds=bounds)
    return retriever.ranking_profile(queries_ids, queries_embeddings, block_dict, lazy=lazy)
