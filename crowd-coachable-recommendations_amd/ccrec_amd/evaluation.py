"""Retrieval quality straight from the device id tensor (the consumer side of ranking_profile,
scripts/al_0_rank.py:130-133): MRR@k as BEIR's custom mrr computes it (first relevant hit within k, averaged over
the queries of `qrels`, rounded to 5 decimals) and Recall@k, without materialising {qid: {pid: score}} dicts.
Also the tensor form of ranking_profile.pt: (query ids, corpus ids, ids [Q,k], scores [Q,k]) <-> nested dict."""
import ctypes

import torch

from . import _lib, ops


def rank_metrics(ids, qrel_lists, k_values=(1, 5, 10, 100), n_qrels=None):
    """ids: [Q, k] int64 cuda (rank order, global ids); qrel_lists: per query iterable of relevant ids (relevance > 0 only).
    n_qrels: the number of queries in the caller's `qrels` -- BEIR's mrr sums over the queries of `results` and divides by len(qrels)
    (the reference passes its full qrels, scripts/al_0_rank.py:130-133), so a profile that holds fewer queries than the qrels is
    averaged over the qrels' count; None = the Q queries passed in.
    -> {"MRR@k": float, "Recall@k": float} per cut-off."""
    lib = ops.require_gpu()
    assert ids.is_cuda and ids.dtype == torch.int64 and ids.dim() == 2
    n_q, k = ids.shape
    assert len(qrel_lists) == n_q
    flat = [sorted(set(int(j) for j in rel)) for rel in qrel_lists]
    ptr = torch.zeros(n_q + 1, dtype=torch.int64)
    ptr[1:] = torch.cumsum(torch.tensor([len(f) for f in flat], dtype=torch.int64), 0)
    idx = torch.tensor([j for f in flat for j in f] or [0], dtype=torch.int64)
    kv = torch.tensor(list(k_values), dtype=torch.int32)
    dev = ids.device
    rr = torch.empty(n_q, len(kv), dtype=torch.float32, device=dev)
    hits = torch.empty(n_q, len(kv), dtype=torch.int32, device=dev)
    ids = ids.contiguous()
    ptr_d, idx_d, kv_d = ptr.to(dev), idx.to(dev), kv.to(dev)
    with torch.cuda.device(dev):
        _lib.check(lib.ccr_rank_metrics(ctypes.c_void_p(ids.data_ptr()), n_q, k, ctypes.c_void_p(ptr_d.data_ptr()),
                                        ctypes.c_void_p(idx_d.data_ptr()), ctypes.c_void_p(kv_d.data_ptr()), len(kv),
                                        ctypes.c_void_p(rr.data_ptr()), ctypes.c_void_p(hits.data_ptr()),
                                        ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "ccr_rank_metrics")
    rr_sum = rr.double().sum(0).cpu()
    nrel = torch.tensor([max(1, len(f)) for f in flat], dtype=torch.float64)
    has = torch.tensor([len(f) > 0 for f in flat])
    rec = (hits.cpu().double() / nrel[:, None])[has].mean(0) if bool(has.any()) else torch.zeros(len(kv), dtype=torch.float64)
    out = {}
    for j, kk in enumerate(k_values):
        out[f"MRR@{kk}"] = round(float(rr_sum[j]) / max(1, n_q if n_qrels is None else int(n_qrels)), 5)
        out[f"Recall@{kk}"] = round(float(rec[j]), 5)
    return out


def profile_to_tensors(ranking_profile, corpus_ids, positions=None):
    """{qid: {pid: score}} (rank-ordered; a nested dict or a lazy RankingProfile) -> (query ids, ids [Q,k] int64 into
    corpus_ids, scores [Q,k] fp32) on the host.  positions: {pid: row}, if the caller has it already."""
    from .ranking_profile import as_tensors
    return as_tensors(ranking_profile, corpus_ids, positions)


def tensors_to_profile(qids, corpus_ids, ids, scores):
    """Inverse of profile_to_tensors: rebuilds the rank-ordered nested dict the AL scripts consume."""
    ids, scores = ids.cpu().tolist(), scores.cpu().tolist()
    return {q: dict(zip([corpus_ids[j] for j in ri], rs)) for q, ri, rs in zip(qids, ids, scores)}
