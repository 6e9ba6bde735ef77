"""rime_lite surface of the same computation (src/rime_lite/util/__init__.py:117-155, score_array.py:320-339):
_assign_topk(S, k) -> CSR with k ones per row whose column order is the rank order.

When S is a low-rank product of two dense factors (MatMulExpression(left @ right), or an object with
.left/.right dense children, or a (U, V) tuple meaning U @ V.T) the product is never materialised:
the factors are packed to bf16 and sent through the fused top-k.  tie_breaker is accepted for
signature compatibility; the canonical (score desc, index asc) rule replaces the 1e-10 random jitter
(which is below one fp32 ulp for |score| > 1e-3: SURVEY App. B)."""
import numpy as np
import scipy.sparse as sps
import torch

from . import ops


def _dense(x):
    if hasattr(x, "c"):          # LazyDenseMatrix keeps its ndarray in .c (score_array.py:219-220)
        x = x.c
    elif hasattr(x, "numpy") and not isinstance(x, torch.Tensor):
        x = x.numpy()
    return torch.as_tensor(np.asarray(x) if not isinstance(x, torch.Tensor) else x, dtype=torch.float32)


def _factors(S):
    """-> (U [n_users, d], V [n_items, d]) such that S == U @ V.T, or None."""
    if isinstance(S, tuple) and len(S) == 2:
        return _dense(S[0]), _dense(S[1])
    if hasattr(S, "left") and hasattr(S, "right"):
        left, right = _dense(S.left), _dense(S.right)   # right is [d, n_items]
        return left, right.T.contiguous()
    return None


def _csr_of_ones(indices, shape):
    return sps.csr_matrix((np.ones(indices.size), np.ravel(indices), np.arange(0, indices.size + 1, indices.shape[1])),
                          shape=shape)


def _low_rank_plus_sparse(S):
    """rime_lite's ElementWiseExpression(add, [low-rank, LazySparseMatrix]) (score_array.py:300-318) -> our lazy type."""
    import operator
    from .bbpr_transform import LowRankPlusSparse, LowRankScore
    if getattr(S, "op", None) is not operator.add or len(getattr(S, "children", ())) != 2:
        return None
    low, prior = None, None
    for c in S.children:
        if sps.issparse(getattr(c, "c", None)):
            prior = c.c
        elif isinstance(c, LowRankScore):
            low = c
        else:
            fac = _factors(c)
            if fac is not None:
                ops.require_gpu()
                U, V = fac      # any factor width: the pack zero-pads to a multiple of 8, the fused kernels zero-fill their K tail
                low = LowRankScore(ops.pack_bf16(U.cuda()), ops.pack_bf16(V.cuda()))
    return LowRankPlusSparse(low, prior) if low is not None and prior is not None else None


def _assign_topk(S, k, tie_breaker=1e-10, device="cpu", batch_size=None):
    if hasattr(S, "topk") and hasattr(S, "shape") and not isinstance(S, torch.Tensor):   # LowRankScore / LowRankPlusSparse
        _, ids = S.topk(k)
        return _csr_of_ones(ids.cpu().numpy(), S.shape)
    lps = _low_rank_plus_sparse(S)
    if lps is not None:
        _, ids = lps.topk(k)
        return _csr_of_ones(ids.cpu().numpy(), lps.shape)
    fac = _factors(S)
    if fac is None:
        raise NotImplementedError("ccrec_amd._assign_topk handles low-rank (left @ right) scores with an optional sparse "
                                  "prior; other LazyScore expressions stay on rime_lite's own path")
    U, V = fac
    ops.require_gpu()
    index = ops.CorpusIndex(ops.pack_bf16(V.cuda()))     # any factor width (zero-padded to a multiple of 8 by the pack)
    _, ids = index.search(ops.pack_bf16(U.cuda()), k)
    indices = ids.cpu().numpy()
    shape = (U.shape[0], V.shape[0])
    return sps.csr_matrix((np.ones(indices.size), np.ravel(indices), np.arange(0, indices.size + 1, indices.shape[1])),
                          shape=shape)


assign_topk = _assign_topk


def perplexity(x):
    """exp(entropy) of the normalised counts (src/rime_lite/util/__init__.py:158-160)."""
    x = np.ravel(x) / np.sum(x)
    return float(np.exp(-x @ np.log(np.where(x > 0, x, 1e-10))))


def _dense_sum(x, axis):
    return np.asarray(x.sum(axis)).ravel() if axis is not None else float(x.sum())


def evaluate_assigned(target_csr, assigned_csr, score_mat=None, axis=None, min_total_recs=0, device="cpu"):
    """src/rime_lite/metrics/__init__.py:52-84 for a sparse / dense target and a sparse assignment: precision, coverage,
    perplexity (and recall along `axis`, objective mean when a score is given).  score_mat may be a low-rank lazy score
    (bbpr_transform.LowRankScore, a (U, V) pair or an object with .left/.right): only the assigned cells are scored."""
    target = sps.csr_matrix(target_csr)
    assigned = sps.csr_matrix(assigned_csr)
    hit = target.multiply(assigned)
    hit_axis = _dense_sum(hit, axis) if axis is not None else float(hit.sum())
    sum0, sum1 = _dense_sum(assigned, 0), _dense_sum(assigned, 1)
    min_total_recs = max(min_total_recs, sum0.sum())
    out = {
        "prec": np.sum(hit_axis) / min_total_recs,
        "recs/user": sum1.mean(),
        "item_cov": (sum0 > 0).mean(),
        "item_ppl": perplexity(sum0),
        "user_cov": (sum1 > 0).mean(),
        "user_ppl": perplexity(sum1),
    }
    if score_mat is not None:
        coo = assigned.tocoo()
        prior = None
        if hasattr(score_mat, "low") and hasattr(score_mat, "prior"):   # bbpr_transform.LowRankPlusSparse
            score_mat, prior = score_mat.low, score_mat.prior
        if hasattr(score_mat, "user") and hasattr(score_mat, "item"):
            U, V = score_mat.user.float(), score_mat.item.float()
        else:
            fac = _factors(score_mat)
            U, V = (fac[0], fac[1]) if fac is not None else (None, None)
        if U is not None:
            rows = torch.as_tensor(coo.row, dtype=torch.long, device=U.device)
            cols = torch.as_tensor(coo.col, dtype=torch.long, device=U.device)
            w = torch.as_tensor(coo.data, dtype=torch.float64, device=U.device)
            obj_sum = float(((U[rows].double() * V[cols].double()).sum(1) * w).sum())
            if prior is not None:   # the assigned cells' share of the sparse prior
                obj_sum += float(prior.multiply(assigned).sum())
        else:
            dense = np.asarray(score_mat.numpy() if hasattr(score_mat, "numpy") else score_mat)
            obj_sum = float((dense[coo.row, coo.col] * coo.data).sum())
        out["obj_mean"] = float(obj_sum / min_total_recs)
    if axis is not None:
        ideal = np.ravel(target.sum(axis=axis))
        # as the reference computes it (metrics/__init__.py:80-82): the hit counts keep the [n, 1] / [1, n] shape of a
        # sparse .sum(axis), so along axis=1 the division by the flat `ideal` broadcasts to an n x n table before the mean
        hits_2d = np.asarray(hit.sum(axis))
        out["recall"] = (hits_2d / np.fmax(1, ideal)).mean()
    return out


def evaluate_item_rec(target_csr, score_mat, topk, device="cpu", **kw):
    """src/rime_lite/metrics/__init__.py:87-89: top-k assignment of the (lazy) score, then evaluate_assigned along users."""
    assigned_csr = _assign_topk(score_mat, topk, device=device, **kw)
    return evaluate_assigned(target_csr, assigned_csr, score_mat, axis=1, device=device)
