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


def _assign_topk(S, k, tie_breaker=1e-10, device="cpu", batch_size=None):
    if hasattr(S, "topk") and hasattr(S, "user") and hasattr(S, "item"):   # bbpr_transform.LowRankScore
        _, ids = S.topk(k)
        indices = ids.cpu().numpy()
        return sps.csr_matrix((np.ones(indices.size), np.ravel(indices), np.arange(0, indices.size + 1, indices.shape[1])),
                              shape=S.shape)
    fac = _factors(S)
    if fac is None:
        raise NotImplementedError("ccrec_amd._assign_topk handles low-rank (left @ right) scores; "
                                  "dense/sparse LazyScore expressions stay on rime_lite's own path")
    U, V = fac
    ops.require_gpu()
    dim = U.shape[1]
    pad = (-dim) % 8   # the kernels want dim % 8 == 0; zero columns do not change any dot product
    if pad:
        U = torch.nn.functional.pad(U, (0, pad))
        V = torch.nn.functional.pad(V, (0, pad))
    index = ops.CorpusIndex(ops.pack_bf16(V.cuda()))
    _, ids = index.search(ops.pack_bf16(U.cuda()), k)
    indices = ids.cpu().numpy()
    shape = (U.shape[0], V.shape[0])
    return sps.csr_matrix((np.ones(indices.size), np.ravel(indices), np.arange(0, indices.size + 1, indices.shape[1])),
                          shape=shape)


assign_topk = _assign_topk
