"""Ranker-API twin of the retrieval path: src/ccrec/models/bbpr.py:466-550 (BertBPR.get_all_embeddings /
BertBPR.transform).  The reference encodes every item title into a host fp32 [n_items, 768] matrix, indexes
user rows, fills a dense host [n_users, n_items] score matrix chunk by chunk and wraps it in a
LazyDenseMatrix.  Here the embeddings are packed to bf16 on device as they are produced and transform()
returns a LOW-RANK lazy score (user rows x item rows); nothing of size n_users x n_items exists unless a
caller asks for .as_tensor().  rime_util._assign_topk consumes it through the fused top-k.

`transform(D) + D.prior_score` (the post-fit expression of bbpr.py:592-595 / bert_mt.py:375-377) stays lazy too:
LowRankScore + scipy-sparse -> LowRankPlusSparse, whose top-k runs on ccr_search_sparse_prior."""
import os

import numpy as np
import scipy.sparse as sps
import torch

from . import ops


def get_all_embeddings(item_titles, tokenizer, tokenizer_kw, model, batch_size, output_step="embedding", sim_type=None):
    """bbpr.py:466-483: encode all titles in batches of `batch_size` -> packed bf16 [n_items, dim] on device
    (cos: rows L2-normalised by the pack kernel)."""
    ops.require_gpu()
    if sim_type is None:
        sim_type = os.environ["CCREC_SIM_TYPE"]
    if output_step == "embedding":
        output_step = os.environ["CCREC_EMBEDDING_TYPE"]
    all_texts = list(item_titles)
    num = len(all_texts)
    num_batches = int(np.ceil(num / batch_size))
    out = None
    with torch.no_grad():
        for step in range(num_batches):
            if (step % 100) == 0:
                print("Processing", step, "|", num_batches)
            text_batch = all_texts[step * batch_size:(step + 1) * batch_size]
            tokens = tokenizer(text_batch, **tokenizer_kw)
            emb = torch.as_tensor(model(**tokens, output_step=output_step)).cuda()
            if out is None:
                out = torch.empty(num, ops.padded_dim(emb.shape[1]), dtype=torch.bfloat16, device=emb.device)
            ops.pack_bf16(emb, normalize=(sim_type == "cos"), out=out[step * batch_size:step * batch_size + len(text_batch)])
    return out


def _row_index(key, n, device):
    if isinstance(key, slice):
        key = range(n)[key]
    idx = torch.as_tensor(np.atleast_1d(np.asarray(key)), dtype=torch.long)
    return (idx % n).to(device)   # rime_lite's dense slicing takes the row index modulo the row count (score_array.py:227-231)


class LowRankScore:
    """Lazy [n_users, n_items] score = U V^T over packed bf16 rows (duck-types the .shape / .T / as_tensor
    surface of rime_lite's LazyScoreBase that the evaluation code touches)."""

    def __init__(self, user_bf16, item_bf16):
        assert user_bf16.dtype == torch.bfloat16 and item_bf16.dtype == torch.bfloat16
        self.user, self.item = user_bf16.contiguous(), item_bf16.contiguous()
        self.shape = (self.user.shape[0], self.item.shape[0])
        self._index = None

    def __len__(self):
        return self.shape[0]

    @property
    def size(self):
        return int(np.prod(self.shape))

    @property
    def T(self):
        return LowRankScore(self.item, self.user)

    def index(self):
        if self._index is None:
            self._index = ops.CorpusIndex(self.item)
        return self._index

    def __getitem__(self, key):
        """Row (user) subset, as rime_lite's lazy scores are sliced for batching (score_array.py:98-100, 332-333):
        an int, a slice or an index array."""
        rows = _row_index(key, self.shape[0], self.user.device)
        out = LowRankScore(self.user[rows], self.item)
        out._index = self._index   # same item rows: the search index carries over
        return out

    @staticmethod
    def collate_fn(batch):
        """Stack row subsets of ONE score back together (score_array.py:335-338)."""
        out = LowRankScore(torch.cat([b.user for b in batch], 0), batch[0].item)
        out._index = batch[0]._index
        return out

    def as_tensor(self, device=None):
        """Dense scores (this materialises n_users x n_items fp32): canonical fp64-ordered values for small problems,
        the MFMA tile kernel (same bf16 rows, fp32 accumulation) above 4e9 multiply-adds."""
        big = self.shape[0] * self.shape[1] * self.user.shape[1] > 4e9
        s = self.index().scores(self.user, "mfma" if big else "canonical")
        return s if device is None else s.to(device)

    def numpy(self):
        return self.as_tensor().cpu().numpy()

    def topk(self, k):
        """-> (scores [n_users, k], item rows [n_users, k]) in canonical order."""
        return self.index().search(self.user, k)

    def __add__(self, other):
        """+ a sparse prior (scipy.sparse, or an object carrying one in .c like rime_lite's LazySparseMatrix)."""
        if other is None or _is_scalar_zero(other):   # rime_lite's datasets start with prior_score = 0 (dataset/base.py:217)
            return self
        c = other.c if hasattr(other, "c") else other
        if sps.issparse(c):
            return LowRankPlusSparse(self, c)
        if isinstance(c, (np.ndarray, torch.Tensor)) and tuple(c.shape) == tuple(self.shape):
            # a dense addend: the sum is dense (the reference's ElementWiseExpression would materialise it as well)
            return self.as_tensor() + torch.as_tensor(c, device=self.user.device)
        return NotImplemented

    __radd__ = __add__


def _is_scalar_zero(x):
    return isinstance(x, (int, float, np.integer, np.floating)) and x == 0


class LowRankPlusSparse:
    """Lazy U V^T + P with P sparse [n_users, n_items] (the reference's ElementWiseExpression(add, [dense, sparse]),
    score_array.py:300-318).  final = (double) canonical_score + prior, as torch promotes fp32 + fp64."""

    def __init__(self, low, prior):
        prior = sps.csr_matrix(prior, dtype=np.float64)
        assert prior.shape == low.shape, f"shape mismatch {prior.shape} vs {low.shape}"
        prior.sum_duplicates()
        prior.sort_indices()
        self.low, self.prior, self.shape = low, prior, low.shape

    def __len__(self):
        return self.shape[0]

    @property
    def size(self):
        return int(np.prod(self.shape))

    @property
    def T(self):
        return LowRankPlusSparse(self.low.T, self.prior.T.tocsr())

    def __add__(self, other):
        if other is None or _is_scalar_zero(other):
            return self
        c = other.c if hasattr(other, "c") else other
        if sps.issparse(c):
            return LowRankPlusSparse(self.low, self.prior + sps.csr_matrix(c, dtype=np.float64))
        if isinstance(c, (np.ndarray, torch.Tensor)) and tuple(c.shape) == tuple(self.shape):
            return self.as_tensor() + torch.as_tensor(c, device=self.low.user.device)
        return NotImplemented

    __radd__ = __add__

    def __getitem__(self, key):
        rows = _row_index(key, self.shape[0], "cpu").numpy()
        return LowRankPlusSparse(self.low[rows], self.prior[rows])

    @staticmethod
    def collate_fn(batch):
        return LowRankPlusSparse(LowRankScore.collate_fn([b.low for b in batch]), sps.vstack([b.prior for b in batch], "csr"))

    def as_tensor(self, device=None):
        dense = self.low.as_tensor().double()
        coo = self.prior.tocoo()
        if coo.nnz:
            r = torch.as_tensor(coo.row, dtype=torch.long, device=dense.device)
            c = torch.as_tensor(coo.col, dtype=torch.long, device=dense.device)
            dense.index_put_((r, c), torch.as_tensor(coo.data, dtype=torch.float64, device=dense.device), accumulate=True)
        return dense if device is None else dense.to(device)

    def numpy(self):
        return self.as_tensor().cpu().numpy()

    def topk(self, k):
        """-> (final scores [n_users, k] fp64, item rows [n_users, k]) by (final desc, row asc)."""
        p = self.prior
        return self.low.index().search_sparse_prior(self.low.user, k, p.indptr.astype(np.int64), p.indices.astype(np.int64),
                                                    p.data)


def score_op(S, op, device=None, reduce_fn=None):
    """score_array.py:460-474 (max / min / sum over the whole matrix) for the lazy scores of this module, without
    materialising any batch: max / min = the best of every row's top-1 (fused search; with a prior: the sparse-prior
    search), sum = colsum(U) . colsum(V) (+ the prior's sum).  Returns a python float."""
    low = S.low if isinstance(S, LowRankPlusSparse) else S
    assert isinstance(low, LowRankScore), "score_op handles LowRankScore / LowRankPlusSparse"
    prior = S.prior if isinstance(S, LowRankPlusSparse) else None
    if op == "sum":
        total = float(torch.dot(ops.colsum_bf16(low.user), ops.colsum_bf16(low.item)))
        return total + (float(prior.sum()) if prior is not None else 0.0)
    if op not in ("max", "min"):
        raise NotImplementedError(f"score_op: {op}")
    if op == "min":   # min S = -max(-S): negating a bf16 row is exact
        neg = LowRankScore(-low.user, low.item)
        negs = neg if prior is None else LowRankPlusSparse(neg, -prior)
        return -score_op(negs, "max")
    vals, _ = (S.topk(1) if prior is not None else low.topk(1))
    return float(vals.max())


def transform(all_emb_bf16, i_to_ptr, j_to_ptr):
    """bbpr.py:526-550 without the dense matrix: user rows = all_emb[i_to_ptr], item rows = all_emb[j_to_ptr]."""
    i = torch.as_tensor(np.asarray(i_to_ptr), dtype=torch.long, device=all_emb_bf16.device)
    j = torch.as_tensor(np.asarray(j_to_ptr), dtype=torch.long, device=all_emb_bf16.device)
    return LowRankScore(all_emb_bf16[i], all_emb_bf16[j])


class BertBPR:
    """The transform half of the reference's ranker (src/ccrec/models/bbpr.py:328-550): same attribute names
    (item_titles, tokenizer, tokenizer_kw, model.item_tower, batch_size) and the same `transform(D)` contract --
    D is a rime-style dataset with `user_in_test["_hist_items"]`, `item_in_test.index` (and `prior_score`) --
    but the returned score is lazy low-rank, so `transform(D) + D.prior_score` -> `evaluate_item_rec(..., k)` never
    builds the [n_users, n_items] matrix.  fit() is the reference's Lightning loop and is out of scope.

    item_df: DataFrame with a TITLE column indexed by item id (bbpr.py:346); model: an object with .item_tower
    (or the tower itself)."""

    def __init__(self, item_df, model, tokenizer, max_length=None, batch_size=None, query_item_position_in_user_history=0):
        self.item_titles = item_df["TITLE"]
        self.model = model if hasattr(model, "item_tower") else _TowerHolder(model)
        self.tokenizer = tokenizer
        if max_length is None:
            max_length = int(os.environ.get("CCREC_MAX_LENGTH", 128))
        self.max_length = max_length
        self.tokenizer_kw = dict(truncation=True, padding="max_length", max_length=max_length, return_tensors="pt")  # bbpr.py:357-362
        self.batch_size = batch_size
        self.query_item_position = query_item_position_in_user_history

    def _pointers(self, D):
        """bbpr.py:287-291 (_DataModule): rows of the title table for every test user (its query item) and test item."""
        item_to_ptr = {k: ptr for ptr, k in enumerate(self.item_titles.index)}
        i_to_ptr = [item_to_ptr[hist[self.query_item_position]] for hist in D.user_in_test["_hist_items"]]
        j_to_ptr = [item_to_ptr[item] for item in D.item_in_test.index]
        return i_to_ptr, j_to_ptr

    def get_all_embeddings(self, model, batch_size, output_step="embedding"):
        """bbpr.py:466-483 -> packed bf16 [n_items, dim] on the device."""
        return get_all_embeddings(self.item_titles.tolist(), self.tokenizer, self.tokenizer_kw, model, batch_size, output_step)

    @torch.no_grad()
    def transform(self, D):
        """bbpr.py:494-550.  Branches kept: `oracle_dir` (qrels -> a sparse 0/1 score, :510-518), `random` (:520-521),
        the encoder path (:523-545) as a LowRankScore."""
        i_to_ptr, j_to_ptr = self._pointers(D)
        n_users, n_items = len(i_to_ptr), len(j_to_ptr)
        if hasattr(self, "oracle_dir"):
            ptr_list = self.item_titles.index.to_list()
            col_of = {p: c for c, p in enumerate(j_to_ptr)}
            rows, cols = [], []
            for step, user_index in enumerate(i_to_ptr):
                qid = ptr_list[user_index].split("_")[-1]
                for pid in self.oracle_dir[qid]:
                    c = col_of.get(ptr_list.index("p_{}".format(pid)))
                    if c is not None:
                        rows.append(step)
                        cols.append(c)
            return sps.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(n_users, n_items))
        if hasattr(self, "random"):
            return torch.rand(n_users, n_items)
        tower = self.model.item_tower
        tower.eval()
        batch = self.batch_size or 256 * max(1, torch.cuda.device_count())
        all_emb = self.get_all_embeddings(tower, batch, output_step="embedding")
        return transform(all_emb, i_to_ptr, j_to_ptr)


class _TowerHolder:
    def __init__(self, tower):
        self.item_tower = tower
