"""Tensor-level wrappers over the C ABI.  torch is plumbing here (device memory, streams); every
arithmetic step runs in libccr_hip.so.  All ops raise if there is no ROCm device."""
import contextlib
import ctypes

import torch

from . import _lib

_DTYPES = {torch.float32: _lib.DTYPE_F32, torch.float16: _lib.DTYPE_F16, torch.bfloat16: _lib.DTYPE_BF16}


def require_gpu():
    if not torch.cuda.is_available():
        raise _lib.CcrError("ccrec_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback")
    return _lib.load()


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


_NO_SWITCH = contextlib.nullcontext()


def _on(t):
    """Run the call with t's device current (HIP kernels launch on the current device).  The usual case -- it already
    is -- costs one integer compare instead of a device-guard round trip (the calls sit between a stream sync and the
    next kernel: host time here is GPU idle time)."""
    if t.device.index == torch.cuda.current_device():
        return _NO_SWITCH
    return torch.cuda.device(t.device)


def _check_bounds(norm_bounds, rows):
    assert norm_bounds.is_cuda and norm_bounds.dtype == torch.float32 and norm_bounds.is_contiguous()
    assert norm_bounds.dim() == 1 and norm_bounds.numel() >= rows, "norm_bounds: one fp32 per (destination) row"


def padded_dim(dim):
    """Width of the packed rows: embeddings of any width are stored zero-padded to a multiple of 8 (16-byte chunks)."""
    return (int(dim) + 7) // 8 * 8


def pack_bf16(x, normalize=False, out=None, return_norms=False, norm_bounds=None):
    """fp32 [rows, dim] (cuda) -> bf16 [rows, padded_dim(dim)]; normalize=True applies x / max(||x||, 1e-12) first
    (the cos_sim rule, ms_marco_eval.py:160-161).  `out` may be a slice of a preallocated shard.
    A width that is not a multiple of 8 (the reference takes any factor width, score_array.py:320-339) is zero-padded to the
    next one: scores, norms and cosines are unchanged, queries and corpus pad alike.
    norm_bounds: optional [rows] fp32 cuda tensor (the matching slice of a shard-sized one; no initialisation needed) that
    receives an upper bound of every packed row's norm; hand the whole array to CorpusIndex(norm_bounds=...) and the
    index build needs no pass over the shard (the search uses the bounds, per 256-row tile, in its filter margins)."""
    lib = require_gpu()
    assert x.is_cuda and x.dim() == 2, "pack_bf16 expects a 2-d cuda tensor"
    x = x.contiguous()
    if x.dtype != torch.float32:
        x = x.float()  # autocast may hand over fp16 encoder outputs (al_0_rank.py:125)
    rows, dim = x.shape
    pdim = padded_dim(dim)
    if out is None:
        out = torch.empty(rows, pdim, dtype=torch.bfloat16, device=x.device)
    assert out.is_contiguous() and out.dtype == torch.bfloat16 and tuple(out.shape) == (rows, pdim)
    norms = torch.empty(rows, dtype=torch.float32, device=x.device) if return_norms else None
    if norm_bounds is not None:
        _check_bounds(norm_bounds, rows)
    with _on(x):
        if pdim != dim:
            _lib.check(lib.ccr_pack_bf16_padded(_ptr(x), rows, dim, _ptr(out), pdim, _ptr(norms), _ptr(norm_bounds),
                                                int(bool(normalize)), _stream(x)), "ccr_pack_bf16_padded")
        else:
            _lib.check(lib.ccr_pack_bf16_ex(_ptr(x), _ptr(out), _ptr(norms), _ptr(norm_bounds), rows, dim, int(bool(normalize)),
                                            _stream(x)), "ccr_pack_bf16")
    return (out, norms) if return_norms else out


def meanpool_pack(hidden, mask, normalize=False, want_f32=True, want_bf16=True, out_bf16=None, out_f32=None, dst_rows=None,
                  norm_bounds=None):
    """Fused masked mean pooling (item_tower.py:141-146) + bf16 pack of [B, L, dim] hidden states.
    Returns (pooled_f32 or None, packed_bf16 or None).

    out_bf16 / out_f32: write into these [rows, dim] tensors (e.g. the resident shard) instead of fresh [B, dim] ones;
    dst_rows: [B] int64 destination rows inside them (length-sorted batches scatter back to corpus order);
    norm_bounds: fp32 cuda tensor indexed like the destination rows: bound of each packed row's norm (see pack_bf16)."""
    lib = require_gpu()
    assert hidden.is_cuda and hidden.dim() == 3 and hidden.dtype in _DTYPES
    hidden = hidden.contiguous()
    mask = mask.to(device=hidden.device, dtype=torch.int64).contiguous()
    B, L, dim = hidden.shape
    assert tuple(mask.shape) == (B, L)
    f32 = out_f32 if out_f32 is not None else (torch.empty(B, dim, dtype=torch.float32, device=hidden.device) if want_f32 else None)
    b16 = out_bf16 if out_bf16 is not None else (torch.empty(B, dim, dtype=torch.bfloat16, device=hidden.device) if want_bf16 else None)
    for t, dt in ((f32, torch.float32), (b16, torch.bfloat16)):
        if t is not None:
            assert t.is_cuda and t.dtype == dt and t.dim() == 2 and t.shape[1] == dim and t.is_contiguous()
            assert dst_rows is not None or t.shape[0] >= B
    if dst_rows is not None:
        dst_rows = dst_rows.to(device=hidden.device, dtype=torch.int64).contiguous()
        assert dst_rows.numel() == B
    if norm_bounds is not None:
        _check_bounds(norm_bounds, b16.shape[0] if (b16 is not None and dst_rows is not None) else B)
    with _on(hidden):
        _lib.check(lib.ccr_meanpool_pack_bf16_ex(_ptr(hidden), _DTYPES[hidden.dtype], _ptr(mask), _ptr(b16), _ptr(f32),
                                                 _ptr(dst_rows), _ptr(norm_bounds), B, L, dim, int(bool(normalize)),
                                                 _stream(hidden)), "ccr_meanpool_pack_bf16")
    return f32, b16


def meanpool_pack_packed(hidden, seq_start, seq_len, normalize=False, out_bf16=None, out_f32=None, dst_rows=None, norm_bounds=None):
    """meanpool_pack for a packed token array: hidden [T, dim], sequence s = rows seq_start[s] .. + seq_len[s] - 1 (int32 cuda
    tensors).  Writes into out_bf16 / out_f32 ([rows, dim], at dst_rows or rows 0 .. n_seq - 1); returns (out_f32, out_bf16)."""
    lib = require_gpu()
    assert hidden.is_cuda and hidden.dim() == 2 and hidden.dtype in _DTYPES and hidden.is_contiguous()
    n_seq, dim = seq_len.numel(), hidden.shape[1]
    for t in (seq_start, seq_len):
        assert t.is_cuda and t.dtype == torch.int32 and t.is_contiguous() and t.numel() == n_seq
    if out_bf16 is None and out_f32 is None:
        out_bf16 = torch.empty(n_seq, dim, dtype=torch.bfloat16, device=hidden.device)
    for t, dt in ((out_f32, torch.float32), (out_bf16, torch.bfloat16)):
        if t is not None:
            assert t.is_cuda and t.dtype == dt and t.dim() == 2 and t.shape[1] == dim and t.is_contiguous()
            assert dst_rows is not None or t.shape[0] >= n_seq
    if dst_rows is not None:
        dst_rows = dst_rows.to(device=hidden.device, dtype=torch.int64).contiguous()
        assert dst_rows.numel() == n_seq
    if norm_bounds is not None:
        _check_bounds(norm_bounds, out_bf16.shape[0] if (out_bf16 is not None and dst_rows is not None) else n_seq)
    with _on(hidden):
        _lib.check(lib.ccr_meanpool_pack_bf16_packed(_ptr(hidden), _DTYPES[hidden.dtype], _ptr(seq_start), _ptr(seq_len), _ptr(out_bf16),
                                                     _ptr(out_f32), _ptr(dst_rows), _ptr(norm_bounds), n_seq, dim, int(bool(normalize)),
                                                     _stream(hidden)), "ccr_meanpool_pack_bf16_packed")
    return out_f32, out_bf16


class _MeanPool(torch.autograd.Function):
    """Differentiable masked mean pooling (item_tower.py:141-146): forward = the fused pooling kernel (fp32 rows),
    backward = ccr_meanpool_bwd (grad / count broadcast over the unmasked tokens)."""

    @staticmethod
    def forward(ctx, hidden, mask):
        pooled, _ = meanpool_pack(hidden.detach(), mask, normalize=False, want_f32=True, want_bf16=False)
        ctx.save_for_backward(mask.to(device=hidden.device, dtype=torch.int64).contiguous())
        ctx.shape, ctx.dtype = tuple(hidden.shape), hidden.dtype
        return pooled

    @staticmethod
    def backward(ctx, grad):
        lib = require_gpu()
        (mask,) = ctx.saved_tensors
        B, L, dim = ctx.shape
        g = grad.detach().to(torch.float32).contiguous()
        dh = torch.empty(B, L, dim, dtype=ctx.dtype, device=g.device)
        with _on(g):
            _lib.check(lib.ccr_meanpool_bwd(_ptr(g), _ptr(mask), _ptr(dh), _DTYPES[ctx.dtype], B, L, dim, _stream(g)),
                       "ccr_meanpool_bwd")
        return dh, None


def meanpool(hidden, mask):
    """Masked mean pooling [B, L, dim] -> fp32 [B, dim] that autograd can differentiate (the training forward)."""
    return _MeanPool.apply(hidden, mask)


MAX_QUERIES_PER_SEARCH = 16384   # larger batches are searched in pieces (last_stats() then describes the last piece)


_HALF_DTYPES = {torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}


def _half_code(dtype):
    """CCR_DTYPE_* of one of the two 16-bit operand types of the encoder layer kernels."""
    assert dtype in _HALF_DTYPES, f"{dtype}: the encoder layer kernels take torch.bfloat16 or torch.float16 operands"
    return _HALF_DTYPES[dtype]


def attention(qkv, seq_start, seq_len, n_heads, max_len, pad_len=0, scale=0.125, out=None):
    """Multi-head self-attention of a token array (ccr_attention_half; head width 64): qkv [T, 3 * n_heads * 64] bf16 or fp16 = the
    stacked query | key | value projection of every token, seq_start / seq_len [n_seq] int32 = each sequence's first row and real tokens,
    max_len = the longest seq_len (host int, <= 512), pad_len = L for a right-padded [n_seq, L] batch (its padding rows get
    zeros) or 0 for a packed array.  -> context rows [T, n_heads * 64] of qkv's dtype."""
    lib = require_gpu()
    code = _half_code(qkv.dtype)
    assert qkv.is_cuda and qkv.dim() == 2 and qkv.is_contiguous()
    T, width = qkv.shape
    assert width == 3 * n_heads * 64, f"qkv rows are {width} wide, expected 3 x {n_heads} heads x 64"
    assert seq_start.dtype == torch.int32 and seq_len.dtype == torch.int32 and seq_start.is_cuda and seq_len.is_cuda
    assert seq_start.is_contiguous() and seq_len.is_contiguous() and seq_start.numel() == seq_len.numel()
    if out is None:
        out = torch.empty(T, n_heads * 64, dtype=qkv.dtype, device=qkv.device)
    assert out.is_cuda and out.dtype == qkv.dtype and tuple(out.shape) == (T, n_heads * 64) and out.is_contiguous()
    with _on(qkv):
        _lib.check(lib.ccr_attention_half(_ptr(qkv), _ptr(seq_start), _ptr(seq_len), _ptr(out), seq_len.numel(), int(n_heads),
                                          int(max_len), int(pad_len), float(scale), code, _stream(qkv)), "ccr_attention_half")
    return out


def embed_layernorm(word_table, position_table, type_table, token_ids, positions, token_types, gamma, beta, eps, dtype=torch.bfloat16):
    """LayerNorm((word_table[ids] + type_table[types]) + position_table[positions]) per token (ccr_embed_layernorm_half): fp32 tables
    [n, dim], int64 index vectors [T] (token_types may be None = type 0).  -> (fp32 [T, dim], its copy in `dtype` (bf16 / fp16) [T, dim])."""
    lib = require_gpu()
    code = _half_code(dtype)
    dim = word_table.shape[1]
    for t in (word_table, position_table, type_table):
        assert t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.shape[1] == dim and t.is_contiguous()
    for t in (gamma, beta):
        assert t.is_cuda and t.dtype == torch.float32 and tuple(t.shape) == (dim,) and t.is_contiguous()
    T = token_ids.numel()
    for t in (token_ids, positions) + ((token_types,) if token_types is not None else ()):
        assert t.is_cuda and t.dtype == torch.int64 and t.dim() == 1 and t.numel() == T and t.is_contiguous()
    f32 = torch.empty(T, dim, dtype=torch.float32, device=word_table.device)
    b16 = torch.empty(T, dim, dtype=dtype, device=word_table.device)
    with _on(word_table):
        _lib.check(lib.ccr_embed_layernorm_half(_ptr(word_table), word_table.shape[0], _ptr(position_table), position_table.shape[0],
                                                _ptr(type_table), type_table.shape[0], _ptr(token_ids), _ptr(positions),
                                                _ptr(token_types), _ptr(gamma), _ptr(beta), float(eps), _ptr(f32), _ptr(b16), T, dim,
                                                code, _stream(word_table)), "ccr_embed_layernorm_half")
    return f32, b16


def gelu_(x):
    """Exact (erf) GELU of a contiguous bf16 or fp16 cuda tensor, IN PLACE (ccr_gelu_half); returns x."""
    lib = require_gpu()
    code = _half_code(x.dtype)
    assert x.is_cuda and x.is_contiguous() and x.numel() % 8 == 0
    with _on(x):
        _lib.check(lib.ccr_gelu_half(_ptr(x), _ptr(x), x.numel(), code, _stream(x)), "ccr_gelu_half")
    return x


def add_layernorm(x, residual, gamma, beta, eps, want_f32=True, want_bf16=True):
    """LayerNorm(x + residual) * gamma + beta per row (ccr_add_layernorm_half): x [rows, dim] bf16 or fp16, residual [rows, dim] fp32 or
    None, gamma / beta [dim] fp32, dim a multiple of 256 (<= 2048).  -> (fp32 rows or None, their copy in x's dtype or None:
    want_bf16 asks for that 16-bit copy whichever of the two types x has)."""
    lib = require_gpu()
    code = _half_code(x.dtype)
    assert x.is_cuda and x.dim() == 2 and x.is_contiguous()
    rows, dim = x.shape
    if residual is not None:
        assert residual.is_cuda and residual.dtype == torch.float32 and tuple(residual.shape) == (rows, dim) and residual.is_contiguous()
    for t in (gamma, beta):
        assert t.is_cuda and t.dtype == torch.float32 and tuple(t.shape) == (dim,) and t.is_contiguous()
    assert want_f32 or want_bf16
    f32 = torch.empty(rows, dim, dtype=torch.float32, device=x.device) if want_f32 else None
    b16 = torch.empty(rows, dim, dtype=x.dtype, device=x.device) if want_bf16 else None
    with _on(x):
        _lib.check(lib.ccr_add_layernorm_half(_ptr(x), _ptr(residual), _ptr(gamma), _ptr(beta), float(eps), _ptr(f32), _ptr(b16),
                                              rows, dim, code, _stream(x)), "ccr_add_layernorm_half")
    return f32, b16


class CorpusIndex:
    """A resident bf16 corpus shard + its search state (ccr_index).  Build once per AL step, query many.

    corpus_bf16: [n_rows, dim] bf16 cuda tensor (kept alive by this object; the C index borrows it).
    global_row_offset: id of row 0 in the whole corpus (row-sharded multi-GPU search).
    norm_bounds: [n_rows] fp32 cuda tensor written by pack_bf16 / meanpool_pack(norm_bounds=...) for THESE rows (optional;
    kept alive by this object and not to be rewritten while the index is in use).
    workspace: optional uint8 cuda tensor to search in (not to be shared by two indices that are in use at the same time).
    """

    def __init__(self, corpus_bf16, global_row_offset=0, norm_bounds=None, workspace=None):
        self._lib = require_gpu()
        assert corpus_bf16.is_cuda and corpus_bf16.dtype == torch.bfloat16 and corpus_bf16.dim() == 2
        self.corpus = corpus_bf16.contiguous()
        self.n_rows, self.dim = self.corpus.shape
        self.offset = int(global_row_offset)
        self._h = ctypes.c_void_p()
        # workspace: a uint8 cuda tensor to adopt (e.g. the previous step's `index.workspace`: an index built per step then
        # allocates nothing in steady state); replaced by a larger one when a search needs more
        assert workspace is None or (workspace.is_cuda and workspace.dtype == torch.uint8 and workspace.device == corpus_bf16.device)
        self._ws = workspace
        self._ws_need = {}
        self._deferred = None
        with _on(self.corpus):
            if norm_bounds is None:
                _lib.check(self._lib.ccr_index_create(_ptr(self.corpus), self.n_rows, self.dim, self.offset,
                                                      _stream(self.corpus), ctypes.byref(self._h)), "ccr_index_create")
            else:  # row-norm bounds from ccr_pack_bf16_ex: no extra pass over the shard, no synchronisation
                _check_bounds(norm_bounds, self.n_rows)
                self._norm_bounds = norm_bounds   # borrowed by the C index, like the corpus
                _lib.check(self._lib.ccr_index_create_with_norms(_ptr(self.corpus), self.n_rows, self.dim, self.offset,
                                                                 _ptr(norm_bounds), _stream(self.corpus),
                                                                 ctypes.byref(self._h)), "ccr_index_create_with_norms")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            # a deferred search nobody finished: ccr_index_destroy waits for that search's own event before the index's
            # arrays go back to the block cache (the workspace tensor is released after this call returns)
            self._lib.ccr_index_destroy(h)

    @property
    def workspace(self):
        return self._ws

    def _grow_ws(self, need):
        """The workspace, at least `need` bytes.  A deferred search still owns the current one: complete it before the
        tensor is replaced (or handed to another search)."""
        if getattr(self, "_deferred", None) is not None:
            self.finish()
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.corpus.device)
        return self._ws

    def _workspace(self, n_q, k):
        need = self._ws_need.get((n_q, k))
        if need is None:   # the planner runs three times in there: once per (n_q, k) is enough
            need = self._ws_need[(n_q, k)] = int(self._lib.ccr_search_workspace_bytes(self._h, n_q, k))
        return self._grow_ws(need)

    def search(self, queries_bf16, k, flags=_lib.SEARCH_DEFAULT, out=None, defer=False):
        """-> (scores [n_q, k] fp32, ids [n_q, k] int64 global), canonical order.
        out=(scores, ids): contiguous [n_q, k] destinations (e.g. views of a packed all-gather message).
        defer=True: CCR_SEARCH_ASYNC -- the call returns without synchronising the stream; call finish() before trusting
        the result (it completes the -- rare -- flagged queries and fills last_stats())."""
        q = queries_bf16
        assert q.is_cuda and q.dtype == torch.bfloat16 and q.dim() == 2 and q.shape[1] == self.dim
        q = q.contiguous()
        n_q = q.shape[0]
        if out is None:
            scores = torch.empty(n_q, k, dtype=torch.float32, device=q.device)
            ids = torch.empty(n_q, k, dtype=torch.int64, device=q.device)
        else:
            scores, ids = out
            assert scores.shape == (n_q, k) and ids.shape == (n_q, k) and scores.is_contiguous() and ids.is_contiguous()
            assert scores.dtype == torch.float32 and ids.dtype == torch.int64 and scores.device == q.device == ids.device
        if n_q == 0:
            return scores, ids
        if n_q > MAX_QUERIES_PER_SEARCH:   # the workspace grows with the query count (~0.4 MB per query at k = 100): bounded batches
            assert not defer, "a deferred search takes at most MAX_QUERIES_PER_SEARCH queries"
            for lo in range(0, n_q, MAX_QUERIES_PER_SEARCH):
                hi = min(n_q, lo + MAX_QUERIES_PER_SEARCH)
                self.search(q[lo:hi], k, flags, out=(scores[lo:hi], ids[lo:hi]))
            return scores, ids
        ws = self._workspace(n_q, k)
        if defer:
            flags = int(flags) | _lib.SEARCH_ASYNC
            self._deferred = (q, scores, ids)   # keep the operands alive until finish()
        with _on(q):
            _lib.check(self._lib.ccr_search(self._h, _ptr(q), n_q, k, _ptr(scores), _ptr(ids), _ptr(ws), ws.numel(),
                                            int(flags), _stream(q)), "ccr_search")
        return scores, ids

    def finish(self):
        """Complete a deferred search: waits for THAT search's stream work (its own event -- not for work enqueued after it),
        fills last_stats(), and re-does the queries the search flagged (only then does it synchronise the stream)."""
        with _on(self.corpus):
            _lib.check(self._lib.ccr_search_finish(self._h), "ccr_search_finish")
        self._deferred = None

    def stream_wait_main_pass(self, stream):
        """ccr_search_stream_wait_main_pass: `stream` (a torch.cuda.Stream) waits for the main pass of this index's last search -- deferred
        or not --: work on it that does not touch the search's buffers then runs beside the search's select stage."""
        _lib.check(self._lib.ccr_search_stream_wait_main_pass(self._h, ctypes.c_void_p(stream.cuda_stream)), "ccr_search_stream_wait_main_pass")

    def search_shard(self, queries_bf16, k, message, defer=False, flags=_lib.SEARCH_DEFAULT):
        """ccr_search_shard: the search writes the packed shard message (header | scores | u32 local rows) that ONE all-gather
        moves (dist.ShardMessage.send).  `message`: uint8 cuda tensor of ccr_shard_message_bytes(n_q, k).  defer=True: no host
        synchronisation; the header's n_flagged is written on the stream, finish() completes the search."""
        q = queries_bf16
        assert q.is_cuda and q.dtype == torch.bfloat16 and q.dim() == 2 and q.shape[1] == self.dim
        q = q.contiguous()
        n_q = q.shape[0]
        assert 0 < n_q <= MAX_QUERIES_PER_SEARCH and 1 <= k <= self.n_rows
        assert message.is_cuda and message.dtype == torch.uint8 and message.is_contiguous() and message.device == q.device
        assert message.numel() >= shard_message_bytes(n_q, k)
        ws = self._workspace(n_q, k)
        if defer:
            flags = int(flags) | _lib.SEARCH_ASYNC
            self._deferred = (q, message)
        with _on(q):
            _lib.check(self._lib.ccr_search_shard(self._h, _ptr(q), n_q, k, _ptr(message), _ptr(ws), ws.numel(), int(flags),
                                                  _stream(q)), "ccr_search_shard")
        return message

    def _special_args(self, queries_bf16, ptr, idx):
        q = queries_bf16
        assert q.is_cuda and q.dtype == torch.bfloat16 and q.dim() == 2 and q.shape[1] == self.dim
        q = q.contiguous()
        ptr = torch.as_tensor(ptr, dtype=torch.int64).cpu().contiguous()
        assert ptr.numel() == q.shape[0] + 1
        idx = torch.as_tensor(idx, dtype=torch.int64).to(q.device).contiguous()
        assert idx.numel() == int(ptr[-1])
        return q, ptr, idx

    def _special_ws(self, need):
        return self._grow_ws(need)

    def search_blocked(self, queries_bf16, k, block_ptr, block_idx, flags=_lib.SEARCH_DEFAULT):
        """Search with per-query blocked GLOBAL ids (CSR: block_ptr [n_q + 1] on the host, block_idx ascending and unique
        inside a query; any length -- ms_marco_eval.py:224-227): blocked columns score -1e6, they are kept not removed.
        Ids outside this shard are ignored.  -> (scores [n_q, k], ids [n_q, k])."""
        q, ptr, idx = self._special_args(queries_bf16, block_ptr, block_idx)
        n_q = q.shape[0]
        scores = torch.empty(n_q, k, dtype=torch.float32, device=q.device)
        ids = torch.empty(n_q, k, dtype=torch.int64, device=q.device)
        if n_q == 0:
            return scores, ids
        if n_q > MAX_QUERIES_PER_SEARCH:
            for lo in range(0, n_q, MAX_QUERIES_PER_SEARCH):
                hi = min(n_q, lo + MAX_QUERIES_PER_SEARCH)
                a, b = int(ptr[lo]), int(ptr[hi])
                s_part, i_part = self.search_blocked(q[lo:hi], k, ptr[lo:hi + 1] - a, idx[a:b], flags)
                scores[lo:hi], ids[lo:hi] = s_part, i_part
            return scores, ids
        pp = ctypes.c_void_p(ptr.data_ptr())
        need = int(self._lib.ccr_search_blocked_workspace_bytes(self._h, n_q, k, pp))
        if need == 0:
            raise _lib.CcrError("ccr_search_blocked: " + self._lib.ccr_last_error().decode("utf-8", "replace"))
        ws = self._special_ws(need)
        with _on(q):
            _lib.check(self._lib.ccr_search_blocked(self._h, _ptr(q), n_q, k, pp, _ptr(idx), _ptr(scores), _ptr(ids), _ptr(ws),
                                                    ws.numel(), int(flags), _stream(q)), "ccr_search_blocked")
        return scores, ids

    def search_sparse_prior(self, queries_bf16, k, prior_ptr, prior_idx, prior_val, flags=_lib.SEARCH_DEFAULT):
        """Top-k of (low-rank score + sparse prior) (bbpr.py:592-595).  CSR prior over GLOBAL column ids (ascending, unique
        per row, <= 4096 per row), fp64 values.  -> (final scores [n_q, k] fp64, ids [n_q, k]) by (final desc, id asc)."""
        q, ptr, idx = self._special_args(queries_bf16, prior_ptr, prior_idx)
        val = torch.as_tensor(prior_val, dtype=torch.float64).to(q.device).contiguous()
        assert val.numel() == idx.numel()
        n_q = q.shape[0]
        scores = torch.empty(n_q, k, dtype=torch.float64, device=q.device)
        ids = torch.empty(n_q, k, dtype=torch.int64, device=q.device)
        if n_q == 0:
            return scores, ids
        if n_q > MAX_QUERIES_PER_SEARCH:
            for lo in range(0, n_q, MAX_QUERIES_PER_SEARCH):
                hi = min(n_q, lo + MAX_QUERIES_PER_SEARCH)
                a, b = int(ptr[lo]), int(ptr[hi])
                s_part, i_part = self.search_sparse_prior(q[lo:hi], k, ptr[lo:hi + 1] - a, idx[a:b], val[a:b], flags)
                scores[lo:hi], ids[lo:hi] = s_part, i_part
            return scores, ids
        pp = ctypes.c_void_p(ptr.data_ptr())
        need = int(self._lib.ccr_search_sparse_prior_workspace_bytes(self._h, n_q, k, pp))
        if need == 0:
            raise _lib.CcrError("ccr_search_sparse_prior: " + self._lib.ccr_last_error().decode("utf-8", "replace"))
        ws = self._special_ws(need)
        with _on(q):
            _lib.check(self._lib.ccr_search_sparse_prior(self._h, _ptr(q), n_q, k, pp, _ptr(idx), _ptr(val), _ptr(scores),
                                                         _ptr(ids), _ptr(ws), ws.numel(), int(flags), _stream(q)),
                       "ccr_search_sparse_prior")
        return scores, ids

    def last_stats(self):
        st = _lib.SearchStats()
        _lib.check(self._lib.ccr_search_last_stats(self._h, ctypes.byref(st)), "ccr_search_last_stats")
        return {f: getattr(st, f) for f, _ in st._fields_}

    def scores(self, queries_bf16, mode="canonical"):
        """Dense score matrix [n_q, n_rows] fp32 (ccr_scores).  mode "canonical": the fp64-ordered values the ranking
        reports; "mfma": the same bf16 rows through the MFMA tile kernel (fp32 accumulation)."""
        q = queries_bf16.contiguous()
        assert q.is_cuda and q.dtype == torch.bfloat16 and q.dim() == 2 and q.shape[1] == self.dim
        out = torch.empty(q.shape[0], self.n_rows, dtype=torch.float32, device=q.device)
        if q.shape[0] == 0:
            return out
        m = {"canonical": _lib.SCORES_CANONICAL, "mfma": _lib.SCORES_MFMA}[mode]
        with _on(q):
            _lib.check(self._lib.ccr_scores(self._h, _ptr(q), q.shape[0], m, _ptr(out), _stream(q)), "ccr_scores")
        return out


def _rank_strided(t):
    """[R, n_q, k] tensor whose [n_q, k] lists are dense; only the rank stride is free (a view of a gathered message)."""
    return t.dim() == 3 and t.stride(2) == 1 and t.stride(1) == t.shape[2] and t.stride(0) >= t.shape[1] * t.shape[2]


def merge_topk(scores, ids):
    """[R, n_q, k] per-shard canonical lists -> global ([n_q, k], [n_q, k]).  Rank-strided views are merged in place."""
    lib = require_gpu()
    assert scores.is_cuda and scores.dtype == torch.float32 and ids.dtype == torch.int64 and scores.shape == ids.shape
    if not _rank_strided(scores):
        scores = scores.contiguous()
    if not _rank_strided(ids):
        ids = ids.contiguous()
    R, n_q, k = scores.shape
    os_ = torch.empty(n_q, k, dtype=torch.float32, device=scores.device)
    oi = torch.empty(n_q, k, dtype=torch.int64, device=scores.device)
    with _on(scores):
        _lib.check(lib.ccr_merge_topk_strided(_ptr(scores), _ptr(ids), scores.stride(0), ids.stride(0), R, n_q, k, _ptr(os_),
                                              _ptr(oi), _stream(scores)), "ccr_merge_topk")
    return os_, oi


def shard_message_bytes(n_q, k):
    """Bytes of one packed shard message (ccr_shard_message_bytes): 32-byte header | fp32 scores | u32 local rows."""
    rows_at = (_lib.SHARD_HEADER_BYTES + n_q * k * 4 + 15) // 16 * 16
    return (rows_at + n_q * k * 4 + 15) // 16 * 16


def shard_message_fill(message, n_q, k, scores, ids, row_offset, n_rows):
    """Ordinary results ([n_q, k_valid] fp32 scores, int64 GLOBAL ids; k_valid <= k) -> packed shard message."""
    lib = require_gpu()
    k_valid = scores.shape[1] if scores is not None and scores.dim() == 2 else 0
    assert message.is_cuda and message.dtype == torch.uint8 and message.numel() >= shard_message_bytes(n_q, k)
    if k_valid:
        scores, ids = scores.contiguous(), ids.contiguous()
        assert scores.dtype == torch.float32 and ids.dtype == torch.int64 and scores.shape == ids.shape == (n_q, k_valid)
    with _on(message):
        _lib.check(lib.ccr_shard_message_fill(_ptr(message), n_q, k, k_valid, _ptr(scores) if k_valid else None,
                                              _ptr(ids) if k_valid else None, int(row_offset), int(n_rows), _stream(message)),
                   "ccr_shard_message_fill")
    return message


def merge_shard_messages(gathered, world, n_q, k):
    """R gathered shard messages (uint8 [R * stride]) -> global ([n_q, k] fp32, [n_q, k] int64 global ids)."""
    lib = require_gpu()
    assert gathered.is_cuda and gathered.dtype == torch.uint8 and gathered.is_contiguous() and gathered.numel() % world == 0
    stride = gathered.numel() // world
    os_ = torch.empty(n_q, k, dtype=torch.float32, device=gathered.device)
    oi = torch.empty(n_q, k, dtype=torch.int64, device=gathered.device)
    with _on(gathered):
        _lib.check(lib.ccr_merge_shard_messages(_ptr(gathered), stride, world, n_q, k, _ptr(os_), _ptr(oi), _stream(gathered)),
                   "ccr_merge_shard_messages")
    return os_, oi


SHORT_LIST_LDS_BYTES = 96 * 1024   # ccr_merge_short_lists merges a query's R lists in LDS: R * k_list * 12 bytes must fit


def merge_short_lists(gathered, world, n_q, k_list, k_out, out=None):
    """R gathered SHORT shard messages (k_list entries per query each) -> the k_out best of every query plus the verification of the
    shortcut (ccr_merge_short_lists): ([n_q, k_out] fp32, [n_q, k_out] int64 global ids, flags [n_q] int32 -- 1 where a shard's list
    was consumed to its end although the shard holds more rows: that query must be repeated with full lists --, n_flagged [1] int32)."""
    lib = require_gpu()
    assert gathered.is_cuda and gathered.dtype == torch.uint8 and gathered.is_contiguous() and gathered.numel() % world == 0
    assert 1 <= k_list <= k_out <= world * k_list and world * k_list * 12 <= SHORT_LIST_LDS_BYTES
    stride = gathered.numel() // world
    dev = gathered.device
    if out is not None:      # (scores, ids, zeroed flags, zeroed count) allocated by the caller (dist.ShardExchange: on another stream)
        os_, oi, flags, count = out
    else:
        os_ = torch.empty(n_q, k_out, dtype=torch.float32, device=dev)
        oi = torch.empty(n_q, k_out, dtype=torch.int64, device=dev)
        flags = torch.zeros(max(1, n_q), dtype=torch.int32, device=dev)[:n_q]
        count = torch.zeros(1, dtype=torch.int32, device=dev)
    with _on(gathered):
        _lib.check(lib.ccr_merge_short_lists(_ptr(gathered), stride, world, n_q, k_list, k_out, _ptr(os_), _ptr(oi), _ptr(flags) if n_q else _ptr(count),
                                             _ptr(count), _stream(gathered)), "ccr_merge_short_lists")
    return os_, oi, flags, count


def apply_block(scores, ids, block_ptr, block_idx, k_out, n_rows_total):
    """Post-filter an over-fetched canonical list with per-query blocked ids (ms_marco_eval.py:224-227)."""
    lib = require_gpu()
    n_q, k_in = scores.shape
    os_ = torch.empty(n_q, k_out, dtype=torch.float32, device=scores.device)
    oi = torch.empty(n_q, k_out, dtype=torch.int64, device=scores.device)
    block_ptr = block_ptr.to(device=scores.device, dtype=torch.int64).contiguous()
    block_idx = block_idx.to(device=scores.device, dtype=torch.int64).contiguous()
    if block_idx.numel() == 0:
        block_idx = torch.zeros(1, dtype=torch.int64, device=scores.device)
    with _on(scores):
        _lib.check(lib.ccr_apply_block(_ptr(scores.contiguous()), _ptr(ids.contiguous()), n_q, k_in, _ptr(block_ptr),
                                       _ptr(block_idx), int(n_rows_total), _ptr(os_), _ptr(oi), k_out, _stream(scores)),
                   "ccr_apply_block")
    return os_, oi


def colsum_bf16(x):
    """Column sums of a bf16 [rows, dim] cuda matrix in fp64 (ccr_colsum_bf16) -> [dim] float64."""
    lib = require_gpu()
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2
    x = x.contiguous()
    out = torch.empty(x.shape[1], dtype=torch.float64, device=x.device)
    with _on(x):
        _lib.check(lib.ccr_colsum_bf16(_ptr(x), x.shape[0], x.shape[1], _ptr(out), _stream(x)), "ccr_colsum_bf16")
    return out


_INBATCH_LAYOUT = {}   # (B, dim) -> (bytes of the packed blocks padded to 256, workspace bytes, total bytes)


class _InBatchCE(torch.autograd.Function):
    """loss = CE([Q P^T | Q N^T] * inv_T, arange(B)).mean()   (bbpr.py:205-212), bf16 operands, fp32 accumulate.

    The step is ~0.12 ms of kernels, so the host path is kept to a handful of Python operations: ONE allocation per forward holds the
    packed bf16 operands, the library's workspace and lse (addresses by integer arithmetic, no tensor views), ONE library call rounds
    the three fp32 blocks to bf16 (torch's .to(bfloat16) bits) and runs the forward; the backward is one allocation and one call.
    The buffer belongs to THIS forward until its backward has run (the backward reads the forward's logits there; the library checks
    the workspace's stamp on the device)."""

    @staticmethod
    def forward(ctx, q, p, n, inv_temperature):
        lib = require_gpu()
        B, dim = q.shape
        assert q.is_cuda and p.shape == q.shape == n.shape and p.device == q.device == n.device, "three [B, dim] blocks on one device"
        lay = _INBATCH_LAYOUT.get((B, dim))
        if lay is None:
            ws_bytes = int(lib.ccr_inbatch_ce_workspace_bytes(B, dim))
            head = (3 * B * dim * 2 + 255) // 256 * 256
            lay = _INBATCH_LAYOUT[(B, dim)] = (head, ws_bytes, head + (ws_bytes + 255) // 256 * 256 + B * 4)
        head, ws_bytes, total = lay
        dev = q.device
        buf = torch.empty(total, dtype=torch.uint8, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        base = buf.data_ptr()
        lse_ptr = base + total - B * 4
        stream = torch.cuda.current_stream(dev).cuda_stream
        fast = (q.dtype == p.dtype == n.dtype == torch.float32 and q.is_contiguous() and p.is_contiguous() and n.is_contiguous()
                and dim % 8 == 0 and (q.data_ptr() | p.data_ptr() | n.data_ptr()) % 16 == 0)
        with _on(q):
            if fast:
                _lib.check(lib.ccr_inbatch_ce_fwd_f32(q.data_ptr(), p.data_ptr(), n.data_ptr(), B, dim, float(inv_temperature), base,
                                                      loss.data_ptr(), lse_ptr, base + head, ws_bytes, stream), "ccr_inbatch_ce_fwd_f32")
            else:   # other dtypes / strided inputs: torch copies round to bf16, then the forward on the packed blocks
                packed = buf[:3 * B * dim * 2].view(torch.bfloat16).view(3, B, dim)
                for dst, t in zip(packed, (q, p, n)):
                    dst.copy_(t.detach())
                blk = B * dim * 2
                _lib.check(lib.ccr_inbatch_ce_fwd(base, base + blk, base + 2 * blk, B, dim, float(inv_temperature), loss.data_ptr(), lse_ptr,
                                                  base + head, ws_bytes, stream), "ccr_inbatch_ce_fwd")
        ctx.save_for_backward(buf)
        ctx.lay = (B, dim, head, ws_bytes, total, float(inv_temperature), q.dtype, p.dtype, n.dtype)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        lib = require_gpu()
        (buf,) = ctx.saved_tensors
        B, dim, head, ws_bytes, total, inv_t, tq, tp, tn = ctx.lay
        dev = buf.device
        grads = torch.empty(3, B, dim, dtype=torch.float32, device=dev)
        g = grad_out
        if g.dtype != torch.float32 or g.device != dev or not g.is_contiguous():
            g = g.detach().to(device=dev, dtype=torch.float32).contiguous()   # stays on the device either way
        base, gp, blk = buf.data_ptr(), grads.data_ptr(), B * dim
        with _on(buf):
            _lib.check(lib.ccr_inbatch_ce_bwd_dev(base, base + 2 * blk, base + 4 * blk, base + total - B * 4, B, dim, inv_t, g.data_ptr(),
                                                  gp, gp + 4 * blk, gp + 8 * blk, base + head, ws_bytes, torch.cuda.current_stream(dev).cuda_stream),
                       "ccr_inbatch_ce_bwd")
        dq, dp, dn = grads.unbind(0)
        f32 = torch.float32
        return (dq if tq == f32 else dq.to(tq)), (dp if tp == f32 else dp.to(tp)), (dn if tn == f32 else dn.to(tn)), None


def inbatch_ce(q, p, n, inv_temperature):
    return _InBatchCE.apply(q, p, n, inv_temperature)
