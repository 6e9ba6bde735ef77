"""ctypes binding of libccr_hip.so (include/ccr_retrieval.h).  No CPU fallback: if the library or a
ROCm device is missing, every op raises -- the product path never routes through oracle/ or torch math."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libccr_hip.so")

CCR_OK = 0
CCR_ERR_INVALID, CCR_ERR_HIP, CCR_ERR_WORKSPACE, CCR_ERR_BLOCK_ID = -1, -2, -3, -4
DTYPE_F32, DTYPE_F16, DTYPE_BF16 = 0, 1, 2
SEARCH_DEFAULT, SEARCH_FORCE_DENSE, SEARCH_FORCE_FUSED, SEARCH_ASYNC = 0, 1, 2, 4
SCORES_CANONICAL, SCORES_MFMA = 0, 1

EXPORTS = [
    "ccr_last_error", "ccr_version", "ccr_pack_bf16", "ccr_pack_bf16_ex", "ccr_meanpool_pack_bf16", "ccr_meanpool_pack_bf16_ex", "ccr_index_create",
    "ccr_index_create_with_norms", "ccr_index_destroy",
    "ccr_index_rows", "ccr_index_dim", "ccr_search_workspace_bytes", "ccr_search", "ccr_search_last_stats",
    "ccr_merge_topk", "ccr_merge_topk_strided", "ccr_apply_block", "ccr_inbatch_ce_workspace_bytes", "ccr_inbatch_ce_fwd", "ccr_inbatch_ce_bwd", "ccr_inbatch_ce_bwd_dev", "ccr_rank_metrics",
    "ccr_search_finish", "ccr_scores", "ccr_search_blocked_workspace_bytes", "ccr_search_blocked",
    "ccr_search_sparse_prior_workspace_bytes", "ccr_search_sparse_prior", "ccr_colsum_bf16", "ccr_meanpool_bwd", "ccr_bm25_index_create", "ccr_bm25_index_destroy", "ccr_bm25_search_workspace_bytes",
    "ccr_bm25_search", "ccr_pack_bf16_padded", "ccr_shard_message_bytes", "ccr_search_shard", "ccr_shard_message_fill", "ccr_merge_shard_messages",
    "ccr_attention_bf16", "ccr_add_layernorm", "ccr_meanpool_pack_bf16_packed", "ccr_embed_layernorm", "ccr_gelu_bf16",
    "ccr_attention_half", "ccr_add_layernorm_half", "ccr_embed_layernorm_half", "ccr_gelu_half", "ccr_merge_short_lists",
    "ccr_bm25_search_workspace_bytes_k", "ccr_bm25_search_last_stats", "ccr_bm25_index_set_idf", "ccr_inbatch_pack3_bf16", "ccr_inbatch_ce_fwd_f32",
    "ccr_search_stream_wait_main_pass",
]

SHARD_HEADER_BYTES = 32
SHARD_MAGIC = 0x4D524343


class ShardHeader(ctypes.Structure):
    """ccr_shard_header (include/ccr_retrieval.h)."""
    _fields_ = [("magic", ctypes.c_uint32), ("n_flagged", ctypes.c_uint32), ("k_valid", ctypes.c_uint32),
                ("n_covered", ctypes.c_uint32), ("row_offset", ctypes.c_int64), ("n_rows", ctypes.c_int64)]


class SearchStats(ctypes.Structure):
    _fields_ = [("path", ctypes.c_int32), ("n_fallback", ctypes.c_int32), ("sample_tiles", ctypes.c_int32),
                ("ranges", ctypes.c_int32), ("cap", ctypes.c_int32), ("sublists", ctypes.c_int32),
                ("n_candidates", ctypes.c_int64), ("ms_sample", ctypes.c_float), ("ms_threshold", ctypes.c_float),
                ("ms_main", ctypes.c_float), ("ms_select", ctypes.c_float), ("ms_fallback", ctypes.c_float),
                ("ms_total", ctypes.c_float), ("n_retried", ctypes.c_int32), ("n_dense", ctypes.c_int32),
                ("main_launches", ctypes.c_int32), ("opt_rank", ctypes.c_int32), ("main_tile_queries", ctypes.c_int32),
                ("reserved0", ctypes.c_int32)]


class CcrError(RuntimeError):
    pass


_lib = None


def load():
    """dlopen the in-tree library and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise CcrError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       f"(or make -C crowd-coachable-recommendations_amd/csrc)")
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, f32, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t
    lib.ccr_last_error.restype = ctypes.c_char_p
    lib.ccr_last_error.argtypes = []
    lib.ccr_version.restype = i32
    lib.ccr_pack_bf16.argtypes = [vp, vp, vp, i64, i32, i32, vp]
    lib.ccr_pack_bf16_ex.argtypes = [vp, vp, vp, vp, i64, i32, i32, vp]
    lib.ccr_index_create_with_norms.argtypes = [vp, i64, i32, i64, vp, vp, ctypes.POINTER(vp)]
    lib.ccr_meanpool_pack_bf16.argtypes = [vp, i32, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.ccr_meanpool_pack_bf16_ex.argtypes = [vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.ccr_index_create.argtypes = [vp, i64, i32, i64, vp, ctypes.POINTER(vp)]
    lib.ccr_index_destroy.argtypes = [vp]
    lib.ccr_index_rows.argtypes = [vp]
    lib.ccr_index_rows.restype = i64
    lib.ccr_index_dim.argtypes = [vp]
    lib.ccr_search_workspace_bytes.argtypes = [vp, i32, i32]
    lib.ccr_search_workspace_bytes.restype = sz
    lib.ccr_search.argtypes = [vp, vp, i32, i32, vp, vp, vp, sz, i32, vp]
    lib.ccr_search_last_stats.argtypes = [vp, ctypes.POINTER(SearchStats)]
    lib.ccr_merge_topk.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
    lib.ccr_merge_topk_strided.argtypes = [vp, vp, i64, i64, i32, i32, i32, vp, vp, vp]
    lib.ccr_apply_block.argtypes = [vp, vp, i32, i32, vp, vp, i64, vp, vp, i32, vp]
    lib.ccr_inbatch_ce_workspace_bytes.argtypes = [i32, i32]
    lib.ccr_inbatch_ce_workspace_bytes.restype = sz
    lib.ccr_inbatch_ce_fwd.argtypes = [vp, vp, vp, i32, i32, f32, vp, vp, vp, sz, vp]
    lib.ccr_inbatch_ce_bwd.argtypes = [vp, vp, vp, vp, i32, i32, f32, f32, vp, vp, vp, vp, sz, vp]
    lib.ccr_inbatch_ce_bwd_dev.argtypes = [vp, vp, vp, vp, i32, i32, f32, vp, vp, vp, vp, vp, sz, vp]
    lib.ccr_inbatch_pack3_bf16.argtypes = [vp, vp, vp, i32, i32, vp, vp]
    lib.ccr_inbatch_ce_fwd_f32.argtypes = [vp, vp, vp, i32, i32, f32, vp, vp, vp, vp, sz, vp]
    lib.ccr_rank_metrics.argtypes = [vp, i32, i32, vp, vp, vp, i32, vp, vp, vp]
    lib.ccr_search_finish.argtypes = [vp]
    lib.ccr_search_stream_wait_main_pass.argtypes = [vp, vp]
    lib.ccr_scores.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.ccr_search_blocked_workspace_bytes.argtypes = [vp, i32, i32, vp]
    lib.ccr_search_blocked_workspace_bytes.restype = sz
    lib.ccr_search_blocked.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, vp, sz, i32, vp]
    lib.ccr_search_sparse_prior_workspace_bytes.argtypes = [vp, i32, i32, vp]
    lib.ccr_search_sparse_prior_workspace_bytes.restype = sz
    lib.ccr_search_sparse_prior.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, sz, i32, vp]
    lib.ccr_colsum_bf16.argtypes = [vp, i64, i32, vp, vp]
    lib.ccr_meanpool_bwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
    lib.ccr_bm25_index_create.argtypes = [vp, vp, vp, vp, i64, i64, ctypes.c_double, ctypes.POINTER(vp)]
    lib.ccr_bm25_index_destroy.argtypes = [vp]
    lib.ccr_bm25_search_workspace_bytes.argtypes = [vp, i32, i32]
    lib.ccr_bm25_search_workspace_bytes.restype = sz
    lib.ccr_bm25_search_workspace_bytes_k.argtypes = [vp, i32, i32, i32]
    lib.ccr_bm25_search_workspace_bytes_k.restype = sz
    lib.ccr_bm25_search_last_stats.argtypes = [vp, vp]
    lib.ccr_bm25_index_set_idf.argtypes = [vp, vp, vp, vp]
    lib.ccr_bm25_search.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, vp, sz, vp]
    lib.ccr_pack_bf16_padded.argtypes = [vp, i64, i32, vp, i32, vp, vp, i32, vp]
    lib.ccr_attention_bf16.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, ctypes.c_float, vp]
    lib.ccr_add_layernorm.argtypes = [vp, vp, vp, vp, ctypes.c_float, vp, vp, i64, i32, vp]
    lib.ccr_embed_layernorm.argtypes = [vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, ctypes.c_float, vp, vp, i64, i32, vp]
    lib.ccr_gelu_bf16.argtypes = [vp, vp, i64, vp]
    lib.ccr_attention_half.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, ctypes.c_float, i32, vp]
    lib.ccr_add_layernorm_half.argtypes = [vp, vp, vp, vp, ctypes.c_float, vp, vp, i64, i32, i32, vp]
    lib.ccr_embed_layernorm_half.argtypes = [vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, ctypes.c_float, vp, vp, i64, i32, i32, vp]
    lib.ccr_gelu_half.argtypes = [vp, vp, i64, i32, vp]
    lib.ccr_meanpool_pack_bf16_packed.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]
    lib.ccr_shard_message_bytes.argtypes = [i32, i32]
    lib.ccr_shard_message_bytes.restype = sz
    lib.ccr_search_shard.argtypes = [vp, vp, i32, i32, vp, vp, sz, i32, vp]
    lib.ccr_shard_message_fill.argtypes = [vp, i32, i32, i32, vp, vp, i64, i64, vp]
    lib.ccr_merge_shard_messages.argtypes = [vp, i64, i32, i32, i32, vp, vp, vp]
    lib.ccr_merge_short_lists.argtypes = [vp, i64, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if fn.restype is ctypes.c_int and name not in ("ccr_version", "ccr_index_dim"):
            fn.restype = i32
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != CCR_OK:
        msg = load().ccr_last_error().decode("utf-8", "replace")
        if rc == CCR_ERR_BLOCK_ID:
            raise AssertionError("block id not found")
        raise CcrError(f"{what} failed ({rc}): {msg}")
