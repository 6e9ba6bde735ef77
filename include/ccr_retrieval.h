/*
 * ccr_retrieval.h -- C ABI of the MI355X-native retrieval hot path
 *                    (encode output -> bf16 pack -> exhaustive Q.D^T -> top-k -> shard merge).
 *
 * The reference (awslabs/crowd-coachable-recommendations) has no FFI for this path: it is plain
 * Python over torch ops.  Each entry point below names the reference code it replaces
 * (file:line under the reference tree); INTEGRATION.md shows the ctypes stub a maintainer adds.
 *
 * Conventions
 *   - plain pointers and sizes only; every data pointer is DEVICE memory unless marked host;
 *   - the caller owns every buffer; an index BORROWS the corpus pointer it was created on;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - return value: CCR_OK (0) or a negative CCR_ERR_* code; ccr_last_error() gives the text
 *     (thread-local);
 *   - embeddings are bf16 bit patterns (uint16_t), row-major [rows][dim];
 *   - canonical arithmetic: score(q,d) = (float) sum_{i ascending} (double)q_i*(double)d_i,
 *     rank order = score descending, id ascending on equal scores.  Results of ccr_search are
 *     exactly that ordering for ANY input (near-ties and exact ties included).
 */
#ifndef CCR_RETRIEVAL_H
#define CCR_RETRIEVAL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCR_OK 0
#define CCR_ERR_INVALID (-1)     /* bad argument (null pointer, dim not supported, k > rows, ...) */
#define CCR_ERR_HIP (-2)         /* a HIP runtime call failed */
#define CCR_ERR_WORKSPACE (-3)   /* workspace too small: see ccr_search_workspace_bytes */
#define CCR_ERR_BLOCK_ID (-4)    /* a blocked id is outside the corpus ("block id not found") */

#define CCR_DTYPE_F32 0
#define CCR_DTYPE_F16 1
#define CCR_DTYPE_BF16 2

/* ccr_search flags */
#define CCR_SEARCH_DEFAULT 0
#define CCR_SEARCH_FORCE_DENSE 1   /* fp64 brute-force path for every query (tests; the default path of a tiny corpus scores on the matrix
                                      cores and re-scores only the rows inside the error margin in fp64 -- same result) */
#define CCR_SEARCH_FORCE_FUSED 2   /* MFMA fused path even where the planner would pick dense */
#define CCR_SEARCH_ASYNC 4         /* do not synchronise: flagged queries (rare) are completed by ccr_search_finish() */

/* ccr_scores modes */
#define CCR_SCORES_CANONICAL 0     /* fp64-ordered canonical scores (bit-identical to the ranking's scores) */
#define CCR_SCORES_MFMA 1          /* the same bf16 rows through the MFMA tile kernel (fp32 accumulation, |err| < 1e-6) */

typedef struct ccr_index ccr_index;

/* statistics of the last ccr_search on an index (host struct, filled after the call returns) */
typedef struct ccr_search_stats {
    int32_t path;              /* 0 = dense exact, 1 = fused MFMA */
    int32_t n_fallback;        /* queries the first fused attempt flagged (candidate overflow / mass ties) */
    int32_t sample_tiles;      /* 256-row corpus tiles scored by the threshold (sample) pass */
    int32_t ranges;            /* corpus ranges of the main pass */
    int32_t cap;               /* candidate slots per sub-list */
    int32_t sublists;          /* sub-lists per (range, query): 8 = main pass on v_mfma_f32_16x16x32_bf16, 4 = 32x32x16 */
    int64_t n_candidates;      /* total stage-1 survivors over all queries */
    /* device time of each phase of the last search, from HIP events on the search stream (ms) */
    float ms_sample;           /* sample pass GEMM (group maxima) */
    float ms_threshold;        /* query norms + threshold select */
    float ms_main;             /* main pass GEMM + filter (the dominant kernel) */
    float ms_select;           /* stage-2 select + canonical re-score */
    float ms_fallback;         /* retry pass + dense path for flagged queries (0 if none) */
    float ms_total;
    int32_t n_retried;         /* flagged queries re-done by the fused retry pass (thresholds re-tightened from their own lists) */
    int32_t n_dense;           /* flagged queries finished by an exact whole-row path (margin select or fp64: mass ties, flagged again,
                                  on-stream chunk); path 0: queries the margin select left to the fp64 path */
    int32_t main_launches;     /* launches of the main-pass kernel in this search (phases: thresholds are re-tightened between them) */
    int32_t opt_rank;          /* > 0: the thresholds were ESTIMATED (the opt_rank-th largest sampled group maximum; one launch, verified by
                                  the select stage); 0: conservative lower bounds with re-tightening */
    int32_t main_tile_queries; /* queries per tile of the main pass: 256, or 384 (the 256 x 384 form, gemm_topk16w_kernel); 0 on the
                                  dense path and for the streaming pass of small batches */
    int32_t reserved0;
} ccr_search_stats;

const char *ccr_last_error(void);
int ccr_version(void);

/*
 * fp32 -> bf16 pack of encoder outputs, optionally L2-normalising each row first.
 * Replaces: the fp32 host copy + vstack of scripts/ms_marco_eval.py:141-149 (embeddings stay on
 * device as a packed bf16 shard) and, with normalize=1, the F.normalize calls of cos_sim
 * (scripts/ms_marco_eval.py:160-161; twin src/ccrec/models/bbpr.py:490-491).
 *   src   [rows][dim] fp32          dst  [rows][dim] bf16 (RNE; NaN stays NaN)
 *   norms [rows] fp32 or NULL: L2 norm of the fp32 row (before normalisation)
 * dim % 4 == 0.  normalize: y = x / max(||x||, 1e-12), fixed reduction order (oracle/ccr_oracle.c).
 */
int ccr_pack_bf16(const float *src, uint16_t *dst, float *norms, int64_t rows, int dim, int normalize, void *stream);
/* Same, plus row_norm_bounds ([rows] device floats, may be NULL): entry r receives an upper bound of the L2 norm of the
 * PACKED row r (one plain store per row: no initialisation needed, batches write disjoint slices of one shard-sized
 * array).  Passing the array on to ccr_index_create_with_norms saves the index build's own pass over the shard; the
 * index uses the bounds only inside its filter margins. */
int ccr_pack_bf16_ex(const float *src, uint16_t *dst, float *norms, float *row_norm_bounds, int64_t rows, int dim, int normalize,
                     void *stream);

/* Embeddings of ANY width (the reference takes any factor width: src/rime_lite/util/score_array.py:320-339,
 * src/ccrec/models/bbpr.py:536-540): rows of src_dim floats are packed into rows of dst_dim = src_dim rounded up to a multiple
 * of 8, the tail zero-filled -- inner products, norms and cosines are unchanged, and every search path takes the padded width
 * (the fused kernels zero-fill their last 32-element K step for widths that are not multiples of 32).
 *   norms / row_norm_bounds: as ccr_pack_bf16_ex.  The canonical sum of squares extends to any width by zero padding. */
int ccr_pack_bf16_padded(const float *src, int64_t rows, int src_dim, uint16_t *dst, int dst_dim, float *norms,
                         float *row_norm_bounds, int normalize, void *stream);

/*
 * Fused masked mean pooling (+ optional normalise) + pack of the encoder's last hidden state.
 * Replaces: src/ccrec/models/item_tower.py:137-147 (masked_fill, sum(dim=1), divide) followed by the pack.
 *   hidden [B][L][dim] of hidden_dtype (CCR_DTYPE_*), mask [B][L] int64 (0/1)
 *   dst_bf16 [B][dim] or NULL, dst_f32 [B][dim] or NULL (the un-rounded fp32 pooled rows)
 */
int ccr_meanpool_pack_bf16(const void *hidden, int hidden_dtype, const int64_t *mask, uint16_t *dst_bf16,
                           float *dst_f32, int B, int L, int dim, int normalize, void *stream);
/* Same for length-sorted (variable-length) encoder batches that write straight into the resident shard:
 *   dst_rows [B] int64 or NULL: pooled row b goes to row dst_rows[b] of dst_bf16 / dst_f32 (NULL = row b);
 *   row_norm_bounds (device floats, indexed like the destination rows) or NULL: a bound of each packed row's L2 norm,
 *   as ccr_pack_bf16_ex.
 * The pooled value does not depend on L (masked positions are skipped, real tokens are summed in order), so a batch
 * padded to its own longest text gives the same bits as the reference's fixed max_length padding (item_tower.py:29,
 * tokenizer_kw) whenever the encoder's hidden states for the real tokens are the same. */
int ccr_meanpool_pack_bf16_ex(const void *hidden, int hidden_dtype, const int64_t *mask, uint16_t *dst_bf16,
                              float *dst_f32, const int64_t *dst_rows, float *row_norm_bounds, int B, int L, int dim,
                              int normalize, void *stream);

/* Same for a PACKED token array (the kernel forward of the encoder layers runs on the real tokens only: no padding rows exist):
 *   hidden [T][dim]; sequence s is rows seq_start[s] .. seq_start[s] + seq_len[s] - 1 (int32 device arrays, as ccr_attention_bf16).
 * The sum runs over a sequence's tokens in order -- the same bits as the padded form gives for the same hidden states. */
int ccr_meanpool_pack_bf16_packed(const void *hidden, int hidden_dtype, const int32_t *seq_start, const int32_t *seq_len,
                                  uint16_t *dst_bf16, float *dst_f32, const int64_t *dst_rows, float *row_norm_bounds, int n_seq,
                                  int dim, int normalize, void *stream);

/* Backward of the pooling for the training forward (the reference's tower is called with gradients on in
 * src/ccrec/models/bbpr.py:130-141,195-197): dhidden[b][l][:] = mask[b][l] ? grad[b][:] / sum(mask[b]) : 0.
 *   grad [B][dim] fp32 (gradient w.r.t. the un-normalised pooled rows), dhidden [B][L][dim] of hidden_dtype. */
int ccr_meanpool_bwd(const float *grad, const int64_t *mask, void *dhidden, int hidden_dtype, int B, int L, int dim,
                     void *stream);

/*
 * The non-GEMM pieces of the encoder layer that produces `hidden` (the reference runs transformers' BertModel under autocast:
 * src/ccrec/models/item_tower.py:122 `cls_model(**inputs).last_hidden_state`, scripts/al_0_rank.py:92-101,125).  The
 * projections stay library GEMMs on the host side (hipBLASLt through torch); these two kernels replace the attention call
 * and the residual-add / LayerNorm / cast passes between them.  Arithmetic = the layer's under autocast(bf16): bf16 operands,
 * fp32 scores, softmax and accumulation, probabilities rounded to bf16 for the P.V product, fp32 residual sum and LayerNorm.
 *
 * ccr_attention_bf16: multi-head self-attention over n_seq sequences of a token array, head width 64.
 *   qkv [T][3 * n_heads * 64] bf16: row t = (Q | K | V) of token t, each n_heads * 64 wide -- the output of ONE projection with
 *     the query / key / value weights stacked (transformers' BertSelfAttention.query / .key / .value);
 *   seq_start [n_seq] int32: first row of sequence s in the token array; seq_len [n_seq] int32 (0 .. max_len; device memory, so the
 *     kernel itself cuts a larger entry to max_len and treats a negative one as 0): its real tokens
 *     (keys beyond are masked, as attention_mask = 0 is in the reference's inputs, scripts/al_0_rank.py:76-81 padding=True);
 *   out [T][n_heads * 64] bf16: context rows, heads side by side (the operand of BertSelfOutput.dense);
 *   max_len: longest seq_len (<= 512; sizes the LDS image of one head's keys and values);
 *   pad_len: 0 for a packed token array, or the padded length L of a right-padded [n_seq][L] batch (seq_start[s] = s * L):
 *     rows seq_len[s] .. pad_len - 1 of a sequence receive zeros;
 *   scale: 1 / sqrt(64) for BERT.
 * ccr_add_layernorm: y = LayerNorm(x + residual) * gamma + beta per row (BertSelfOutput / BertOutput: LayerNorm(dense(h) + input)).
 *   x_bf16 [rows][dim] bf16, residual [rows][dim] fp32 or NULL, gamma / beta [dim] fp32, dim % 256 == 0, dim <= 2048;
 *   out_f32 [rows][dim] or NULL (the residual stream), out_bf16 [rows][dim] or NULL (the next projection's operand).
 */
/* ccr_embed_layernorm: the embedding block in front of the layers (transformers BertEmbeddings.forward): row r =
 * LayerNorm((word_table[token_ids[r]] + type_table[token_types[r]]) + position_table[positions[r]]) * gamma + beta, fp32 tables
 * [n][dim], int64 indices (token_types NULL = type 0; an index outside its table is clamped), outputs as ccr_add_layernorm. */
int ccr_embed_layernorm(const float *word_table, int64_t vocab, const float *position_table, int64_t n_positions,
                        const float *type_table, int64_t n_types, const int64_t *token_ids, const int64_t *positions,
                        const int64_t *token_types, const float *gamma, const float *beta, float eps, float *out_f32,
                        uint16_t *out_bf16, int64_t rows, int dim, void *stream);
/* ccr_gelu_bf16: y = GELU(x) elementwise on n bf16 values (n % 8 == 0, 16-byte aligned; y may be x), the exact erf form of
 * transformers' BertIntermediate (hidden_act "gelu") in fp32, rounded to bf16 once. */
int ccr_gelu_bf16(const uint16_t *x, uint16_t *y, int64_t n, void *stream);
int ccr_attention_bf16(const uint16_t *qkv, const int32_t *seq_start, const int32_t *seq_len, uint16_t *out, int n_seq,
                       int n_heads, int max_len, int pad_len, float scale, void *stream);
int ccr_add_layernorm(const uint16_t *x_bf16, const float *residual, const float *gamma, const float *beta, float eps,
                      float *out_f32, uint16_t *out_bf16, int64_t rows, int dim, void *stream);
/* The same four kernels on either 16-bit operand type: half_dtype = CCR_DTYPE_BF16 or CCR_DTYPE_F16, whichever the caller's autocast
 * context names.  The reference encodes under torch.cuda.amp.autocast() (scripts/al_0_rank.py:8,125), whose CUDA default is fp16:
 * fp16 operands and outputs for the projections, the attention call and the GELU, fp32 for the residual sum and LayerNorm
 * (src/ccrec/models/item_tower.py:122 runs transformers' BertLayer under it).  Every "bf16" array above is then an fp16 array; scores,
 * softmax, accumulation, residual stream and LayerNorm stay fp32 either way.  The *_bf16 / un-suffixed entry points above are these
 * with half_dtype = CCR_DTYPE_BF16. */
int ccr_attention_half(const uint16_t *qkv, const int32_t *seq_start, const int32_t *seq_len, uint16_t *out, int n_seq, int n_heads,
                       int max_len, int pad_len, float scale, int half_dtype, void *stream);
int ccr_add_layernorm_half(const uint16_t *x_half, const float *residual, const float *gamma, const float *beta, float eps,
                           float *out_f32, uint16_t *out_half, int64_t rows, int dim, int half_dtype, void *stream);
int ccr_embed_layernorm_half(const float *word_table, int64_t vocab, const float *position_table, int64_t n_positions,
                             const float *type_table, int64_t n_types, const int64_t *token_ids, const int64_t *positions,
                             const int64_t *token_types, const float *gamma, const float *beta, float eps, float *out_f32,
                             uint16_t *out_half, int64_t rows, int dim, int half_dtype, void *stream);
int ccr_gelu_half(const uint16_t *x, uint16_t *y, int64_t n, int half_dtype, void *stream);

/*
 * Build a search index over a resident bf16 corpus shard (borrowed pointer, no copy).
 * Replaces: the host-resident fp32 passage matrix of scripts/ms_marco_eval.py:199-201,208-210.
 *   global_row_offset: id of row 0 of this shard in the whole corpus (multi-GPU row sharding).
 * Synchronises `stream` once (one pass over the shard: the largest row norm of every 256-row tile, for the filter margins).
 */
int ccr_index_create(const uint16_t *D_bf16, int64_t n_rows, int dim, int64_t global_row_offset, void *stream,
                     ccr_index **out);
/* As ccr_index_create, but the caller supplies (device pointer, [n_rows] floats) an upper bound of every row's norm
 * (from ccr_pack_bf16_ex / ccr_meanpool_pack_bf16_ex): no pass over the corpus, no stream synchronisation.  The array is
 * BORROWED like the corpus: it must stay valid and unchanged until the index is destroyed (the main pass uses the largest
 * bound of each 256-row tile, the select stage each candidate row's own bound). */
int ccr_index_create_with_norms(const uint16_t *D_bf16, int64_t n_rows, int dim, int64_t global_row_offset,
                                const float *row_norm_bounds, void *stream, ccr_index **out);
int ccr_index_destroy(ccr_index *index);
int64_t ccr_index_rows(const ccr_index *index);
int ccr_index_dim(const ccr_index *index);

/*
 * Exhaustive inner-product top-k of n_q queries against the shard.
 * Replaces: the score loop, the host [Q,N] matrix and the per-row full sort of
 * scripts/ms_marco_eval.py:203-218,228-230, and the per-batch topk of
 * src/rime_lite/util/__init__.py:136-142.
 *   Q_bf16 [n_q][dim]; out_scores [n_q][k] fp32; out_ids [n_q][k] int64 (global ids)
 *   1 <= k <= min(n_rows, 4096).
 * The call returns after the results are complete on `stream` (it synchronises the stream once
 * to read the fallback count).  With CCR_SEARCH_ASYNC in `flags` it returns without synchronising: the kernels are
 * enqueued and the caller may enqueue more work (e.g. the shard exchange); ccr_search_finish(index) -- which waits for THIS
 * search's own event (not for work enqueued after it), fills the statistics and re-does the queries the search flagged
 * (sub-list overflow, mass ties, an estimated threshold that failed its check: rare) -- must run before the results of
 * flagged queries are trusted (ccr_search_last_stats().n_fallback tells how many there were; the lists of all other queries
 * are final on the stream).  Buffers and workspace must stay valid until then.
 * Embeddings are expected to be finite.  NaN / Inf values do not fault: the filter margins become infinite, every
 * query takes the exact dense path, +-Inf scores rank as numbers and NaN scores rank by bit pattern (not torch.sort's
 * NaN-first rule) -- identically on every path.
 */
size_t ccr_search_workspace_bytes(const ccr_index *index, int n_q, int k);
int ccr_search(ccr_index *index, const uint16_t *Q_bf16, int n_q, int k, float *out_scores, int64_t *out_ids,
               void *workspace, size_t ws_bytes, int flags, void *stream);
int ccr_search_finish(ccr_index *index);
int ccr_search_last_stats(const ccr_index *index, ccr_search_stats *stats /* host */);
/*
 * Makes `stream` wait until the MAIN PASS (the dominant kernel) of the index's last search -- possibly still pending
 * (CCR_SEARCH_ASYNC) -- has completed on the search's stream.  Work that does not depend on that search's results and does not
 * touch its buffers (packing the NEXT corpus shard into another buffer, the reference's next `embedding_func` batch of
 * scripts/ms_marco_eval.py:141-149) can then run beside the search's select stage instead of behind it.  No-op when the
 * last search took the dense path (nothing recorded).  No host synchronisation.
 */
int ccr_search_stream_wait_main_pass(const ccr_index *index, void *stream);

/*
 * Dense score matrix of n_q queries against the shard: out [n_q][n_rows] fp32.
 * Replaces: cos_sim / the chunked `Q @ chunk^T` of scripts/ms_marco_eval.py:155-162,212-215 and
 * src/ccrec/models/bbpr.py:485-492,536-540 when a caller really wants the matrix (small problems; the ranking path
 * never materialises it).  mode: CCR_SCORES_CANONICAL or CCR_SCORES_MFMA.
 */
int ccr_scores(const ccr_index *index, const uint16_t *Q_bf16, int n_q, int mode, float *out, void *stream);

/*
 * Search with per-query blocked ids of ANY length (the reference blocks any number of ids per query:
 * scripts/ms_marco_eval.py:224-227, scores[block_ind] = -1e6, kept not removed).
 *   block_ptr_host [n_q + 1] (HOST), block_idx [block_ptr_host[n_q]] int64 GLOBAL ids (DEVICE), ascending and unique inside
 *   each query; ids outside this shard's [offset, offset + n_rows) are ignored (row-sharded search: every rank passes the
 *   whole list).  Queries with k + len <= min(n_rows, 4096) over-fetch through the fused path and post-filter
 *   (ccr_apply_block); longer lists take the exact dense path with the blocked columns set to -1e6 before the selection.
 *   Result: canonical order of the modified scores, exactly k entries per query.
 */
size_t ccr_search_blocked_workspace_bytes(const ccr_index *index, int n_q, int k, const int64_t *block_ptr_host);
int ccr_search_blocked(ccr_index *index, const uint16_t *Q_bf16, int n_q, int k, const int64_t *block_ptr_host,
                       const int64_t *block_idx, float *out_scores, int64_t *out_ids, void *workspace, size_t ws_bytes,
                       int flags, void *stream);

/*
 * Top-k of (low-rank score + sparse prior), the post-fit expression of the ranker API:
 *   bbpr.transform(gnd) + gnd.prior_score -> evaluate_item_rec(..., k)   (src/ccrec/models/bbpr.py:592-595;
 *   _assign_topk src/rime_lite/util/__init__.py:117-155 over ElementWiseExpression(add, [dense, sparse]),
 *   src/rime_lite/util/score_array.py:300-318).
 *   prior CSR: prior_ptr_host [n_q + 1] (HOST), prior_idx int64 GLOBAL column ids ascending and unique per row (DEVICE),
 *   prior_val fp64 (DEVICE); at most 4096 entries per row.
 *   final(q, j) = (double) canonical_score(q, j) + prior(q, j)  (fp64, as torch promotes fp32 + fp64);
 *   out_scores fp64 [n_q][k], out_ids [n_q][k] in the order (final desc, id asc).
 * Exact for any prior values (negative ones included): the candidates are the top-(k + nnz_q) of the low-rank score
 * plus every prior column of the row (those are re-scored canonically).
 */
size_t ccr_search_sparse_prior_workspace_bytes(const ccr_index *index, int n_q, int k, const int64_t *prior_ptr_host);
int ccr_search_sparse_prior(ccr_index *index, const uint16_t *Q_bf16, int n_q, int k, const int64_t *prior_ptr_host,
                            const int64_t *prior_idx, const double *prior_val, double *out_scores, int64_t *out_ids,
                            void *workspace, size_t ws_bytes, int flags, void *stream);

/*
 * Column sums of a bf16 matrix in fp64: out[c] = sum_r X[r][c] (fixed order per column block, deterministic).
 * Used by score_op(S, "sum") for a low-rank S = U V^T: sum_ij S_ij = colsum(U) . colsum(V)
 * (src/rime_lite/util/score_array.py:460-474 without materialising any batch of S).
 */
int ccr_colsum_bf16(const uint16_t *X, int64_t rows, int dim, double *out /* [dim] device */, void *stream);

/*
 * Merge R per-shard top-k lists (after the RCCL all-gather) into the global top-k.
 * New (the reference scores on one GPU only: SURVEY 2a); order rule as above.
 *   scores [R][n_q][k], ids [R][n_q][k] -> out_scores [n_q][k], out_ids [n_q][k];  k <= 4096.
 */
int ccr_merge_topk(const float *scores, const int64_t *ids, int R, int n_q, int k, float *out_scores,
                   int64_t *out_ids, void *stream);
/* Same merge over rank-strided inputs: list (r, q) starts at scores + r * score_rank_stride + q * k and
 * ids + r * id_rank_stride + q * k (strides in elements).  This is the layout ONE all-gather of a packed per-rank
 * message {scores [n_q][k] fp32 | ids [n_q][k] int64} leaves behind: a single collective instead of two. */
int ccr_merge_topk_strided(const float *scores, const int64_t *ids, int64_t score_rank_stride, int64_t id_rank_stride,
                           int R, int n_q, int k, float *out_scores, int64_t *out_ids, void *stream);

/*
 * Packed per-shard result message of a row-sharded search: ONE all-gather (RCCL) moves it, one kernel merges the gathered copies.
 * New (the reference scores on one GPU only, scripts/ms_marco_eval.py:205-218); what it replaces there is the per-row
 * sort + keep-1001 of :228-230 for a corpus split over the GPUs of a node.
 *   layout: 32-byte ccr_shard_header | scores [n_q][k] fp32 | pad to 16 B | rows [n_q][k] u32 LOCAL row of this shard | pad to 16 B
 *   8 bytes per entry on the wire (12 with int64 global ids); the merge adds header.row_offset.
 * ccr_search_shard == ccr_search writing that message.  With CCR_SEARCH_ASYNC the call does not synchronise and the header's
 * n_flagged is written ON THE STREAM (the select stage's flag count): after the all-gather every rank reads every rank's
 * {n_flagged, n_covered} and all ranks take the same branch -- lists are final iff n_flagged <= n_covered (= 0) on every rank, otherwise
 * the flagged ranks call ccr_search_finish and the exchange is repeated by ALL ranks (a matched second collective).
 * 1 <= k <= min(n_rows, 4096); a shard smaller than k searches k_valid = n_rows entries and fills the message with
 * ccr_shard_message_fill instead.
 */
#define CCR_SHARD_MAGIC 0x4d524343u /* "CCRM" */
typedef struct ccr_shard_header {
    uint32_t magic;
    uint32_t n_flagged;   /* queries flagged by the shard's asynchronous search (0 after a synchronous one) */
    uint32_t k_valid;     /* entries of every list that are real rows; slots [k_valid, k) are padding */
    uint32_t n_covered;   /* flagged queries the search completed on the stream by itself (0: ccr_search_finish completes them) */
    int64_t row_offset;   /* global id of the shard's row 0 */
    int64_t n_rows;       /* rows of the shard */
} ccr_shard_header;
size_t ccr_shard_message_bytes(int n_q, int k);
int ccr_search_shard(ccr_index *index, const uint16_t *Q_bf16, int n_q, int k, void *message, void *workspace, size_t ws_bytes,
                     int flags, void *stream);
/* Build a message from ordinary results (blocked searches, shards smaller than k):
 *   scores [n_q][k_valid] fp32, ids [n_q][k_valid] int64 GLOBAL ids inside [row_offset, row_offset + n_rows), k_valid <= k. */
int ccr_shard_message_fill(void *message, int n_q, int k, int k_valid, const float *scores, const int64_t *ids, int64_t row_offset,
                           int64_t n_rows, void *stream);
/* Merge R gathered messages (message r at messages + r * message_stride_bytes) into the global top-k: out_scores [n_q][k] fp32,
 * out_ids [n_q][k] int64 global ids, order rule as everywhere.  Padding slots rank last (score -inf, distinct ids above 2^62), so
 * every output slot is written even when the whole corpus holds fewer than k rows.  k <= 4096, R <= 64. */
int ccr_merge_shard_messages(const void *messages, int64_t message_stride_bytes, int R, int n_q, int k, float *out_scores,
                             int64_t *out_ids, void *stream);
/* Short-list exchange (the k that scripts/ms_marco_eval.py:230 keeps -- 1001 -- over R shards).  A shard of exchangeable rows holds
 * k / R +- sqrt(k (1/R)(1 - 1/R)) of a global top-k, so every rank searches and sends only its canonical top-k_list, k_list ~ k / R + 6 sigma
 * (ccr_search_shard(..., k_list, ...): the select / re-score stage, which does not shrink with the shard, does k_list instead of k rows per
 * query, and the message is k_list / k of the size).  This merge keeps the k_out best of the R k_list entries and VERIFIES the shortcut: a
 * shard's list is its exact top-k_list, so if its last entry is not among the kept k_out, none of its unsent rows is either and the result
 * is the global top-k_out, bit for bit.  flags[q] = 1 (and *n_flagged counts them; both device memory, n_flagged zeroed by the call) where some
 * list was consumed to its end while its shard (header.n_rows > header.k_valid) holds more rows: the caller repeats THOSE queries with full
 * lists (every rank computes the same flags from the same gathered bytes, so the repeat is a matched collective).
 *   messages: R gathered messages of ccr_shard_message_bytes(n_q, k_list) layout; out_* [n_q][k_out]; k_list <= k_out <= R k_list;
 *   R k_list 12 B <= 96 KiB (the lists of a query are merged in LDS). */
int ccr_merge_short_lists(const void *messages, int64_t message_stride_bytes, int R, int n_q, int k_list, int k_out, float *out_scores,
                          int64_t *out_ids, uint32_t *flags, uint32_t *n_flagged, void *stream);

/*
 * Apply per-query blocked ids to an over-fetched result list.
 * Replaces: scripts/ms_marco_eval.py:224-227 (scores[block_ind] = -1e6, kept not removed).
 *   in_* [n_q][k_in] canonical top-k_in with k_in >= min(n_rows, k_out + max block length);
 *   block_ptr [n_q+1], block_idx [block_ptr[n_q]] int64 global ids, ascending inside each query;
 *   out_* [n_q][k_out]: unblocked entries in order, then (only if fewer than k_out remain)
 *   blocked ids ascending with score -1e6.  n_rows_total bounds valid ids (CCR_ERR_BLOCK_ID).
 */
int ccr_apply_block(const float *in_scores, const int64_t *in_ids, int n_q, int k_in, const int64_t *block_ptr,
                    const int64_t *block_idx, int64_t n_rows_total, float *out_scores, int64_t *out_ids, int k_out,
                    void *stream);

/*
 * In-batch-negative contrastive loss ("multiple_nrl"), forward and backward.
 * Replaces: src/ccrec/models/bbpr.py:205-212 (two mm + cat + scale + CrossEntropyLoss mean).
 *   Qe, Pe, Ne [B][dim] bf16; logits = [Qe Pe^T | Qe Ne^T] * inv_temperature (fp32 accumulate)
 *   fwd: loss (1 float, mean CE with labels arange(B)), lse [B] (saved for bwd)
 *   bwd: dQ, dP, dN [B][dim] fp32 = grad_out * dloss/d(.)
 *   workspace: ccr_inbatch_ce_workspace_bytes(B, dim) bytes of device memory (16-byte aligned); dim % 16 == 0; the embedding
 *   pointers 16-byte aligned.  The FORWARD leaves the scaled logits ([2B][B] fp32) in it and the BACKWARD reads them there
 *   instead of recomputing them: pass the backward the same, unmodified workspace its forward call used (the Python autograd
 *   function saves it with the operands).  The forward also leaves a stamp {magic, B, dim, inv_temperature} behind its logits and
 *   the backward checks it ON THE DEVICE: a workspace that is not this (B, dim, inv_temperature) forward's -- scratch memory, or one
 *   another forward has started on since -- yields NaN gradients, not plausible garbage (no host round trip, so no error code).
 *   Launches: forward = one kernel (+ a memset of its tickets and stamp), backward = two kernels (the gradient of the logits evaluated
 *   and split into three bf16 parts once, in both orientations, + the embeddings' transposes; then one launch for dQ and dP | dN with
 *   every MFMA fragment read straight from L2); deterministic (fixed combine orders, no float atomics).
 *   ccr_inbatch_pack3_bf16: the step's three fp32 blocks [B][dim] -> out [3][B][dim] bf16 (RNE, torch's .to(bfloat16) bits) in ONE
 *   launch (dim % 8 == 0, 16-byte aligned pointers); ccr_inbatch_ce_fwd_f32 = that pack into `packed` + the forward on the packed
 *   blocks in one call (the autograd function's host path: the backward takes packed, packed + B dim, packed + 2 B dim).
 */
size_t ccr_inbatch_ce_workspace_bytes(int B, int dim);
int ccr_inbatch_pack3_bf16(const float *q, const float *p, const float *n, int B, int dim, uint16_t *out, void *stream);
int ccr_inbatch_ce_fwd_f32(const float *q, const float *p, const float *n, int B, int dim, float inv_temperature, uint16_t *packed,
                           float *loss, float *lse, void *workspace, size_t ws_bytes, void *stream);
int ccr_inbatch_ce_fwd(const uint16_t *Qe, const uint16_t *Pe, const uint16_t *Ne, int B, int dim,
                       float inv_temperature, float *loss, float *lse, void *workspace, size_t ws_bytes, void *stream);
int ccr_inbatch_ce_bwd(const uint16_t *Qe, const uint16_t *Pe, const uint16_t *Ne, const float *lse, int B, int dim,
                       float inv_temperature, float grad_out, float *dQ, float *dP, float *dN, void *workspace,
                       size_t ws_bytes, void *stream);
/* Same with the upstream gradient read from device memory (a 1-element fp32 tensor, as autograd hands it over):
 * no host read-back of the scalar, so the training step stays asynchronous. */
int ccr_inbatch_ce_bwd_dev(const uint16_t *Qe, const uint16_t *Pe, const uint16_t *Ne, const float *lse, int B, int dim,
                           float inv_temperature, const float *grad_out_dev, float *dQ, float *dP, float *dN,
                           void *workspace, size_t ws_bytes, void *stream);

/*
 * Reciprocal rank / hit counts of every query at several cut-offs, from the id tensor of ccr_search.
 * Replaces: EvaluateRetrieval.evaluate_custom(qrels, ranking_profile, [1,5,10,100], metric="mrr")
 * (scripts/al_0_rank.py:130-133) and its python-dict traversal; the caller averages over queries.
 *   ids [n_q][k] int64 (rank order); qrel_ptr [n_q+1], qrel_idx ascending per query (relevant ids, score > 0);
 *   k_values [n_k] int32 (n_k <= 16); out_rr [n_q][n_k] fp32; out_hits [n_q][n_k] int32.
 */
int ccr_rank_metrics(const int64_t *ids, int n_q, int k, const int64_t *qrel_ptr, const int64_t *qrel_idx,
                     const int32_t *k_values, int n_k, float *out_rr, int32_t *out_hits, void *stream);

/*
 * BM25 as a sparse scorer (the lexical leg of the candidate builder).
 * Replaces: BM25.transform, scripts/bm_25.py:31-52 (scipy on the host, one query at a time) and the per-query full
 * sort + keep 1001 of ranking_bm25, scripts/ms_marco_eval.py:165-186.
 *   index  = term-major postings of the count matrix: indptr [n_terms + 1] (HOST), doc_ids [nnz] int32 -- STRICTLY ASCENDING
 *            inside a term (scipy's sorted CSC; the scorer walks every list with a cursor) and inside [0, n_docs): checked by
 *            ccr_bm25_index_create in one pass over the postings (CCR_ERR_INVALID names the first violation) -- and tf [nnz] fp32 (DEVICE,
 *            borrowed), doc_k [n_docs] fp64 (DEVICE, borrowed) = k1 * (1 - b + b * len_d / avdl)
 *   query  = CSR on the HOST: q_ptr [n_q + 1], q_terms strictly ascending term ids, q_idf = ln(n / df_t) per entry
 *   score(q, d) = sum_t ascending (tf * idf_t) * (k1 + 1) / (tf + doc_k[d]) in fp64, rounded once to fp32;
 *   out_scores / out_ids [n_q][k] in the order (score desc, document index asc); documents without any query term
 *   score 0 and follow in index order, as a stable sort of the reference's dense score vector would leave them.
 *   Queries of up to 256 distinct terms run the document-tile scorer (fp64 accumulators in LDS; one cursor group per lane up to 64 terms,
 *   four up to 256); from ~30 k documents up the top-k
 *   filter is fused into it (a sampled threshold per query, the documents that pass leave the tile's registers as candidate records,
 *   verified; rows the filter cannot finish are scored again with their fp32 rows stored and ranked exactly), so no [n_q][n_docs]
 *   score row exists and the workspace is ~130 KiB per query + 1 GiB.  Smaller corpora and CCR_BM25_DENSE_SELECT=1 store the fp32 rows
 *   (~8 GiB per batch) and rank them exactly; a call with a longer query runs the round kernels (fp64 rows in the workspace): same
 *   bits on every path.  CCR_BM25_TILE=-1 (read at ccr_bm25_index_create) forces the round kernels.
 *   ccr_bm25_search_workspace_bytes_k sizes the workspace for one k; ccr_bm25_search_workspace_bytes for any k (the larger layout).
 */
typedef struct ccr_bm25_index ccr_bm25_index;
int ccr_bm25_index_create(const int64_t *indptr_host, const int32_t *doc_ids, const float *tf, const double *doc_k,
                          int64_t n_terms, int64_t n_docs, double k1, ccr_bm25_index **out);
int ccr_bm25_index_destroy(ccr_bm25_index *index);
/* Optional, once per index: the idf of every term (HOST, [n_terms]; BM25.fit's ln(n / df), scripts/bm_25.py:21-29) -> a table of the
 * FINISHED contribution of every posting, contrib [nnz] fp64 (DEVICE, caller-owned, filled here; must outlive the index):
 * (tf idf_t)(k1 + 1) / (tf + doc_k[d]) by the scorer's own operations, so the same bits.  A search whose q_idf entries all equal this
 * idf bit for bit streams 12 bytes per posting and adds -- no doc_k gather, no fp64 division; any other search runs as before. */
int ccr_bm25_index_set_idf(ccr_bm25_index *index, const double *idf_host, double *contrib, void *stream);
size_t ccr_bm25_search_workspace_bytes(const ccr_bm25_index *index, int n_q, int max_terms_per_query);
size_t ccr_bm25_search_workspace_bytes_k(const ccr_bm25_index *index, int n_q, int max_terms_per_query, int k);
/* of the index's last search: out4 = {path: 0 round kernels + stored rows, 1 tile scorer + stored rows, 2 tile scorer + fused filter, + 4 if
 * the contribution table was used;
 * rows the filter could not finish (scored again and ranked exactly); batches; rank of the sampled threshold} */
int ccr_bm25_search_last_stats(const ccr_bm25_index *index, int64_t *out4);
int ccr_bm25_search(const ccr_bm25_index *index, const int64_t *q_ptr_host, const int32_t *q_terms_host,
                    const double *q_idf_host, int n_q, int k, float *out_scores, int64_t *out_ids, void *workspace,
                    size_t ws_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CCR_RETRIEVAL_H */
