// ccr_index.h -- the index object and the launchers shared by the host-side translation units
// (ccr_api.hip: planner + ccr_search; ccr_special.hip: blocked / sparse-prior searches).
#pragma once
#include "ccr_common.h"

namespace ccr {

// Tuning / diagnostic knobs from the environment, read ONCE when an index is created (never inside a search).
struct Knobs {
    int qgroups;       // CCR_QGROUPS      0 = planner's choice, else 1/2/4/8
    int progressive;   // CCR_PROGRESSIVE  0 = single main-pass launch
    int max_phases;    // CCR_PHASES       2 = at most one re-tightening (default 3)
    int mfma16;        // CCR_MFMA16       -1 = planner's choice, 0 = 32x32x16 kernel, 1 = 16x16x32 kernel
    int sample_div;    // CCR_SAMPLE_DIV   0 = planner's choice, else the pinned sample fraction 1/div
    int gemm_dbg;      // CCR_GEMM_DBG     timing-only ablations of the main pass (WRONG results when non-zero)
    int stagger;       // CCR_GEMM_STAGGER 0 = both wave groups of the 32x32x16 kernel in phase
    int qdirect;       // CCR_QDIRECT      16x16x32 main pass: 0 = queries through the LDS ring, 1 / 3 / 4 / 5 = query fragments straight from global memory
    int ranges;        // CCR_RANGES       0 = planner's choice, else the pinned range count (rounded to a multiple of 8)
    int item_swap;     // CCR_ITEM_SWAP    1 = co-resident workgroups share the query block instead of the corpus range (honoured with CCR_PROGRESSIVE=0)
    int optimistic;    // CCR_OPTIMISTIC   -1 = planner's choice, 0 = conservative thresholds only, 1 = estimated thresholds wherever the sample allows
    int opt_rank;      // CCR_OPT_RANK     0 = max(48, 3 k fs), else the pinned rank (tests: a small rank makes the verification fail)
    int max_lists;     // CCR_MAX_LISTS    0 = planner's limit, else a cap on ranges x sublists (A/B of the select stage's walk)
    int narrow;        // CCR_NARROW       -1 = planner's choice (n_q <= 64 whose rows fit the LDS), 0 = tile kernels only (the A/B knob)
    int narrow_nt;     // CCR_NARROW_NT    1 = the streaming kernel's corpus loads are non-temporal, 0 = default cache policy (default: 6.0 vs 5.5 TB/s at NQ)
    int narrow_grid;   // CCR_NARROW_GRID  0 = planner's choice, else workgroups of the streaming kernel
    int wide;          // CCR_WIDE         -1 = planner's choice, 0 = 256 x 256 main-pass tiles only (the A/B knob), 1 = 256 x 384 wherever the kernel applies
    int narrow_groups; // CCR_NARROW_GROUPS 2 = batches of 65 .. 128 queries stream as two query groups (default), 1 = tile kernels above 64 queries (the A/B knob)
};
Knobs read_knobs();

constexpr int FALLBACK_ROWS = 16;   // dense score rows reserved for flagged queries (one on-stream chunk)

Plan make_plan(int64_t n_rows, int dim, int n_q, int k, int flags, int num_cu, const Knobs &kn);

// kernels implemented in the other translation units
int launch_gemm_filter(const GemmArgs &a, int grid, hipStream_t s);
int launch_gemm_gmax(const GemmArgs &a, int grid, hipStream_t s);
int launch_gemm_store(const GemmArgs &a, int grid, hipStream_t s);
int launch_gemm16_filter(const GemmArgs &a, int grid, hipStream_t s);
int launch_gemm16_store(const GemmArgs &a, int grid, hipStream_t s);
int launch_gemm16w_filter(const GemmArgs &a, int grid, hipStream_t s);   // 256 x 384 tiles (a.qblocks = blocks of 384 queries), dim % 32 == 0
int launch_shard_header(const ccr_shard_header &h, void *message, hipStream_t s);   // ccr_merge.hip: the 32-byte header, by value
int ensure_dynamic_lds(const void *kernel, size_t lds);   // per (kernel, device) opt-in to > 64 KiB of dynamic LDS
// tile_bits (optional): bit patterns of the largest row norm of every 256-row tile, max-accumulated (zeroed by the caller)
int launch_row_norms_bf16(const uint16_t *X, int64_t rows, int dim, float *norms, uint32_t *max_bits, uint32_t *tile_bits,
                          hipStream_t s);
int launch_tile_norms(const float *row_bounds, int64_t rows, uint32_t *tile_bits, uint32_t *max_bits, hipStream_t s);
// thr[q] = tau_q (lower bound of the k-th largest exact score), cq[q] = gamma ||q|| (margin per unit of row norm)
int launch_threshold(const float *gmax, int64_t n_groups, int n_q, int nq_pad, int k, const float *qnorm,
                     const uint32_t *dmax_bits, int dim, const float *tile_norm, int64_t sample_stride, float *thr, float *cq,
                     hipStream_t s, uint32_t *zero_cnt = nullptr, int zero_per_query = 0);   // zero_cnt: n_q x zero_per_query counters to clear on the way
int select_compact_entries(int dim, int ranges, int rescore_cap, int64_t want);
// thr (device, [n_q], may be null): the thresholds the main pass filtered with.  The select VERIFIES them -- the k-th largest lower
// bound L of the candidates must reach thr[q], else rows below an over-estimated threshold may be missing: the query is flagged
// for the retry pass and thr[q] is replaced by L (a valid bound); with fewer than k candidates thr[q] = -inf and the dense path.
int launch_select_rescore(const uint2 *cand, const uint32_t *cnt, int ranges, int sp, int n_q, int nq_pad, const CandLayout &lay, int k,
                          int rescore_cap, int compact, int64_t n_rows, float *thr, const float *cq, const float *tile_norm, const float *row_norm,
                          const uint32_t *dmax_bits, const uint16_t *Q, const uint16_t *D, int dim, int64_t id_offset, float *out_scores, int64_t *out_ids,
                          uint32_t *flag_count, uint32_t *flag_list, unsigned long long *stat_cand, const uint32_t *out_rows,
                          hipStream_t s);
int launch_partition_flags(const uint32_t *flags, int begin, int n, uint32_t *retry_list, uint32_t *dense_list, uint32_t *counts,
                           hipStream_t s);
int launch_underfilled_to_retry(uint32_t *flags, int begin, int n, const float *thr_safe, float *thr, hipStream_t s);
int launch_scatter_thresholds(const uint32_t *list, int n, const float *thr2, float *thr, hipStream_t s);
int launch_gather_queries(const uint16_t *Q, int dim, const uint32_t *list, int n, const float *thr, const float *cq, uint16_t *Q2,
                          float *thr2, float *cq2, hipStream_t s);
// nsub: sub-lists of the fully scored ranges; queries whose block position inside its XCD group is < part_blocks have nsub_part.
// prev_*: the same at the previous re-tightening of this search (top_in: its k best lower bounds per query are in `top` and only
// the sub-lists completed since are read); top_out: leave the k best for the next one.  top: [nq_pad + n_q * k] u32 or null.
// tile_q: queries per block of the main pass that filled the lists (a query's block position decides which ranges it has seen).
int launch_threshold_update(const uint2 *cand, const uint32_t *cnt, int nsub, int nsub_part, int part_blocks, int prev_nsub, int prev_part,
                            int prev_blocks, int qb_per, int sp, int n_q, int nq_pad, const CandLayout &lay, int k, const float *cq,
                            const float *tile_norm, uint32_t *top, bool top_in, bool top_out, float *thr, hipStream_t s, int tile_q = TILE_Q);
// dense exact path.  qlist: query rows to score (nullptr: q_begin + qi); out_rows: destination rows of the select
// (nullptr: q_begin + qi); count_dev (device, may be null): only the first *count_dev - q_begin entries of the list exist
// (the on-stream fallback chunk of an asynchronous search -- the host does not know the count yet).
int launch_dense_scores(const uint16_t *D, int64_t n_rows, int dim, const uint16_t *Q, const uint32_t *qlist,
                        int q_begin, int nq_chunk, const uint32_t *count_dev, float *out, hipStream_t s);
int launch_dense_select(const float *scores, int64_t n_rows, int k, const uint32_t *out_rows, int q_begin, int nq_chunk,
                        const uint32_t *count_dev, int64_t id_offset, float *out_scores, int64_t *out_ids, hipStream_t s,
                        bool aggregate = true, const uint32_t *in_rows = nullptr, bool in_rows_compact = false);
// in_rows: block b ranks the score row in_rows[b] (in_rows_compact: score row b) and writes output row q_begin + in_rows[b]; with count_dev
// only the first *count_dev blocks

// Exact top-k from MFMA score rows [nq_chunk][pitch] + error margins (ccr_dense.hip); the chunk's query rows are contiguous at Q.
// hint (per chunk query, or null): a valid lower bound of its k-th largest score -- one scan of the row instead of five.
// Queries it cannot finish (more than 8 192 rows inside the margin, non-finite margins) are appended to flag_list with FLAG_DENSE.
int launch_margin_select(const float *scores, int64_t pitch, int64_t n_rows, int k, int dim, const uint16_t *Q, const uint16_t *D,
                         const float *tile_norm, const float *row_norm, const uint32_t *dmax_bits, const float *hint, const uint32_t *out_rows,
                         int q_begin, int nq_chunk, int64_t id_offset, float *out_scores, int64_t *out_ids, uint32_t *flag_count, uint32_t *flag_list,
                         hipStream_t s);

}  // namespace ccr

struct ccr_index {
    const uint16_t *D;
    int64_t n_rows;
    int dim;
    int64_t offset;
    int64_t id_out;       // what the running search writes as ids: `offset` (int64 global ids) or ID_LOCAL_U32 (u32 local rows of a shard message)
    uint32_t *dmax_bits;  // device: bits of the max row norm (a slot of the per-device slab)
    float *tile_norm;     // device [ceil(n_rows / 256)]: bound of the row norms of each 256-row tile (per-device block cache)
    size_t tile_bytes;
    const float *row_norm;   // device [n_rows]: norm bound of every row -- the caller's array (borrowed) or the index's own pass
    float *row_norm_own;     // the latter (per-device block cache), else null
    size_t row_bytes;
    bool have_events;
    bool main_pass_recorded;   // ev[4] was recorded by the last search (fused path): ccr_search_stream_wait_main_pass has something to wait for
    int num_cu;
    int device;
    ccr::Knobs knobs;
    hipEvent_t ev[8];     // phase boundaries of the last search; ev[7]: end of an asynchronous search's stream work (flag line copied)
    volatile uint32_t *host_flags;   // pinned host line {flag count, -, candidate count (8 B)} of the per-device slab
    ccr::Plan plan;       // plan of the last search and its key (the planner simulates item assignments: ~25 us)
    int plan_nq, plan_k, plan_flags;
    ccr_search_stats stats;
    // an asynchronous search (CCR_SEARCH_ASYNC) that ccr_search_finish has not completed yet
    struct {
        bool active;
        const uint16_t *Q;
        int n_q, k;
        float *out_scores;
        int64_t *out_ids;
        char *ws;
        hipStream_t stream;
    } pending;
};
