// ccr_common.h -- shared host/device helpers for the gfx950 retrieval library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "ccr_retrieval.h"

namespace ccr {

void set_error(const char *fmt, ...);

#define CCR_HIP_CHECK(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ccr::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return CCR_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

#define CCR_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            ccr::set_error(__VA_ARGS__);  \
            return CCR_ERR_INVALID;       \
        }                                 \
    } while (0)

#define CCR_LAUNCH_CHECK()                                                              \
    do {                                                                                \
        hipError_t _e = hipGetLastError();                                              \
        if (_e != hipSuccess) {                                                         \
            ccr::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return CCR_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)

// ---- geometry of the fused MFMA path (one definition for kernels and planner)
constexpr int TILE_DOCS = 256;      // corpus rows per GEMM tile
constexpr int TILE_Q = 256;         // queries per GEMM tile
constexpr int WIDE_Q = 384;         // ... of the 256 x 384 form of the main pass (gemm_topk16w_kernel)
constexpr int TILE_K = 32;          // K granularity of the fused kernels (one LDS sub-stage = 32 bf16 = 64 B per row): dim % 32 == 0
constexpr int GEMM_THREADS = 512;   // 8 waves: 2 (doc halves) x 4 (query quarters)
constexpr int GROUPS_PER_TILE = 16; // group maxima per (sample tile, query): 2 wave rows x 4 MFMA tiles x 2 lane halves
constexpr int MAX_K = 4096;
constexpr uint32_t FLAG_DENSE = 0x80000000u;   // flagged-list entry: the query needs the exact dense path (not a retry)
constexpr int NUM_XCD = 8;

// Candidate storage of the main pass.  The ranges of one launch (phase) form a segment with its own sub-list capacity:
// the later phases run under re-tightened thresholds and need far fewer slots per sub-list than phase A.
// The sp sub-lists of one (range r, query q) share a CELL of sp * cap[g] records at base[g] + ((r - begin(g)) * nq_pad + q) * sp * cap[g],
// stored SLOT-MAJOR: record `slot` of sub-list (part) s sits at cell + slot * sp + s.  Lists hold one or two records on
// average, so a cell's records fill its first one or two 128-byte lines instead of one line per sub-list -- the readers
// (threshold update, select) are bound by the number of scattered lines they touch, not by bytes.
struct CandLayout {
    int nseg;
    int seg_end[3];     // exclusive end range of each segment (ascending; unused entries = INT_MAX)
    int cap[3];
    long long base[3];  // in 8-byte records from the start of the candidate area
};
// segment of range r: capacity, first range, first record
__host__ __device__ __forceinline__ void cand_segment(const CandLayout &L, int r, int &cap, int &r0, long long &base) {
    const int g = (r >= L.seg_end[0] ? 1 : 0) + (r >= L.seg_end[1] ? 1 : 0);
    r0 = g ? L.seg_end[g - 1] : 0;
    cap = L.cap[g];
    base = L.base[g];
}
// first record (slot 0) of sub-list (r, q, s); slot sl is sp * sl records further
__host__ __device__ __forceinline__ long long cand_sublist(const CandLayout &L, int r, int q, int s, int nq_pad, int sp, int &cap) {
    const int g = (r >= L.seg_end[0] ? 1 : 0) + (r >= L.seg_end[1] ? 1 : 0);
    const int r0 = g ? L.seg_end[g - 1] : 0;
    cap = L.cap[g];
    return L.base[g] + (((long long)(r - r0) * nq_pad + q) * sp) * (long long)cap + s;
}

struct Plan {
    int fused;            // 1 = fused MFMA path usable
    int nq_pad;           // n_q rounded up to TILE_Q (256 x 384 main pass: at least main_qblocks * 384)
    int qblocks;          // nq_pad / TILE_Q: the query blocks of the sample pass (and of every 256 x 256 kernel)
    int tile_q;           // queries per tile of the MAIN pass: TILE_Q, or 384 (gemm_topk16w_kernel)
    int main_qblocks;     // query blocks of the main pass: ceil(n_q / tile_q)
    int main_qgroups;     // ... and their groups over the XCDs (divides main_qblocks)
    int64_t tiles;        // ceil(n_rows / TILE_DOCS)
    int64_t full_tiles;   // floor(n_rows / TILE_DOCS)
    int sample_tiles;     // tiles scored by the threshold pass
    int64_t sample_stride;
    int sample_ranges;    // range count of the sample pass (chosen by simulating its item assignment)
    int ranges;           // corpus ranges of the main pass (multiple of NUM_XCD)
    int item_a;           // phase A = work items [0, item_a) of every XCD set (0 = single phase); thresholds are re-tightened after it
    int item_b;           // end of the second phase in items (0: two phases)
    int opt_rank;         // > 0: estimated thresholds = the opt_rank-th largest sampled group maximum (single launch; the select verifies)
    int qgroups;          // query-block groups over the XCDs
    int cap;              // candidate slots per sub-list (the largest segment's: statistics)
    CandLayout cand;      // per-phase segments of the candidate area
    int grid;             // persistent workgroups (multiple of NUM_XCD)
    int rescore_cap;      // max rows re-scored per query (power of two)
    int select_compact;   // candidates per query the select kernel gathers into LDS
    int mfma16;           // 1: main pass on the 16x16x32 MFMA kernel
    int sublists;         // candidate sub-lists per (range, query): 4 (32x32x16 kernel) or 8 (16x16x32 kernel)
    // small query batches (n_q <= 64, query rows resident in LDS): the FIRST main pass is the streaming kernel (ccr_narrow.hip) with
    // its own list layout -- one range, two sub-lists per query; the retry pass of flagged queries keeps the tile kernels' layout above
    int narrow;           // 0, or the query tiles of 16 the streaming kernel computes (1, 2, 4)
    int narrow_groups;    // 1, or 2: 65 .. 128 queries as two groups of <= 64, each streamed by half of the workgroups (pairs on one XCD)
    int first_nsub;       // sub-lists per query the first main pass fills (ranges * sublists, or 2) ...
    int first_sp;         // ... per cell (sublists, or 2) ...
    CandLayout first_lay; // ... and where they are (cand, or one segment of narrow capacity)
    // workspace layout (byte offsets)
    size_t off_qnorm, off_thr, off_gmax, off_cnt, off_cand, off_flag, off_dense, off_retry, off_top, off_safe, total;
    int64_t dense_rows_per_chunk;  // queries per dense chunk
};

// arguments of the fused GEMM kernel (ccr_fused.hip)
struct GemmArgs {
    const uint16_t *D;
    int64_t n_rows;
    int dim;
    const uint16_t *Q;
    int n_q;
    int nq_pad;
    int qblocks;
    int64_t n_vt;         // virtual tiles; real tile = vt * tile_stride
    int64_t tile_stride;
    int ranges;           // item (r, qb) covers virtual tiles r, r + ranges, ...
    int qgroups;          // query-block groups spread over the XCDs (1, 2, 4 or 8; divides qblocks)
    // Work items of an XCD set are numbered item = (range / classes) * blocks_per_group + block_in_group; a launch covers the
    // items [item_begin, item_end) of every XCD set, so a phase can be EXACTLY whole rounds of items whatever the range count
    int item_begin, item_end;
    // EPI_FILTER
    const float *thr;     // [nq_pad] tau_q: a lower bound of the query's k-th largest EXACT score
    const float *cq;      // [nq_pad] margin coefficient gamma * ||q||: |mfma - exact| <= cq * ||d||
    const float *tile_norm;   // [ceil(n_rows / TILE_DOCS)] bound of the row norms of each 256-row tile (the index's)
    uint2 *cand;          // candidate area {score bits, local row}; sub-list addresses and capacities from `lay`
    uint32_t *cnt;        // [ranges][nq_pad][sublists]
    CandLayout lay;
    // EPI_GMAX
    float *gmax;          // [n_vt * 16][nq_pad]
    // EPI_STORE (debug)
    float *store;         // [n_q][store_pitch]
    int64_t store_pitch;  // floats per stored query row (0: n_rows); a multiple of 4 lets a lane store its 4 consecutive rows as 16 bytes
    int dbg;              // timing-only ablations (CCR_GEMM_DBG; results are WRONG when non-zero)
    int stagger;          // 32x32x16 kernel: 1 = the two wave groups run one barrier interval apart (production), 0 = in phase
    int item_swap;        // experiment (CCR_ITEM_SWAP, single-launch plans only): co-resident workgroups share the query block, not the range
    int64_t dbg_alloc_rows;   // diagnostic library, with dbg_pitch: rows the array behind D holds (CCR_DBG_ALLOC_ROWS)
    int dbg_alloc_q;      // ... and the rows the array behind Q holds (CCR_DBG_ALLOC_Q)
    int dbg_pitch;        // diagnostic library, CCR_GEMM_DBG & 1024: row pitch (elements) the DMA addresses are generated with
    int qdirect;          // 16x16x32 kernel: 0 = queries through the LDS ring, 1 / 3 / 4 / 5 = query fragments straight from global memory (gemm_topk16q_kernel)
};


// ---- device helpers
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// monotone map float -> uint32 (larger float <-> larger uint); NaN maps above +inf (never produced here)
__device__ __forceinline__ uint32_t f32_orderable(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float orderable_to_f32(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// Result ids are written as int64 global ids (row + id_offset) or, for a packed shard message (ccr_search_shard), as the u32 LOCAL row:
// the exchange then moves 8 instead of 12 bytes per entry and the merge adds the shard's row offset from the message header.
constexpr int64_t ID_LOCAL_U32 = INT64_MIN;
__device__ __forceinline__ void store_id(int64_t *out_ids, int64_t pos, int64_t id_offset, uint32_t row) {
    if (id_offset == ID_LOCAL_U32)
        reinterpret_cast<uint32_t *>(out_ids)[pos] = row;
    else
        out_ids[pos] = id_offset + (int64_t)row;
}

// worst-case |mfma fp32 score - exact| <= gamma(dim) * ||q|| * ||d||  (any summation order, each
// fp32 add within 2^-23 relative: see DESIGN.md "filter margins")
__host__ __device__ __forceinline__ float mfma_gamma(int dim) { return (float)dim * 1.1920929e-7f * 1.02f; }

}  // namespace ccr
