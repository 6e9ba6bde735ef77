// ccr_narrow.h -- the streaming main pass for small query batches (ccr_narrow.hip): arguments, geometry, launcher.
#pragma once
#include "ccr_common.h"

namespace ccr {

constexpr int NARROW_THREADS = 512;                 // 8 waves, each streaming its own 16-row groups
constexpr int NARROW_WAVES = NARROW_THREADS / 64;
constexpr int NARROW_MAX_Q = 64;                    // query rows resident in LDS (4 MFMA tiles of 16)
constexpr int NARROW_MAX_GROUPS = 2;                // query groups of <= NARROW_MAX_Q a launch serves (each group streamed by 1 / groups of the workgroups)
constexpr int NARROW_LDS_CAP = 64;                  // records a workgroup stages per query before it flushes (overflow: straight to global)
constexpr int NARROW_LDS_CAP_WIDE = 8;              // ... with 96 resident query rows (6 tiles: 147 KiB of the 160 at dim 768)
__host__ __device__ constexpr int narrow_lds_cap(int nqt) { return nqt > 4 ? NARROW_LDS_CAP_WIDE : NARROW_LDS_CAP; }
constexpr int NARROW_SUBLISTS = 2;                  // sub-lists per query in the candidate area: workgroups of even / odd index

// LDS bytes per resident query row: >= the row, == 32 (mod 256): the 16 lanes of every ds_read_b128 group (rows l15, chunks lq) then
// hit 16 distinct 16-byte slots of the 256-byte LDS line
__host__ __device__ inline int narrow_query_stride(int dim) {
    const int row = dim * 2;
    return row + ((32 - row % 256) + 256) % 256;
}
size_t narrow_lds_bytes(int nqt, int dim);

struct NarrowArgs {
    const uint16_t *D;
    int64_t n_rows;
    int dim;
    const uint16_t *Q;
    int n_q;
    int q_stride;            // narrow_query_stride(dim)
    const float *thr;        // [>= n_q] tau_q
    const float *cq;         // [>= n_q] gamma ||q||
    const float *tile_norm;  // [ceil(n_rows / 256)]
    uint2 *cand;             // candidate area: query q's cell = [cap][2] records {score bits, local row}, slot-major
    uint32_t *cnt;           // [n_q][2], zeroed by the caller
    int cap;                 // slots per sub-list
    int groups;              // 1, or 2: workgroup b serves query group (b >> 3) & 1 (queries [64 g, 64 g + 64)) and row stream (b >> 4) * 8 + (b & 7):
                             // blocks b and b + 8 -- one XCD -- walk the SAME rows, so the second reader finds them in that XCD's L2
};

int launch_narrow_filter(const NarrowArgs &a, int nqt, int grid, bool nt, hipStream_t s);

}  // namespace ccr
