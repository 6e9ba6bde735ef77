// ccr_topk_device.h -- workgroup-level selection primitives shared by the dense and fused paths.
//   * block_radix_select : exact k-th largest of M 32-bit orderable keys (4 x 8-bit MSB-first passes,
//                          256-bin LDS histogram);
//   * block_bitonic_sort_desc : in-LDS bitonic sort of 64-bit keys, descending.
// 64-bit result keys are (orderable(score) << 32) | ~local_idx : unique per document, so "score
// descending, index ascending" is plain descending key order.
#pragma once
#include "ccr_common.h"

namespace ccr {

// Wave-parallel digit pick of one radix pass (called by lanes 0..63 of ONE wave): the largest digit d whose
// suffix count hist[d] + ... + hist[255] reaches `remaining`; ctl[0] = d, ctl[1] = remaining - (count above d).
// Lane l owns bins 4l..4l+3; a shuffle suffix-scan replaces the 256-step serial walk (d = 0 if never reached).
__device__ __forceinline__ void radix_pick_digit(const uint32_t *hist, int lane, uint32_t remaining, uint32_t *ctl) {
    uint32_t h[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = hist[4 * lane + j];
    const uint32_t own = h[0] + h[1] + h[2] + h[3];
    uint32_t incl = own;   // sum over lanes >= lane
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_down(incl, off, 64);
        if (lane + off < 64) incl += t;
    }
    const uint32_t above = incl - own;
    const bool cross = above < remaining && remaining <= incl;
    if (cross) {
        uint32_t cum = above;
        int d = 4 * lane;   // falls through to the lane's lowest bin
#pragma unroll
        for (int j = 3; j >= 1; --j) {
            if (cum + h[j] >= remaining) {
                d = 4 * lane + j;
                break;
            }
            cum += h[j];
        }
        ctl[0] = (uint32_t)d;
        ctl[1] = remaining - cum;
    }
    if (lane == 0 && incl < remaining) {  // fewer items than requested (contract violation): same answer as the serial walk
        ctl[0] = 0u;
        ctl[1] = remaining - (incl - h[0]);
    }
}

// Histogram increment with wave-level aggregation (all 64 lanes of the wave must call; `active` masks the lane).
// Scores of one row fall into a handful of the 256 top-byte bins, so plain LDS atomics from 64 lanes hit the same address and
// serialise (2.7 M of them per pass over an NQ-sized row).  Up to three rounds: the first active lane's bin is broadcast, the
// lanes that share it are counted with a ballot and ONE atomic adds the count; lanes still active afterwards (flat
// distributions) fall back to their own atomic.
__device__ __forceinline__ void hist_add_aggregated(uint32_t *hist, uint32_t bin, bool active) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(active);
#pragma unroll 1
    for (int r = 0; r < 3 && todo != 0ull; ++r) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t lb = (uint32_t)__shfl((int)bin, leader, 64);
        const bool mine = active && bin == lb;
        const unsigned long long same = __ballot(mine);
        if (lane == leader) atomicAdd(&hist[lb], (uint32_t)__popcll(same));
        if (mine) active = false;
        todo &= ~same;
    }
    if (active) atomicAdd(&hist[bin], 1u);
}

// All threads of the block must call.  get(i) -> uint32 orderable key of item i (0 <= i < M) or
// skip == true to ignore the slot.  On return (uniform across the block):
//   kth      = key of the k-th largest item
//   need_eq  = how many items equal to kth belong to the top-k (1 <= need_eq)
// Requires the number of non-skipped items >= k.  s_hist: 256 uint32, s_ctl: 4 uint32 (LDS).
// AGG: wave-aggregated histogram updates (long rows: M in the millions, skewed bins); the block size must be a multiple of 64.
template <bool AGG = false, class Get>
__device__ __forceinline__ void block_radix_select(Get get, int64_t M, int k, uint32_t *s_hist, uint32_t *s_ctl,
                                                   uint32_t &kth, int &need_eq) {
    const int tid = threadIdx.x;
    const int nt = blockDim.x;
    uint32_t prefix = 0, mask = 0;
    int remaining = k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int b = tid; b < 256; b += nt) s_hist[b] = 0;
        __syncthreads();
        if (AGG) {
            for (int64_t base = 0; base < M; base += nt) {   // every lane of a wave runs the same trips
                const int64_t i = base + tid;
                bool skip = false;
                uint32_t o = 0;
                if (i < M) o = get(i, skip);
                hist_add_aggregated(s_hist, (o >> shift) & 255u, i < M && !skip && (o & mask) == prefix);
            }
        } else {
            for (int64_t i = tid; i < M; i += nt) {
                bool skip = false;
                uint32_t o = get(i, skip);
                if (!skip && (o & mask) == prefix) atomicAdd(&s_hist[(o >> shift) & 255u], 1u);
            }
        }
        __syncthreads();
        if (tid < 64) radix_pick_digit(s_hist, tid, (uint32_t)remaining, s_ctl);
        __syncthreads();
        prefix |= s_ctl[0] << shift;
        mask |= 0xffu << shift;
        remaining = (int)s_ctl[1];
        __syncthreads();
    }
    kth = prefix;
    need_eq = remaining;
}

// keys[0..n) in LDS, n a power of two; all threads of the block must call.
__device__ __forceinline__ void block_bitonic_sort_desc(unsigned long long *keys, int n) {
    const int tid = threadIdx.x;
    const int nt = blockDim.x;
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = tid; i < (n >> 1); i += nt) {
                const int pos = 2 * i - (i & (stride - 1));
                const unsigned long long a = keys[pos], b = keys[pos + stride];
                const bool desc = ((pos & size) == 0);
                if ((a < b) == desc) {
                    keys[pos] = b;
                    keys[pos + stride] = a;
                }
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ unsigned long long make_key(float score, uint32_t local_idx) {
    return ((unsigned long long)f32_orderable(score) << 32) | (unsigned long long)(~local_idx);
}
__device__ __forceinline__ float key_score(unsigned long long key) { return orderable_to_f32((uint32_t)(key >> 32)); }
__device__ __forceinline__ uint32_t key_idx(unsigned long long key) { return ~(uint32_t)key; }

__host__ __device__ __forceinline__ int pow2_ceil(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// canonical score of one (query, doc) pair: fp64 accumulate in increasing element order.
// q: bf16 row in LDS (or global), d: bf16 row in global; dim % 8 == 0; both 16-byte aligned.
__device__ __forceinline__ float canonical_dot(const uint16_t *__restrict__ q, const uint16_t *__restrict__ d, int dim) {
    double acc = 0.0;
    // the loads of 4 steps are independent of the fp64 chain: unrolled so that they are in flight together
#pragma unroll 4
    for (int c = 0; c < dim; c += 8) {
        const uint4 dv = *reinterpret_cast<const uint4 *>(d + c);
        const uint4 qv = *reinterpret_cast<const uint4 *>(q + c);
        const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w};
        const uint32_t qw[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc = fma((double)__uint_as_float(qw[e] << 16), (double)__uint_as_float(dw[e] << 16), acc);
            acc = fma((double)__uint_as_float(qw[e] & 0xffff0000u), (double)__uint_as_float(dw[e] & 0xffff0000u), acc);
        }
    }
    return (float)acc;
}


}  // namespace ccr
