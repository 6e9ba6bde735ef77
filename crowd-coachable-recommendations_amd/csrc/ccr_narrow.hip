// ccr_narrow.hip -- the main pass for SMALL query batches (n_q <= 64): HBM-bound, so built as a stream, not as a tiled GEMM.
//
// scripts/ms_marco_eval.py:205-218 scores every query batch against the whole corpus; with a handful of queries (an interactive
// ranking() call, the retry of a few flagged queries, SURVEY 8d's Q = 1 / 16 / 64) the arithmetic intensity is n_q flop per corpus
// byte -- far below the ridge (~315) -- and the pass can at best run at the rate the corpus streams out of HBM.  The 256 x 256 tile
// kernel (ccr_fused.hip) is the wrong shape there: it computes a full 256-query tile whatever n_q is (256 x the MFMA work at n_q = 1),
// moves a 16-KiB query slice through the LDS-DMA path beside every 16 KiB of corpus, and rendezvouses its eight waves twice per 32
// K-elements; measured 5.0 / 4.8 / 4.0 TB/s of corpus bytes at n_q = 1 / 16 / 64 (profiles/r03_small_batches_pmc.txt).
//
// This kernel instead
//   * keeps the query rows RESIDENT in LDS for the whole launch (64 x 1 536 B = 96 KiB at dim 768; rows 32 bytes apart modulo 256 so
//     that the 16 lanes of every ds_read_b128 group hit 16 distinct 16-byte slots);
//   * streams the corpus straight from global memory into the MFMA A-operand registers: lane (row l15, chunk lq) of a wave loads 16
//     bytes of row l15 of its 16-row group -- exactly the fragment v_mfma_f32_16x16x32_bf16 wants, so there is no LDS round trip and
//     NO barrier in the loop: every wave runs its own stream with P loads (P KiB) in flight, refilling a slot as soon as it is consumed;
//   * computes only ceil(n_q / 16) query tiles (1, 2 or 4 MFMAs per K-step instead of 32 per wave);
//   * filters in the epilogue like the tile kernel (score + cq * tile norm >= tau_q) and appends the rare hits to per-query lists in
//     LDS (one LDS atomic per hit), flushed to the candidate area once per workgroup (one global atomic per (workgroup, query)).
// The candidate area uses the select stage's own layout with ONE range and TWO sub-lists per query (workgroups of even / odd index),
// so select_rescore_kernel, the retry pass and the exact paths are unchanged: results are the canonical bits either way.
// r5: 65 .. 128 queries run as TWO groups of <= 64 (NarrowArgs::groups): workgroups b and b + 8 -- the same XCD -- hold the two groups'
// query rows and walk the SAME corpus rows, so HBM still delivers the corpus once (the partner finds the rows in the XCD's L2 or in the
// Infinity Cache) instead of the tile kernels' full 256-query tile per corpus tile.
#include <stdlib.h>

#include "ccr_gemm_common.h"
#include "ccr_index.h"
#include "ccr_narrow.h"

namespace ccr {

typedef float f32x4n __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float narrow_uniform_f32(const float *p) {
    float v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

template <bool NT>
__device__ __forceinline__ bf16x8 stream_load(const char *p) {
    if constexpr (NT)
        return __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(p));
    else
        return *reinterpret_cast<const bf16x8 *>(p);
}

// NQT: query tiles of 16 (1, 2, 4; 6 = 65 .. 96 queries in one group, with a short staging list per query); P: loads in flight per wave =
// K-steps per chunk (divides dim / 32)
template <int NQT, int P, bool NT>
__global__ __launch_bounds__(NARROW_THREADS) void narrow_filter_kernel(const NarrowArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int row_bytes = a.dim * 2;
    const int S = a.q_stride;                      // LDS bytes per query row (== 32 mod 256)
    constexpr int NQ = NQT * 16;
    constexpr int LCAP = narrow_lds_cap(NQT);
    // query group and row stream of this workgroup (groups == 1: every workgroup serves all queries and has its own stream)
    const int grp = a.groups == 2 ? (int)((blockIdx.x >> 3) & 1u) : 0;
    const int stream_id = a.groups == 2 ? (int)((blockIdx.x >> 4) * 8 + (blockIdx.x & 7u)) : (int)blockIdx.x;
    const int n_streams = a.groups == 2 ? (int)(gridDim.x >> 1) : (int)gridDim.x;
    const int q_base = grp * NARROW_MAX_Q;
    const int nq_here = a.n_q - q_base < NQ ? a.n_q - q_base : NQ;      // >= 1: the planner pairs groups only above NARROW_MAX_Q queries
    char *s_q = smem;
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(smem + (size_t)NQ * S);
    uint2 *s_list = reinterpret_cast<uint2 *>(smem + (size_t)NQ * S + NQ * 4);   // [NQ][LCAP]

    // ---- query rows -> LDS (rows beyond n_q: zeros; their thresholds are NaN, nothing of theirs is ever recorded)
    const int chunks = row_bytes / 16;
    for (int i = tid; i < NQ * chunks; i += NARROW_THREADS) {
        const int q = i / chunks, c = i - q * chunks;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (q < nq_here) v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(a.Q) + (size_t)(q_base + q) * row_bytes + c * 16);
        *reinterpret_cast<uint4 *>(s_q + (size_t)q * S + c * 16) = v;
    }
    for (int i = tid; i < NQ; i += NARROW_THREADS) s_cnt[i] = 0u;
    __syncthreads();

    float thr[NQT], cqv[NQT];
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
        const int q = qt * 16 + l15;
        thr[qt] = (q < nq_here) ? a.thr[q_base + q] : __builtin_nanf("");
        cqv[qt] = (q < nq_here) ? a.cq[q_base + q] : 0.f;
    }

    // ---- this wave's stream: 16-row groups g = (i * gridDim.x + blockIdx.x) * 8 + wv, each KS = dim / 32 K-steps of 64 bytes per row,
    // walked in chunks of P steps (P divides KS): the loads of chunk c + 1 are issued while chunk c is consumed
    const int KS = a.dim / SUB_K;
    const int cpg = KS / P;                                            // chunks per group
    const int64_t n_groups = (a.n_rows + 15) / 16;
    const int64_t g_first = (int64_t)stream_id * NARROW_WAVES + wv;
    const int64_t g_step = (int64_t)n_streams * NARROW_WAVES;
    const int64_t my_groups = g_first < n_groups ? (n_groups - g_first + g_step - 1) / g_step : 0;
    const int64_t n_chunks = my_groups * cpg;
    const char *Dbytes = reinterpret_cast<const char *>(a.D);
    auto row_ptr = [&](int64_t group) -> const char * {
        int64_t row = group * 16 + l15;
        if (row > a.n_rows - 1) row = a.n_rows - 1;                     // tail group: clamped copies, dropped by the filter
        return Dbytes + row * row_bytes + lq * 16;
    };

    bf16x8 buf[P];
    int64_t lgroup = g_first;            // load stream position: group, chunk inside the group
    int lc = 0;
    const char *lp = my_groups > 0 ? row_ptr(lgroup) : Dbytes;
    if (n_chunks > 0) {
#pragma unroll
        for (int j = 0; j < P; ++j) buf[j] = stream_load<NT>(lp + j * 64);
        if (++lc == cpg) {
            lc = 0;
            lgroup += g_step;
            lp = row_ptr(lgroup < n_groups ? lgroup : g_first);
        } else {
            lp += P * 64;
        }
    }

    f32x4n acc[NQT];
    int cc = 0;                          // compute stream: chunk inside the group
    int64_t cgroup = g_first;
    const char *bq = s_q + (size_t)l15 * S + lq * 16;
    // the query fragments of K-step s: lane (column l15, chunk lq) reads 16 bytes of query row qt * 16 + l15.  They are read ONE STEP
    // AHEAD of the MFMAs that use them (bcur / bnext), so the LDS latency sits behind the previous step's matrix work
    auto read_b = [&](bf16x8 (&dst)[NQT], int kstep) {
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) dst[qt] = *reinterpret_cast<const bf16x8 *>(bq + (size_t)qt * 16 * S + kstep * 64);
    };
    bf16x8 bcur[NQT];
    read_b(bcur, 0);
    for (int64_t c = 0; c < n_chunks; ++c) {
        const bool more = c + 1 < n_chunks;       // wave-uniform
        const int kbase = cc * P;
        const int knext = (cc + 1 == cpg) ? 0 : kbase + P;   // first K-step of the next chunk
#pragma unroll
        for (int j = 0; j < P; ++j) {
            bf16x8 bnext[NQT];
            read_b(bnext, j + 1 < P ? kbase + j + 1 : knext);
            if (j == 0 && cc == 0) {
                const f32x4n z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int qt = 0; qt < NQT; ++qt) acc[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(buf[j], bcur[qt], z, 0, 0, 0);
            } else {
#pragma unroll
                for (int qt = 0; qt < NQT; ++qt) acc[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(buf[j], bcur[qt], acc[qt], 0, 0, 0);
            }
            // refill the slot just consumed (chunk c + 1, step j) RIGHT behind the MFMAs that read it: a rolling window of P loads.
            // UNCONDITIONAL (behind the last chunk `lp` names rows of the wave's first group again: P KiB read for nothing, once per
            // wave): a branch around the load makes hipcc count vmcnt for the path without it, which drains the window at the end of
            // every chunk.  The scheduling barrier keeps hipcc from collecting the P refills at the end of the chunk (a burst: the
            // window would run empty while the chunk is consumed).
            buf[j] = stream_load<NT>(lp + j * 64);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt) bcur[qt] = bnext[qt];
        }
        if (more) {
            if (++lc == cpg) {
                lc = 0;
                lgroup += g_step;
                lp = row_ptr(lgroup < n_groups ? lgroup : g_first);
            } else {
                lp += P * 64;
            }
        }
        if (++cc == cpg) {
            cc = 0;
            // ---- epilogue of a finished 16-row group.  C layout: lane -> query column l15, register e -> row 4 lq + e.
            const int64_t r0 = cgroup * 16;
            const float nt = narrow_uniform_f32(a.tile_norm + r0 / TILE_DOCS);
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt) {
                const float t = fmaf(-cqv[qt], nt, thr[qt]);    // a row can reach tau_q only if mfma + cq ||d|| >= tau_q, ||d|| <= nt
                const float m = fmaxf(fmaxf(acc[qt][0], acc[qt][1]), fmaxf(acc[qt][2], acc[qt][3]));
                if (__ballot(m >= t) != 0ull) {
                    if (m >= t) {   // rare, divergent
                        const int q = qt * 16 + l15;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = acc[qt][e];
                            const int64_t doc = r0 + 4 * lq + e;
                            if (v >= t && doc < a.n_rows) {
                                const uint32_t pos = atomicAdd(&s_cnt[q], 1u);
                                const uint2 rec = make_uint2(__float_as_uint(v), (uint32_t)doc);
                                if (pos < (uint32_t)LCAP) {
                                    s_list[q * LCAP + pos] = rec;
                                } else {   // the workgroup's staging list is full (a flooded list): straight to the candidate area
                                    const int sl = stream_id & 1;
                                    const uint32_t gp = atomicAdd(&a.cnt[(q_base + q) * 2 + sl], 1u);
                                    if (gp < (uint32_t)a.cap) a.cand[((int64_t)(q_base + q) * a.cap + gp) * 2 + sl] = rec;
                                }
                            }
                        }
                    }
                }
            }
            cgroup += g_step;
        }
    }

    // ---- flush: one block reservation per (workgroup, query), records go to sub-list (workgroup parity) of the query's cell
    __syncthreads();
    const int sl = stream_id & 1;
    for (int q = wv; q < NQ && q < nq_here; q += NARROW_WAVES) {
        uint32_t n = s_cnt[q];
        if (n > (uint32_t)LCAP) n = LCAP;
        if (n == 0) continue;                               // wave-uniform
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&a.cnt[(q_base + q) * 2 + sl], n);
        base = __shfl(base, 0);
        for (uint32_t i = lane; i < n; i += 64) {
            const uint32_t p = base + i;
            if (p < (uint32_t)a.cap) a.cand[((int64_t)(q_base + q) * a.cap + p) * 2 + sl] = s_list[q * LCAP + i];
        }
    }
}

size_t narrow_lds_bytes(int nqt, int dim) {
    const int S = narrow_query_stride(dim);
    return (size_t)nqt * 16 * S + (size_t)nqt * 16 * 4 + (size_t)nqt * 16 * narrow_lds_cap(nqt) * 8;
}

template <int NQT, int P>
static int launch_narrow_np(const NarrowArgs &a, int grid, bool nt, hipStream_t s) {
    const size_t lds = narrow_lds_bytes(NQT, a.dim);
    if (nt) {
        const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&narrow_filter_kernel<NQT, P, true>), 160 * 1024);
        if (rc != CCR_OK) return rc;
        hipLaunchKernelGGL((narrow_filter_kernel<NQT, P, true>), dim3(grid), dim3(NARROW_THREADS), lds, s, a);
    } else {
        const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&narrow_filter_kernel<NQT, P, false>), 160 * 1024);
        if (rc != CCR_OK) return rc;
        hipLaunchKernelGGL((narrow_filter_kernel<NQT, P, false>), dim3(grid), dim3(NARROW_THREADS), lds, s, a);
    }
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

template <int NQT>
static int launch_narrow_n(const NarrowArgs &a, int grid, bool nt, hipStream_t s) {
    const int KS = a.dim / SUB_K;
    // two query groups: every corpus row is pulled by two CUs (one of them from the L2 / the Infinity Cache), i.e. a CU moves twice the
    // bytes of the one-group launch -- twice the window (24 KiB per wave in flight) keeps the pull rate per CU from binding
    if (NQT == 4 && a.groups == 2 && KS % 24 == 0 && getenv("CCR_NARROW_P12") == nullptr) return launch_narrow_np<NQT, 24>(a, grid, nt, s);
    if (KS % 12 == 0) return launch_narrow_np<NQT, 12>(a, grid, nt, s);
    if (KS % 8 == 0) return launch_narrow_np<NQT, 8>(a, grid, nt, s);
    if (KS % 4 == 0) return launch_narrow_np<NQT, 4>(a, grid, nt, s);
    if (KS % 2 == 0) return launch_narrow_np<NQT, 2>(a, grid, nt, s);
    return launch_narrow_np<NQT, 1>(a, grid, nt, s);
}

int launch_narrow_filter(const NarrowArgs &a, int nqt, int grid, bool nt, hipStream_t s) {
    CCR_REQUIRE(a.dim % SUB_K == 0 && a.n_q >= 1 && (nqt == 1 || nqt == 2 || nqt == 4 || nqt == 6) && (a.groups == 1 || a.groups == 2) &&
                    (a.groups == 1 ? a.n_q <= nqt * 16 : (nqt == 4 && a.n_q > NARROW_MAX_Q && a.n_q <= 2 * NARROW_MAX_Q && grid % 16 == 0)),
                "narrow main pass: bad shape (internal)");
    CCR_REQUIRE(narrow_lds_bytes(nqt, a.dim) <= 160 * 1024, "narrow main pass: %zu bytes of LDS (internal)", narrow_lds_bytes(nqt, a.dim));
    if (nqt == 1) return launch_narrow_n<1>(a, grid, nt, s);
    if (nqt == 2) return launch_narrow_n<2>(a, grid, nt, s);
    if (nqt == 6) return launch_narrow_n<6>(a, grid, nt, s);
    return launch_narrow_n<4>(a, grid, nt, s);
}

}  // namespace ccr
