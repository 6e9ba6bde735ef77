// ccr_fused.hip -- the fused MFMA path: S^T = D . Q^T on the matrix cores with the top-k filter
// in the epilogue, so the [Q, N] score matrix never exists in HBM.
//
//   sample pass  (EPI_GMAX)  : score every `stride`-th 256-row corpus tile, reduce each 16-row MFMA
//                              fragment to its maximum -> gmax[group][query]
//   threshold    (threshold_kernel): tau_q = k-th largest group maximum (a lower bound of the k-th
//                              largest score: k disjoint groups hold a score >= tau_q), minus the
//                              MFMA error margins (cq * tile norm)
//   main pass    (EPI_FILTER): score the whole shard; accumulators >= thr_q are appended to the
//                              (range, query) candidate list (LDS counter, 8-byte records)
//   select       (select_rescore_kernel): per query radix-select the k-th largest MFMA score among
//                              the candidates, keep everything whose error interval reaches it, re-score those
//                              canonically (fp64 ordered) and sort by (score desc, id asc).
//
// GEMM geometry (gfx950): 256 docs x 256 queries per workgroup tile, 512 threads = 8 waves as
// 2 (doc halves) x 4 (query quarters); each wave owns 128 docs x 64 queries = 4 x 2 tiles of
// v_mfma_f32_32x32x16_bf16 (docs on the accumulator rows/registers, queries on the lanes, so a lane
// compares its 16 registers against ONE per-lane threshold).  Staging: global_load_lds_dwordx4 into a
// ring of four 32-KiB sub-stages (32 K elements = 64-B rows), 16-byte chunk index XOR-swizzled with
// (row>>2)&3 on the SOURCE address and on the fragment read (conflict-free ds_read_b128).
#include <stdlib.h>

#include <mutex>
#include <set>
#include <utility>

#include "ccr_gemm_common.h"
#include "ccr_topk_device.h"

namespace ccr {

// 16 zero bytes: what the LDS-DMA reads for the K chunks of the LAST sub-stage that lie beyond the row end when dim % 32 != 0
// (TAIL instantiations; rows are dim * 2 bytes, dim % 8 == 0, so a 16-byte chunk is wholly inside or wholly outside a row)
__device__ __attribute__((aligned(16))) uint32_t g_zero_chunk[4];

// Wave-uniform 4-byte load through the scalar cache (lgkmcnt, not vmcnt), complete on return.
__device__ __forceinline__ float load_uniform_f32(const float *p) {
    float v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// =============================================================================================
// GEMM + top-k filter kernel.  Ping-pong schedule: K is walked in 32-element sub-stages through a
// ring of four 32-KiB LDS buffers (up to 3 sub-stages of LDS-DMA in flight, counted vmcnt, raw
// s_barrier).  The two waves that share a SIMD (wave w and w+4) run one barrier interval apart:
// while one executes its 16 MFMAs, the other reads its next operands from LDS, issues the next
// sub-stage's DMA and runs the top-k filter of a finished tile (VALU beside the partner's MFMAs).
//   per wave and sub-stage u:
//     mem(u):  [filter of a finished tile] wait OWN DMA of u+1 | 12 ds_read_b128 of u | DMA of u+3 | lgkmcnt(0)
//     barrier A_u | 16 x v_mfma_f32_32x32x16_bf16 | barrier B_u
// LDS ring protocol (why this is race free):
//   RAW  a wave confirms (vmcnt) its own DMA of sub-stage u in mem(u-1), i.e. before its barrier
//        A_{u-1}; every reader of u starts mem(u) after a later barrier instance.
//   WAR  DMA of u+3 overwrites the buffer of u-1.  It is issued in mem(u); every wave's reads of
//        u-1 were retired (lgkmcnt(0)) before its barrier A_{u-1}, and for both groups that
//        barrier instance precedes every mem(u).

// s_waitcnt vmcnt(N) with a compile-time N
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define CCR_STAMP(idx)                                   \
    if constexpr ((DBG & 16) != 0) {                     \
        __builtin_amdgcn_sched_barrier(0);               \
        const unsigned long long _t = stamp();           \
        __builtin_amdgcn_sched_barrier(0);               \
        seg[idx] += _t - tprev;                          \
        tprev = _t;                                      \
    }

// DBG (compile-time, diagnostic instantiations only; results are WRONG when non-zero):
//   1 corpus rows always tile 0, 2 query slice always 0, 4 no DMA, 8 no MFMA, 16 cycle stamps
template <int EPI, bool STAGGER, int DBG, bool TAIL = false>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_topk_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wd = wv >> 2;   // doc half of the tile
    const int wq = wv & 3;    // query quarter of the tile
    const int l31 = lane & 31;
    const int h = lane >> 5;
    const bool g1 = STAGGER && (wv >= 4);  // the trailing half of the ping-pong (wave-uniform)
    const int KS2 = TAIL ? (a.dim + SUB_K - 1) / SUB_K : a.dim / SUB_K;   // TAIL: the last sub-stage is partly zero-filled
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // DBG 16: cycles per segment of the inner loop
    unsigned long long tprev = 0;
    if constexpr ((DBG & 16) != 0) tprev = stamp();
    if constexpr ((DBG & 4) != 0) {  // no-DMA ablation: zero operands (scores 0 stay below the thresholds)
        for (int i = tid; i < RING * SUB_BYTES / 16; i += GEMM_THREADS) reinterpret_cast<uint4 *>(smem)[i] = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }

    // DMA role: 4 x 1-KiB pieces per sub-stage (16 rows x 64 B each); LDS image is lane-linear, the
    // 16-byte chunk swizzle (chunk ^ (row>>2)&3) is applied on the SOURCE address and on the read
    const int srow = wv * 16 + (lane >> 2);  // + piece*128
    const int schunk = (lane & 3) ^ ((srow >> 2) & 3);
    const int swz = (lane >> 2) & 3;
    int cofs[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) cofs[ks] = (((2 * ks + h) ^ swz) << 4);
    const int a_base = (wd * 128 + l31) * 64;                // + dt*2048
    const int b_base = SUB_Q_REGION + (wq * 64 + l31) * 64;  // + qt*2048

    // Work items (range r, query block qb).  Blocks b and b+8 share an XCD (its 4-MiB L2), so XCD x takes
    // query-block group x % qgroups (its query rows stay L2-resident) and the ranges r = x / qgroups
    // (mod 8 / qgroups); co-resident workgroups walk the same corpus tiles for different query blocks.
    const int xcd = blockIdx.x & (NUM_XCD - 1);
    const int jx = blockIdx.x >> 3;
    const int per_x = gridDim.x >> 3;
    const int qg = xcd % a.qgroups;
    const int rc = xcd / a.qgroups;
    const int nrc = NUM_XCD / a.qgroups;           // range classes
    const int qb_per = a.qblocks / a.qgroups;      // query blocks of this XCD (qgroups divides qblocks)
    const int count_x = (a.ranges / nrc) * qb_per;   // items of this XCD set over the whole pass
    const int item_end = a.item_end < count_x ? a.item_end : count_x;

    for (int item = a.item_begin + jx; item < item_end; item += per_x) {
        // item order: consecutive items (the co-resident workgroups of an XCD) share a RANGE and walk its corpus tiles for
        // different query blocks; item_swap (experiment, single-launch plans): they share the QUERY BLOCK instead
        const int n_rl = a.ranges / nrc;
        const int rl = a.item_swap ? item % n_rl : item / qb_per;
        const int qb = qg * qb_per + (a.item_swap ? item / n_rl : item % qb_per);
        const int r = rc + nrc * rl;
        const int64_t ntile = (a.n_vt - r + a.ranges - 1) / a.ranges;
        if (ntile <= 0) continue;
        const int q0 = qb * TILE_Q;

        // Each lane owns ONE (query, lane-half, wave-row) candidate sub-list per query column, so the
        // append counter is a plain register: no atomics anywhere in the filter.
        float thr[2] = {0.f, 0.f}, cqv[2] = {0.f, 0.f};
        uint32_t ncand[2] = {0u, 0u};
        uint2 *clist[2] = {nullptr, nullptr};
        int cap = 0;
        if (EPI == EPI_FILTER) {
            int seg_r0;
            long long seg_base;
            cand_segment(a.lay, r, cap, seg_r0, seg_base);   // this range's segment of the candidate area (wave-uniform)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const int q = q0 + wq * 64 + qt * 32 + l31;
                thr[qt] = (q < a.n_q) ? a.thr[q] : __builtin_nanf("");
                cqv[qt] = (q < a.n_q) ? a.cq[q] : 0.f;
                clist[qt] = a.cand + seg_base + ((int64_t)(r - seg_r0) * a.nq_pad + q) * 4 * cap + (wd * 2 + h);   // slot-major cell
            }
            // make hipcc wait for the threshold loads HERE, before any LDS-DMA is in flight: its own
            // wait at the first use inside the tile epilogue would be vmcnt(0) and drain the DMA ring
            asm volatile("" : "+v"(thr[0]), "+v"(thr[1]), "+v"(cqv[0]), "+v"(cqv[1]));
        }
        const uint16_t *qsrc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int qrow = q0 + i * 128 + srow;
            if (qrow > a.n_q - 1) qrow = a.n_q - 1;
            qsrc[i] = a.Q + (int64_t)qrow * a.dim + schunk * 8;
        }

        f32x16 acc[4][2];
        const int64_t U = ntile * KS2;

        // ---- DMA issue stream (runs up to 3 sub-stages ahead of the consumer).  Per-thread source pointers
        // are recomputed once per tile; a sub-stage adds only the K offset.
        int64_t iu = 0, it = 0;
        int iks = 0;
        const uint16_t *dsrc[2];
        auto tile_ptrs = [&]() {
            const int64_t row0 = (DBG & 1) ? 0 : (r + it * a.ranges) * a.tile_stride * TILE_DOCS;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int64_t drow = row0 + i * 128 + srow;
                if (drow > a.n_rows - 1) drow = a.n_rows - 1;
                dsrc[i] = a.D + drow * a.dim + schunk * 8;
            }
        };
        tile_ptrs();
        auto issue = [&]() {
            char *buf = smem + (int)(iu & (RING - 1)) * SUB_BYTES;
            const int k0 = iks * SUB_K;
            if constexpr (!(DBG & 4)) {
                if (TAIL && iks == KS2 - 1) {   // wave-uniform: chunks beyond the row end come from the zero chunk
                    const bool in = k0 + schunk * 8 < a.dim;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        glds16(in ? (const void *)(dsrc[i] + k0) : (const void *)g_zero_chunk, buf + (i * 512 + wv * 64) * 16);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        glds16(in ? (const void *)(qsrc[i] + k0) : (const void *)g_zero_chunk, buf + SUB_Q_REGION + (i * 512 + wv * 64) * 16);
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i) glds16(dsrc[i] + k0, buf + (i * 512 + wv * 64) * 16);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        glds16(qsrc[i] + ((DBG & 2) ? 0 : k0), buf + SUB_Q_REGION + (i * 512 + wv * 64) * 16);
                }
            }
            ++iu;
            if (++iks == KS2) {
                iks = 0;
                ++it;
                tile_ptrs();
            }
        };

        // ---- epilogue of a finished 256x256 tile (accumulators still live)
        // C layout of v_mfma_f32_32x32x16: lane -> query column (lane & 31); register e -> corpus row
        // (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) of the 32-row MFMA tile.
        auto epilogue = [&](int64_t vt, float nt) {   // nt: norm bound of the tile's rows (wave-uniform)
            const int64_t row_base = vt * a.tile_stride * TILE_DOCS + wd * 128 + 4 * h;
            const uint32_t row32 = (uint32_t)row_base;
            const int64_t left = a.n_rows - row_base;
            const uint32_t rows_left = left <= 0 ? 0u : (left > 128 ? 128u : (uint32_t)left);   // rows of this block inside the shard
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                float sub[4][4], mdt[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        sub[dt][g] = fmaxf(fmaxf(acc[dt][qt][4 * g], acc[dt][qt][4 * g + 1]),
                                           fmaxf(acc[dt][qt][4 * g + 2], acc[dt][qt][4 * g + 3]));
                    mdt[dt] = fmaxf(fmaxf(sub[dt][0], sub[dt][1]), fmaxf(sub[dt][2], sub[dt][3]));
                }
                if constexpr ((DBG & 16) != 0) {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) asm volatile("" : "+v"(mdt[dt]));
                    CCR_STAMP(6)  // max trees (the hit path is charged to segment 1)
                }
                if (EPI == EPI_GMAX) {
                    const int ql = wq * 64 + qt * 32 + l31;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
                        a.gmax[(vt * GROUPS_PER_TILE + wd * 8 + dt * 2 + h) * a.nq_pad + q0 + ql] = mdt[dt];
                } else if (EPI == EPI_FILTER) {
                    // a row of this tile can reach tau_q only if mfma + cq * ||d|| >= tau_q, and ||d|| <= nt
                    const float t = fmaf(-cqv[qt], nt, thr[qt]);
                    const float mall = fmaxf(fmaxf(mdt[0], mdt[1]), fmaxf(mdt[2], mdt[3]));
                    if (__ballot(mall >= t) != 0ull) {
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) {
                            if (__ballot(mdt[dt] >= t) == 0ull) continue;
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                if (sub[dt][g] >= t) {  // rare, divergent: <= 4 rows to test
#pragma unroll
                                    for (int e2 = 0; e2 < 4; ++e2) {
                                        const float v = acc[dt][qt][4 * g + e2];
                                        // 32-bit row arithmetic (local rows < 2^32; rows_left = n_rows - first row of this
                                        // lane's block, clamped): the hit path is instruction-bound beside the partner's MFMAs
                                        const uint32_t off = (uint32_t)(dt * 32 + 8 * g + e2);
                                        if (v >= t && off < rows_left) {
                                            if (ncand[qt] < (uint32_t)cap)
                                                clist[qt][ncand[qt] * 4] = make_uint2(__float_as_uint(v), row32 + off);
                                            ++ncand[qt];
                                        }
                                    }
                                }
                            }
                        }
                    }
                } else {  // EPI_STORE
                    const int q = q0 + wq * 64 + qt * 32 + l31;
                    const int64_t pitch = a.store_pitch ? a.store_pitch : a.n_rows;
                    if (q < a.n_q) {
                        float *dst = a.store + (int64_t)q * pitch;
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {   // registers 4g .. 4g+3 are four consecutive corpus rows
                                const int64_t doc = row_base + dt * 32 + 8 * g;
                                if ((pitch & 3) == 0 && doc + 3 < a.n_rows) {
                                    *reinterpret_cast<float4 *>(dst + doc) = make_float4(acc[dt][qt][4 * g], acc[dt][qt][4 * g + 1],
                                                                                         acc[dt][qt][4 * g + 2], acc[dt][qt][4 * g + 3]);
                                } else {
#pragma unroll
                                    for (int e2 = 0; e2 < 4; ++e2)
                                        if (doc + e2 < a.n_rows) dst[doc + e2] = acc[dt][qt][4 * g + e2];
                                }
                            }
                    }
                }
            }
        };

        // ---- prologue: up to 3 sub-stages in flight, sub-stage 0 confirmed and published
        const int npro = U < 3 ? (int)U : 3;
        for (int i = 0; i < npro; ++i) issue();
        if (npro == 3)
            CCR_WAIT_VM(8);
        else if (npro == 2)
            CCR_WAIT_VM(4);
        else
            CCR_WAIT_VM(0);
        CCR_BARRIER();
        if (g1) CCR_BARRIER();

        int cks = 0;
        int64_t ct = 0;
        bool pending = false;
        int64_t pending_vt = 0;
        float pending_nt = 0.f;
        if constexpr ((DBG & 16) != 0) tprev = stamp();
        for (int64_t u = 0; u < U; ++u) {
            // ================= mem phase
            CCR_STAMP(0)  // barrier B wait (+ loop overhead)
            if (pending) {
                epilogue(pending_vt, pending_nt);
                pending = false;
            }
            CCR_STAMP(1)  // tile epilogue
            if (u + 1 < U) {  // confirm OWN DMA of sub-stage u+1 (published by the barrier below)
                if (u + 2 < U)
                    CCR_WAIT_VM(4);
                else
                    CCR_WAIT_VM(0);
            }
            CCR_STAMP(2)  // DMA wait
            const char *buf = smem + (int)(u & (RING - 1)) * SUB_BYTES;
            bf16x8 af[2][4], bfr[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    af[ks][dt] = *reinterpret_cast<const bf16x8 *>(buf + a_base + dt * 2048 + cofs[ks]);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
                    bfr[ks][qt] = *reinterpret_cast<const bf16x8 *>(buf + b_base + qt * 2048 + cofs[ks]);
            }
            if (u + 3 < U) issue();  // into the buffer of sub-stage u-1 (see the WAR note above)
            CCR_WAIT_LGKM0();        // operands in registers BEFORE the barrier; free behind the DMA issue
            CCR_STAMP(3)  // LDS reads + DMA issue
            CCR_BARRIER();
            CCR_STAMP(4)  // barrier A wait
            // ================= mfma phase
            if constexpr ((DBG & 8) != 0) {  // no matrix work: keep the operands alive, skip the MFMAs
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) asm volatile("" ::"v"(af[ks][dt]));
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt) asm volatile("" ::"v"(bfr[ks][qt]));
                }
                if (cks == 0) {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                            for (int e = 0; e < 16; ++e) acc[dt][qt][e] = -1e30f;
                }
            } else {
                // no s_setprio(1) here: the partner wave's mem phase carries VALU work (filter, addresses) that a
                // raised MFMA wave would starve (MI355X_MICROARCH 'Two waves per SIMD', item 2)
                if (cks == 0) {  // first sub-stage of a tile: C = 0 (no accumulator clearing pass)
                    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                        for (int qt = 0; qt < 2; ++qt)
                            acc[dt][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][dt], bfr[0][qt], z, 0, 0, 0);
                } else {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                        for (int qt = 0; qt < 2; ++qt)
                            acc[dt][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][dt], bfr[0][qt], acc[dt][qt], 0, 0, 0);
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt)
                        acc[dt][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][dt], bfr[1][qt], acc[dt][qt], 0, 0, 0);
            }
            CCR_STAMP(5)  // MFMA phase
            if (++cks == KS2) {
                cks = 0;
                // Tile finished.  The leading group filters it at the start of its next mem phase; the trailing
                // group (one barrier interval behind) filters it right here, so that BOTH filters fall into the
                // same interval instead of stalling the partner twice per tile.
                // (the tile's norm bound: a SCALAR load + wait right here, behind the MFMAs just issued -- a vector load would
                // sit in vmcnt among the LDS-DMA pieces and hipcc's own wait for it would drain the ring)
                const int64_t vt_done = r + ct * a.ranges;
                const float nt_done = EPI == EPI_FILTER ? load_uniform_f32(a.tile_norm + vt_done * a.tile_stride) : 0.f;
                if (g1) {
                    epilogue(vt_done, nt_done);
                } else {
                    pending = true;
                    pending_vt = vt_done;
                    pending_nt = nt_done;
                }
                ++ct;
            }
            CCR_BARRIER();
        }
        if (pending) epilogue(pending_vt, pending_nt);
        if (STAGGER && !g1) CCR_BARRIER();  // every wave executes the same number of barriers

        if (EPI == EPI_FILTER) {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
                a.cnt[((int64_t)r * a.nq_pad + q0 + wq * 64 + qt * 32 + l31) * 4 + wd * 2 + h] = ncand[qt];
        }
        __syncthreads();  // LDS ring free for the next item
    }
    if constexpr ((DBG & 16) != 0) {
        if (lane == 0 && a.store) {
            unsigned long long *o = reinterpret_cast<unsigned long long *>(a.store) + ((size_t)blockIdx.x * 8 + wv) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = seg[i];
        }
    }
}


// =============================================================================================
// Variant of the main-pass kernel on v_mfma_f32_16x16x32_bf16 (same tile, same ring, same ping-pong).
// The chip holds a higher clock on this MFMA shape in power-limited loops (MI355X_MICROARCH 'DVFS
// give-back' item 7), so it is A/B-ed against the 32x32x16 kernel by wall time (CCR_MFMA16).
//   wave tile 128 x 64 = 8 x 4 MFMA tiles, one K step (32) per sub-stage: 32 MFMAs, 8 + 4 ds_read_b128
//   C layout: lane -> query column (lane & 15), register e -> corpus row 4 * (lane >> 4) + e of the 16-row tile
//   -> a lane owns 4 query columns x 32 rows; 8 candidate sub-lists per (range, query): (wave row, lane >> 4)
//   LDS image: 64-B rows, 16-byte chunk c of row r stored at chunk c ^ (((r >> 2) & 1) << 1) (conflict-free for
//   the 16x16 fragment read pattern; brute-forced over all xor tables).
typedef float f32x4v __attribute__((ext_vector_type(4)));

// Maximum of a 16x16 MFMA tile's four registers in two VALU instructions (fmaxf() costs four: hipcc canonicalises the operands first).
// Every VALU instruction of the filter counts: it is issued while the partner wave of the SIMD runs its MFMAs and gets a slot only now
// and then (in-kernel stamps: the tile epilogue was 17 % of a wave's time with fmaxf trees and the first hit path).
__device__ __forceinline__ float max4_asm(const f32x4v &c) {
    float m;
    asm("v_max3_f32 %0, %1, %2, %3\n\tv_max_f32 %0, %0, %4" : "=&v"(m) : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]));
    return m;
}
__device__ __forceinline__ float max3_asm(float a, float b, float c) {
    float m;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(a), "v"(b), "v"(c));
    return m;
}

template <int EPI, int DBG, bool TAIL = false>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_topk16_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wd = wv >> 2;
    const int wq = wv & 3;
    const int l15 = lane & 15;
    const int lq = lane >> 4;  // 0..3: K chunk of the operand fragments, row quad of the accumulator
    const bool g1 = (wv >= 4);
    const int KS2 = TAIL ? (a.dim + SUB_K - 1) / SUB_K : a.dim / SUB_K;
    // DBG (diagnostic instantiations only, WRONG results): 4 no DMA at all, 8 no MFMA, 32 corpus-only DMA (the query region of the ring stays
    // zero), 64 query-only DMA (the corpus region stays zero), 128 thresholds +inf (the complete kernel without a single hit: the baseline
    // of the others, whose zero scores never pass), 256 every DMA piece reads whole 128-byte lines (8 rows x 128 B; garbage operands: use 384 = 256 + 128);
    // the counted waits follow the pieces a wave issues per sub-stage
    constexpr bool DMA_D = !(DBG & 4) && !(DBG & 64);
    constexpr bool DMA_Q = !(DBG & 4) && !(DBG & 32);
    constexpr int PIECES = (DMA_D ? 2 : 0) + (DMA_Q ? 2 : 0);
    if constexpr ((DBG & (4 | 32 | 64)) != 0) {
        for (int i = tid; i < RING * SUB_BYTES / 16; i += GEMM_THREADS) reinterpret_cast<uint4 *>(smem)[i] = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }

    const int srow = wv * 16 + (lane >> 2);  // + piece*128
    const int schunk = (lane & 3) ^ (((srow >> 2) & 1) << 1);
    const int cofs = ((lq ^ (((lane >> 2) & 1) << 1)) << 4);
    const int a_base = (wd * 128 + l15) * 64 + cofs;                // + dt*1024
    const int b_base = SUB_Q_REGION + (wq * 64 + l15) * 64 + cofs;  // + qt*1024

    const int xcd = blockIdx.x & (NUM_XCD - 1);
    const int jx = blockIdx.x >> 3;
    const int per_x = gridDim.x >> 3;
    const int qg = xcd % a.qgroups;
    const int rc = xcd / a.qgroups;
    const int nrc = NUM_XCD / a.qgroups;
    const int qb_per = a.qblocks / a.qgroups;
    const int count_x = (a.ranges / nrc) * qb_per;   // items of this XCD set over the whole pass
    const int item_end = a.item_end < count_x ? a.item_end : count_x;

    for (int item = a.item_begin + jx; item < item_end; item += per_x) {
        // item order: consecutive items (the co-resident workgroups of an XCD) share a RANGE and walk its corpus tiles for
        // different query blocks; item_swap (experiment, single-launch plans): they share the QUERY BLOCK instead
        const int n_rl = a.ranges / nrc;
        const int rl = a.item_swap ? item % n_rl : item / qb_per;
        const int qb = qg * qb_per + (a.item_swap ? item / n_rl : item % qb_per);
        const int r = rc + nrc * rl;
        const int64_t ntile = (a.n_vt - r + a.ranges - 1) / a.ranges;
        if (ntile <= 0) continue;
        const int q0 = qb * TILE_Q;

        float thr[4] = {0.f, 0.f, 0.f, 0.f}, cqv[4] = {0.f, 0.f, 0.f, 0.f};
        uint32_t ncand[4] = {0u, 0u, 0u, 0u};
        uint2 *clist[4] = {nullptr, nullptr, nullptr, nullptr};
        int cap = 0;
        if (EPI == EPI_FILTER) {
            int seg_r0;
            long long seg_base;
            cand_segment(a.lay, r, cap, seg_r0, seg_base);   // this range's segment of the candidate area (wave-uniform)
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                const int q = q0 + wq * 64 + qt * 16 + l15;
                thr[qt] = (q < a.n_q) ? a.thr[q] : __builtin_nanf("");
                cqv[qt] = (q < a.n_q) ? a.cq[q] : 0.f;
                clist[qt] = a.cand + seg_base + ((int64_t)(r - seg_r0) * a.nq_pad + q) * 8 * cap + (wd * 4 + lq);   // slot-major cell
            }
            if constexpr ((DBG & 128) != 0) {   // everything runs, nothing passes the filter: the baseline of the ablations
#pragma unroll
                for (int qt = 0; qt < 4; ++qt) thr[qt] = INFINITY;
            }
            asm volatile("" : "+v"(thr[0]), "+v"(thr[1]), "+v"(thr[2]), "+v"(thr[3]), "+v"(cqv[0]), "+v"(cqv[1]), "+v"(cqv[2]), "+v"(cqv[3]));
        }
        const uint16_t *qsrc[(DBG & 256) ? 4 : 2];
#pragma unroll
        for (int i = 0; i < ((DBG & 256) ? 4 : 2); ++i) {
            int qrow = q0 + ((DBG & 256) ? i * 64 + wv * 8 + (lane >> 3) : i * 128 + srow);
            if (qrow > a.n_q - 1) qrow = a.n_q - 1;
            if constexpr ((DBG & 1024) != 0) {
                const int last = (int)(((int64_t)(a.dbg_alloc_q > a.n_q ? a.dbg_alloc_q : a.n_q) * a.dim - a.dim) / a.dbg_pitch);
                if (qrow > last) qrow = last;
                qsrc[i] = a.Q + (int64_t)qrow * a.dbg_pitch + schunk * 8;
            } else
            qsrc[i] = a.Q + (int64_t)qrow * a.dim + ((DBG & 256) ? (lane & 7) : schunk) * 8;
        }

        f32x4v acc[8][4];
        const int64_t U = ntile * KS2;

        int64_t iu = 0, it = 0;
        int iks = 0;
        const uint16_t *dsrc[(DBG & 256) ? 4 : 2];
        auto tile_ptrs = [&]() {
            const int64_t row0 = (r + it * a.ranges) * a.tile_stride * TILE_DOCS;
#pragma unroll
            for (int i = 0; i < ((DBG & 256) ? 4 : 2); ++i) {
                // DBG 256 (timing only): a piece reads 8 rows x 128 B (whole cache lines) instead of 16 rows x 64 B; the LDS image is garbage
                int64_t drow = row0 + ((DBG & 256) ? i * 64 + wv * 8 + (lane >> 3) : i * 128 + srow);
                if (drow > a.n_rows - 1) drow = a.n_rows - 1;
                if constexpr ((DBG & 1024) != 0) {   // timing only: the traffic of rows at another PITCH (a.dbg_pitch elements), kept inside the array
                    // (a.dbg_alloc_rows: rows of `dim` elements the array behind D really holds -- the experiment hands the index the first
                    // n_rows = alloc * dim / pitch rows of a larger array, so that no address is clamped)
                    const int64_t last = ((a.dbg_alloc_rows > a.n_rows ? a.dbg_alloc_rows : a.n_rows) * a.dim - a.dim) / a.dbg_pitch;
                    if (drow > last) drow = last;
                    dsrc[i] = a.D + drow * a.dbg_pitch + schunk * 8;
                } else
                dsrc[i] = a.D + drow * a.dim + ((DBG & 256) ? (lane & 7) : schunk) * 8;
            }
        };
        tile_ptrs();
        auto issue = [&]() {
            char *buf = smem + (int)(iu & (RING - 1)) * SUB_BYTES;
            const int k0 = iks * SUB_K;
            if (TAIL && iks == KS2 - 1) {   // wave-uniform: chunks beyond the row end come from the zero chunk
                const bool in = k0 + schunk * 8 < a.dim;
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    glds16(in ? (const void *)(dsrc[i] + k0) : (const void *)g_zero_chunk, buf + (i * 512 + wv * 64) * 16);
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    glds16(in ? (const void *)(qsrc[i] + k0) : (const void *)g_zero_chunk, buf + SUB_Q_REGION + (i * 512 + wv * 64) * 16);
            } else if constexpr ((DBG & 512) != 0) {
                // DBG 512 (timing only): the traffic of a K-TILED layout -- the 16 KiB a sub-stage takes from the corpus tile (and from the
                // query block) are ONE contiguous block, every piece one contiguous KiB: piece (iks * 16 + i * 8 + wv) of the tile's /
                // block's 384 contiguous KiB in the row-major arrays.  Every byte is still requested exactly once; operands are garbage.
                const int64_t row0 = (r + it * a.ranges) * a.tile_stride * TILE_DOCS;
                const int64_t dtile = (int64_t)(a.n_rows - row0 < TILE_DOCS ? a.n_rows - row0 : TILE_DOCS) * a.dim * 2;   // bytes of this tile
                const int64_t qblk = (int64_t)(a.n_q - q0 < TILE_Q ? a.n_q - q0 : TILE_Q) * a.dim * 2;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    int64_t o = (int64_t)(iks * 16 + i * 8 + wv) * 1024;
                    if (o > dtile - 1024) o = dtile - 1024;
                    glds16(reinterpret_cast<const char *>(a.D + row0 * a.dim) + o + lane * 16, buf + (i * 512 + wv * 64) * 16);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    int64_t o = (int64_t)(iks * 16 + i * 8 + wv) * 1024;
                    if (o > qblk - 1024) o = qblk - 1024;
                    glds16(reinterpret_cast<const char *>(a.Q + (int64_t)q0 * a.dim) + o + lane * 16, buf + SUB_Q_REGION + (i * 512 + wv * 64) * 16);
                }
            } else if constexpr ((DBG & 256) != 0) {
                // even sub-stage: rows 0 .. 127 of the tile, K [k0, k0 + 64); odd: rows 128 .. 255, the same K range: every byte of the tile
                // is still requested exactly once, now as whole 128-byte lines
                const int ofs = (iks & ~1) * SUB_K, hf = (iks & 1) * 2;
#pragma unroll
                for (int i = 0; i < 2; ++i) glds16((hf ? dsrc[(i + 2) & 3] : dsrc[i]) + ofs, buf + (i * 512 + wv * 64) * 16);
#pragma unroll
                for (int i = 0; i < 2; ++i) glds16((hf ? qsrc[(i + 2) & 3] : qsrc[i]) + ofs, buf + SUB_Q_REGION + (i * 512 + wv * 64) * 16);
            } else {
                // DBG 2048 / 4096 / 8192 (timing AND results stay right): cache policy sc1 / sc0 / sc0 sc1 on every piece
                constexpr int AUX = ((DBG & 2048) ? 16 : 0) | ((DBG & 4096) ? 1 : 0) | ((DBG & 8192) ? 17 : 0);
                if constexpr (DMA_D) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) glds16_aux<AUX>(dsrc[i] + k0, buf + (i * 512 + wv * 64) * 16);
                }
                if constexpr (DMA_Q) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) glds16_aux<AUX>(qsrc[i] + k0, buf + SUB_Q_REGION + (i * 512 + wv * 64) * 16);
                }
            }
            ++iu;
            if (++iks == KS2) {
                iks = 0;
                ++it;
                tile_ptrs();
            }
        };

        auto epilogue = [&](int64_t vt, float nt) {   // nt: norm bound of the tile's rows (wave-uniform)
            const int64_t row_base = vt * a.tile_stride * TILE_DOCS + wd * 128 + 4 * lq;  // + dt*16 + e
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                if (EPI == EPI_FILTER) {
                    float sub[8];
#pragma unroll
                    for (int dt = 0; dt < 8; ++dt) sub[dt] = max4_asm(acc[dt][qt]);   // (two VALU instructions per tile instead of fmaxf's four)
                    const float t = fmaf(-cqv[qt], nt, thr[qt]);   // per-tile margin: mfma + cq * ||d|| >= tau_q with ||d|| <= nt
                    float mall = max3_asm(sub[0], sub[1], sub[2]);
                    mall = max3_asm(mall, sub[3], sub[4]);
                    mall = max3_asm(mall, sub[5], sub[6]);
                    mall = fmaxf(mall, sub[7]);
                    if (__ballot(mall >= t) != 0ull) {
#pragma unroll
                        for (int dt = 0; dt < 8; ++dt) {
                            if (sub[dt] >= t) {  // rare, divergent: 4 rows to test
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const float v = acc[dt][qt][e];
                                    const int64_t doc = row_base + dt * 16 + e;
                                    if (v >= t && doc < a.n_rows) {
                                        if (ncand[qt] < (uint32_t)cap)
                                            clist[qt][ncand[qt] * 8] = make_uint2(__float_as_uint(v), (uint32_t)doc);
                                        ++ncand[qt];
                                    }
                                }
                            }
                        }
                    }
                } else {  // EPI_STORE
                    const int q = q0 + wq * 64 + qt * 16 + l15;
                    const int64_t pitch = a.store_pitch ? a.store_pitch : a.n_rows;
                    if (q < a.n_q) {
                        float *dst = a.store + (int64_t)q * pitch;
#pragma unroll
                        for (int dt = 0; dt < 8; ++dt) {   // the four registers of a tile are four consecutive corpus rows
                            const int64_t doc = row_base + dt * 16;
                            if ((pitch & 3) == 0 && doc + 3 < a.n_rows) {
                                *reinterpret_cast<float4 *>(dst + doc) =
                                    make_float4(acc[dt][qt][0], acc[dt][qt][1], acc[dt][qt][2], acc[dt][qt][3]);
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (doc + e < a.n_rows) dst[doc + e] = acc[dt][qt][e];
                            }
                        }
                    }
                }
            }
        };

        const int npro = U < 3 ? (int)U : 3;
        for (int i = 0; i < npro; ++i) issue();
        if (npro == 3)
            wait_vm<2 * PIECES>();
        else if (npro == 2)
            wait_vm<PIECES>();
        else
            CCR_WAIT_VM(0);
        CCR_BARRIER();
        if (g1) CCR_BARRIER();

        int cks = 0;
        int64_t ct = 0;
        bool pending = false;
        int64_t pending_vt = 0;
        float pending_nt = 0.f;
        for (int64_t u = 0; u < U; ++u) {
            if (pending) {
                epilogue(pending_vt, pending_nt);
                pending = false;
            }
            if (u + 1 < U) {
                if (u + 2 < U)
                    wait_vm<PIECES>();
                else
                    CCR_WAIT_VM(0);
            }
            const char *buf = smem + (int)(u & (RING - 1)) * SUB_BYTES;
            bf16x8 af[8], bfr[4];
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) af[dt] = *reinterpret_cast<const bf16x8 *>(buf + a_base + dt * 1024);
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) bfr[qt] = *reinterpret_cast<const bf16x8 *>(buf + b_base + qt * 1024);
            if (u + 3 < U) issue();
            CCR_WAIT_LGKM0();
            CCR_BARRIER();
            if constexpr ((DBG & 8) != 0) {   // no matrix work: keep the operands alive, skip the MFMAs
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) asm volatile("" ::"v"(af[dt]));
#pragma unroll
                for (int qt = 0; qt < 4; ++qt) asm volatile("" ::"v"(bfr[qt]));
                if (cks == 0) {
#pragma unroll
                    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
                        for (int qt = 0; qt < 4; ++qt)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[dt][qt][e] = -1e30f;
                }
#pragma unroll
                for (int dt = 0; dt < 8; ++dt)
#pragma unroll
                    for (int qt = 0; qt < 4; ++qt) asm volatile("" : "+v"(acc[dt][qt]));   // opaque: the filter trees stay
            } else if (cks == 0) {
                const f32x4v z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dt = 0; dt < 8; ++dt)
#pragma unroll
                    for (int qt = 0; qt < 4; ++qt)
                        acc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[dt], bfr[qt], z, 0, 0, 0);
            } else {
#pragma unroll
                for (int dt = 0; dt < 8; ++dt)
#pragma unroll
                    for (int qt = 0; qt < 4; ++qt)
                        acc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[dt], bfr[qt], acc[dt][qt], 0, 0, 0);
            }
            if (++cks == KS2) {
                cks = 0;
                // (the tile's norm bound: a SCALAR load + wait right here, behind the MFMAs just issued -- a vector load would
                // sit in vmcnt among the LDS-DMA pieces and hipcc's own wait for it would drain the ring)
                const int64_t vt_done = r + ct * a.ranges;
                const float nt_done = EPI == EPI_FILTER ? load_uniform_f32(a.tile_norm + vt_done * a.tile_stride) : 0.f;
                if (g1) {
                    epilogue(vt_done, nt_done);
                } else {
                    pending = true;
                    pending_vt = vt_done;
                    pending_nt = nt_done;
                }
                ++ct;
            }
            CCR_BARRIER();
        }
        if (pending) epilogue(pending_vt, pending_nt);
        if (!g1) CCR_BARRIER();

        if (EPI == EPI_FILTER) {
#pragma unroll
            for (int qt = 0; qt < 4; ++qt)
                a.cnt[((int64_t)r * a.nq_pad + q0 + wq * 64 + qt * 16 + l15) * 8 + wd * 4 + lq] = ncand[qt];
        }
        __syncthreads();
    }
}

// =============================================================================================
// WIDE form of the 16x16x32 main pass (round 6): a 256 x 384 tile on the same eight waves.
// Why: the 256 x 256 kernel is bound by the CU's L1 miss path (~48 cycles per 1-KiB LDS-DMA piece, 32 pieces per K step of
// 32 against 1 070 cycles of MFMA).  A wave tile of 128 x 96 -- 192 accumulator registers of the wave's 256 -- takes (256 + 384) x 64 B
// per K step for 1.5x the multiply-adds: -17 % bytes per flop through that path, 14 instead of 18 operand reads per 48 MFMAs, and a
// query count like 3 452 pads to 3 456 (nine blocks) instead of 3 584.  Same occupancy (two waves per SIMD), same ping-pong of the two
// wave groups, same LDS image and swizzle, same candidate layout (8 sub-lists per (range, query): wave row x lane >> 4).
//   registers: 192 accumulators + 8 corpus fragments (32) + 6 query fragments (24) = 248 -- so nothing else may live in registers across
//     the K loop but two lane offsets (WIDE_BFR = 3, the first form: three query fragments, the other three refilled IN PLACE behind
//     the MFMAs of the first three query tiles; 1.4 % slower -- the refills queue behind the partner's operand reads): the thresholds and margin
//     coefficients of the block's 384 queries and the sub-list counters sit in LDS, candidate addresses are rebuilt on a hit, the lane
//     id is recomputed where it is needed (fresh_lane) and the DMA source of every piece is a wave-uniform base + one clamped per-lane
//     offset computed at issue;
//   LDS: a ring of three 40-KiB slots ([256 corpus rows x 64 B][384 query rows x 64 B]; two K steps = 80 KiB in flight) + 3 KiB
//     thresholds + 12 KiB counters = 135 KiB; 5 LDS-DMA pieces (16 rows x 64 B = 1 KiB) per wave and K step.  A second form with
//     separate rings -- corpus four slots deep, fetched by group 0, queries three, fetched by group 1 -- measured 2-3 % slower;
//   protocol, K step u (A_u / B_u: the barriers in front of and behind the step's MFMAs; g1 runs one barrier behind: g1's A_u is the
//     barrier instance of g0's B_u):
//     mem(u): pieces of u + 2 into the slot of u - 1 | [g0: filter of a finished tile] | 8 + 6 ds_read_b128 of u | own pieces of u + 1
//             landed (vmcnt) | lgkmcnt(0) | A_u | 48 MFMAs | [g1: filter] | lgkmcnt(0) | B_u
//     RAW: a wave confirms its pieces of u + 1 before its A_u; the first reader of u + 1 starts behind a later barrier instance.
//     WAR: the slot of u - 1 is rewritten in mem(u).  Fragments read in mem(u - 1) were retired before the reader's A_{u-1}, which
//          for both groups precedes every mem(u).  With WIDE_BFR = 3 the REFILLS (query tiles 3-5 of a wave = rows 48-95 of its
//          96-row group) are read during the MFMAs and retired only before B_{u-1}: g1's B_{u-1} is behind the start of g0's mem(u),
//          so GROUP 0 FETCHES NO REFILL ROW -- its waves take the 16 corpus pieces and rows 0-15 of each query group, group 1 (whose
//          mem(u) starts behind its own B_{u-1} and behind g0's) the query rows 16-95.  (The assignment is kept with six resident
//          fragments: it costs nothing.)
// dim % 32 == 0 only (the planner keeps the 256 x 256 kernel elsewhere).  DBG 128 (diagnostic library): thresholds +inf.

// The lane id, computed where it is needed (two VALU instructions) instead of living in a register across the K loop: the wide kernel
// has no register to spare for lane coordinates and the addresses derived from them.
__device__ __forceinline__ int fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
// LDS accesses of the filter as inline asm.  hipcc (ROCm 7.2) tracks LDS-DMA: in front of a C++ load from LDS that it cannot prove
// disjoint from the destination of an LDS-DMA instruction in flight it inserts s_waitcnt vmcnt(0) -- in the pending epilogue, which runs
// right behind the DMA issue of its memory phase, that drained the whole ring once per tile (cycle stamps: 8 500 cycles per epilogue
// WITHOUT a single hit).  The thresholds and counters are never the target of a DMA piece; these accessors say so by not being C++ loads.
__device__ __forceinline__ uint32_t lds_offset(const void *p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
__device__ __forceinline__ float2 lds_read_f2(uint32_t addr) {   // complete on return
    uint64_t v;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return make_float2(__uint_as_float((uint32_t)v), __uint_as_float((uint32_t)(v >> 32)));
}
__device__ __forceinline__ uint32_t lds_add_rtn(uint32_t addr, uint32_t x) {   // complete on return
    uint32_t r;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr), "v"(x) : "memory");
    return r;
}
constexpr int WIDE_SUB_BYTES = (TILE_DOCS + WIDE_Q) * SUB_K * 2;   // 40960
constexpr int WIDE_Q_REGION = TILE_DOCS * SUB_K * 2;               // 16384
constexpr int WIDE_RING = 3;
constexpr int WIDE_PIECES = 5;                                     // 40 pieces of 1 KiB per K step over 8 waves
constexpr int WIDE_QT = 6;                                         // query tiles of 16 per wave
#ifndef WIDE_BFR
#define WIDE_BFR 6                                                 // query fragments resident per K step: 6, or 3 + three refills (the A/B)
#endif
constexpr int WIDE_RING_BYTES = WIDE_RING * WIDE_SUB_BYTES;        // 122880
constexpr size_t WIDE_LDS = (size_t)WIDE_RING_BYTES + WIDE_Q * 8 + (size_t)(GEMM_THREADS / 64) * WIDE_QT * 64 * 4;   // 138240

template <int DBG>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_topk16w_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *s_tc = reinterpret_cast<float2 *>(smem + WIDE_RING_BYTES);                     // [384] {tau_q, cq} of the item's query block
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(smem + WIDE_RING_BYTES + WIDE_Q * 8);   // [wave][query tile][lane]
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int wd = wv >> 2;
    const int wq = wv & 3;
    const bool g1 = (wv >= 4);
    const int KS2 = a.dim / SUB_K;
    const uint32_t pitch = (uint32_t)a.dim * 2u;    // bytes per row
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // DBG 16 (diagnostic library): cycles per segment of the K loop
    unsigned long long tprev = 0;
    // TWO lane-dependent registers live across the K loop -- the DMA source offset of a lane inside a piece and its operand-read offset
    // inside a slot --; everything else that depends on the lane is rebuilt from fresh_lane() where it is needed.  (VALU instructions of
    // the memory phase are issued beside the partner wave's MFMAs and wait for a slot: 34 of them per K step for addresses were 30 % of
    // a wave's time.)
    uint32_t lane_src, lane_lds;
    {
        const int ln = fresh_lane();
        const int pr = ln >> 2;                        // row of a DMA piece this lane fetches a 16-byte chunk of
        lane_src = (uint32_t)pr * pitch + (uint32_t)(((ln & 3) ^ (((pr >> 2) & 1) << 1)) << 4);
        lane_lds = (uint32_t)((ln & 15) * 64 + (((ln >> 4) ^ (((ln >> 2) & 1) << 1)) << 4));
    }

    const int xcd = blockIdx.x & (NUM_XCD - 1);
    const int jx = blockIdx.x >> 3;
    const int per_x = gridDim.x >> 3;
    const int qg = xcd % a.qgroups;
    const int rc = xcd / a.qgroups;
    const int nrc = NUM_XCD / a.qgroups;
    const int qb_per = a.qblocks / a.qgroups;
    const int count_x = (a.ranges / nrc) * qb_per;
    const int item_end = a.item_end < count_x ? a.item_end : count_x;

    for (int item = a.item_begin + jx; item < item_end; item += per_x) {
        const int n_rl = a.ranges / nrc;
        // item order: consecutive items (the co-resident workgroups of an XCD) share a RANGE and walk its corpus tiles for different
        // query blocks.  (Measured at NQ, nine blocks = 5 MiB of query rows per XCD against 4 MiB of L2: sharing the query block instead
        // -- CCR_ITEM_SWAP -- main pass 12.05 ms against 11.49; blocks in two halves, all ranges x first half then x second: 11.75.)
        const int rl = a.item_swap ? item % n_rl : item / qb_per;
        const int qb = qg * qb_per + (a.item_swap ? item / n_rl : item % qb_per);
        const int r = rc + nrc * rl;
        const int64_t ntile = (a.n_vt - r + a.ranges - 1) / a.ranges;
        if (ntile <= 0) continue;
        const int q0 = qb * WIDE_Q;

        int cap, seg_r0;
        long long seg_base;
        cand_segment(a.lay, r, cap, seg_r0, seg_base);   // this range's segment of the candidate area (wave-uniform)
        {
            const int ln = fresh_lane();
            const int tid = wv * 64 + ln;
            if (tid < WIDE_Q) {
                const int q = q0 + tid;
                float t = (q < a.n_q) ? a.thr[q] : __builtin_nanf("");
                if constexpr ((DBG & 128) != 0) t = INFINITY;
                s_tc[tid] = make_float2(t, (q < a.n_q) ? a.cq[q] : 0.f);
            }
#pragma unroll
            for (int qt = 0; qt < WIDE_QT; ++qt) s_cnt[(wv * WIDE_QT + qt) * 64 + ln] = 0u;
        }
        // (the first read of s_tc is behind at least one barrier; the vector loads above are complete before the first DMA piece is counted)
        CCR_WAIT_VM(0);

        const int qlimit = (a.n_q - 1 - q0 < WIDE_Q - 1) ? a.n_q - 1 - q0 : WIDE_Q - 1;   // last valid row of the query block
        const char *qblk = reinterpret_cast<const char *>(a.Q) + (int64_t)q0 * pitch;

        f32x4v acc[8][WIDE_QT];
        const int U = (int)(ntile * KS2);   // K steps of the item (< 2^31: tiles / ranges x dim / 32)
        // issue state, wave-uniform and advanced incrementally (no per-step multiplies): ring slot, K step inside the tile and -- group 0 --
        // the tile's base address and its last valid row
        int islot = 0, iks = 0;
        int64_t it = 0;
        const char *dtile = nullptr;
        int dlimit = 0;
        auto tile_base = [&]() __attribute__((always_inline)) {
            const int64_t row0 = (r + it * a.ranges) * a.tile_stride * TILE_DOCS;
            dlimit = (a.n_rows - 1 - row0 < TILE_DOCS - 1) ? (int)(a.n_rows - 1 - row0) : TILE_DOCS - 1;
            dtile = reinterpret_cast<const char *>(a.D) + row0 * pitch;
        };
        tile_base();
        // one K step's pieces of this wave (a piece = 16 image rows x 64 B): group 0, wave w: corpus rows (4 w + i) * 16 .. + 15 (i < 4) and
        // query rows 96 w .. + 15; group 1, wave w: query rows 96 w + 16 (1 + i) .. + 15 (i < 5).  The per-lane offsets are rebuilt at every
        // issue (hoisted out of the K loop they would cost registers the accumulators need); rows beyond the end re-read the last row
        // (their scores are never recorded)
        // do ALL the query rows this wave fetches exist (wave-uniform, fixed for the item)?  Group 0, wave w: rows 96 w .. + 15; group 1: 96 w + 16 .. + 95
        const bool q_whole = g1 ? ((wv - 4) * 96 + 95 <= qlimit) : (wv * 96 + 15 <= qlimit);
        auto issue = [&]() __attribute__((always_inline)) {
            char *buf = smem + islot * WIDE_SUB_BYTES;
            const uint32_t kb = (uint32_t)iks * (SUB_K * 2);
            if (q_whole && (g1 || dlimit == TILE_DOCS - 1)) {
                // every row of every piece exists: wave-uniform bases (scalar arithmetic) + ONE per-lane offset for all the pieces
                const uint32_t voff = lane_src + kb;
                if constexpr ((DBG & 4) == 0) {
                    if (!g1) {
                        if constexpr (!(DBG & 64)) {
                            const char *rows = dtile + (size_t)(wv * 64) * pitch;
#pragma unroll
                            for (int i = 0; i < 4; ++i) glds16(rows + (size_t)(i * 16) * pitch + voff, buf + (wv * 4 + i) * 1024);
                        }
                        if constexpr (!(DBG & 32)) glds16(qblk + (size_t)(wv * 96) * pitch + voff, buf + WIDE_Q_REGION + wv * 96 * 64);
                    } else if constexpr (!(DBG & 32)) {
                        const char *rows = qblk + (size_t)((wv - 4) * 96 + 16) * pitch;
                        char *dst = buf + WIDE_Q_REGION + ((wv - 4) * 96 + 16) * 64;
#pragma unroll
                        for (int i = 0; i < WIDE_PIECES; ++i) glds16(rows + (size_t)(i * 16) * pitch + voff, dst + i * 1024);
                    }
                }
            } else {
                // the last tile of the shard / the last block of the batch: rows beyond the end re-read the last row (never recorded)
                const int iln = fresh_lane();
                const int pr = iln >> 2;
                const uint32_t co = (uint32_t)(((iln & 3) ^ (((pr >> 2) & 1) << 1)) << 4) + kb;
                auto piece = [&](const char *rows, int row0, int limit, char *dst) __attribute__((always_inline)) {
                    int row = row0 + pr;
                    row = row < limit ? row : limit;
                    glds16(rows + ((uint32_t)row * pitch + co), dst);
                };
                if constexpr ((DBG & 4) == 0) {
                    if (!g1) {
                        if constexpr (!(DBG & 64)) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) piece(dtile, (wv * 4 + i) * 16, dlimit, buf + (wv * 4 + i) * 1024);
                        }
                        if constexpr (!(DBG & 32)) piece(qblk, wv * 96, qlimit, buf + WIDE_Q_REGION + wv * 96 * 64);
                    } else if constexpr (!(DBG & 32)) {
#pragma unroll
                        for (int i = 0; i < WIDE_PIECES; ++i)
                            piece(qblk, (wv - 4) * 96 + 16 * (1 + i), qlimit, buf + WIDE_Q_REGION + ((wv - 4) * 96 + 16 * (1 + i)) * 64);
                    }
                }
            }
            islot = islot == WIDE_RING - 1 ? 0 : islot + 1;
            if (++iks == KS2) {
                iks = 0;
                ++it;
                if (!g1) tile_base();
            }
        };

        auto epilogue = [&](int64_t vt, float nt) __attribute__((always_inline)) {   // nt: norm bound of the tile's rows (wave-uniform)
            // (every address below is rebuilt here instead of living in registers across the K loop)
            const int eln = fresh_lane();
            const int l15o = eln & 15, lqo = eln >> 4;
            // candidate cell of (this range, query q0 + ql, sub-list wd * 4 + lqo): a wave-uniform base + a 32-bit byte offset
            // (slot-major: slot n of the cell is 8 records further; 384 queries x 8 x cap <= 8192 x 8 B < 2^32)
            char *const cbase = reinterpret_cast<char *>(a.cand + seg_base + ((int64_t)(r - seg_r0) * a.nq_pad + q0) * 8 * cap + wd * 4);
            const uint32_t row_lo = (uint32_t)(wd * 128 + 4 * lqo);       // + dt * 16 + e: row inside the tile
            const uint32_t tc_addr = lds_offset(s_tc) + (uint32_t)l15o * 8u;            // + (wq * 96 + qt * 16) * 8
            const uint32_t cnt_addr = lds_offset(s_cnt) + (uint32_t)(lqo * 16 + l15o) * 4u;   // + ((wv * 6 + qt) * 64) * 4
            const int64_t tile_row0 = vt * a.tile_stride * TILE_DOCS;      // wave-uniform
            const uint32_t rows_left = a.n_rows - tile_row0 < TILE_DOCS ? (uint32_t)(a.n_rows - tile_row0) : (uint32_t)TILE_DOCS;
#pragma unroll
            for (int qt = 0; qt < WIDE_QT; ++qt) {
                float sub[8];
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) sub[dt] = max4_asm(acc[dt][qt]);
                const float2 tc = lds_read_f2(tc_addr + (uint32_t)(wq * 96 + qt * 16) * 8u);
                const float t = fmaf(-tc.y, nt, tc.x);   // per-tile margin: mfma + cq * ||d|| >= tau_q with ||d|| <= nt
                float mall = max3_asm(sub[0], sub[1], sub[2]);
                mall = max3_asm(mall, sub[3], sub[4]);
                mall = max3_asm(mall, sub[5], sub[6]);
                mall = fmaxf(mall, sub[7]);
                if (__ballot(mall >= t) != 0ull) {
                    const uint32_t ql = (uint32_t)(wq * 96 + qt * 16 + l15o);
                    const uint32_t cell = (ql * 8u * (uint32_t)cap + (uint32_t)lqo) * 8u;    // byte offset of slot 0
                    const uint32_t cn = cnt_addr + (uint32_t)((wv * WIDE_QT + qt) * 64) * 4u;
#pragma unroll
                    for (int dt = 0; dt < 8; ++dt) {
                        if (sub[dt] >= t) {  // rare, divergent: 4 rows to test
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float v = acc[dt][qt][e];
                                const uint32_t row = row_lo + dt * 16 + e;
                                if (v >= t && row < rows_left) {
                                    const uint32_t n = lds_add_rtn(cn, 1u);   // lane-private cell
                                    if (n < (uint32_t)cap)
                                        *reinterpret_cast<uint2 *>(cbase + (cell + n * 64u)) = make_uint2(__float_as_uint(v), (uint32_t)tile_row0 + row);
                                }
                            }
                        }
                    }
                }
            }
        };

        // pieces in flight per wave and K step: 5 (production); the DMA ablations issue 0 / 4 or 0 / 1 or 5 per group
        auto wait_all_but_one_step = [&]() __attribute__((always_inline)) {
            if constexpr ((DBG & 4) != 0) {
            } else if constexpr ((DBG & 32) != 0) {
                if (!g1) wait_vm<4>();
            } else if constexpr ((DBG & 64) != 0) {
                if (!g1)
                    wait_vm<1>();
                else
                    wait_vm<WIDE_PIECES>();
            } else {
                wait_vm<WIDE_PIECES>();
            }
        };
        const int npro = U < 2 ? U : 2;
        for (int i = 0; i < npro; ++i) issue();
        if (npro == 2)
            wait_all_but_one_step();
        else
            CCR_WAIT_VM(0);
        CCR_BARRIER();
        if (g1) CCR_BARRIER();

        int cks = 0;
        int64_t ct = 0;
        bool pending = false;
        int64_t pending_vt = 0;
        float pending_nt = 0.f;
        int uslot = 0;
        if constexpr ((DBG & 16) != 0) tprev = stamp();
        for (int u = 0; u < U; ++u) {
            CCR_STAMP(0)  // barrier B wait (+ loop overhead)
            // the pieces of u + 2 go out FIRST (into the slot of u - 1, free since the barrier just passed): the L1 miss path, which bounds
            // the pass, gets its work at the start of the phase instead of behind the filter and the operand reads (-1 %; two pieces in
            // front of the operand reads and three behind them: +3 %)
            if (u + 2 < U) issue();
            __builtin_amdgcn_sched_barrier(0);
            CCR_STAMP(3)  // DMA issue
            if (pending) {
                epilogue(pending_vt, pending_nt);
                pending = false;
            }
            CCR_STAMP(1)  // tile epilogue (group 0)
            const char *abuf = smem + uslot * WIDE_SUB_BYTES;
            const char *bbuf = abuf + WIDE_Q_REGION;
            uslot = uslot == WIDE_RING - 1 ? 0 : uslot + 1;
            const int a_base = wd * 128 * 64 + (int)lane_lds;   // + dt * 1024
            const int b_base = wq * 96 * 64 + (int)lane_lds;    // + qt * 1024
            bf16x8 af[8], bfr[WIDE_BFR];
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) af[dt] = *reinterpret_cast<const bf16x8 *>(abuf + a_base + dt * 1024);
#pragma unroll
            for (int qt = 0; qt < WIDE_BFR; ++qt) bfr[qt] = *reinterpret_cast<const bf16x8 *>(bbuf + b_base + qt * 1024);
            if (u + 2 < U)
                wait_all_but_one_step();           // own pieces of u + 1 have landed
            else
                CCR_WAIT_VM(0);
            CCR_STAMP(2)  // DMA wait
            // (group 0 could confirm its pieces as late as its B_u -- the barrier instance of g1's A_u --: measured 3 % slower)
            CCR_WAIT_LGKM0();
            CCR_STAMP(6)  // operand reads landed
            CCR_BARRIER();
            CCR_STAMP(4)  // barrier A wait
            if constexpr ((DBG & 8) != 0) {   // no matrix work: the operand reads (refills included) stay, the accumulators are opaque
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) asm volatile("" ::"v"(af[dt]));
#pragma unroll
                for (int qt = 0; qt < WIDE_QT; ++qt) {
                    asm volatile("" ::"v"(bfr[qt % WIDE_BFR]));
                    if (WIDE_BFR == 3 && qt < 3) {
                        __builtin_amdgcn_sched_barrier(0);
                        bfr[qt] = *reinterpret_cast<const bf16x8 *>(bbuf + b_base + (qt + 3) * 1024);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (cks == 0) {
#pragma unroll
                    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
                        for (int qt = 0; qt < WIDE_QT; ++qt) acc[dt][qt] = f32x4v{-1e30f, -1e30f, -1e30f, -1e30f};
                }
#pragma unroll
                for (int dt = 0; dt < 8; ++dt)
#pragma unroll
                    for (int qt = 0; qt < WIDE_QT; ++qt) asm volatile("" : "+v"(acc[dt][qt]));
            } else if (cks == 0) {
                const f32x4v z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int qt = 0; qt < WIDE_QT; ++qt) {
#pragma unroll
                    for (int dt = 0; dt < 8; ++dt) acc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[dt], bfr[qt % WIDE_BFR], z, 0, 0, 0);
                    if (WIDE_BFR == 3 && qt < 3) {
                        __builtin_amdgcn_sched_barrier(0);
                        bfr[qt] = *reinterpret_cast<const bf16x8 *>(bbuf + b_base + (qt + 3) * 1024);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
#pragma unroll
                for (int qt = 0; qt < WIDE_QT; ++qt) {
#pragma unroll
                    for (int dt = 0; dt < 8; ++dt)
                        acc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[dt], bfr[qt % WIDE_BFR], acc[dt][qt], 0, 0, 0);
                    if (WIDE_BFR == 3 && qt < 3) {
                        __builtin_amdgcn_sched_barrier(0);
                        bfr[qt] = *reinterpret_cast<const bf16x8 *>(bbuf + b_base + (qt + 3) * 1024);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            CCR_STAMP(5)  // MFMA phase (issue of the 48 MFMAs and the refills)
            if (++cks == KS2) {
                cks = 0;
                // (the tile's norm bound: a SCALAR load + wait right here, behind the MFMAs just issued)
                const int64_t vt_done = r + ct * a.ranges;
                const float nt_done = load_uniform_f32(a.tile_norm + vt_done * a.tile_stride);
                if (g1) {
                    epilogue(vt_done, nt_done);
                } else {
                    pending = true;
                    pending_vt = vt_done;
                    pending_nt = nt_done;
                }
                ++ct;
            }
            CCR_STAMP(1)  // tile epilogue (group 1)
            CCR_WAIT_LGKM0();   // the refills of this K step are retired before the barrier behind which group 1 rewrites their rows
            CCR_STAMP(7)  // MFMA drain: the step's last refills / MFMAs
            CCR_BARRIER();
        }
        if (pending) epilogue(pending_vt, pending_nt);
        if (!g1) CCR_BARRIER();

        {
            const int ln = fresh_lane();
#pragma unroll
            for (int qt = 0; qt < WIDE_QT; ++qt)
                a.cnt[((int64_t)r * a.nq_pad + q0 + wq * 96 + qt * 16 + (ln & 15)) * 8 + wd * 4 + (ln >> 4)] = s_cnt[(wv * WIDE_QT + qt) * 64 + ln];
        }
        __syncthreads();
    }
    if constexpr ((DBG & 16) != 0) {
        if (fresh_lane() == 0 && a.store) {
            unsigned long long *o = reinterpret_cast<unsigned long long *>(a.store) + ((size_t)blockIdx.x * 8 + wv) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = seg[i];
        }
    }
}

// =============================================================================================
// QUERY-DIRECT form of the 16x16x32 main pass (round 6) -- an EXPERIMENT kept for the record, compiled only into the diagnostic library
// (make DIAG=1 / tools/build_diag.sh, CCR_QDIRECT = 1 / 3 / 4 / 5).  Exact (every form returns the production kernel's ids and score bits)
// and 38-48 % SLOWER: NQ main pass 16.8-18.2 ms against 12.1 (profiles/r06_qdirect_ab.txt).  Why: what bounds the main pass is the CU's
// vector-memory pipeline (L1 / TCP miss handling: ~0.3 64-byte requests per clock, profiles/r06_main_pass_memory_pipeline_pmc.txt), not the
// LDS-DMA issue; fragments fetched straight into registers are requested by BOTH waves that share a query quarter, so the CU moves
// 16 + 32 KiB per sub-stage through that pipeline instead of 16 + 16.
#ifdef CCR_DIAGNOSTICS
// (The reasoning it was built on, before the measurement.)  The ablation of the kernel above (profiles/r06_main_pass_ablation.txt:
// full 11.35 ms, corpus-only DMA 8.0, no DMA 7.75) says that the LDS-DMA *issue* -- four 1-KiB pieces per wave and sub-stage, two of
// them the query slice that is re-streamed for every corpus tile -- is what keeps the mem phase longer than the partner's MFMA phase.
// Here the ring carries the CORPUS alone (two pieces per wave and sub-stage) and a lane fetches its four B-operand fragments -- 16 bytes
// of query row (q0 + wq*64 + qt*16 + l15) at K chunk lq: exactly what v_mfma_f32_16x16x32_bf16 wants in SrcB -- straight from global
// memory (the item's query block stays in the XCD's L2) one sub-stage ahead: no LDS round trip for the queries, no ds_read for them.
//   QD = 1  fragments of u + 1 requested at the TAIL of the MFMA phase of u (behind the 32 MFMAs), ONE register set
//   QD = 3  requested BETWEEN the MFMAs of u: query tile qt's eight MFMAs, then qt's next fragment, ONE register set
//   QD = 4 / 5  the same two placements with TWO register sets: the fragments of u + 2 are requested during u
// vmcnt order per wave, one set: ... DMA(u+2)[2] | Q(u)[4] | DMA(u+3)[2] | Q(u+1)[4] ...: `vmcnt(2)` before barrier A_u confirms Q(u) and, being
// older, this wave's DMA of u + 1 AND u + 2 (a sub-stage of corpus has one period to land).  Two sets: ... Q(u)[4] | DMA(u+2)[2] Q(u+1)[4] DMA(u+3)[2]:
// `vmcnt(8)` confirms Q(u) and the DMA of u + 1 -- the prefetch distance of the kernel above.
// Same ring protocol, same ping-pong, same epilogues and candidate layout as gemm_topk16_kernel; dim % 32 == 0 only.
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

template <int EPI, int QD>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_topk16q_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wd = wv >> 2;
    const int wq = wv & 3;
    const int l15 = lane & 15;
    const int lq = lane >> 4;
    const bool g1 = (wv >= 4);
    const int KS2 = a.dim / SUB_K;

    const int srow = wv * 16 + (lane >> 2);  // + piece*128
    const int schunk = (lane & 3) ^ (((srow >> 2) & 1) << 1);
    const int cofs = ((lq ^ (((lane >> 2) & 1) << 1)) << 4);
    const int a_base = (wd * 128 + l15) * 64 + cofs;                // + dt*1024

    const int xcd = blockIdx.x & (NUM_XCD - 1);
    const int jx = blockIdx.x >> 3;
    const int per_x = gridDim.x >> 3;
    const int qg = xcd % a.qgroups;
    const int rc = xcd / a.qgroups;
    const int nrc = NUM_XCD / a.qgroups;
    const int qb_per = a.qblocks / a.qgroups;
    const int count_x = (a.ranges / nrc) * qb_per;
    const int item_end = a.item_end < count_x ? a.item_end : count_x;

    for (int item = a.item_begin + jx; item < item_end; item += per_x) {
        const int n_rl = a.ranges / nrc;
        const int rl = a.item_swap ? item % n_rl : item / qb_per;
        const int qb = qg * qb_per + (a.item_swap ? item / n_rl : item % qb_per);
        const int r = rc + nrc * rl;
        const int64_t ntile = (a.n_vt - r + a.ranges - 1) / a.ranges;
        if (ntile <= 0) continue;
        const int q0 = qb * TILE_Q;

        float thr[4] = {0.f, 0.f, 0.f, 0.f}, cqv[4] = {0.f, 0.f, 0.f, 0.f};
        uint32_t ncand[4] = {0u, 0u, 0u, 0u};
        uint2 *clist[4] = {nullptr, nullptr, nullptr, nullptr};
        int cap = 0;
        if (EPI == EPI_FILTER) {
            int seg_r0;
            long long seg_base;
            cand_segment(a.lay, r, cap, seg_r0, seg_base);
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                const int q = q0 + wq * 64 + qt * 16 + l15;
                thr[qt] = (q < a.n_q) ? a.thr[q] : __builtin_nanf("");
                cqv[qt] = (q < a.n_q) ? a.cq[q] : 0.f;
                clist[qt] = a.cand + seg_base + ((int64_t)(r - seg_r0) * a.nq_pad + q) * 8 * cap + (wd * 4 + lq);
            }
            asm volatile("" : "+v"(thr[0]), "+v"(thr[1]), "+v"(thr[2]), "+v"(thr[3]), "+v"(cqv[0]), "+v"(cqv[1]), "+v"(cqv[2]), "+v"(cqv[3]));
        }
        // this lane's four query rows (B fragments): byte offsets from a.Q, K chunk lq folded in (n_q * dim * 2 < 2^31: checked by the launcher)
        uint32_t qoff[4];
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            int q = q0 + wq * 64 + qt * 16 + l15;
            if (q > a.n_q - 1) q = a.n_q - 1;
            qoff[qt] = (uint32_t)q * (uint32_t)(a.dim * 2) + (uint32_t)(lq * 16);
        }
        // The fragment loads and their waits are inline asm on purpose: hipcc's own wait for a C++ load is `vmcnt(0)` in front of the
        // first MFMA (it cannot bound the conditional DMA issue in between), which drains the ring every sub-stage.  The asm load
        // updates its destination in place ("+v": one physical register set across the loop), the counted wait names the same
        // registers, so nothing reads them in between.
        const char *qbase = reinterpret_cast<const char *>(a.Q);
        auto ldq = [&](u32x4v &dst, int qt, int ks) {   // ks: K sub-stage inside the tile (wave-uniform)
            const char *sb = qbase + ks * (SUB_K * 2);
            asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(qoff[qt]), "s"(sb));
        };
#define CCR_WAIT_VM_Q(n, b) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])::"memory")

        f32x4v acc[8][4];
        const int64_t U = ntile * KS2;

        int64_t iu = 0, it = 0;
        int iks = 0;
        const uint16_t *dsrc[2];
        auto tile_ptrs = [&]() {
            const int64_t row0 = (r + it * a.ranges) * a.tile_stride * TILE_DOCS;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int64_t drow = row0 + i * 128 + srow;
                if (drow > a.n_rows - 1) drow = a.n_rows - 1;
                dsrc[i] = a.D + drow * a.dim + schunk * 8;
            }
        };
        tile_ptrs();
        auto issue = [&]() {   // the corpus slice of one sub-stage: two 1-KiB pieces per wave
            char *buf = smem + (int)(iu & (RING - 1)) * SUB_BYTES;
            const int k0 = iks * SUB_K;
#pragma unroll
            for (int i = 0; i < 2; ++i) glds16(dsrc[i] + k0, buf + (i * 512 + wv * 64) * 16);
            ++iu;
            if (++iks == KS2) {
                iks = 0;
                ++it;
                tile_ptrs();
            }
        };

        auto epilogue = [&](int64_t vt, float nt) __attribute__((always_inline)) {
            const int64_t row_base = vt * a.tile_stride * TILE_DOCS + wd * 128 + 4 * lq;  // + dt*16 + e
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                if (EPI == EPI_FILTER) {
                    float sub[8];
#pragma unroll
                    for (int dt = 0; dt < 8; ++dt)
                        sub[dt] = fmaxf(fmaxf(acc[dt][qt][0], acc[dt][qt][1]), fmaxf(acc[dt][qt][2], acc[dt][qt][3]));
                    const float t = fmaf(-cqv[qt], nt, thr[qt]);
                    const float mall = fmaxf(fmaxf(fmaxf(sub[0], sub[1]), fmaxf(sub[2], sub[3])),
                                             fmaxf(fmaxf(sub[4], sub[5]), fmaxf(sub[6], sub[7])));
                    if (__ballot(mall >= t) != 0ull) {
#pragma unroll
                        for (int dt = 0; dt < 8; ++dt) {
                            if (sub[dt] >= t) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const float v = acc[dt][qt][e];
                                    const int64_t doc = row_base + dt * 16 + e;
                                    if (v >= t && doc < a.n_rows) {
                                        if (ncand[qt] < (uint32_t)cap)
                                            clist[qt][ncand[qt] * 8] = make_uint2(__float_as_uint(v), (uint32_t)doc);
                                        ++ncand[qt];
                                    }
                                }
                            }
                        }
                    }
                } else {  // EPI_STORE
                    const int q = q0 + wq * 64 + qt * 16 + l15;
                    const int64_t pitch = a.store_pitch ? a.store_pitch : a.n_rows;
                    if (q < a.n_q) {
                        float *dst = a.store + (int64_t)q * pitch;
#pragma unroll
                        for (int dt = 0; dt < 8; ++dt) {
                            const int64_t doc = row_base + dt * 16;
                            if ((pitch & 3) == 0 && doc + 3 < a.n_rows) {
                                *reinterpret_cast<float4 *>(dst + doc) =
                                    make_float4(acc[dt][qt][0], acc[dt][qt][1], acc[dt][qt][2], acc[dt][qt][3]);
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (doc + e < a.n_rows) dst[doc + e] = acc[dt][qt][e];
                            }
                        }
                    }
                }
            }
        };

        // ---- prologue: the fragments of sub-stage 0 (and 1) first -- oldest in vmcnt -- then up to three sub-stages of corpus DMA
        constexpr bool TWO = (QD >= 4);          // two register sets: fragments requested TWO sub-stages ahead
        constexpr bool INTER = (QD == 3 || QD == 5);   // requested between the MFMAs (query tile by query tile) instead of behind them
        u32x4v bq[4] = {}, bq2[4] = {};
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) ldq(bq[qt], qt, 0);
        if constexpr (TWO) {
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) ldq(bq2[qt], qt, KS2 > 1 ? 1 : 0);
        }
        // The DMA issue is UNCONDITIONAL (three sub-stages here, one per sub-stage in the loop -- the last three of an item fetch clamped
        // rows into ring buffers nobody reads again): every counted wait below is then ONE straight-line statement.  A wait inside a branch
        // makes hipcc merge the fragment registers of the two paths with copies, and it places such copies IN FRONT of the wait.
        for (int i = 0; i < 3; ++i) issue();
        if constexpr (TWO)
            CCR_WAIT_VM_Q(2, bq);   // Q(0) Q(1) DMA(0) DMA(1) | DMA(2): sub-stage 1 has landed too, so that vmcnt(8) is enough at u = 0
        else
            CCR_WAIT_VM_Q(4, bq);   // Q(0) DMA(0) | DMA(1) DMA(2)
        CCR_BARRIER();
        if (g1) CCR_BARRIER();

        int cks = 0;
        int64_t ct = 0;
        bool pending = false;
        int64_t pending_vt = 0;
        float pending_nt = 0.f;
        // one sub-stage; cur = the fragments of u (requested one -- TWO: two -- sub-stages ago), nxt = where this sub-stage's requests go
        // (ONE set: cur itself, for u + 1; TWO: cur as well -- it is free once the MFMAs of u are issued -- for u + 2, while the other
        // set holds u + 1 in flight)
        auto substage = [&](int64_t u, u32x4v (&cur)[4]) __attribute__((always_inline)) {
            if (pending) {
                epilogue(pending_vt, pending_nt);
                pending = false;
            }
            int nks = cks + (TWO ? 2 : 1);   // K sub-stage (inside its tile) of the fragments requested during u
            if (nks >= KS2) nks -= KS2;
            if (nks >= KS2) nks = 0;        // (KS2 == 1)
            const char *buf = smem + (int)(u & (RING - 1)) * SUB_BYTES;
            bf16x8 af[8];
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) af[dt] = *reinterpret_cast<const bf16x8 *>(buf + a_base + dt * 1024);
            issue();   // sub-stage u + 3 (past the item's end: see the prologue)
            // confirm, before this wave's barrier A_u: the fragments of u and -- older in vmcnt -- this wave's corpus DMA of u + 1
            if constexpr (TWO)
                CCR_WAIT_VM_Q(8, cur);   // ... Q(u) | DMA(u+2)[2] Q(u+1)[4] DMA(u+3)[2]
            else
                CCR_WAIT_VM_Q(2, cur);   // ... DMA(u+2)[2] Q(u) | DMA(u+3)[2]
            CCR_WAIT_LGKM0();
            CCR_BARRIER();
            if constexpr (INTER) {   // query tile by query tile; a tile's next fragment is requested as soon as its MFMAs are issued
#pragma unroll
                for (int qt = 0; qt < 4; ++qt) {
                    if (cks == 0) {
                        const f32x4v z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int dt = 0; dt < 8; ++dt) acc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[dt], __builtin_bit_cast(bf16x8, cur[qt]), z, 0, 0, 0);
                    } else {
#pragma unroll
                        for (int dt = 0; dt < 8; ++dt) acc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[dt], __builtin_bit_cast(bf16x8, cur[qt]), acc[dt][qt], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    ldq(cur[qt], qt, nks);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                if (cks == 0) {
                    const f32x4v z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
                        for (int qt = 0; qt < 4; ++qt)
                            acc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[dt], __builtin_bit_cast(bf16x8, cur[qt]), z, 0, 0, 0);
                } else {
#pragma unroll
                    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
                        for (int qt = 0; qt < 4; ++qt)
                            acc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[dt], __builtin_bit_cast(bf16x8, cur[qt]), acc[dt][qt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int qt = 0; qt < 4; ++qt) ldq(cur[qt], qt, nks);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (++cks == KS2) {
                cks = 0;
                const int64_t vt_done = r + ct * a.ranges;
                const float nt_done = EPI == EPI_FILTER ? load_uniform_f32(a.tile_norm + vt_done * a.tile_stride) : 0.f;
                if (g1) {
                    epilogue(vt_done, nt_done);
                } else {
                    pending = true;
                    pending_vt = vt_done;
                    pending_nt = nt_done;
                }
                ++ct;
            }
            CCR_BARRIER();
        };
        if constexpr (TWO) {
            int64_t u = 0;
            for (; u + 1 < U; u += 2) {
                substage(u, bq);
                substage(u + 1, bq2);
            }
            if (u < U) substage(u, bq);
        } else {
            for (int64_t u = 0; u < U; ++u) substage(u, bq);
        }
        // The DMA and the fragments requested past the item's end land here, unused.  The wait NAMES the fragment registers: they stay
        // allocated until nothing is in flight into them (hipcc would otherwise hand them to the epilogue below as temporaries).
        if constexpr (TWO) CCR_WAIT_VM_Q(0, bq2);
        CCR_WAIT_VM_Q(0, bq);
        if (pending) epilogue(pending_vt, pending_nt);
        if (!g1) CCR_BARRIER();

        if (EPI == EPI_FILTER) {
#pragma unroll
            for (int qt = 0; qt < 4; ++qt)
                a.cnt[((int64_t)r * a.nq_pad + q0 + wq * 64 + qt * 16 + l15) * 8 + wd * 4 + lq] = ncand[qt];
        }
        __syncthreads();
    }
}
#endif   // CCR_DIAGNOSTICS

// ---------------------------------------------------------------------------------------------
// bf16 row norms (fp32 accumulate; used only for error margins, inflated by the caller).  With tile_bits (the index's own
// pass over a shard it was given without norms) a wave takes 64 CONSECUTIVE rows at a time and max-accumulates their norms
// into the word of their 256-row tile (four adds per tile); norms of non-negative floats order as unsigned bit patterns and a
// NaN (row with a NaN) stays on top.
__global__ __launch_bounds__(256) void row_norms_bf16_kernel(const uint16_t *__restrict__ X, int64_t rows, int dim,
                                                            float *__restrict__ norms, uint32_t *__restrict__ max_bits,
                                                            uint32_t *__restrict__ tile_bits) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int chunk = tile_bits ? 64 : 1;
    uint32_t wbits = 0u;
    for (int64_t r0 = wave * chunk; r0 < rows; r0 += nwaves * chunk) {
        uint32_t cbits = 0u;
        const int64_t r1 = r0 + chunk < rows ? r0 + chunk : rows;
        for (int64_t r = r0; r < r1; ++r) {
            const uint16_t *x = X + r * dim;
            float s = 0.f;
            for (int c = lane * 8; c < dim; c += 64 * 8) {
                const uint4 v = *reinterpret_cast<const uint4 *>(x + c);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = __uint_as_float(w[e] << 16), hi = __uint_as_float(w[e] & 0xffff0000u);
                    s = fmaf(lo, lo, s);
                    s = fmaf(hi, hi, s);
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
            const float n = sqrtf(s);
            if (norms && lane == 0) norms[r] = n;
            const uint32_t nb = __float_as_uint(n) & 0x7fffffffu;
            cbits = nb > cbits ? nb : cbits;
        }
        if (tile_bits && lane == 0) atomicMax(tile_bits + r0 / TILE_DOCS, cbits);
        wbits = cbits > wbits ? cbits : wbits;
    }
    // one look first: only a wave that would raise the maximum touches the atomic
    if (max_bits && lane == 0 && wbits > __hip_atomic_load(max_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(max_bits, wbits);
}

// tile_norm[t] = the largest of the row-norm bounds of rows [256 t, 256 t + 256) the pack kernels wrote (bit patterns: a NaN
// bound stays on top).  grid = tiles, block = 256.  No shared maximum here: the ~2 K blocks resident at the start would all
// see the initial zero and queue their atomics on one address (measured: 38 us for this kernel instead of 4).
__global__ __launch_bounds__(256) void tile_norms_kernel(const float *__restrict__ row_bounds, int64_t rows,
                                                        uint32_t *__restrict__ tile_bits) {
    __shared__ uint32_t s_w[4];
    const int64_t r = (int64_t)blockIdx.x * TILE_DOCS + threadIdx.x;
    uint32_t b = r < rows ? (__float_as_uint(row_bounds[r]) & 0x7fffffffu) : 0u;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t o = __shfl_xor(b, off, 64);
        b = o > b ? o : b;
    }
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t m01 = s_w[0] > s_w[1] ? s_w[0] : s_w[1], m23 = s_w[2] > s_w[3] ? s_w[2] : s_w[3];
        tile_bits[blockIdx.x] = m01 > m23 ? m01 : m23;
    }
}

// max_bits = the largest tile bound (a NaN / Inf poisons the thresholds -> dense path).  One workgroup of 1 024 threads.
__global__ __launch_bounds__(1024) void max_tile_norm_kernel(const uint32_t *__restrict__ tile_bits, int64_t tiles,
                                                            uint32_t *__restrict__ max_bits) {
    __shared__ uint32_t s_w[16];
    uint32_t b = 0u;
    for (int64_t t = threadIdx.x; t < tiles; t += 1024) {
        const uint32_t v = tile_bits[t];
        b = v > b ? v : b;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t o = __shfl_xor(b, off, 64);
        b = o > b ? o : b;
    }
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t m = 0u;
        for (int w = 0; w < 16; ++w) m = s_w[w] > m ? s_w[w] : m;
        *max_bits = m;
    }
}

// Sample thresholds.  A sampled 16-row group whose maximum MFMA score is m holds a row whose EXACT score is at least
// m - cq * nt (cq = gamma * ||q||, nt = norm bound of the group's tile), so tau_q = the k-th largest of those lower bounds is
// a lower bound of the query's k-th largest exact score: thr[q] = tau_q, cq[q] = the margin coefficient the later stages use.
// One workgroup = 16 queries x 16 group phases: a gmax row is read as 64-byte segments (query-contiguous
// layout), every query has its own 256-bin LDS histogram (no same-address atomics), 4 MSB-first passes.
// grid = nq_pad / 16, block = 256.
__global__ __launch_bounds__(256) void threshold_kernel(const float *__restrict__ gmax, int64_t n_groups, int n_q, int nq_pad,
                                                       int k, const float *__restrict__ qnorm,
                                                       const uint32_t *__restrict__ dmax_bits, float gamma,
                                                       const float *__restrict__ tile_norm, int64_t sample_stride,
                                                       float *__restrict__ thr, float *__restrict__ cq) {
    __shared__ uint32_t s_hist[16][256];
    __shared__ uint32_t s_prefix[16], s_remaining[16];
    const int tid = threadIdx.x;
    const int ql = tid & 15, ph = tid >> 4;
    const int q = blockIdx.x * 16 + ql;
    if (tid < 16) {
        s_prefix[tid] = 0;
        s_remaining[tid] = (uint32_t)k;
    }
    // |mfma - exact| <= gamma * ||q|| * ||d||; a non-finite row norm anywhere in the shard (NaN / Inf embeddings) poisons
    // every threshold, so that all queries take the exact dense path
    const float dmax = __uint_as_float(*dmax_bits);
    const float c = (q < n_q) ? ((dmax < INFINITY) ? gamma * (qnorm[q] * 1.001f) * 1.001f : __builtin_nanf("")) : 0.f;
    // norm bound of the sampled tile of group g (plain vector loads: staging them in LDS, or scalar loads of the wave-uniform
    // index, both made this kernel slower -- 84 vs 70 us at NQ: they share lgkmcnt with the histogram's LDS atomics)
    auto tile_of = [&](int64_t g) { return tile_norm[(g / GROUPS_PER_TILE) * sample_stride]; };
    uint32_t mask = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int b = tid; b < 16 * 256; b += 256) (&s_hist[0][0])[b] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix[ql];
        // 4 independent loads in flight per thread (the loop is latency-bound otherwise)
        int64_t i = ph;
        for (; i + 240 < n_groups; i += 256) {   // 16 independent loads in flight per thread
            float v[16], tn[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                v[j] = gmax[(i + 16 * j) * nq_pad + q];
                tn[j] = tile_of(i + 16 * j);
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t o = f32_orderable(fmaf(-c, tn[j], v[j]));
                if ((o & mask) == prefix) atomicAdd(&s_hist[ql][(o >> shift) & 255u], 1u);
            }
        }
        for (; i < n_groups; i += 16) {
            const uint32_t o = f32_orderable(fmaf(-c, tile_of(i), gmax[i * nq_pad + q]));
            if ((o & mask) == prefix) atomicAdd(&s_hist[ql][(o >> shift) & 255u], 1u);
        }
        __syncthreads();
        {   // digit pick: 16 lanes per query (16 bins each), shuffle suffix-scan instead of a 256-step serial walk
            const int qi = tid >> 4, part = tid & 15;
            const uint32_t remaining = s_remaining[qi];
            uint32_t hb[16];
            uint32_t own = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                hb[j] = s_hist[qi][16 * part + j];
                own += hb[j];
            }
            uint32_t incl = own;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const uint32_t t = __shfl_down(incl, off, 16);
                if (part + off < 16) incl += t;
            }
            const uint32_t above = incl - own;
            const uint32_t total = __shfl(incl, 0, 16);
            __syncthreads();   // every lane has read s_remaining before the crossing lane rewrites it
            if (above < remaining && remaining <= incl) {
                uint32_t cum = above;
                int d = 16 * part;
#pragma unroll
                for (int j = 15; j >= 1; --j) {
                    if (cum + hb[j] >= remaining) {
                        d = 16 * part + j;
                        break;
                    }
                    cum += hb[j];
                }
                s_prefix[qi] |= (uint32_t)d << shift;
                s_remaining[qi] = remaining - cum;
            } else if (part == 0 && total < remaining) {   // fewer groups than k (cannot happen: the planner samples >= 2k)
                s_remaining[qi] = remaining - (total - hb[0]);
            }
        }
        mask |= 0xffu << shift;
        __syncthreads();
    }
    if (tid < 16) {
        const int qq = blockIdx.x * 16 + tid;
        if (qq < n_q) {
            const float tau = orderable_to_f32(s_prefix[tid]);
            cq[qq] = c;   // tid < 16: q == qq
            thr[qq] = (c < INFINITY) ? tau : __builtin_nanf("");
        }
    }
}


// Visit every record (sub-list j, slot sl) with first <= sl < count[j]: the (sub-list, slot) rectangle is padded to a
// power of two per sub-list, BATCH independent 8-byte loads are issued before the first record is consumed.
template <int THREADS, int BATCH, class At, class Want, class Put>
__device__ __forceinline__ void sweep_sublists(int tid, int n_lists, int first, int longest, const uint32_t *s_cnt, At at,
                                               Want want, Put put) {
    if (longest <= first) return;
    int lw = 0;
    while ((1 << lw) < longest - first) ++lw;
    const int sweep = n_lists << lw;
    for (int i0 = tid; i0 < sweep; i0 += THREADS * BATCH) {
        uint2 e[BATCH];
        int jj[BATCH], ss[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int i = i0 + u * THREADS;
            const int j = i >> lw, sl = first + (i & ((1 << lw) - 1));
            jj[u] = -1;
            ss[u] = sl;
            if (i < sweep && (uint32_t)sl < s_cnt[j] && want(j, sl)) {
                jj[u] = j;
                e[u] = at(j, sl);
            }
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u)
            if (jj[u] >= 0) put(jj[u], ss[u], e[u]);
    }
}

// Progressive thresholds: after a phase of the main pass the k-th largest LOWER BOUND (mfma - cq * tile norm) among the
// candidates found so far is a tighter valid lower bound of the k-th largest exact score (the candidates are real rows with
// their real MFMA scores, whether or not a sub-list overflowed).  thr[q] = max(thr[q], that).
// The candidate lists are one or two 128-byte lines per sub-list scattered over the shard-sized candidate area: reading them
// is what this kernel costs (measured at NQ: 1.4 TB/s of line traffic, 52 us for the 160 sub-lists per query after phase A,
// 138-185 us for the 448 after phase B1).  So the second re-tightening does not read phase A's lists again: the first one
// leaves the k best lower bounds of every query in `top` (k contiguous values), the second merges them with the sub-lists
// completed since.
// grid = n_q, block = 256.  top: [nq_pad] valid counts, then [n_q][k] values (may be null when top_in == top_out == 0).
__global__ __launch_bounds__(256) void threshold_update_kernel(const uint2 *__restrict__ cand, const uint32_t *__restrict__ cnt,
                                                              int nsub_full, int nsub_part, int part_blocks, int prev_full,
                                                              int prev_part, int prev_blocks, int qb_per, int sp, int nq_pad,
                                                              const CandLayout lay, int k, int compact,
                                                              const float *__restrict__ cq, const float *__restrict__ tile_norm,
                                                              uint32_t *__restrict__ top, int top_in, int top_out,
                                                              float *__restrict__ thr, int tile_q) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_val[];   // [compact] orderable lower bounds of the candidates found so far
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_ctl[4];
    __shared__ uint32_t s_cnt[2048];
    __shared__ uint32_t s_total, s_maxc, s_fill, s_eq;
    const int tid = threadIdx.x;
    const int q = blockIdx.x;
    // sub-lists complete so far: those of the fully scored ranges, plus one more row of ranges for the queries whose block
    // (position inside its XCD group) was already scored in the partially finished one
    const int qpos = (q / tile_q) % qb_per;
    const int nsub = qpos < part_blocks ? nsub_part : nsub_full;
    uint32_t *top_n = top;
    uint32_t *top_v = top ? top + nq_pad + (int64_t)q * k : nullptr;
    const bool have_prev = top_in && top_n[q] == (uint32_t)k;
    const int j0 = have_prev ? (qpos < prev_blocks ? prev_part : prev_full) : 0;   // lists the previous update already covered
    if (tid == 0) {
        s_total = have_prev ? (uint32_t)k : 0u;
        s_maxc = 0;
        s_fill = 0;
    }
    __syncthreads();
    uint32_t my = 0, mymax = 0;
    for (int j = j0 + tid; j < nsub; j += blockDim.x) {
        uint32_t c = cnt[((int64_t)(j / sp) * nq_pad + q) * sp + (j % sp)];
        int cap;
        (void)cand_sublist(lay, j / sp, q, j % sp, nq_pad, sp, cap);
        if (c > (uint32_t)cap) c = (uint32_t)cap;
        s_cnt[j - j0] = c;
        my += c;
        mymax = c > mymax ? c : mymax;
    }
    if (my) {
        atomicAdd(&s_total, my);
        atomicMax(&s_maxc, mymax);
    }
    __syncthreads();
    if (s_total < (uint32_t)k) {   // not enough rows seen yet: keep the sample threshold
        if (top_out && tid == 0) top_n[q] = 0u;
        return;
    }
    auto sub_base = [&](int j) -> int64_t {
        int cap;
        return cand_sublist(lay, (j0 + j) / sp, q, (j0 + j) % sp, nq_pad, sp, cap);
    };
    uint32_t kth = 0;
    int need_eq = 0, M = 0;
    bool selected = false;   // s_val[0, M) is the set kth was selected from
    {
        // The sub-lists are sparse: one sweep over (sub-list, slot < longest list) with independent loads gathers the
        // scores into LDS, the four radix passes then never touch global memory.  If more candidates exist than the LDS
        // holds, the first `compact` the sweep meets give a first bound (the k-th largest score of ANY k or more real rows
        // is a valid lower bound of the query's k-th largest score); the sweep is then repeated over the records at or above
        // that bound -- far fewer -- until they all fit, so the final bound is the k-th largest of EVERYTHING recorded (a
        // corpus in topical order leaves the good rows in a few sub-lists that a first-come subset would miss).
        const int maxc = (int)s_maxc;
        const float c = cq[q];
        uint32_t keep = 0u;   // orderable bound: records below it are skipped
        auto append = [&](uint32_t o) {
            if (o >= keep) {
                const uint32_t p = atomicAdd(&s_fill, 1u);
                if (p < (uint32_t)compact) s_val[p] = o;
            }
        };
        for (int round = 0; round < 4; ++round) {
            __syncthreads();
            if (tid == 0) s_fill = 0;
            __syncthreads();
            selected = false;
            if (have_prev)
                for (int i = tid; i < k; i += blockDim.x) append(top_v[i]);
            sweep_sublists<256, 8>(
                tid, nsub - j0, 0, maxc, s_cnt, [&](int j, int sl) -> uint2 { return cand[sub_base(j) + (int64_t)sl * sp]; },
                [](int, int) { return true; },
                [&](int, int, uint2 e) { append(f32_orderable(fmaf(-c, tile_norm[e.y / TILE_DOCS], __uint_as_float(e.x)))); });
            __syncthreads();
            const uint32_t filled = s_fill;
            M = (int)(filled < (uint32_t)compact ? filled : (uint32_t)compact);
            if (M < k) break;   // (ties at the bound cut off by the LDS size: keep the bound of the previous round)
            block_radix_select(
                [&](int64_t i, bool &skip) -> uint32_t {
                    (void)skip;
                    return s_val[i];
                },
                (int64_t)M, k, s_hist, s_ctl, kth, need_eq);
            selected = true;
            if (filled <= (uint32_t)compact || kth <= keep) break;   // everything at or above the bound was seen / no progress
            keep = kth;
        }
    }
    if (top_out) {   // the k best lower bounds seen so far, for the next re-tightening (any order)
        if (!selected) {
            if (tid == 0) top_n[q] = 0u;
        } else {
            __syncthreads();
            if (tid == 0) s_fill = s_eq = 0u;
            __syncthreads();
            for (int i = tid; i < M; i += blockDim.x) {
                const uint32_t v = s_val[i];
                bool take = v > kth;
                if (v == kth) take = atomicAdd(&s_eq, 1u) < (uint32_t)need_eq;
                if (take) top_v[atomicAdd(&s_fill, 1u)] = v;
            }
            if (tid == 0) top_n[q] = (uint32_t)k;
        }
    }
    if (tid == 0) {
        const float t1 = orderable_to_f32(kth);
        if (t1 > thr[q]) thr[q] = t1;
    }
}

// Stage-2: per query, candidates -> exact canonical top-k.  grid = n_q, block = THREADS (256; 1024 for large k).
// dyn LDS: [dim bf16 query row][sub-list counts][sub-list offsets][rescore_cap u64 keys][compact x 8-byte candidates]
template <int THREADS>
__global__ __launch_bounds__(THREADS) void select_rescore_kernel(const uint2 *__restrict__ cand, const uint32_t *__restrict__ cnt,
                                                            int ranges, int sp, int nq_pad, const CandLayout lay, int k, int rescore_cap, int compact,
                                                            int64_t n_rows, float *__restrict__ thr, const float *__restrict__ cq,
                                                            const float *__restrict__ tile_norm, const float *__restrict__ row_norm,
                                                            const uint32_t *__restrict__ dmax_bits,
                                                            const uint16_t *__restrict__ Q, const uint16_t *__restrict__ D,
                                                            int dim, int64_t id_offset, float *__restrict__ out_scores,
                                                            int64_t *__restrict__ out_ids, uint32_t *__restrict__ flag_count,
                                                            uint32_t *__restrict__ flag_list,
                                                            unsigned long long *__restrict__ stat_cand,
                                                            const uint32_t *__restrict__ out_rows) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const size_t cnt_bytes = ((size_t)(ranges + 1) * 4 + 15) & ~(size_t)15;
    uint16_t *s_q = reinterpret_cast<uint16_t *>(sm);
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(sm + (((size_t)dim * 2 + 15) & ~(size_t)15));
    uint32_t *s_off = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(s_cnt) + cnt_bytes);
    unsigned long long *s_keys = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(s_off) + cnt_bytes);
    uint2 *s_comp = reinterpret_cast<uint2 *>(s_keys + rescore_cap);   // [compact] gathered candidates
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_ctl[4];
    __shared__ uint32_t s_wsum[THREADS / 64];
    __shared__ int s_flag;
    __shared__ uint32_t s_total, s_maxc, s_fill;
    __shared__ uint32_t s_ncoll;

    const int tid = threadIdx.x;
    const int q = blockIdx.x;
    if (tid == 0) {
        s_flag = 0;
        s_total = 0;
        s_ncoll = 0;
        s_maxc = 0;
        s_fill = 0;
    }
    __syncthreads();
    // `ranges` counts SUB-LISTS here (sp per (range, query)); thread t owns the contiguous sub-lists [t*per, t*per+per)
    // so that the exclusive scan of the counts below is a plain block scan of per-thread sums
    const int per = (ranges + THREADS - 1) / THREADS;
    uint32_t my = 0, mymax = 0;
    for (int j = tid * per; j < min(ranges, tid * per + per); ++j) {
        uint32_t c = cnt[((int64_t)(j / sp) * nq_pad + q) * sp + (j % sp)];
        int cap;
        (void)cand_sublist(lay, j / sp, q, j % sp, nq_pad, sp, cap);
        if (c > (uint32_t)cap) {
            s_flag = 1;  // overflow: some survivors were dropped
            c = (uint32_t)cap;
        }
        s_cnt[j] = c;
        my += c;
        mymax = c > mymax ? c : mymax;
    }
    {   // block exclusive scan of `my` -> s_off[]
        const int lane = tid & 63, wid = tid >> 6;
        uint32_t incl = my;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) s_wsum[wid] = incl;
        if (my) atomicMax(&s_maxc, mymax);
        for (int c = tid; c < dim / 8; c += THREADS)
            reinterpret_cast<uint4 *>(s_q)[c] = reinterpret_cast<const uint4 *>(Q + (int64_t)q * dim)[c];
        __syncthreads();
        uint32_t base = incl - my;
        for (int w = 0; w < wid; ++w) base += s_wsum[w];
        for (int j = tid * per; j < min(ranges, tid * per + per); ++j) {
            s_off[j] = base;
            base += s_cnt[j];
        }
        if (tid == THREADS - 1) s_total = base;
        __syncthreads();
    }
    if (tid == 0 && stat_cand) atomicAdd(stat_cand, (unsigned long long)s_total);
    // A query the fused path cannot finish is flagged.  Two kinds: candidates were DROPPED (a sub-list overflowed, or fewer
    // than k rows passed) -- the caller retries the main pass for it under the re-tightened threshold its truncated lists
    // still give (FLAG_RETRY); or the selection itself cannot be done here (mass ties around the cut, k beyond the LDS
    // budget) -- only the exact dense path helps (FLAG_DENSE, the top bit of the list entry).
    bool bad = (s_flag != 0) || (s_total < (uint32_t)k);
    bool dense_only = (s_flag == 0) && bad;   // nothing was dropped and still fewer than k rows passed: a retry cannot find more
    // (fewer than k rows passed an ESTIMATED threshold: it was too high, and no list gives a bound -- the exact path, without a hint)
    if (dense_only && thr && tid == 0) thr[q] = -INFINITY;

    // sub-list j = (range j / sp, wave-row / lane part j % sp): cand_sublist() gives its first record and capacity
    auto sub_base = [&](int j) -> int64_t {
        int cap;
        return cand_sublist(lay, j / sp, q, j % sp, nq_pad, sp, cap);
    };
    auto at = [&](int j, int sl) -> uint2 { return cand[sub_base(j) + (int64_t)sl * sp]; };
    const int coll_cap = rescore_cap;
    uint32_t kth = 0;
    int need_eq = 0;
    int n_lds = 0;   // candidates resident in s_comp
    // Every candidate (mfma score m, row d of tile t) has its exact score inside [m - c ||d||, m + c ||d||] (c = gamma ||q||), and
    // ||d|| <= nt.  The LDS records carry the lower bound m - c nt; the k-th largest of them, L, is a lower bound of the k-th
    // largest exact score, and only candidates whose UPPER bound m + c ||d|| -- the ROW's own norm here, so one huge row widens
    // nobody else's interval -- reaches L can be in the result.
    const float c = cq[q];
    bool row_lb = false;   // the records' lower bounds use the row's own norm (the rare overflow sweep) instead of its tile's

    if (!bad && s_total <= (uint32_t)compact) {
        // The sub-lists are sparse: one sweep over (sub-list, slot < longest list) gathers them into LDS at their
        // scanned offsets (independent loads, no atomics); the select then never touches global memory.
        // (1) the first FIRST slots of every cell -- whole 16-byte pairs (slot, parts 2p and 2p + 1), contiguous and
        //     independent of the counts: a cell's slots 0..3 are its first one (sp = 4) or two (sp = 8) lines (issuing these
        //     loads at kernel start, ahead of the count scan, was measured: 398 -> 488 us, the 32 live registers cost more);
        // (2) only lists longer than that are swept: (sub-list, slot FIRST + s), s < W = pow2 >= longest - FIRST.
        // (large k -- the 1 024-thread form -- has lists of three to six records: eight unconditional slots there)
        constexpr int FIRST = THREADS >= 1024 ? 8 : 4, MAXL = 8;   // ranges * FIRST / 2 <= MAXL * THREADS (1 024 / 2 048 sub-lists)
        {
            const int per_cell = FIRST * sp / 2;               // 16-byte pairs per cell
            const int nload = (ranges / sp) * per_cell;
            uint4 v[MAXL];
#pragma unroll
            for (int t = 0; t < MAXL; ++t) {
                const int x = tid + t * THREADS;
                if (x < nload) {
                    const int cell = x / per_cell, w = x - cell * per_cell;
                    int cap;
                    v[t] = *reinterpret_cast<const uint4 *>(cand + cand_sublist(lay, cell, q, 0, nq_pad, sp, cap) + 2 * w);
                }
            }
#pragma unroll
            for (int t = 0; t < MAXL; ++t) {
                const int x = tid + t * THREADS;
                if (x < nload) {
                    const int cell = x / per_cell, w = x - cell * per_cell;
                    const int slot = (2 * w) / sp, part = (2 * w) % sp;
                    const int j = cell * sp + part;
                    if ((uint32_t)slot < s_cnt[j]) s_comp[s_off[j] + slot] = make_uint2(v[t].x, v[t].y);
                    if ((uint32_t)slot < s_cnt[j + 1]) s_comp[s_off[j + 1] + slot] = make_uint2(v[t].z, v[t].w);
                }
            }
        }
        sweep_sublists<THREADS, 8>(tid, ranges, FIRST, (int)s_maxc, s_cnt, at, [](int, int) { return true; },
                                   [&](int j, int sl, uint2 e) { s_comp[s_off[j] + sl] = e; });
        __syncthreads();
        n_lds = (int)s_total;
        for (int i0 = tid; i0 < n_lds; i0 += 4 * THREADS) {   // lower bounds with the TILE's norm (a 42-KB table: cache hits)
            uint2 e[4];
            float rn[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * THREADS;
                e[u] = s_comp[i < n_lds ? i : i0];
                rn[u] = tile_norm[e[u].y / TILE_DOCS];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * THREADS;
                if (i < n_lds) s_comp[i] = make_uint2(__float_as_uint(fmaf(-c, rn[u], __uint_as_float(e[u].x))), e[u].y);
            }
        }
        __syncthreads();
    } else if (!bad && compact < k) {
        bad = dense_only = true;   // the LDS budget cannot even hold k records (huge dim * k): exact dense path
    } else if (!bad) {
        // More candidates than the LDS holds.  The k-th largest lower bound of ANY `compact` of them is a valid lower bound of
        // the query's k-th largest, so: keep the first `compact` records whose upper bound passes the current bound (none at
        // first), and while more than that passed, tighten the bound to the k-th largest lower bound of the kept ones and sweep again.
        // Every sweep at least halves what passes (the kept ones are a random part of what passed); 4 sweeps cover
        // 16 x the LDS capacity, beyond that the exact dense path takes the query.
        float keep = -INFINITY;
        bool fits = false;
        row_lb = true;
        for (int round = 0; round < 4 && !fits; ++round) {
            __syncthreads();   // everybody is done with s_comp / s_fill of the previous round
            if (tid == 0) s_fill = 0u;
            __syncthreads();
            sweep_sublists<THREADS, 8>(tid, ranges, 0, (int)s_maxc, s_cnt, at, [](int, int) { return true; },
                                       [&](int, int, uint2 e) {
                                           // (the ROW's own norm on both sides here: records of a tile that holds one huge row
                                           // would otherwise pass every sweep and contribute nothing to the bound)
                                           const float cn = c * row_norm[e.y];
                                           if (__uint_as_float(e.x) + cn >= keep) {
                                               const uint32_t p = atomicAdd(&s_fill, 1u);
                                               if (p < (uint32_t)compact) s_comp[p] = make_uint2(__float_as_uint(__uint_as_float(e.x) - cn), e.y);
                                           }
                                       });
            __syncthreads();
            if (s_fill <= (uint32_t)compact) {
                fits = true;
            } else {
                uint32_t kth0 = 0;
                int eq0 = 0;
                block_radix_select(
                    [&](int64_t i, bool &skip) -> uint32_t {
                        (void)skip;
                        return f32_orderable(__uint_as_float(s_comp[i].x));
                    },
                    compact, k, s_hist, s_ctl, kth0, eq0);
                keep = orderable_to_f32(kth0);
            }
        }
        if (fits)
            n_lds = (int)s_fill;
        else
            bad = dense_only = true;   // (near-)constant scores or k close to the LDS capacity: the exact dense path takes the query
    }
    if (!bad) {
        const int M = n_lds;
        block_radix_select(
            [&](int64_t i, bool &skip) -> uint32_t {
                (void)skip;
                return f32_orderable(__uint_as_float(s_comp[i].x));
            },
            M, k, s_hist, s_ctl, kth, need_eq);
        const float low = orderable_to_f32(kth);                                  // L: the k-th largest lower bound
        // Verification of the threshold the main pass filtered with.  Every row whose exact score reaches tau was recorded; if k
        // candidates have lower bounds >= tau (L >= tau), the k-th largest exact score is >= tau and nothing is missing.  A valid
        // lower bound (conservative thresholds, re-tightened ones) always passes; an ESTIMATED threshold that came out too high
        // fails: the query is retried under L, which is a valid bound.
        if (thr && low < thr[q]) {
            __syncthreads();
            if (tid == 0) {
                thr[q] = low;
                const uint32_t p = atomicAdd(flag_count, 1u);
                flag_list[p] = (out_rows ? out_rows[q] : (uint32_t)q);
            }
            return;
        }
        const float loose = fmaf(-2.f * c, __uint_as_float(*dmax_bits), low);    // no row's upper bound is further above its lower one
        for (int i = tid; i < M; i += THREADS) {
            const uint2 e = s_comp[i];
            const float lb = __uint_as_float(e.x);
            // (the row's own norm only here, for the few records near the cut: 4-byte reads scattered over an n_rows table --
            // taken for every record they cost the k = 1001 select +0.25 ms)
            if (lb >= loose && fmaf(c, row_norm[e.y], fmaf(c, row_lb ? row_norm[e.y] : tile_norm[e.y / TILE_DOCS], lb)) >= low) {
                const uint32_t p = atomicAdd(&s_ncoll, 1u);
                if (p < (uint32_t)coll_cap) s_keys[p] = (unsigned long long)e.y;  // local row for now
            }
        }
        __syncthreads();
        if (s_ncoll > (uint32_t)coll_cap) bad = dense_only = true;  // mass ties around the cut
    }
    if (bad) {
        if (tid == 0) {
            const uint32_t p = atomicAdd(flag_count, 1u);
            flag_list[p] = (out_rows ? out_rows[q] : (uint32_t)q) | (dense_only ? FLAG_DENSE : 0u);
        }
        return;
    }
    const int ncoll = (int)s_ncoll;
    const int np2 = pow2_ceil(ncoll);
    // Canonical re-score.  The candidate rows are scattered over the shard; a thread walking "its" row 16 bytes at a
    // time makes every wave-load touch 64 different cache lines (measured 1.5 TB/s at k = 1000).  Instead the rows are
    // staged through the (now free) candidate area of the LDS in K slices of SB bytes per row: lanes_per_row adjacent
    // lanes fetch one row's slice (whole 64/128-byte segments, all loads of a slice in flight), then every thread runs
    // the fp64 chain of its own row(s) over the slice from LDS.  Element order inside a row is unchanged.
    const size_t stage_bytes = (size_t)compact * 8;
    // (both variants: before the slices were software-pipelined the barriers cost the 256-thread form more than they saved;
    // with the pipeline the NQ select goes 405 -> 308 us -- its re-score 228 -> 130 us -- and a 1/8 shard, whose rows sit in the
    // Infinity Cache, is unchanged)
    int SB = (size_t)ncoll * 144 <= stage_bytes ? 128 : ((size_t)ncoll * 80 <= stage_bytes ? 64 : 0);
    constexpr int PRE = 8;                                               // 16-byte pieces a thread keeps in flight for the next slice
    if (SB == 128 && ncoll * 8 > PRE * THREADS) SB = 64;
    if (SB != 0 && ncoll <= 2 * THREADS && ncoll * (SB / 16) <= PRE * THREADS) {
        char *stage = reinterpret_cast<char *>(s_comp);
        const int stride = SB + 16;                 // +16: consecutive rows start 4 banks apart (conflict-free b128 reads)
        const int lpr = SB / 16;                    // lanes per row and slice (THREADS % lpr == 0: a thread's pieces share one column)
        const int row_bytes = dim * 2;
        const char *Dbytes = reinterpret_cast<const char *>(D);
        const int col = (tid % lpr) * 16;           // byte column of this thread's pieces inside a slice
        // Software pipeline: the pieces of slice s + 1 are loaded into registers while slice s is consumed from the LDS, so the
        // scattered row reads (the bulk of this stage: ~1.5 KB per re-scored row) overlap the fp64 chains instead of sitting
        // between two barriers 24 times per query.
        // (every piece is loaded unconditionally -- an unused slot reads row 0, a column past the row end re-reads column 0 --
        // so that the prefetched values stay in registers; only the store into the LDS is conditional)
        // piece j of this thread belongs to candidate (tid + j * THREADS) / lpr; only its row pointer is kept.  Eight named
        // values, not an array: hipcc leaves a uint4[8] in scratch memory under the 128-register budget of 1 024 threads.
#define CCR_PIECES(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
        // address column: a column past the row end (dim = 32 under a 128-byte slice) reads column 0 -- the FIRST load too, not only
        // the refills below: the last corpus row may end exactly at the end of the caller's allocation
        const int acol = col < row_bytes ? col : 0;
        const char *Dcol = Dbytes + acol;
        const int npiece = ncoll * lpr;
#define CCR_DECL(j)                                                                                              \
    const int idx##j = tid + j * THREADS;                                                                        \
    const char *src##j = Dcol + (int64_t)(uint32_t)s_keys[idx##j < npiece ? idx##j / lpr : 0] * row_bytes;      \
    uint4 pre##j = *reinterpret_cast<const uint4 *>(src##j);
        CCR_PIECES(CCR_DECL)
#undef CCR_DECL
        double acc[2] = {0.0, 0.0};
        for (int k0 = 0; k0 < row_bytes; k0 += SB) {
            __syncthreads();                        // the previous slice has been consumed (first pass: s_comp is free)
            const bool col_in = k0 + col < row_bytes;
#define CCR_PUT(j) \
    if (col_in && idx##j < npiece) *reinterpret_cast<uint4 *>(stage + (idx##j / lpr) * stride + col) = pre##j;
            CCR_PIECES(CCR_PUT)
#undef CCR_PUT
            __syncthreads();
            {   // next slice: in flight during the fp64 chains below
                const int kn = (k0 + SB + col < row_bytes) ? k0 + SB : -acol;
#define CCR_GET(j) pre##j = *reinterpret_cast<const uint4 *>(src##j + kn);
                CCR_PIECES(CCR_GET)
#undef CCR_GET
            }
#undef CCR_PIECES
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int i = tid + j * THREADS;
                if (i < ncoll) {
                    double a = acc[j];
                    for (int c = 0; c < lpr && k0 + c * 16 < row_bytes; ++c) {
                        const uint4 dv = *reinterpret_cast<const uint4 *>(stage + (size_t)i * stride + c * 16);
                        const uint4 qv = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(s_q) + k0 + c * 16);
                        const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w};
                        const uint32_t qw[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            a = fma((double)__uint_as_float(qw[e] << 16), (double)__uint_as_float(dw[e] << 16), a);
                            a = fma((double)__uint_as_float(qw[e] & 0xffff0000u), (double)__uint_as_float(dw[e] & 0xffff0000u), a);
                        }
                    }
                    acc[j] = a;
                }
            }
        }
        __syncthreads();                            // every loader is done reading the row ids in s_keys
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * THREADS;
            if (i < np2) s_keys[i] = i < ncoll ? make_key((float)acc[j], (uint32_t)s_keys[i]) : 0ull;
        }
    } else {
        // each thread reads and rewrites only its own key slots
        for (int i = tid; i < np2; i += blockDim.x) {
            unsigned long long key = 0ull;
            if (i < ncoll) {
                const uint32_t row = (uint32_t)s_keys[i];
                key = make_key(canonical_dot(s_q, D + (int64_t)row * dim, dim), row);
            }
            s_keys[i] = key;
        }
    }
    block_bitonic_sort_desc(s_keys, np2);
    const int64_t orow = out_rows ? (int64_t)out_rows[q] : (int64_t)q;   // retry pass: compacted query i -> original row
    for (int i = tid; i < k; i += blockDim.x) {
        const unsigned long long key = s_keys[i];
        out_scores[orow * k + i] = key_score(key);
        store_id(out_ids, orow * k + i, id_offset, key_idx(key));
    }
}

// Retry pass set-up: split the flagged list into the queries to retry and the ones that need the dense path, and gather
// the retried queries' rows, thresholds and margins into compact arrays.  One workgroup; counts[0] = retried, counts[1] = dense.
__global__ __launch_bounds__(256) void partition_flags_kernel(const uint32_t *__restrict__ flags, int begin, int n,
                                                             uint32_t *__restrict__ retry_list, uint32_t *__restrict__ dense_list,
                                                             uint32_t *__restrict__ counts) {   // dense_list: the caller's append position
    __shared__ uint32_t s_n[2];
    if (threadIdx.x == 0) s_n[0] = s_n[1] = 0;
    __syncthreads();
    // order inside the two lists does not matter (each entry is a whole query)
    for (int i = begin + threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t f = flags[i];
        if (f & FLAG_DENSE)
            dense_list[atomicAdd(&s_n[1], 1u)] = f & ~FLAG_DENSE;
        else
            retry_list[atomicAdd(&s_n[0], 1u)] = f;
    }
    __syncthreads();
    if (threadIdx.x < 2) counts[threadIdx.x] = s_n[threadIdx.x];
}

// Estimated thresholds: a query for which FEWER than k rows passed (the select left thr[q] = -inf and FLAG_DENSE) gets the conservative
// bound of the sample instead -- a valid lower bound of its k-th largest score -- and becomes an ordinary retry.
__global__ __launch_bounds__(256) void underfilled_to_retry_kernel(uint32_t *__restrict__ flags, int begin, int n,
                                                                  const float *__restrict__ thr_safe, float *__restrict__ thr) {
    const int i = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t f = flags[i];
    const uint32_t q = f & ~FLAG_DENSE;
    if ((f & FLAG_DENSE) && thr[q] == -INFINITY && thr_safe[q] > -INFINITY) {   // (a NaN bound -- non-finite embeddings -- stays on the exact path)
        thr[q] = thr_safe[q];
        flags[i] = q;
    }
}

// thr[list[i]] = thr2[i]: the re-tightened thresholds of a retry round go back to the original query order
__global__ __launch_bounds__(256) void scatter_thresholds_kernel(const uint32_t *__restrict__ list, int n, const float *__restrict__ thr2,
                                                                float *__restrict__ thr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) thr[list[i]] = thr2[i];
}

__global__ __launch_bounds__(256) void gather_queries_kernel(const uint16_t *__restrict__ Q, int dim, const uint32_t *__restrict__ list,
                                                            int n, const float *__restrict__ thr, const float *__restrict__ delta,
                                                            uint16_t *__restrict__ Q2, float *__restrict__ thr2,
                                                            float *__restrict__ delta2) {
    const int i = blockIdx.x;
    if (i >= n) return;
    const uint32_t q = list[i];
    for (int c = threadIdx.x; c < dim / 8; c += blockDim.x)
        reinterpret_cast<uint4 *>(Q2 + (int64_t)i * dim)[c] = reinterpret_cast<const uint4 *>(Q + (int64_t)q * dim)[c];
    if (threadIdx.x == 0) {
        thr2[i] = thr[q];
        delta2[i] = delta[q];
    }
}

// ---------------------------------------------------------------------------------------------
// The dynamic-LDS opt-in (> 64 KiB) is a per-device function attribute: it is set the first time a kernel is launched
// on a device and remembered per (kernel, device) -- not on every launch.
int ensure_dynamic_lds(const void *kernel, size_t lds) {
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    CCR_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({kernel, dev})) return CCR_OK;
    CCR_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    done.insert({kernel, dev});
    return CCR_OK;
}

template <class K>
static int launch_kernel(K kernel, size_t lds, const GemmArgs &a, int grid, hipStream_t s) {
    const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), lds);
    if (rc != CCR_OK) return rc;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(GEMM_THREADS), lds, s, a);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

template <int EPI>
static int launch_gemm(const GemmArgs &a, int grid, hipStream_t s) {
    const size_t lds = RING * (size_t)SUB_BYTES;
#ifdef CCR_DIAGNOSTICS
    // Timing-only ablations of the main pass (CCR_GEMM_DBG, read once at index creation): they return WRONG results, so they exist
    // only in a library built with -DCCR_DIAGNOSTICS (make DIAG=1); the shipped library has no such mode.
    if (EPI == EPI_FILTER && a.dbg != 0) {
        switch (a.dbg) {
            case 1: return launch_kernel(&gemm_topk_kernel<EPI, true, 1>, lds, a, grid, s);
            case 2: return launch_kernel(&gemm_topk_kernel<EPI, true, 2>, lds, a, grid, s);
            case 3: return launch_kernel(&gemm_topk_kernel<EPI, true, 3>, lds, a, grid, s);
            case 4: return launch_kernel(&gemm_topk_kernel<EPI, true, 4>, lds, a, grid, s);
            case 8: return launch_kernel(&gemm_topk_kernel<EPI, true, 8>, lds, a, grid, s);
            case 11: return launch_kernel(&gemm_topk_kernel<EPI, true, 11>, lds, a, grid, s);
            case 16: return launch_kernel(&gemm_topk_kernel<EPI, true, 16>, lds, a, grid, s);
            default: break;   // unknown value: the production kernel
        }
    }
#endif
    if (a.dim % SUB_K != 0) return launch_kernel(&gemm_topk_kernel<EPI, true, 0, true>, lds, a, grid, s);   // zero-filled last K sub-stage
    if (!a.stagger) return launch_kernel(&gemm_topk_kernel<EPI, false, 0>, lds, a, grid, s);   // CCR_GEMM_STAGGER=0 (A/B of the ping-pong)
    return launch_kernel(&gemm_topk_kernel<EPI, true, 0>, lds, a, grid, s);
}

int launch_gemm_filter(const GemmArgs &a, int grid, hipStream_t s) { return launch_gemm<EPI_FILTER>(a, grid, s); }
#ifdef CCR_DIAGNOSTICS
// query-direct forms (a.qdirect = 1 / 3 / 4 / 5: gemm_topk16q_kernel; diagnostic library only); dim % 32 == 0 and a query block of less than 2^31 bytes
template <int EPI>
static int launch_gemm16q(const GemmArgs &a, int grid, hipStream_t s) {
    const size_t lds = RING * (size_t)SUB_BYTES;
    switch (a.qdirect) {
        case 1: return launch_kernel(&gemm_topk16q_kernel<EPI, 1>, lds, a, grid, s);
        case 3: return launch_kernel(&gemm_topk16q_kernel<EPI, 3>, lds, a, grid, s);
        case 4: return launch_kernel(&gemm_topk16q_kernel<EPI, 4>, lds, a, grid, s);
        default: return launch_kernel(&gemm_topk16q_kernel<EPI, 5>, lds, a, grid, s);
    }
}
static bool qdirect_ok(const GemmArgs &a) {
    return a.qdirect > 0 && a.dim % SUB_K == 0 && (int64_t)a.n_q * a.dim * 2 < (int64_t)1 << 31;
}
#endif

int launch_gemm16_filter(const GemmArgs &a, int grid, hipStream_t s) {
#ifdef CCR_DIAGNOSTICS
    // timing-only ablations of the 16x16x32 main pass (CCR_GEMM_DBG; WRONG results): diagnostic library only (make DIAG=1)
    switch (a.dbg) {
        case 4: return launch_kernel(&gemm_topk16_kernel<EPI_FILTER, 4>, RING * (size_t)SUB_BYTES, a, grid, s);
        case 8: return launch_kernel(&gemm_topk16_kernel<EPI_FILTER, 8>, RING * (size_t)SUB_BYTES, a, grid, s);
        case 12: return launch_kernel(&gemm_topk16_kernel<EPI_FILTER, 12>, RING * (size_t)SUB_BYTES, a, grid, s);
        case 32: return launch_kernel(&gemm_topk16_kernel<EPI_FILTER, 32>, RING * (size_t)SUB_BYTES, a, grid, s);
        case 40: return launch_kernel(&gemm_topk16_kernel<EPI_FILTER, 40>, RING * (size_t)SUB_BYTES, a, grid, s);
        case 64: return launch_kernel(&gemm_topk16_kernel<EPI_FILTER, 64>, RING * (size_t)SUB_BYTES, a, grid, s);
        case 128: return launch_kernel(&gemm_topk16_kernel<EPI_FILTER, 128>, RING * (size_t)SUB_BYTES, a, grid, s);
        case 384: return launch_kernel(&gemm_topk16_kernel<EPI_FILTER, 384>, RING * (size_t)SUB_BYTES, a, grid, s);
        case 640: return launch_kernel(&gemm_topk16_kernel<EPI_FILTER, 640>, RING * (size_t)SUB_BYTES, a, grid, s);
        case 1152: return launch_kernel(&gemm_topk16_kernel<EPI_FILTER, 1152>, RING * (size_t)SUB_BYTES, a, grid, s);
        case 2176: return launch_kernel(&gemm_topk16_kernel<EPI_FILTER, 2176>, RING * (size_t)SUB_BYTES, a, grid, s);
        case 4224: return launch_kernel(&gemm_topk16_kernel<EPI_FILTER, 4224>, RING * (size_t)SUB_BYTES, a, grid, s);
        case 8320: return launch_kernel(&gemm_topk16_kernel<EPI_FILTER, 8320>, RING * (size_t)SUB_BYTES, a, grid, s);
        default: break;   // 0 or unknown: the production kernel
    }
#endif
#ifdef CCR_DIAGNOSTICS
    if (qdirect_ok(a)) return launch_gemm16q<EPI_FILTER>(a, grid, s);
#endif
    if (a.dim % SUB_K != 0) return launch_kernel(&gemm_topk16_kernel<EPI_FILTER, 0, true>, RING * (size_t)SUB_BYTES, a, grid, s);
    return launch_kernel(&gemm_topk16_kernel<EPI_FILTER, 0>, RING * (size_t)SUB_BYTES, a, grid, s);
}
// the 256 x 384 form (a.qblocks counts blocks of WIDE_Q queries; dim % 32 == 0)
int launch_gemm16w_filter(const GemmArgs &a, int grid, hipStream_t s) {
    if (a.dim % SUB_K != 0 || a.dim < SUB_K) {
        set_error("launch_gemm16w_filter: dim %d is not a multiple of %d", a.dim, SUB_K);
        return CCR_ERR_INVALID;
    }
#ifdef CCR_DIAGNOSTICS
    switch (a.dbg) {   // timing-only ablations (thresholds +inf in all of them)
        case 16: return launch_kernel(&gemm_topk16w_kernel<16>, WIDE_LDS, a, grid, s);     // cycle stamps (results stay right)
        case 144: return launch_kernel(&gemm_topk16w_kernel<144>, WIDE_LDS, a, grid, s);   // cycle stamps, thresholds +inf
        case 128: return launch_kernel(&gemm_topk16w_kernel<128>, WIDE_LDS, a, grid, s);
        case 132: return launch_kernel(&gemm_topk16w_kernel<132>, WIDE_LDS, a, grid, s);   // no DMA
        case 136: return launch_kernel(&gemm_topk16w_kernel<136>, WIDE_LDS, a, grid, s);   // no MFMA
        case 140: return launch_kernel(&gemm_topk16w_kernel<140>, WIDE_LDS, a, grid, s);   // neither: LDS reads + barriers + filter trees
        case 160: return launch_kernel(&gemm_topk16w_kernel<160>, WIDE_LDS, a, grid, s);   // corpus pieces only
        case 192: return launch_kernel(&gemm_topk16w_kernel<192>, WIDE_LDS, a, grid, s);   // query pieces only
        case 168: return launch_kernel(&gemm_topk16w_kernel<168>, WIDE_LDS, a, grid, s);   // corpus pieces only, no MFMA
        default: break;
    }
#endif
    return launch_kernel(&gemm_topk16w_kernel<0>, WIDE_LDS, a, grid, s);
}
int launch_gemm16_store(const GemmArgs &a, int grid, hipStream_t s) {
#ifdef CCR_DIAGNOSTICS
    if (qdirect_ok(a)) return launch_gemm16q<EPI_STORE>(a, grid, s);
#endif
    if (a.dim % SUB_K != 0) return launch_kernel(&gemm_topk16_kernel<EPI_STORE, 0, true>, RING * (size_t)SUB_BYTES, a, grid, s);
    return launch_kernel(&gemm_topk16_kernel<EPI_STORE, 0>, RING * (size_t)SUB_BYTES, a, grid, s);
}
int launch_gemm_gmax(const GemmArgs &a, int grid, hipStream_t s) { return launch_gemm<EPI_GMAX>(a, grid, s); }
int launch_gemm_store(const GemmArgs &a, int grid, hipStream_t s) { return launch_gemm<EPI_STORE>(a, grid, s); }

int launch_row_norms_bf16(const uint16_t *X, int64_t rows, int dim, float *norms, uint32_t *max_bits, uint32_t *tile_bits,
                          hipStream_t s) {
    if (rows <= 0) return CCR_OK;
    int64_t blocks = tile_bits ? (rows + 255) / 256 : (rows + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(row_norms_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, X, rows, dim, norms, max_bits, tile_bits);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

int launch_tile_norms(const float *row_bounds, int64_t rows, uint32_t *tile_bits, uint32_t *max_bits, hipStream_t s) {
    if (rows <= 0) return CCR_OK;
    const int64_t tiles = (rows + TILE_DOCS - 1) / TILE_DOCS;
    hipLaunchKernelGGL(tile_norms_kernel, dim3((unsigned)tiles), dim3(256), 0, s, row_bounds, rows, tile_bits);
    hipLaunchKernelGGL(max_tile_norm_kernel, dim3(1), dim3(1024), 0, s, tile_bits, tiles, max_bits);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

// The same thresholds for a SMALL batch (the streaming main pass, n_q <= 128): threshold_kernel gives a query 16 threads, which walk its
// ~4 000 sampled lower bounds four times from global memory -- 70 us for ONE query, a tenth of the whole search.  Here a workgroup
// per query: the lower bounds are staged in LDS once (every load independent, one round trip), the four radix passes read LDS.
// zero_cnt (may be null): the streaming pass's sub-list counters of this query are cleared here instead of by a memset node.
// grid = n_q, block = 256, dyn LDS = n_groups * 4 bytes.
__global__ __launch_bounds__(256) void threshold_small_kernel(const float *__restrict__ gmax, int n_groups, int nq_pad, int k,
                                                             const float *__restrict__ qnorm, const uint32_t *__restrict__ dmax_bits,
                                                             float gamma, const float *__restrict__ tile_norm, int64_t sample_stride,
                                                             float *__restrict__ thr, float *__restrict__ cq, uint32_t *__restrict__ zero_cnt,
                                                             int zero_per_query) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_low[];
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_ctl[4];
    const int tid = threadIdx.x, q = blockIdx.x;
    const float dmax = __uint_as_float(*dmax_bits);
    const float c = (dmax < INFINITY) ? gamma * (qnorm[q] * 1.001f) * 1.001f : __builtin_nanf("");
    for (int g = tid; g < n_groups; g += 256)
        s_low[g] = f32_orderable(fmaf(-c, tile_norm[(int64_t)(g / GROUPS_PER_TILE) * sample_stride], gmax[(int64_t)g * nq_pad + q]));
    if (zero_cnt && tid < zero_per_query) zero_cnt[q * zero_per_query + tid] = 0u;
    __syncthreads();
    uint32_t kth = 0;
    int need_eq = 0;
    block_radix_select<true>(
        [&](int64_t i, bool &skip) -> uint32_t {
            (void)skip;
            return s_low[i];
        },
        n_groups, k, s_hist, s_ctl, kth, need_eq);
    if (tid == 0) {
        cq[q] = c;
        thr[q] = (c < INFINITY) ? orderable_to_f32(kth) : __builtin_nanf("");
    }
}

int launch_threshold(const float *gmax, int64_t n_groups, int n_q, int nq_pad, int k, const float *qnorm,
                     const uint32_t *dmax_bits, int dim, const float *tile_norm, int64_t sample_stride, float *thr, float *cq,
                     hipStream_t s, uint32_t *zero_cnt, int zero_per_query) {
    // small batches: a workgroup per query (needs k <= n_groups -- the planner samples >= 2k groups -- and the bounds in 64 KiB of LDS);
    // *zero_cnt tells the caller whether its counters have been cleared here
    if (n_q <= 128 && n_groups >= k && n_groups <= 16384 && zero_per_query <= 256) {
        if (n_groups * 4 > 32 * 1024) {   // (the kernel's static LDS sits on top of the dynamic part: opt in with head room)
            const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&threshold_small_kernel), 72 * 1024);
            if (rc != CCR_OK) return rc;
        }
        hipLaunchKernelGGL(threshold_small_kernel, dim3(n_q), dim3(256), (size_t)n_groups * 4, s, gmax, (int)n_groups, nq_pad, k, qnorm, dmax_bits,
                           mfma_gamma(dim), tile_norm, sample_stride, thr, cq, zero_cnt, zero_per_query);
        CCR_LAUNCH_CHECK();
        return CCR_OK;
    }
    if (zero_cnt) CCR_HIP_CHECK(hipMemsetAsync(zero_cnt, 0, (size_t)n_q * zero_per_query * 4, s));
    hipLaunchKernelGGL(threshold_kernel, dim3(nq_pad / 16), dim3(256), 0, s, gmax, n_groups, n_q, nq_pad, k, qnorm, dmax_bits,
                       mfma_gamma(dim), tile_norm, sample_stride, thr, cq);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

int launch_threshold_update(const uint2 *cand, const uint32_t *cnt, int nsub, int nsub_part, int part_blocks, int prev_nsub, int prev_part,
                            int prev_blocks, int qb_per, int sp, int n_q, int nq_pad, const CandLayout &lay, int k, const float *cq,
                            const float *tile_norm, uint32_t *top, bool top_in, bool top_out, float *thr, hipStream_t s, int tile_q) {
    if (nsub_part < nsub) nsub_part = nsub;
    if (prev_part < prev_nsub) prev_part = prev_nsub;
    if (nsub_part > 2048) {
        set_error("threshold_update: %d sub-lists exceed 2048", nsub_part);
        return CCR_ERR_INVALID;
    }
    if (!top) top_in = top_out = false;
    // LDS score buffer: comfortably more than k (the bound tightens with the number of rows seen), 16 KiB at least
    int compact = std::max(4096, 8 * pow2_ceil(k));
    if (compact > 32768) compact = 32768;
    const size_t lds = (size_t)compact * 4;
    if (lds > 48 * 1024) {
        const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&threshold_update_kernel), 128 * 1024);
        if (rc != CCR_OK) return rc;
    }
    hipLaunchKernelGGL(threshold_update_kernel, dim3(n_q), dim3(256), lds, s, cand, cnt, nsub, nsub_part, part_blocks, prev_nsub, prev_part,
                       prev_blocks, qb_per > 0 ? qb_per : 1, sp, nq_pad, lay, k, compact, cq, tile_norm, top, top_in ? 1 : 0, top_out ? 1 : 0,
                       thr, tile_q > 0 ? tile_q : TILE_Q);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

constexpr size_t SELECT_LDS_BUDGET = 150 * 1024;   // dynamic LDS the select kernel may ask for (160 KiB per CU)

static size_t select_fixed_lds(int dim, int ranges, int rescore_cap) {
    const size_t cnt_bytes = ((size_t)(ranges + 1) * 4 + 15) & ~(size_t)15;
    return (((size_t)dim * 2 + 15) & ~(size_t)15) + 2 * cnt_bytes + (size_t)rescore_cap * 8;
}

// Candidates per query the select kernel can gather into LDS: `want` (the planner's expectation with head room),
// at least 2048 (a smaller area lets more queries share a CU: the re-score is latency-bound, 0.39 -> 0.37 ms at NQ and
// 0.25 -> 0.21 ms on a 1/8 shard against a 4096 floor), never more than the LDS budget allows.
int select_compact_entries(int dim, int ranges, int rescore_cap, int64_t want) {
    const size_t fixed = select_fixed_lds(dim, ranges, rescore_cap);
    const int64_t fit = fixed + 2048 < SELECT_LDS_BUDGET ? (int64_t)((SELECT_LDS_BUDGET - fixed) / 8) : 256;
    return (int)std::min<int64_t>(std::max<int64_t>(want, 2048), fit) / 256 * 256;
}

int launch_partition_flags(const uint32_t *flags, int begin, int n, uint32_t *retry_list, uint32_t *dense_list, uint32_t *counts,
                           hipStream_t s) {
    hipLaunchKernelGGL(partition_flags_kernel, dim3(1), dim3(256), 0, s, flags, begin, n, retry_list, dense_list, counts);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

int launch_underfilled_to_retry(uint32_t *flags, int begin, int n, const float *thr_safe, float *thr, hipStream_t s) {
    if (n <= begin) return CCR_OK;
    hipLaunchKernelGGL(underfilled_to_retry_kernel, dim3((n - begin + 255) / 256), dim3(256), 0, s, flags, begin, n, thr_safe, thr);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

int launch_scatter_thresholds(const uint32_t *list, int n, const float *thr2, float *thr, hipStream_t s) {
    if (n <= 0) return CCR_OK;
    hipLaunchKernelGGL(scatter_thresholds_kernel, dim3((n + 255) / 256), dim3(256), 0, s, list, n, thr2, thr);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

int launch_gather_queries(const uint16_t *Q, int dim, const uint32_t *list, int n, const float *thr, const float *delta, uint16_t *Q2,
                          float *thr2, float *delta2, hipStream_t s) {
    if (n <= 0) return CCR_OK;
    hipLaunchKernelGGL(gather_queries_kernel, dim3(n), dim3(256), 0, s, Q, dim, list, n, thr, delta, Q2, thr2, delta2);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

int launch_select_rescore(const uint2 *cand, const uint32_t *cnt, int ranges, int sp, int n_q, int nq_pad, const CandLayout &lay, int k,
                          int rescore_cap, int compact, int64_t n_rows, float *thr, const float *cq, const float *tile_norm, const float *row_norm,
                          const uint32_t *dmax_bits, const uint16_t *Q,
                          const uint16_t *D, int dim, int64_t id_offset, float *out_scores, int64_t *out_ids,
                          uint32_t *flag_count, uint32_t *flag_list, unsigned long long *stat_cand, const uint32_t *out_rows,
                          hipStream_t s) {
    const size_t lds = select_fixed_lds(dim, ranges, rescore_cap) + (size_t)compact * 8;
    const bool wide = rescore_cap > 512 || compact > 8192 || ranges > 1024;   // 1024 threads: one re-scored row per thread at large k
    auto go = [&](auto kernel, int threads) -> int {
        if (lds > 48 * 1024) {
            const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), SELECT_LDS_BUDGET + 8192);
            if (rc != CCR_OK) return rc;
        }
        hipLaunchKernelGGL(kernel, dim3(n_q), dim3(threads), lds, s, cand, cnt, ranges, sp, nq_pad, lay, k, rescore_cap, compact,
                           n_rows, thr, cq, tile_norm, row_norm, dmax_bits, Q, D, dim, id_offset, out_scores, out_ids, flag_count, flag_list, stat_cand,
                           out_rows);
        CCR_LAUNCH_CHECK();
        return CCR_OK;
    };
    if (wide) return go(&select_rescore_kernel<1024>, 1024);
    return go(&select_rescore_kernel<256>, 256);
}

}  // namespace ccr
