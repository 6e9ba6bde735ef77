// ccr_fused12.hip -- twelve-wave variant of the fused GEMM + top-k main pass: THREE waves per SIMD take turns on
// the matrix pipe, so a wave's memory work (LDS-DMA issue, operand reads, tile filter) may take up to twice its MFMA
// time before the pipe idles (the eight-wave ping-pong of ccr_fused.hip leaves it idle ~40 % of the time: its
// memory phase is ~1,100 cycles against ~500 cycles of MFMAs).
//
// Geometry: 384 corpus rows x 192 queries per workgroup tile, 768 threads = 12 waves as 4 (row quarters of 96 rows)
// x 3 (query thirds of 64); a wave owns 96 x 64 outputs = 6 x 4 tiles of v_mfma_f32_16x16x32_bf16 (96 accumulator
// registers; 3 waves per SIMD leave 168 registers per wave).  Waves w, w + 4, w + 8 share a SIMD and form the three
// rotation groups g = w / 4.  K is walked in 32-element sub-stages through a ring of four 36-KiB LDS buffers
// [384 corpus rows x 64 B][192 query rows x 64 B] (same row swizzle as the 16x16x32 kernel of ccr_fused.hip).
//
// Schedule.  Per sub-stage u every wave runs three phases separated by workgroup barriers:
//     P0(u): [filter of a finished tile] . wait OWN DMA of u + 1 (counted vmcnt) . 10 ds_read_b128 of u
//     P1(u): issue DMA of u + 3 . lgkmcnt(0)
//     P2(u): 24 MFMAs
// and group g runs g barrier intervals behind group 0: wave group g executes phase p of sub-stage u in interval
// I = 3u + p + g, so every interval has exactly one group on the matrix pipe.
// Ring protocol:
//   RAW  DMA of u is confirmed by its issuing wave in P0(u - 1), interval <= 3u - 1; the first read of u is group
//        0's P0(u) in interval 3u: at least one barrier in between.
//   WAR  DMA of u + 3 overwrites the buffer of u - 1.  It is issued in P1(u), interval >= 3u + 1; the last reads of
//        u - 1 are issued in group 2's P0(u - 1) (interval 3u - 1) and retired (lgkmcnt(0)) in its P1(u - 1), before
//        the barrier that ends interval 3u.
#include <stdlib.h>

#include "ccr_gemm_common.h"
#include "ccr_index.h"

namespace ccr {

typedef float f32x4w __attribute__((ext_vector_type(4)));

template <int EPI>
__global__ __launch_bounds__(W12_THREADS, 3) void gemm_topk12_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..11
    const int grp = wv >> 2;    // rotation group = query third
    const int wd = wv & 3;      // row quarter
    const int wq = grp;
    const int l15 = lane & 15;
    const int lq = lane >> 4;
    const int KS2 = a.dim / SUB_K;

    const int srow = wv * 16 + (lane >> 2);   // row of this lane inside a 192-row DMA slab
    const int schunk = (lane & 3) ^ (((srow >> 2) & 1) << 1);
    const int cofs = ((lq ^ (((lane >> 2) & 1) << 1)) << 4);
    const int a_base = (wd * 96 + l15) * 64 + cofs;                  // + dt * 1024
    const int b_base = W12_Q_REGION + (wq * 64 + l15) * 64 + cofs;   // + qt * 1024

    const int xcd = blockIdx.x & (NUM_XCD - 1);
    const int jx = blockIdx.x >> 3;
    const int per_x = gridDim.x >> 3;
    const int qg = xcd % a.qgroups;
    const int rc = xcd / a.qgroups;
    const int nrc = NUM_XCD / a.qgroups;
    const int qb_per = a.qblocks / a.qgroups;
    const int rl0 = a.range_begin / nrc;
    const int rl_x = (a.range_end - a.range_begin) / nrc;
    const int count_x = rl_x * qb_per;

    for (int item = jx; item < count_x; item += per_x) {
        const int rl = rl0 + item / qb_per;
        const int qb = qg * qb_per + item % qb_per;
        const int r = rc + nrc * rl;
        const int ntile = (int)((a.n_vt - r + a.ranges - 1) / a.ranges);
        if (ntile <= 0) continue;
        const int q0 = qb * W12_TILE_Q;

        float thr[4] = {0.f, 0.f, 0.f, 0.f};
        uint32_t ncand[4] = {0u, 0u, 0u, 0u};
        // first record of the lane's sub-list for its first query column, relative to a.cand (32-bit: the candidate area
        // holds < 2^32 records); the lane's other columns are 16 queries = 256 sub-lists further each
        uint32_t coff0 = 0u;
        if (EPI == EPI_FILTER) {
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                const int q = q0 + wq * 64 + qt * 16 + l15;
                thr[qt] = (q < a.n_q) ? a.thr[q] : __builtin_nanf("");
            }
            coff0 = (uint32_t)((((int64_t)(r - a.cand_range0) * a.nq_pad + (q0 + wq * 64 + l15)) * 16 + wd * 4 + lq) * a.cap);
            asm volatile("" : "+v"(thr[0]), "+v"(thr[1]), "+v"(thr[2]), "+v"(thr[3]));
        }
        // DMA sources as 32-bit offsets in 16-byte units from the (scalar) base pointers (a shard of < 64 GiB)
        const uint32_t dim16 = (uint32_t)a.dim >> 3;
        int qrow = q0 + srow;
        if (qrow > a.n_q - 1) qrow = a.n_q - 1;
        const uint32_t qoff = (uint32_t)qrow * dim16 + schunk;

        f32x4w acc[6][4];
        const int U = ntile * KS2;

        int iu = 0, it = 0, iks = 0;
        uint32_t doff[2];
        auto tile_ptrs = [&]() {
            const uint32_t row0 = (uint32_t)(((int64_t)r + (int64_t)it * a.ranges) * a.tile_stride) * W12_TILE_DOCS;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                uint32_t drow = row0 + i * 192 + srow;
                if (drow > (uint32_t)(a.n_rows - 1)) drow = (uint32_t)(a.n_rows - 1);
                doff[i] = drow * dim16 + schunk;
            }
        };
        tile_ptrs();
        auto issue = [&]() {
            char *buf = smem + (iu & (RING - 1)) * W12_SUB_BYTES;
            const uint32_t k0 = (uint32_t)iks * (SUB_K / 8);
#pragma unroll
            for (int i = 0; i < 2; ++i) glds16(a.D + (size_t)(doff[i] + k0) * 8, buf + (i * 768 + wv * 64) * 16);
            glds16(a.Q + (size_t)(qoff + k0) * 8, buf + W12_Q_REGION + (wv * 64) * 16);
            ++iu;
            if (++iks == KS2) {
                iks = 0;
                ++it;
                tile_ptrs();
            }
        };

        // C layout of v_mfma_f32_16x16x32: lane -> query column (lane & 15), register e -> corpus row 4 * (lane >> 4) + e
        auto epilogue = [&](int vt) {
            const uint32_t row_base = (uint32_t)((int64_t)vt * a.tile_stride) * W12_TILE_DOCS + wd * 96 + 4 * lq;   // + dt * 16 + e
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                if (EPI == EPI_FILTER) {
                    float sub[6];
#pragma unroll
                    for (int dt = 0; dt < 6; ++dt)
                        sub[dt] = fmaxf(fmaxf(acc[dt][qt][0], acc[dt][qt][1]), fmaxf(acc[dt][qt][2], acc[dt][qt][3]));
                    const float t = thr[qt];
                    const float mall = fmaxf(fmaxf(fmaxf(sub[0], sub[1]), fmaxf(sub[2], sub[3])), fmaxf(sub[4], sub[5]));
                    if (__ballot(mall >= t) != 0ull) {
#pragma unroll
                        for (int dt = 0; dt < 6; ++dt) {
                            if (sub[dt] >= t) {  // rare, divergent: 4 rows to test
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const float v = acc[dt][qt][e];
                                    const uint32_t doc = row_base + dt * 16 + e;
                                    if (v >= t && doc < (uint32_t)a.n_rows) {
                                        if (ncand[qt] < (uint32_t)a.cap)
                                            a.cand[(size_t)(coff0 + (uint32_t)(qt * 256) * (uint32_t)a.cap + ncand[qt])] =
                                                make_uint2(__float_as_uint(v), doc);
                                        ++ncand[qt];
                                    }
                                }
                            }
                        }
                    }
                } else {  // EPI_STORE
                    const int q = q0 + wq * 64 + qt * 16 + l15;
#pragma unroll
                    for (int dt = 0; dt < 6; ++dt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t doc = row_base + dt * 16 + e;
                            if (q < a.n_q && doc < (uint32_t)a.n_rows) a.store[(int64_t)q * a.n_rows + doc] = acc[dt][qt][e];
                        }
                }
            }
        };

        // ---- prologue: up to 3 sub-stages in flight, sub-stage 0 confirmed and published
        const int npro = U < 3 ? U : 3;
        for (int i = 0; i < npro; ++i) issue();
        if (npro == 3)
            CCR_WAIT_VM(6);
        else if (npro == 2)
            CCR_WAIT_VM(3);
        else
            CCR_WAIT_VM(0);
        CCR_BARRIER();
        for (int i = 0; i < grp; ++i) CCR_BARRIER();   // group g runs g intervals behind group 0

        int cks = 0, ct = 0;
        bool pending = false;
        int pending_vt = 0;
        for (int u = 0; u < U; ++u) {
            // ================= P0: filter of a finished tile, confirm own DMA of u + 1, operand reads of u
            if (pending) {
                epilogue(pending_vt);
                pending = false;
            }
            if (u + 1 < U) {
                if (u + 2 < U)
                    CCR_WAIT_VM(3);
                else
                    CCR_WAIT_VM(0);
            }
            const char *buf = smem + (u & (RING - 1)) * W12_SUB_BYTES;
            bf16x8 af[6], bfr[4];
#pragma unroll
            for (int dt = 0; dt < 6; ++dt) af[dt] = *reinterpret_cast<const bf16x8 *>(buf + a_base + dt * 1024);
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) bfr[qt] = *reinterpret_cast<const bf16x8 *>(buf + b_base + qt * 1024);
            CCR_BARRIER();
            // ================= P1: DMA of u + 3 (its buffer's last readers retired their reads one interval ago), operands landed
            if (u + 3 < U) issue();
            CCR_WAIT_LGKM0();
            CCR_BARRIER();
            // ================= P2
            if (cks == 0) {
                const f32x4w z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dt = 0; dt < 6; ++dt)
#pragma unroll
                    for (int qt = 0; qt < 4; ++qt)
                        acc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[dt], bfr[qt], z, 0, 0, 0);
            } else {
#pragma unroll
                for (int dt = 0; dt < 6; ++dt)
#pragma unroll
                    for (int qt = 0; qt < 4; ++qt)
                        acc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[dt], bfr[qt], acc[dt][qt], 0, 0, 0);
            }
            if (++cks == KS2) {
                cks = 0;
                pending = true;
                pending_vt = r + ct * a.ranges;
                ++ct;
            }
            CCR_BARRIER();
        }
        if (pending) epilogue(pending_vt);
        for (int i = grp; i < 2; ++i) CCR_BARRIER();   // every wave executes the same number of barriers

        if (EPI == EPI_FILTER) {
#pragma unroll
            for (int qt = 0; qt < 4; ++qt)
                a.cnt[((int64_t)r * a.nq_pad + q0 + wq * 64 + qt * 16 + l15) * 16 + wd * 4 + lq] = ncand[qt];
        }
        __syncthreads();
    }
}

static int launch12(const void *kernel, const GemmArgs &a, int grid, hipStream_t s, void (*k)(const GemmArgs)) {
    const size_t lds = RING * (size_t)W12_SUB_BYTES;
    int rc = ensure_dynamic_lds(kernel, lds);
    if (rc != CCR_OK) return rc;
    hipLaunchKernelGGL(k, dim3(grid), dim3(W12_THREADS), lds, s, a);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

int launch_gemm12_filter(const GemmArgs &a, int grid, hipStream_t s) {
    return launch12(reinterpret_cast<const void *>(&gemm_topk12_kernel<EPI_FILTER>), a, grid, s, &gemm_topk12_kernel<EPI_FILTER>);
}
int launch_gemm12_store(const GemmArgs &a, int grid, hipStream_t s) {
    return launch12(reinterpret_cast<const void *>(&gemm_topk12_kernel<EPI_STORE>), a, grid, s, &gemm_topk12_kernel<EPI_STORE>);
}

}  // namespace ccr
