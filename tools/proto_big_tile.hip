// proto_big_tile.hip -- TIMING-ONLY prototype (round 6): does a larger register tile lift the main pass off the L1 miss path?
//
// gemm_topk16_kernel is bound by the CU's L1 miss path (~0.3 64-byte requests per clock); its 256 x 256 tile moves
// (256 + 256) x 64 B per 32-deep K step.  The tile is fixed by the accumulator budget of EIGHT waves x 128 registers.  ONE wave per SIMD
// owns all 512 registers of its lane: 256 AGPRs + 128 VGPRs of accumulators = a 128 x 192 wave tile, a 256 x 384 workgroup tile (four
// waves as 2 x 2), (256 + 384) x 64 B per K step for 1.5x the multiply-adds: -17 % bytes per flop.  hipcc cannot allocate that many
// accumulators through the MFMA builtin (round 1: it spills); here every MFMA is an inline-asm statement whose accumulator operand is
// constrained to the AGPR ("+a") or VGPR ("+v") class, so the allocation is forced.
//
// What it does: S^T tiles of D [rows][768] x Q [3456][768] bf16 exactly like the main pass (same LDS image and swizzle, LDS-DMA ring of
// three 40-KiB slots, one barrier per K step, persistent items (corpus range, query block), XCD-aware order), a max-tree epilogue per
// tile (a hit would be stored; the threshold is +inf) -- no candidate lists, no thresholds: a number to hold against the production
// kernel's no-hit baseline (profiles/r06_main_pass_ablation.txt, dbg = 128) on the same box.  Checks its own arithmetic on one tile.
//
//   hipcc -O3 --offload-arch=gfx950 tools/proto_big_tile.hip -o /tmp/proto_big_tile && /tmp/proto_big_tile
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

constexpr int DIM = 768, KS2 = DIM / 32;
constexpr int TD = 256, TQ = 384;                 // workgroup tile
constexpr int SUB = (TD + TQ) * 64;               // 40 KiB per K step
constexpr int QREG = TD * 64;
constexpr int RINGB = 3;
constexpr int NXCD = 8;

__device__ __forceinline__ void glds16(const void *g, char *lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}
#define BARRIER() asm volatile("s_barrier" ::: "memory")
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// accumulator tiles: AGPR class for the first 64, VGPR class for the last 32
__device__ __forceinline__ void mfma_a(f32x4 &c, bf16x8 a, bf16x8 b) { asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void mfma_a0(f32x4 &c, bf16x8 a, bf16x8 b) { asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&a"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void mfma_v(f32x4 &c, bf16x8 a, bf16x8 b) { asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void mfma_v0(f32x4 &c, bf16x8 a, bf16x8 b) { asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void zero_a(f32x4 &c) {
    asm("v_accvgpr_write_b32 %0, 0\n\tv_accvgpr_write_b32 %1, 0\n\tv_accvgpr_write_b32 %2, 0\n\tv_accvgpr_write_b32 %3, 0" : "=a"(c[0]), "=a"(c[1]), "=a"(c[2]), "=a"(c[3]));
}
__device__ __forceinline__ f32x4 from_a(const f32x4 &c) {   // AGPR tile -> VGPRs
    f32x4 r;
    asm("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
        : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3])
        : "a"(c[0]), "a"(c[1]), "a"(c[2]), "a"(c[3]));
    return r;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ void lds_read(u32x4 &dst, uint32_t addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF)); }
template <int... Is, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Is...>, F f) { (f(std::integral_constant<int, Is>{}), ...); }
#define SEQ(n) std::make_integer_sequence<int, n>{}
__device__ __forceinline__ bf16x8 bits(const u32x4 &x) { return __builtin_bit_cast(bf16x8, x); }

struct Args {
    const uint16_t *D, *Q;
    int64_t n_rows;
    int n_q, ranges, qblocks;
    int64_t n_vt;
    float thr;
    float *out;        // hits (never) + the check tile
    int check;         // 1: store the accumulators of item 0's first tile to out (256 x 384 fp32)
};

template <int SPREAD>
__global__ __launch_bounds__(256, 1) void big_tile_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wd = wv >> 1, wq = wv & 1;
    const int l15 = lane & 15, lq = lane >> 4;
    const int cofs = ((lq ^ (((lane >> 2) & 1) << 1)) << 4);
    const int a_base = (wd * 128 + l15) * 64 + cofs;              // + dt * 1024
    const int b_base = QREG + (wq * 192 + l15) * 64 + cofs;      // + qt * 1024
    // DMA pieces of this wave: 40 per K step, 10 per wave: piece p = wv * 10 + i covers image rows p * 16 .. + 15 (rows 0-255 corpus, 256-639 queries)
    const int prow = lane >> 2;
    const int xcd = blockIdx.x & (NXCD - 1), jx = blockIdx.x >> 3, per_x = gridDim.x >> 3;
    const int n_rl = a.ranges / NXCD;
    const int items = n_rl * a.qblocks;
    for (int item = jx; item < items; item += per_x) {
        const int rl = item / a.qblocks, qb = item % a.qblocks;    // consecutive items (co-resident workgroups of the XCD): one range, different query blocks
        const int r = xcd + NXCD * rl;
        const int64_t ntile = (a.n_vt - r + a.ranges - 1) / a.ranges;
        if (ntile <= 0) continue;
        const int q0 = qb * TQ;
        // piece p = wv * 10 + i of a K step covers image rows p * 16 .. + 15: a wave-uniform base (corpus tile or query block + p * 16 rows +
        // the K offset) + ONE per-lane offset (row lane >> 2 of the piece, source chunk by the swizzle) -- the buffers are padded to whole
        // tiles / query blocks, so no clamp is needed
        const uint32_t lane_off = (uint32_t)(prow * (DIM * 2) + (((lane & 3) ^ (((prow >> 2) & 1) << 1)) << 4));
        int64_t it = 0;
        int iks = 0;
        int64_t iu = 0;
        const char *qblk = reinterpret_cast<const char *>(a.Q) + (int64_t)q0 * (DIM * 2);
        auto issue_piece = [&](int i) __attribute__((always_inline)) {   // piece i (0 .. 9) of this wave for the K step (it, iks)
            char *buf = smem + (int)(iu % RINGB) * SUB;
            int64_t vt = r + it * a.ranges;                       // the K steps requested past the item's last tile re-read that tile
            if (vt > a.n_vt - 1) vt = a.n_vt - 1;
            const char *dtile = reinterpret_cast<const char *>(a.D) + vt * (int64_t)TD * (DIM * 2);
            const int kb = iks * 64;
            const int p = wv * 10 + i;                             // wave-uniform
            const char *base = (p < TD / 16 ? dtile + (int64_t)p * 16 * (DIM * 2) : qblk + (int64_t)(p - TD / 16) * 16 * (DIM * 2)) + kb;
            glds16(base + lane_off, buf + p * 1024);
        };
        auto advance = [&]() __attribute__((always_inline)) {
            ++iu;
            if (++iks == KS2) {
                iks = 0;
                ++it;
            }
        };
        auto issue = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 10; ++i) issue_piece(i);
            advance();
        };

        f32x4 acc_a[8][8], acc_v[8][4];     // [dt][qt]: query tiles 0-7 in AGPRs, 8-11 in VGPRs
        auto zero_all = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
#pragma unroll
                for (int qt = 0; qt < 8; ++qt) zero_a(acc_a[dt][qt]);
#pragma unroll
                for (int qt = 0; qt < 4; ++qt) acc_v[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        zero_all();
        const int64_t U = ntile * KS2;
        issue();
        issue();                             // (unconditional: an item has >= 24 K steps)
        int cks = 0;
        int64_t ct = 0;
        for (int64_t u = 0; u < U; ++u) {
            wait_vm<10>();                   // this wave's pieces of K step u have landed (those of u + 1 may be in flight)
            BARRIER();                       // ... and everybody's; every wave has finished with the slot of u - 1
            if (SPREAD == 0) issue();        // K step u + 2 into the slot of u - 1 (past the item's end: the last tile again, never read)
            const char *buf = smem + (int)(u % RINGB) * SUB;
            bf16x8 af[8], bq[12];
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) af[dt] = *reinterpret_cast<const bf16x8 *>(buf + a_base + dt * 1024);
#pragma unroll
            for (int qt = 0; qt < 12; ++qt) bq[qt] = *reinterpret_cast<const bf16x8 *>(buf + b_base + qt * 1024);
#pragma unroll
            for (int qt = 0; qt < 12; ++qt) {
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) {
                    if (qt < 8)
                        mfma_a(acc_a[dt][qt < 8 ? qt : 0], af[dt], bq[qt]);
                    else
                        mfma_v(acc_v[dt][qt >= 8 ? qt - 8 : 0], af[dt], bq[qt]);
                }
                if (SPREAD == 1 && qt < 10) {   // one DMA piece of K step u + 2 behind every eight MFMAs
                    __builtin_amdgcn_sched_barrier(0);
                    issue_piece(qt);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (SPREAD == 1) advance();
            if (++cks == KS2) {
                cks = 0;
                // epilogue of the finished tile: the filter's max trees (per query tile: the maximum of the lane's 32 rows), a hit is stored
                float m = -INFINITY;
#pragma unroll
                for (int qt = 0; qt < 8; ++qt) {
                    float mq = -INFINITY;
#pragma unroll
                    for (int dt = 0; dt < 8; ++dt) {
                        const f32x4 c = from_a(acc_a[dt][qt]);
                        mq = fmaxf(mq, fmaxf(fmaxf(c[0], c[1]), fmaxf(c[2], c[3])));
                    }
                    m = fmaxf(m, mq);
                }
#pragma unroll
                for (int qt = 0; qt < 4; ++qt) {
                    float mq = -INFINITY;
#pragma unroll
                    for (int dt = 0; dt < 8; ++dt) {
                        const f32x4 c = acc_v[dt][qt];
                        mq = fmaxf(mq, fmaxf(fmaxf(c[0], c[1]), fmaxf(c[2], c[3])));
                    }
                    m = fmaxf(m, mq);
                }
                if (m >= a.thr) a.out[(size_t)TD * TQ + blockIdx.x * 256 + tid] = m;   // (never: thr = +inf)
                if (a.check && item == 0 && blockIdx.x == 0 && ct == 0) {
                    // C layout of 16x16x32: lane -> query column (lane & 15), register e -> corpus row 4 * (lane >> 4) + e
#pragma unroll
                    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
                        for (int qt = 0; qt < 12; ++qt) {
                            const f32x4 c = qt < 8 ? from_a(acc_a[dt][qt < 8 ? qt : 0]) : acc_v[dt][qt >= 8 ? qt - 8 : 0];
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                a.out[(size_t)(wd * 128 + dt * 16 + 4 * lq + e) * TQ + wq * 192 + qt * 16 + l15] = c[e];
                        }
                }
                ++ct;
                zero_all();      // ONE code path for the MFMAs (a C = 0 variant in a branch makes hipcc merge 96 tiles with copies and spill them)
            }
        }
        wait_vm<0>();
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Variant 2: the SAME tile as a software-pipelined single-wave stream.  A K step's 96 MFMAs run in two halves -- corpus tiles dt 0-3,
// then dt 4-7 -- and the operand registers are refilled IN PLACE from the next K step's ring slot as they fall free: af[0..3] behind the
// first half, bq[qt] behind the four MFMAs of query tile qt in the second half, af[4..7] at the end (the next step needs them only in ITS
// second half).  The rendezvous of K step u + 1 (own DMA landed -> barrier) sits between the halves of step u; the ten DMA pieces of
// step u + 3 go out between the query tiles of the second half.  No operand double buffer: 249 VGPRs are in use as it is.
__global__ __launch_bounds__(256, 1) void big_tile_pipe_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wd = wv >> 1, wq = wv & 1;
    const int l15 = lane & 15, lq = lane >> 4;
    const int cofs = ((lq ^ (((lane >> 2) & 1) << 1)) << 4);
    const int a_base = (wd * 128 + l15) * 64 + cofs;
    const int b_base = QREG + (wq * 192 + l15) * 64 + cofs;
    const int prow = lane >> 2;
    const int xcd = blockIdx.x & (NXCD - 1), jx = blockIdx.x >> 3, per_x = gridDim.x >> 3;
    const int n_rl = a.ranges / NXCD;
    const int items = n_rl * a.qblocks;
    for (int item = jx; item < items; item += per_x) {
        const int rl = item / a.qblocks, qb = item % a.qblocks;
        const int r = xcd + NXCD * rl;
        const int64_t ntile = (a.n_vt - r + a.ranges - 1) / a.ranges;
        if (ntile <= 0) continue;
        const int q0 = qb * TQ;
        const uint32_t lane_off = (uint32_t)(prow * (DIM * 2) + (((lane & 3) ^ (((prow >> 2) & 1) << 1)) << 4));
        int64_t it = 0;
        int iks = 0;
        int64_t iu = 0;
        const char *qblk = reinterpret_cast<const char *>(a.Q) + (int64_t)q0 * (DIM * 2);
        auto issue_piece = [&](int i) __attribute__((always_inline)) {
            char *buf = smem + (int)(iu % RINGB) * SUB;
            int64_t vt = r + it * a.ranges;
            if (vt > a.n_vt - 1) vt = a.n_vt - 1;
            const char *dtile = reinterpret_cast<const char *>(a.D) + vt * (int64_t)TD * (DIM * 2);
            const int kb = iks * 64;
            const int p = wv * 10 + i;
            const char *base = (p < TD / 16 ? dtile + (int64_t)p * 16 * (DIM * 2) : qblk + (int64_t)(p - TD / 16) * 16 * (DIM * 2)) + kb;
            glds16(base + lane_off, buf + p * 1024);
        };
        auto advance = [&]() __attribute__((always_inline)) {
            ++iu;
            if (++iks == KS2) {
                iks = 0;
                ++it;
            }
        };
        f32x4 acc_a[8][8], acc_v[8][4];
        auto zero_all = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
#pragma unroll
                for (int qt = 0; qt < 8; ++qt) zero_a(acc_a[dt][qt]);
#pragma unroll
                for (int qt = 0; qt < 4; ++qt) acc_v[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        auto mm = [&](auto first, int dt, int qt, bf16x8 fa, bf16x8 fb) __attribute__((always_inline)) {
            if constexpr (decltype(first)::value) {      // the first K step of a tile: C = 0 (no zeroing pass)
                if (qt < 8)
                    mfma_a0(acc_a[dt][qt < 8 ? qt : 0], fa, fb);
                else
                    mfma_v0(acc_v[dt][qt >= 8 ? qt - 8 : 0], fa, fb);
            } else {
                if (qt < 8)
                    mfma_a(acc_a[dt][qt < 8 ? qt : 0], fa, fb);
                else
                    mfma_v(acc_v[dt][qt >= 8 ? qt - 8 : 0], fa, fb);
            }
        };
        zero_all();
        const int64_t U = ntile * KS2;
        for (int j = 0; j < 3; ++j) {          // K steps 0, 1, 2 into the three ring slots
#pragma unroll
            for (int i = 0; i < 10; ++i) issue_piece(i);
            advance();
        }
        wait_vm<20>();                          // this wave's pieces of K step 0
        BARRIER();
        // (Operand reads as inline asm with hand-counted lgkmcnt waits -- the step's own af[4..7] requested first thing, the top wait leaving
        // exactly those in flight -- measured SLOWER: 13.9 ms against 11.7 with hipcc's own waits; so did a peeled first K step with C = 0
        // instead of the zeroing pass: 13.2 ms.)
        bf16x8 af[8], bq[12];
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) af[dt] = *reinterpret_cast<const bf16x8 *>(smem + a_base + dt * 1024);
#pragma unroll
        for (int qt = 0; qt < 12; ++qt) bq[qt] = *reinterpret_cast<const bf16x8 *>(smem + b_base + qt * 1024);
        int cks = 0;
        int64_t ct = 0;
        const auto first = std::false_type{};
        // (ONE flat loop over the K steps with the tile's epilogue behind a counter: the same body in a (tile, K step) loop nest runs 13.7 ms
        // against 11.7 -- hipcc's waits and placement differ)
        for (int64_t u = 0; u < U; ++u) {
            // ---- first half: corpus tiles 0-3 x all query tiles
#pragma unroll
            for (int qt = 0; qt < 12; ++qt)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) mm(first, dt, qt, af[dt], bq[qt]);
            __builtin_amdgcn_sched_barrier(0);
            // ---- rendezvous of K step u + 1: its pieces have landed (those of u + 2 may be in flight); every wave has read ALL of slot u
            wait_vm<10>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            BARRIER();
            const char *nxt = smem + (int)((u + 1) % RINGB) * SUB;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) af[dt] = *reinterpret_cast<const bf16x8 *>(nxt + a_base + dt * 1024);
            __builtin_amdgcn_sched_barrier(0);
            // ---- second half: corpus tiles 4-7, query tile by query tile; behind each: its next fragment, and a DMA piece of K step u + 3
#pragma unroll
            for (int qt = 0; qt < 12; ++qt) {
#pragma unroll
                for (int dt = 4; dt < 8; ++dt) mm(first, dt, qt, af[dt], bq[qt]);
                __builtin_amdgcn_sched_barrier(0);
                bq[qt] = *reinterpret_cast<const bf16x8 *>(nxt + b_base + qt * 1024);
                if (qt < 10) issue_piece(qt);       // into the slot of u (free since the barrier above)
                __builtin_amdgcn_sched_barrier(0);
            }
            advance();
#pragma unroll
            for (int dt = 4; dt < 8; ++dt) af[dt] = *reinterpret_cast<const bf16x8 *>(nxt + a_base + dt * 1024);
            __builtin_amdgcn_sched_barrier(0);
            if (++cks == KS2) {
                cks = 0;
                float m = -INFINITY;
#pragma unroll
                for (int qt = 0; qt < 8; ++qt) {
                    float mq = -INFINITY;
#pragma unroll
                    for (int dt = 0; dt < 8; ++dt) {
                        const f32x4 c = from_a(acc_a[dt][qt]);
                        mq = fmaxf(mq, fmaxf(fmaxf(c[0], c[1]), fmaxf(c[2], c[3])));
                    }
                    m = fmaxf(m, mq);
                }
#pragma unroll
                for (int qt = 0; qt < 4; ++qt) {
                    float mq = -INFINITY;
#pragma unroll
                    for (int dt = 0; dt < 8; ++dt) {
                        const f32x4 c = acc_v[dt][qt];
                        mq = fmaxf(mq, fmaxf(fmaxf(c[0], c[1]), fmaxf(c[2], c[3])));
                    }
                    m = fmaxf(m, mq);
                }
                if (m >= a.thr) a.out[(size_t)TD * TQ + blockIdx.x * 256 + tid] = m;
                if (a.check && item == 0 && blockIdx.x == 0 && ct == 0) {
#pragma unroll
                    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
                        for (int qt = 0; qt < 12; ++qt) {
                            const f32x4 c = qt < 8 ? from_a(acc_a[dt][qt < 8 ? qt : 0]) : acc_v[dt][qt >= 8 ? qt - 8 : 0];
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                a.out[(size_t)(wd * 128 + dt * 16 + 4 * lq + e) * TQ + wq * 192 + qt * 16 + l15] = c[e];
                        }
                }
                ++ct;
                zero_all();     // (a peeled first K step with C = 0 instead: 13.2 ms against 11.7 -- twice the loop body)
            }
        }
        wait_vm<0>();
        __syncthreads();
    }
}

__global__ void fill_kernel(uint16_t *p, int64_t n, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed;
        x ^= x >> 16;
        x *= 0x7feb352du;
        x ^= x >> 15;
        x *= 0x846ca68bu;
        x ^= x >> 16;
        // roughly N(0, 1/768): sum of four uniforms, centred
        const float f = (((x & 255) + ((x >> 8) & 255) + ((x >> 16) & 255) + (x >> 24)) - 510.0f) * (1.0f / 147.8f) * 0.0361f;
        p[i] = (uint16_t)(__float_as_uint(f) >> 16);
    }
}

int main(int argc, char **argv) {
    const int64_t n_rows = argc > 1 ? atoll(argv[1]) : 2681468;
    const int n_q = argc > 2 ? atoi(argv[2]) : 3452;
    uint16_t *D, *Q;
    float *out;
    const int64_t rows_pad = (n_rows + TD - 1) / TD * TD + 3 * TD;          // whole tiles (+ the K steps requested past an item's end)
    const int q_pad = (n_q + TQ - 1) / TQ * TQ;
    CK(hipMalloc(&D, (size_t)rows_pad * DIM * 2));
    CK(hipMalloc(&Q, (size_t)q_pad * DIM * 2));
    CK(hipMalloc(&out, ((size_t)TD * TQ + 256 * 256) * 4));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, D, rows_pad * DIM, 0x1234u);
    hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, 0, Q, (int64_t)q_pad * DIM, 0x4321u);
    CK(hipDeviceSynchronize());
    Args a;
    a.D = D, a.Q = Q, a.n_rows = n_rows, a.n_q = n_q;
    a.qblocks = (n_q + TQ - 1) / TQ;
    a.n_vt = (n_rows + TD - 1) / TD;
    a.ranges = 256;                       // 32 per XCD: 32 x 9 items over the XCD's 32 workgroups = 9 each at NQ
    a.thr = INFINITY;
    a.out = out;
    a.check = 1;
    const size_t lds = (size_t)RINGB * SUB;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&big_tile_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&big_tile_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&big_tile_pipe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    std::vector<uint16_t> hD((size_t)TD * DIM), hQ((size_t)TQ * DIM);
    std::vector<float> hO((size_t)TD * TQ);
    CK(hipMemcpy(hD.data(), D, hD.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hQ.data(), Q, hQ.size() * 2, hipMemcpyDeviceToHost));
    auto f = [](uint16_t b) {
        union {
            uint32_t u;
            float x;
        } w;
        w.u = (uint32_t)b << 16;
        return (double)w.x;
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 3; ++variant) {
        auto launch = [&]() {
            if (variant == 0)
                hipLaunchKernelGGL(big_tile_kernel<0>, dim3(256), dim3(256), lds, 0, a);
            else if (variant == 1)
                hipLaunchKernelGGL(big_tile_kernel<1>, dim3(256), dim3(256), lds, 0, a);
            else
                hipLaunchKernelGGL(big_tile_pipe_kernel, dim3(256), dim3(256), lds, 0, a);
        };
        a.check = 1;
        CK(hipMemset(out, 0xff, (size_t)TD * TQ * 4));
        launch();
        CK(hipDeviceSynchronize());
        // the first tile of item 0 (corpus tile 0, query block 0) against fp64 on the host
        CK(hipMemcpy(hO.data(), out, hO.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, big = 0;
        int bad = 0;
        for (int d = 0; d < TD; d += 7)
            for (int q = 0; q < TQ; q += 5) {
                double s = 0;
                for (int k = 0; k < DIM; ++k) s += f(hD[(size_t)d * DIM + k]) * f(hQ[(size_t)q * DIM + k]);
                const double df = fabs(s - (double)hO[(size_t)d * TQ + q]);
                if (!(df < 1e-3)) ++bad;
                else worst = fmax(worst, df);
                big = fmax(big, fabs(s));
            }
        printf("variant %d check tile: max |mfma - fp64| = %.3g, %d bad of the sampled outputs (largest |score| %.3g)\n", variant, worst, bad, big);
        a.check = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 5; ++i) launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double flops = 2.0 * n_rows * (double)n_q * DIM;
            printf("variant %d (%s): %.3f ms per pass = %.0f TFLOP/s of algorithmic work (%.3f of 2.5 PF); padded queries %d\n", variant,
                   variant == 2 ? "software-pipelined: operands refilled in place behind the MFMAs" : variant ? "DMA pieces spread between the MFMAs" : "DMA issue in front of the K step", ms / 5, flops / (ms / 5 * 1e-3) / 1e12,
                   flops / (ms / 5 * 1e-3) / 2.5e15, a.qblocks * TQ);
        }
    }
    return 0;
}
