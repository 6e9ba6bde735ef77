// ccr_inbatch.hip -- in-batch-negative contrastive loss ("multiple_nrl", bbpr.py:205-212), fwd + bwd.
//
//   logits[i][j] = inv_T * <Q_i, K_j>,  K = [P ; N]  (2B keys),  loss = mean_i( lse_i - logits[i][i] )
//
// The logits tile is the retrieval kernel's MFMA tile (v_mfma_f32_32x32x16_bf16, keys on the
// accumulator rows, queries on the lanes), so a lane owns one query column and the softmax over keys
// is an in-register online reduction.  B x 2B is small (the encoder dominates the step): operands are
// read straight from L2 in fragment layout, no LDS staging.
//   fwd : ONE launch.  Per (query tile, key split) partial (max, sum, diagonal) and the scaled logits S[2B][B] (kept for the
//         backward); the last split of a query tile to arrive (ticket) combines the tile's partials in split order -> lse, the
//         tile's loss share; the last tile to arrive adds the shares in tile order -> loss.  Deterministic.
//   bwd : TWO launches.  dQ = G^T K and dK = G Q with G[j][i] = (exp(S[j][i] - lse_i) - [j == i]) * inv_T * grad_out / B, split into
//         THREE bf16 parts (g = hi + mid + lo exactly to 2^-24 relative) so that the products with the bf16 embeddings run on
//         v_mfma_f32_32x32x16_bf16 (3 MFMAs of 32 cycles per 16 K instead of 8 fp32 MFMAs of 64): fp32-MFMA accuracy at a fifth of
//         its time.  r5: (1) inbatch_prep_kernel evaluates and splits every G element ONCE (r4 did it while staging, once per
//         64-column output tile = 12 times at dim 768, which bounded the kernels: 61 + 41 us) and leaves the parts in BOTH
//         orientations (keys x queries and queries x keys, 6 bytes each) plus the transposes of the embeddings, all with the
//         contraction index contiguous; (2) inbatch_gemm3_kernel then reads every MFMA fragment straight from L2 -- no LDS staging,
//         no barrier inside the K loop -- one workgroup per 32 x 64 output tile, its four waves taking a quarter of K each and
//         adding their partial tiles through LDS in wave order.  Deterministic, no atomics.
#include <stdlib.h>

#include <algorithm>

#include "ccr_common.h"

namespace ccr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ const uint16_t *key_row(const uint16_t *P, const uint16_t *N, int B, int j, int dim) {
    return (j < B) ? P + (int64_t)j * dim : N + (int64_t)(j - B) * dim;
}

// Hand-off of a wave's partials to the last arriver WITHOUT a release fence: a fence (buffer_wbl2) writes back every dirty line of
// the XCD's L2, and this kernel keeps 8 MB of freshly written logits there -- measured 65 us for the fused forward with fences against
// 52 us without.  MI355X_MICROARCH 'Valid forms', first row of its table: every handed-off byte is
// stored sc1 (write-through: __hip_atomic_store relaxed / agent), the storing wave drains its stores (s_waitcnt vmcnt(0)), ONE lane
// then adds to an agent-scope counter, and the wave whose add came last reads the bytes with sc1 loads (__hip_atomic_load relaxed /
// agent) after its add has returned.  A workgroup here is one wave, so the signalling lane signals for its own wave's stores only.
__device__ __forceinline__ void store_handoff(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float load_handoff(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned int drain_and_take_ticket(unsigned int *ticket) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// grid = (query tiles of 32, key splits), block = one wave.
//   pm / pl / pd [splits][B]: partial max, sum, diagonal; S [2B][ldS]: the scaled logits (row = key, column = query)
//   tile_ticket [query tiles] + loss_ticket [1] + tile_part [query tiles] doubles: zeroed by the caller's memset
// FRAG (the fp32 entry point, B % 32 == 0, dim % 128 == 0): Q points at the FRAGMENT-MAJOR copy of [Q ; P ; N] the pack kernel left in
// the workspace (3B rows, contraction index = dim; layout: frag_major below) -- every operand load of a wave is one contiguous KiB
// instead of 16 bytes from each of 32 rows (32 cache lines per instruction: what bounded this kernel at 52 us).
template <bool FRAG>
__global__ __launch_bounds__(64) void inbatch_fwd_kernel(const uint16_t *__restrict__ Q, const uint16_t *__restrict__ P,
                                                        const uint16_t *__restrict__ N, int B, int dim, float inv_t, int splits,
                                                        float *__restrict__ pm, float *__restrict__ pl, float *__restrict__ pd,
                                                        float *__restrict__ S, int ldS, float *__restrict__ lse,
                                                        float *__restrict__ loss, unsigned int *__restrict__ tile_ticket,
                                                        unsigned int *__restrict__ loss_ticket, double *__restrict__ tile_part,
                                                        uint32_t *__restrict__ stamp) {
    const int lane = threadIdx.x;
    const int l31 = lane & 31, h = lane >> 5;
    const int i0 = blockIdx.x * 32;
    const int s = blockIdx.y;
    const int ktiles = (2 * B + 31) / 32;
    const int t_lo = (int)((int64_t)s * ktiles / splits), t_hi = (int)((int64_t)(s + 1) * ktiles / splits);
    const int i = i0 + l31;
    const int iq = i < B ? i : B - 1;
    // FRAG: row block rb of the 3B-row matrix starts at rb (dim / 64) 2 048 elements; inside a 64-wide chunk k-step ks is at ks 512 + lane 8,
    // so the 8 k-steps of a 128-wide chunk c are at c 4 096 + u 512 (u = 0 .. 7)
    const uint16_t *qrow = FRAG ? Q + (int64_t)blockIdx.x * (dim / 64) * 2048 + lane * 8 : Q + (int64_t)iq * dim + 8 * h;
    constexpr int KSTEP = FRAG ? 512 : 16;      // elements between consecutive k-steps of a lane
    constexpr int KCHUNK = FRAG ? 4096 : 128;   // ... and between 128-wide chunks
    float m = -INFINITY, l = 0.f, dg = -INFINITY;
    for (int t = t_lo; t < t_hi; ++t) {
        int jr = t * 32 + l31;
        if (jr > 2 * B - 1) jr = 2 * B - 1;
        const uint16_t *krow = FRAG ? Q + (int64_t)(B / 32 + t) * (dim / 64) * 2048 + lane * 8 : key_row(P, N, B, jr, dim) + 8 * h;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        // operands come straight from L2 in fragment layout, in chunks of 8 K steps (16 loads); the NEXT chunk's loads are issued before
        // the current chunk's 8 MFMAs (two register sets; the last prefetch repeats the last chunk -- unconditional loads keep the
        // wait counts exact), so a tile costs ~dim / 256 + 1 L2 round trips instead of dim / 128
        const int nfull = dim / 128;
        bf16x8 a[2][8], b[2][8];
        auto load = [&](int set, int c) {
            const int kk = (c < nfull ? c : nfull - 1) * KCHUNK;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[set][u] = *reinterpret_cast<const bf16x8 *>(krow + kk + KSTEP * u);
                b[set][u] = *reinterpret_cast<const bf16x8 *>(qrow + kk + KSTEP * u);
            }
        };
        if (nfull > 0) {
            load(0, 0);
            for (int c = 0; c < nfull; c += 2) {
                load(1, c + 1);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][u], b[0][u], acc, 0, 0, 0);
                load(0, c + 2);
                if (c + 1 < nfull) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][u], b[1][u], acc, 0, 0, 0);
                }
            }
        }
        int k0 = FRAG ? dim : nfull * 128;   // (FRAG: dim % 128 == 0, no tail)
        for (; k0 < dim; k0 += 16) {
            const bf16x8 a = *reinterpret_cast<const bf16x8 *>(krow + k0);
            const bf16x8 b = *reinterpret_cast<const bf16x8 *>(qrow + k0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        float tm = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int j = t * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            const float v = (j < 2 * B) ? acc[e] * inv_t : -INFINITY;
            acc[e] = v;
            tm = fmaxf(tm, v);
            if (j == i) dg = v;
            if (j < 2 * B && i < B) S[(int64_t)j * ldS + i] = v;   // 32 consecutive queries of one key row per half wave: 128-byte segments
        }
        const float mn = fmaxf(m, tm);
        if (mn > -INFINITY) {
            float sum = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) sum += __expf(acc[e] - mn);
            l = l * __expf(m - mn) + sum;
            m = mn;
        }
    }
    {   // the two lane halves hold different key rows of the same query column
        const float m2 = __shfl_xor(m, 32, 64), l2 = __shfl_xor(l, 32, 64), d2 = __shfl_xor(dg, 32, 64);
        const float M = fmaxf(m, m2);
        float L = 0.f;
        if (M > -INFINITY) L = (m > -INFINITY ? l * __expf(m - M) : 0.f) + (m2 > -INFINITY ? l2 * __expf(m2 - M) : 0.f);
        if (h == 0 && i < B) {
            store_handoff(pm + (int64_t)s * B + i, M);
            store_handoff(pl + (int64_t)s * B + i, L);
            store_handoff(pd + (int64_t)s * B + i, fmaxf(dg, d2));
        }
    }
    // ---- the last split of this query tile combines the tile's partials in split order (the wave is the whole workgroup: the
    // ticket is taken by lane 0 behind the wave's own stores, the result broadcast)
    unsigned int t = 0;
    if (lane == 0) t = drain_and_take_ticket(&tile_ticket[blockIdx.x]);   // (s_waitcnt is per wave: every lane's stores are drained)
    t = __shfl(t, 0, 64);
    if (t != (unsigned int)splits - 1) return;
    double part = 0.0;
    if (h == 0 && i < B) {
        float M = -INFINITY, dgc = -INFINITY, L = 0.f;
        for (int s0 = 0; s0 < splits; s0 += 16) {   // 3 x 16 independent loads per round
            float vm[16], vl[16], vd[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const bool in = s0 + u < splits;
                const int64_t at = (int64_t)(in ? s0 + u : 0) * B + i;
                vm[u] = in ? load_handoff(pm + at) : -INFINITY;
                vl[u] = in ? load_handoff(pl + at) : 0.f;
                vd[u] = in ? load_handoff(pd + at) : -INFINITY;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {   // online combine in split order
                const float mn = fmaxf(M, vm[u]);
                if (mn > -INFINITY) L = (M > -INFINITY ? L * __expf(M - mn) : 0.f) + (vm[u] > -INFINITY ? vl[u] * __expf(vm[u] - mn) : 0.f);
                M = mn;
                dgc = fmaxf(dgc, vd[u]);
            }
        }
        const float v = M + __logf(L);
        lse[i] = v;
        part = (double)(v - dgc);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
    unsigned int t2 = 0;
    if (lane == 0) {
        __hip_atomic_store(tile_part + blockIdx.x, part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t2 = drain_and_take_ticket(loss_ticket);
    }
    t2 = __shfl(t2, 0, 64);
    if (t2 != gridDim.x - 1) return;
    if (lane == 0) {   // the last tile adds the shares in tile order
        double tot = 0.0;
        for (unsigned int b = 0; b < gridDim.x; ++b) tot += __hip_atomic_load(tile_part + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        loss[0] = (float)(tot / (double)B);
        // every logit of this forward is in the workspace now: the stamp the backward checks (InbatchStamp below)
        stamp[0] = 0x43434942u, stamp[1] = (uint32_t)B, stamp[2] = (uint32_t)dim, stamp[3] = __float_as_uint(inv_t);
    }
}

// ---- backward -------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split3(float g, uint16_t &hi, uint16_t &mid, uint16_t &lo) {
    const __bf16 a = (__bf16)g;
    const float r1 = g - (float)a;
    const __bf16 b = (__bf16)r1;
    const __bf16 c = (__bf16)(r1 - (float)b);
    hi = __builtin_bit_cast(uint16_t, a);
    mid = __builtin_bit_cast(uint16_t, b);
    lo = __builtin_bit_cast(uint16_t, c);
}

// Stamp the forward leaves in the workspace and the backward checks ON THE DEVICE (a mismatch -- a workspace another forward has
// used in between, or none -- poisons the gradients with NaN instead of returning plausible garbage; no host round trip).
struct InbatchStamp {
    uint32_t magic, B, dim, inv_t_bits;
};
constexpr uint32_t INBATCH_MAGIC = 0x43434942u;   // 'CCIB'

// Operands of the backward's GEMMs, written here in FRAGMENT-MAJOR tiles: a matrix [rows][K] (K = the contraction index; rows and K
// padded to multiples of 64 with zeros) is stored as [rows / 32][K / 64][4 k-steps][2 halves][32 rows][8 elements], i.e. the 16 bytes
// lane (half h, row l) of a v_mfma_f32_32x32x16_bf16 operand wants for k-step ks of a 64-wide K chunk sit at byte
// ((ks 2 + h) 32 + l) 16 of the chunk's 4-KiB block: one 16-byte load per lane fetches the whole fragment as ONE contiguous KiB.
// (Row-major rows with 16 bytes per lane from 32 different rows touch 32 cache lines per load instruction: the first form of the
// GEMM below ran at 5 TB/s of fragment loads, 91 us.)
//   Gk [3] x (keys x queries)   parts of G[j][i]      (dK: m = key j,   k = query i),  ldk = round_up(B, 64)
//   Gq [3] x (queries x keys)   parts of G[j][i]^T    (dQ: m = query i, k = key j),    ldq = round_up(2B, 64)
//   QT (dim x queries) = Q^T    (dK's second operand),   KT (dim x keys) = [P ; N]^T   (dQ's)
// grid.x = G tiles (64 keys x 64 queries) + Q tiles + K tiles (64 rows x 64 columns each), block = 256.
constexpr int PREP_T = 64, PREP_P = PREP_T + 8;   // LDS pitch 72 bf16 = 144 B: 16-byte aligned rows, rows 4 banks apart
__device__ __forceinline__ int64_t frag_major(int r, int k, int nchunks) {   // element offset of (row r, contraction index k), k % 8 == 0
    return ((((int64_t)(r >> 5) * nchunks + (k >> 6)) * 8 + ((k >> 3) & 7)) * 32 + (r & 31)) * 8;
}
__global__ __launch_bounds__(256) void inbatch_prep_kernel(const float *__restrict__ S, int ldS, const float *__restrict__ lse,
                                                          const uint16_t *__restrict__ Q, const uint16_t *__restrict__ P,
                                                          const uint16_t *__restrict__ N, int B, int dim, float gscale,
                                                          const float *__restrict__ gscale_dev, const InbatchStamp *__restrict__ stamp,
                                                          uint32_t inv_t_bits, uint16_t *__restrict__ Gk, uint16_t *__restrict__ Gq,
                                                          uint16_t *__restrict__ QT, uint16_t *__restrict__ KT, int ldk, int ldq) {
    __shared__ __attribute__((aligned(16))) uint16_t s_t[3][PREP_T][PREP_P];
    const int tid = threadIdx.x;
    const int gi = (B + PREP_T - 1) / PREP_T, gj = (2 * B + PREP_T - 1) / PREP_T, gd = (dim + PREP_T - 1) / PREP_T;
    int b = blockIdx.x;
    if (b < gi * gj) {
        // ---- G tile: rows j0 .. j0 + 63 (keys), columns i0 .. i0 + 63 (queries); thread -> row tid / 4, 16 consecutive columns
        if (gscale_dev) gscale *= gscale_dev[0];   // upstream gradient read on the device: no host round trip
        const InbatchStamp st = *stamp;
        if (st.magic != INBATCH_MAGIC || st.B != (uint32_t)B || st.dim != (uint32_t)dim || st.inv_t_bits != inv_t_bits) gscale = __builtin_nanf("");
        const int j0 = (b / gi) * PREP_T, i0 = (b % gi) * PREP_T;
        const int r = tid >> 2, c0 = (tid & 3) * 16;
        const int j = j0 + r;
        uint16_t part[3][16];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int ic = i0 + c0 + 4 * q4;
            float sv[4] = {0.f, 0.f, 0.f, 0.f}, lv[4] = {0.f, 0.f, 0.f, 0.f};
            if (j < 2 * B && ic < ldS) {   // ldS is a multiple of 4: an aligned vector inside the row pitch
                const float4 v = *reinterpret_cast<const float4 *>(S + (int64_t)j * ldS + ic);
                sv[0] = v.x, sv[1] = v.y, sv[2] = v.z, sv[3] = v.w;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) lv[u] = lse[ic + u < B ? ic + u : B - 1];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = ic + u;
                float g = __expf(sv[u] - lv[u]);
                if (j == i) g -= 1.f;
                g = (j < 2 * B && i < B) ? g * gscale : 0.f;   // (a select, not a product: positions outside the matrix may hold anything)
                split3(g, part[0][4 * q4 + u], part[1][4 * q4 + u], part[2][4 * q4 + u]);
            }
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            uint32_t w[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = (uint32_t)part[p][2 * e] | ((uint32_t)part[p][2 * e + 1] << 16);
            {   // rows up to the padded height are written (zeros beyond the matrix)
                uint16_t *base = Gk + (int64_t)p * gj * PREP_T * ldk;
                *reinterpret_cast<uint4 *>(base + frag_major(j, i0 + c0, ldk / 64)) = make_uint4(w[0], w[1], w[2], w[3]);
                *reinterpret_cast<uint4 *>(base + frag_major(j, i0 + c0 + 8, ldk / 64)) = make_uint4(w[4], w[5], w[6], w[7]);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) s_t[p][c0 + e][r] = part[p][e];   // transposed: [query][key]
        }
        __syncthreads();
        const int i = i0 + r;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const uint4 *src = reinterpret_cast<const uint4 *>(&s_t[p][r][c0]);
            uint16_t *base = Gq + (int64_t)p * gi * PREP_T * ldq;
            *reinterpret_cast<uint4 *>(base + frag_major(i, j0 + c0, ldq / 64)) = src[0];
            *reinterpret_cast<uint4 *>(base + frag_major(i, j0 + c0 + 8, ldq / 64)) = src[1];
        }
        return;
    }
    // ---- transposes of the embeddings: X [rows][dim] -> XT [dim][ld], rows beyond the matrix zero
    b -= gi * gj;
    const bool isq = b < gi * gd;
    if (!isq) b -= gi * gd;
    const int rows = isq ? B : 2 * B, ld = isq ? ldk : ldq;
    uint16_t *XT = isq ? QT : KT;
    const int r0 = (b / gd) * PREP_T, d0 = (b % gd) * PREP_T;
    {
        const int r = tid >> 2, c0 = (tid & 3) * 16;   // row r0 + r, columns d0 + c0 .. + 16
        const int row = r0 + r;
        uint16_t v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = 0;
        if (row < rows) {
            const uint16_t *src = isq ? Q + (int64_t)row * dim : key_row(P, N, B, row, dim);
#pragma unroll
            for (int h8 = 0; h8 < 2; ++h8) {
                const int dc = d0 + c0 + 8 * h8;
                if (dc < dim) {   // dim % 16 == 0 and dc % 8 == 0: whole 16-byte vectors
                    const uint4 q = *reinterpret_cast<const uint4 *>(src + dc);
                    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[8 * h8 + 2 * e] = (uint16_t)(w[e] & 0xffffu), v[8 * h8 + 2 * e + 1] = (uint16_t)(w[e] >> 16);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) s_t[0][c0 + e][r] = v[e];
    }
    __syncthreads();
    {
        const int r = tid >> 2, c0 = (tid & 3) * 16;   // column d0 + r of X = row of XT, 16 consecutive source rows
        const int d = d0 + r;   // (rows d >= dim of the padded height: zeros)
        const uint4 *src = reinterpret_cast<const uint4 *>(&s_t[0][r][c0]);
        *reinterpret_cast<uint4 *>(XT + frag_major(d, r0 + c0, ld / 64)) = src[0];
        *reinterpret_cast<uint4 *>(XT + frag_major(d, r0 + c0 + 8, ld / 64)) = src[1];
    }
}

// C[M][dim] fp32 = sum over the three parts and over k of A_part[m][k] * BT[n][k], every operand fragment read straight from L2.
//   blocks [0, nq): dQ  (A = Gq parts, M = B,  K = ldq, BT = KT, C = dQ);   blocks [nq, ...): dK (A = Gk parts, M = 2B, K = ldk,
//   BT = QT, rows < B -> dP, the rest -> dN).  Workgroup = 4 waves on one 32 (m) x 64 (n) output tile; wave w contracts the w-th quarter
//   of K (in 64-element chunks: 12 + 8 sixteen-byte loads per lane, the next chunk's loads issued before the 24 MFMAs of the current
//   one) and the four partial tiles are added through LDS in wave order.
constexpr int G3_TM = 32, G3_TN = 64, G3_KC = 64;
__global__ __launch_bounds__(256) void inbatch_gemm3_kernel(const uint16_t *__restrict__ Gq, const uint16_t *__restrict__ Gk,
                                                           const uint16_t *__restrict__ KT, const uint16_t *__restrict__ QT, int B, int dim,
                                                           int ldq, int ldk, int nq_blocks, float *__restrict__ dQ, float *__restrict__ dP,
                                                           float *__restrict__ dN) {
    __shared__ float s_part[4][32][64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    (void)0;
    const int ntn = (dim + G3_TN - 1) / G3_TN;
    int b = blockIdx.x;
    const bool for_q = b < nq_blocks;
    if (!for_q) b -= nq_blocks;
    const int M = for_q ? B : 2 * B, K = for_q ? ldq : ldk;
    const uint16_t *A = for_q ? Gq : Gk, *BT = for_q ? KT : QT;
    const int m0 = (b / ntn) * G3_TM, n0 = (b % ntn) * G3_TN;
    const int chunks = K / G3_KC, per = (chunks + 3) / 4;
    const int c_lo = wv * per, c_hi = c_lo + per < chunks ? c_lo + per : chunks;
    // fragment-major blocks: [row block of 32][chunk of 64 k] = 2 048 elements, the lane's 16 bytes of k-step ks at ks 512 + lane 8
    const int64_t part_stride = (int64_t)((M + 63) / 64 * 64) * K;
    const uint16_t *arow[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) arow[p] = A + p * part_stride + (int64_t)(m0 >> 5) * chunks * 2048 + lane * 8;
    const uint16_t *brow[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) brow[t] = BT + (int64_t)((n0 >> 5) + t) * chunks * 2048 + lane * 8;   // (the padded height of BT covers n0 + 63)
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    bf16x8 a[2][3][4], bb[2][2][4];
    auto fetch = [&](int set, int c) {
        const int k0 = (c < chunks ? c : chunks - 1) * 2048;   // clamped: loads beyond the wave's range are never used
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int p = 0; p < 3; ++p) a[set][p][ks] = *reinterpret_cast<const bf16x8 *>(arow[p] + k0 + 512 * ks);
#pragma unroll
            for (int t = 0; t < 2; ++t) bb[set][t][ks] = *reinterpret_cast<const bf16x8 *>(brow[t] + k0 + 512 * ks);
        }
    };
    auto mma = [&](int set) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[set][p][ks], bb[set][t][ks], acc[t], 0, 0, 0);
    };
    fetch(0, c_lo);
    for (int c = c_lo; c < c_hi; c += 2) {   // wave-uniform
        fetch(1, c + 1);
        mma(0);
        fetch(0, c + 2);
        if (c + 1 < c_hi) mma(1);
    }
    // C layout of v_mfma_f32_32x32x16: column = lane & 31, register e -> row (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) s_part[wv][t * 16 + e][lane] = acc[t][e];
    __syncthreads();
    for (int idx = tid; idx < 32 * 64; idx += 256) {
        const int r = idx >> 6, ln = idx & 63;
        const float v = ((s_part[0][r][ln] + s_part[1][r][ln]) + s_part[2][r][ln]) + s_part[3][r][ln];
        const int t = r >> 4, e = r & 15;
        const int mrow = m0 + (e & 3) + 8 * (e >> 2) + 4 * (ln >> 5), ncol = n0 + 32 * t + (ln & 31);
        if (mrow < M && ncol < dim) {
            float *crow = for_q ? dQ + (int64_t)mrow * dim : (mrow < B ? dP + (int64_t)mrow * dim : dN + (int64_t)(mrow - B) * dim);
            crow[ncol] = v;
        }
    }
}

// fp32 -> bf16 (RNE: torch's .to(bfloat16) bits) of the step's three embedding blocks in ONE launch.  grid = (chunks, 3), block = 256.
// frag (may be null; B % 32 == 0 and dim % 64 == 0): the same values again as ONE fragment-major matrix [Q ; P ; N] of 3B rows (the
// forward's operand layout: a thread's 8 consecutive dim elements are exactly one lane's 16 bytes of a k-step).
__global__ __launch_bounds__(256) void inbatch_pack3_kernel(const float *__restrict__ q, const float *__restrict__ p, const float *__restrict__ n,
                                                           int64_t count, uint16_t *__restrict__ out, uint16_t *__restrict__ frag, int B, int dim,
                                                           uint32_t *__restrict__ zero, int zero_words) {
    // zero (may be null): the forward's tickets and stamp, cleared here instead of by a memset node (one launch less per step)
    if (zero && blockIdx.x == 0 && blockIdx.y == 0)
        for (int i = threadIdx.x; i < zero_words; i += 256) zero[i] = 0u;
    const float *src = blockIdx.y == 0 ? q : (blockIdx.y == 1 ? p : n);
    uint16_t *dst = out + (int64_t)blockIdx.y * count;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; i < count; i += (int64_t)gridDim.x * 256 * 8) {   // count % 8 == 0
        const float4 v0 = *reinterpret_cast<const float4 *>(src + i), v1 = *reinterpret_cast<const float4 *>(src + i + 4);
        const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            w[e] = (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)f[2 * e]) | ((uint32_t)__builtin_bit_cast(uint16_t, (__bf16)f[2 * e + 1]) << 16);
        *reinterpret_cast<uint4 *>(dst + i) = make_uint4(w[0], w[1], w[2], w[3]);
        if (frag) {
            const int row = (int)(i / dim) + (int)blockIdx.y * B, d = (int)(i % dim);
            *reinterpret_cast<uint4 *>(frag + frag_major(row, d, dim / 64)) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

static int pick_splits(int B) {
    // one wave per (32-query tile, key split): aim at ~8 waves per CU, at most one split per 32-key tile and 64 splits
    const int qtiles = (B + 31) / 32, ktiles = (2 * B + 31) / 32;
    int s = 2048 / (qtiles > 0 ? qtiles : 1);
    if (s < 1) s = 1;
    if (s > ktiles) s = ktiles;
    if (s > 64) s = 64;
    return s;
}

}  // namespace ccr

using namespace ccr;

namespace {
// workspace: [partials 3 x 64 x B floats][S: 2B rows of ldS floats][tickets: (query tiles + 1) u32, padded][tile shares: doubles]
//            [stamp][backward: Gk | Gq (three bf16 parts each) | QT | KT][forward, fp32 entry point: fragment-major Q ; P ; N]
struct InbatchWs {
    float *pm, *pl, *pd, *S;
    int ldS, ldk, ldq;
    unsigned int *tile_ticket, *loss_ticket;
    double *tile_part;
    uint32_t *stamp;
    uint16_t *Gk, *Gq, *QT, *KT, *Xf;
    size_t zero_bytes;   // tickets + stamp: zeroed before every forward
    size_t total;
};
size_t up256(size_t n) { return (n + 255) / 256 * 256; }
InbatchWs inbatch_ws(void *workspace, int B, int dim) {
    InbatchWs w;
    const size_t partial = up256((size_t)3 * 64 * B * sizeof(float));
    w.ldS = (B + 3) / 4 * 4;
    w.ldk = (B + 63) / 64 * 64;
    w.ldq = (2 * B + 63) / 64 * 64;
    const size_t sbytes = up256((size_t)2 * B * w.ldS * sizeof(float));
    const int qtiles = (B + 31) / 32;
    const size_t tickets = up256((size_t)(qtiles + 1) * 4);
    w.zero_bytes = tickets + 256;
    const size_t parts = up256((size_t)qtiles * 8);
    char *base = (char *)workspace;
    w.pm = (float *)base;
    w.pl = w.pm + (size_t)64 * B;
    w.pd = w.pl + (size_t)64 * B;
    w.S = (float *)(base + partial);
    w.tile_ticket = (unsigned int *)(base + partial + sbytes);
    w.loss_ticket = w.tile_ticket + qtiles;
    w.stamp = (uint32_t *)(base + partial + sbytes + tickets);
    w.tile_part = (double *)(base + partial + sbytes + w.zero_bytes);
    size_t off = partial + sbytes + w.zero_bytes + parts;
    const size_t dimp = (size_t)(dim + 63) / 64 * 64;   // every operand height and width is padded to a multiple of 64
    w.Gk = (uint16_t *)(base + off), off += up256((size_t)3 * w.ldq * w.ldk * 2);
    w.Gq = (uint16_t *)(base + off), off += up256((size_t)3 * w.ldk * w.ldq * 2);
    w.QT = (uint16_t *)(base + off), off += up256(dimp * w.ldk * 2);
    w.KT = (uint16_t *)(base + off), off += up256(dimp * w.ldq * 2);
    w.Xf = (uint16_t *)(base + off), off += up256((size_t)3 * B * dimp * 2);   // the forward's fragment-major [Q ; P ; N] (fp32 entry point)
    w.total = off + 256;   // + 256: the caller's pointer need only be 16-byte aligned
    return w;
}
char *align256(void *p) { return (char *)p + (256 - (uintptr_t)p % 256) % 256; }
}  // namespace

extern "C" int ccr_inbatch_pack3_bf16(const float *q, const float *p, const float *n, int B, int dim, uint16_t *out, void *stream);

extern "C" size_t ccr_inbatch_ce_workspace_bytes(int B, int dim) {
    if (B <= 0 || dim <= 0) return 0;
    return inbatch_ws(nullptr, B, dim).total;
}

static int inbatch_fwd_impl(const uint16_t *Qe, const uint16_t *Pe, const uint16_t *Ne, bool frag, int B, int dim, float inv_temperature,
                            float *loss, float *lse, void *workspace, size_t ws_bytes, void *stream, bool zeroed = false) {
    CCR_REQUIRE(Qe && Pe && Ne && loss && lse, "ccr_inbatch_ce_fwd: null pointer");
    CCR_REQUIRE(B >= 1 && dim >= 16 && dim % 16 == 0, "ccr_inbatch_ce_fwd: B=%d dim=%d (dim %% 16 == 0)", B, dim);
    CCR_REQUIRE(((uintptr_t)Qe | (uintptr_t)Pe | (uintptr_t)Ne) % 16 == 0, "ccr_inbatch_ce_fwd: embedding pointers must be 16-byte aligned");
    if (!workspace || ws_bytes < ccr_inbatch_ce_workspace_bytes(B, dim)) {
        set_error("ccr_inbatch_ce_fwd: workspace %zu bytes required, got %zu", ccr_inbatch_ce_workspace_bytes(B, dim), ws_bytes);
        return CCR_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const InbatchWs w = inbatch_ws(align256(workspace), B, dim);
    const int splits = pick_splits(B);
    if (!zeroed) CCR_HIP_CHECK(hipMemsetAsync(w.tile_ticket, 0, w.zero_bytes, s));   // tickets, and the stamp: no longer this workspace's logits
    dim3 grid((B + 31) / 32, splits);
    if (frag)
        hipLaunchKernelGGL(inbatch_fwd_kernel<true>, grid, dim3(64), 0, s, w.Xf, w.Xf, w.Xf, B, dim, inv_temperature, splits, w.pm, w.pl, w.pd, w.S, w.ldS,
                           lse, loss, w.tile_ticket, w.loss_ticket, w.tile_part, w.stamp);
    else
        hipLaunchKernelGGL(inbatch_fwd_kernel<false>, grid, dim3(64), 0, s, Qe, Pe, Ne, B, dim, inv_temperature, splits, w.pm, w.pl, w.pd, w.S, w.ldS,
                           lse, loss, w.tile_ticket, w.loss_ticket, w.tile_part, w.stamp);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

extern "C" int ccr_inbatch_ce_fwd(const uint16_t *Qe, const uint16_t *Pe, const uint16_t *Ne, int B, int dim,
                                  float inv_temperature, float *loss, float *lse, void *workspace, size_t ws_bytes, void *stream) {
    return inbatch_fwd_impl(Qe, Pe, Ne, false, B, dim, inv_temperature, loss, lse, workspace, ws_bytes, stream);
}

static int launch_pack3(const float *q, const float *p, const float *n, int B, int dim, uint16_t *out, uint16_t *frag, void *stream,
                        uint32_t *zero = nullptr, int zero_words = 0) {
    CCR_REQUIRE(q && p && n && out, "ccr_inbatch_pack3_bf16: null pointer");
    CCR_REQUIRE(B >= 1 && dim >= 8 && dim % 8 == 0, "ccr_inbatch_pack3_bf16: B=%d dim=%d (dim %% 8 == 0)", B, dim);
    CCR_REQUIRE(((uintptr_t)q | (uintptr_t)p | (uintptr_t)n | (uintptr_t)out) % 16 == 0, "ccr_inbatch_pack3_bf16: pointers must be 16-byte aligned");
    const int64_t count = (int64_t)B * dim;
    hipLaunchKernelGGL(inbatch_pack3_kernel, dim3((unsigned)std::min<int64_t>((count / 8 + 255) / 256, 4096), 3), dim3(256), 0, (hipStream_t)stream, q, p, n,
                       count, out, frag, B, dim, zero, zero_words);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

extern "C" int ccr_inbatch_pack3_bf16(const float *q, const float *p, const float *n, int B, int dim, uint16_t *out, void *stream) {
    return launch_pack3(q, p, n, B, dim, out, nullptr, stream);
}

// pack + forward in one call (the autograd function's host path: one crossing of the C boundary per forward).  Where the shape allows
// (B % 32 == 0, dim % 128 == 0) the pack also leaves the fragment-major copy the forward then reads.
extern "C" int ccr_inbatch_ce_fwd_f32(const float *q, const float *p, const float *n, int B, int dim, float inv_temperature,
                                      uint16_t *packed, float *loss, float *lse, void *workspace, size_t ws_bytes, void *stream) {
    CCR_REQUIRE(packed && workspace, "ccr_inbatch_ce_fwd_f32: null pointer");
    CCR_REQUIRE(B >= 1 && dim >= 16 && dim % 16 == 0, "ccr_inbatch_ce_fwd_f32: B=%d dim=%d (dim %% 16 == 0)", B, dim);
    if (ws_bytes < ccr_inbatch_ce_workspace_bytes(B, dim)) {
        set_error("ccr_inbatch_ce_fwd_f32: workspace %zu bytes required, got %zu", ccr_inbatch_ce_workspace_bytes(B, dim), ws_bytes);
        return CCR_ERR_WORKSPACE;
    }
    const bool frag = B % 32 == 0 && dim % 128 == 0 && getenv("CCR_INBATCH_ROWMAJOR") == nullptr;
    const InbatchWs w = inbatch_ws(align256(workspace), B, dim);
    const int rc = launch_pack3(q, p, n, B, dim, packed, frag ? w.Xf : nullptr, stream, w.tile_ticket, (int)(w.zero_bytes / 4));
    if (rc != CCR_OK) return rc;
    const size_t blk = (size_t)B * dim;
    return inbatch_fwd_impl(packed, packed + blk, packed + 2 * blk, frag, B, dim, inv_temperature, loss, lse, workspace, ws_bytes, stream, true);
}

static int inbatch_bwd_impl(const uint16_t *Qe, const uint16_t *Pe, const uint16_t *Ne, const float *lse, int B, int dim,
                            float inv_temperature, float grad_out, const float *grad_out_dev, float *dQ, float *dP, float *dN,
                            void *workspace, size_t ws_bytes, void *stream) {
    CCR_REQUIRE(Qe && Pe && Ne && lse && dQ && dP && dN, "ccr_inbatch_ce_bwd: null pointer");
    CCR_REQUIRE(B >= 1 && dim >= 16 && dim % 16 == 0, "ccr_inbatch_ce_bwd: B=%d dim=%d (dim %% 16 == 0)", B, dim);
    CCR_REQUIRE(((uintptr_t)Qe | (uintptr_t)Pe | (uintptr_t)Ne) % 16 == 0, "ccr_inbatch_ce_bwd: embedding pointers must be 16-byte aligned");
    if (!workspace || ws_bytes < ccr_inbatch_ce_workspace_bytes(B, dim)) {
        set_error("ccr_inbatch_ce_bwd: workspace %zu bytes required, got %zu", ccr_inbatch_ce_workspace_bytes(B, dim), ws_bytes);
        return CCR_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const InbatchWs w = inbatch_ws(align256(workspace), B, dim);   // the FORWARD's workspace: it holds the scaled logits S and the stamp
    const float gscale = inv_temperature * grad_out / (float)B;
    // dQ[i][:] = sum_j G[j][i] K[j][:];   dK[j][:] = sum_i G[j][i] Q[i][:], rows < B -> dP, the rest -> dN
    const int gi = (B + PREP_T - 1) / PREP_T, gj = (2 * B + PREP_T - 1) / PREP_T, gd = (dim + PREP_T - 1) / PREP_T;
    hipLaunchKernelGGL(inbatch_prep_kernel, dim3((unsigned)(gi * gj + gi * gd + gj * gd)), dim3(256), 0, s, w.S, w.ldS, lse, Qe, Pe, Ne, B, dim, gscale,
                       grad_out_dev, reinterpret_cast<const InbatchStamp *>(w.stamp), __builtin_bit_cast(uint32_t, inv_temperature), w.Gk, w.Gq, w.QT,
                       w.KT, w.ldk, w.ldq);
    CCR_LAUNCH_CHECK();
    const int ntn = (dim + G3_TN - 1) / G3_TN;
    const int nq_blocks = (B + G3_TM - 1) / G3_TM * ntn, nk_blocks = (2 * B + G3_TM - 1) / G3_TM * ntn;
    hipLaunchKernelGGL(inbatch_gemm3_kernel, dim3((unsigned)(nq_blocks + nk_blocks)), dim3(256), 0, s, w.Gq, w.Gk, w.KT, w.QT, B, dim, w.ldq, w.ldk,
                       nq_blocks, dQ, dP, dN);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

extern "C" int ccr_inbatch_ce_bwd(const uint16_t *Qe, const uint16_t *Pe, const uint16_t *Ne, const float *lse, int B, int dim,
                                  float inv_temperature, float grad_out, float *dQ, float *dP, float *dN, void *workspace,
                                  size_t ws_bytes, void *stream) {
    return inbatch_bwd_impl(Qe, Pe, Ne, lse, B, dim, inv_temperature, grad_out, nullptr, dQ, dP, dN, workspace, ws_bytes, stream);
}

extern "C" int ccr_inbatch_ce_bwd_dev(const uint16_t *Qe, const uint16_t *Pe, const uint16_t *Ne, const float *lse, int B, int dim,
                                      float inv_temperature, const float *grad_out_dev, float *dQ, float *dP, float *dN,
                                      void *workspace, size_t ws_bytes, void *stream) {
    CCR_REQUIRE(grad_out_dev, "ccr_inbatch_ce_bwd_dev: null grad_out_dev");
    return inbatch_bwd_impl(Qe, Pe, Ne, lse, B, dim, inv_temperature, 1.f, grad_out_dev, dQ, dP, dN, workspace, ws_bytes, stream);
}
