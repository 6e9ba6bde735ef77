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
//   bwd : TWO launches, no gradient matrix in memory: dQ = G^T K and dK = G Q with G[j][i] = (exp(S[j][i] - lse_i) - [j == i]) *
//         inv_T * grad_out / B evaluated while the operand is staged, split into THREE bf16 parts (g = hi + mid + lo exactly to
//         2^-24 relative) so that the products with the bf16 embeddings run on v_mfma_f32_32x32x16_bf16 (3 MFMAs of 32 cycles
//         per 16 K instead of 8 fp32 MFMAs of 64): fp32-MFMA accuracy at a fifth of its time; LDS-staged 64x64 tiles,
//         deterministic, no atomics.  (r3: a separate G kernel + two v_mfma_f32_32x32x2_f32 GEMMs, 139 of the step's 175 us; now
//         61 + 40 us.  What bounds them now is the VALU work of evaluating and splitting the gradient while it is staged -- every
//         element once per 64-column output tile, 12 times at dim 768 -- at one wave per SIMD: a variant with one 16-byte LDS store per
//         operand part and thread, conflict free, but 16 scalar loads per thread and chunk, measured SLOWER: 85 + 53 us.)
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
__global__ __launch_bounds__(64) void inbatch_fwd_kernel(const uint16_t *__restrict__ Q, const uint16_t *__restrict__ P,
                                                        const uint16_t *__restrict__ N, int B, int dim, float inv_t, int splits,
                                                        float *__restrict__ pm, float *__restrict__ pl, float *__restrict__ pd,
                                                        float *__restrict__ S, int ldS, float *__restrict__ lse,
                                                        float *__restrict__ loss, unsigned int *__restrict__ tile_ticket,
                                                        unsigned int *__restrict__ loss_ticket, double *__restrict__ tile_part) {
    const int lane = threadIdx.x;
    const int l31 = lane & 31, h = lane >> 5;
    const int i0 = blockIdx.x * 32;
    const int s = blockIdx.y;
    const int ktiles = (2 * B + 31) / 32;
    const int t_lo = (int)((int64_t)s * ktiles / splits), t_hi = (int)((int64_t)(s + 1) * ktiles / splits);
    const int i = i0 + l31;
    const int iq = i < B ? i : B - 1;
    const uint16_t *qrow = Q + (int64_t)iq * dim + 8 * h;
    float m = -INFINITY, l = 0.f, dg = -INFINITY;
    for (int t = t_lo; t < t_hi; ++t) {
        int jr = t * 32 + l31;
        if (jr > 2 * B - 1) jr = 2 * B - 1;
        const uint16_t *krow = key_row(P, N, B, jr, dim) + 8 * h;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        // operands come straight from L2 in fragment layout: 8 K steps (16 loads) are issued before the first MFMA
        int k0 = 0;
        for (; k0 + 128 <= dim; k0 += 128) {
            bf16x8 a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u] = *reinterpret_cast<const bf16x8 *>(krow + k0 + 16 * u);
                b[u] = *reinterpret_cast<const bf16x8 *>(qrow + k0 + 16 * u);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[u], acc, 0, 0, 0);
        }
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
    }
}

// C[M][dim] (fp32) = sum_k A(k, m) X(k, :) with A = the gradient of the loss w.r.t. the scaled logits, evaluated from S and lse while
// it is staged and split into three bf16 parts; X rows are bf16 embedding rows.
//   FOR_Q:  m = query i, k = key j:   A(k, m) = g(S[k][m]),  X(k) = key_row(k)          -> dQ   (M = B, K = 2B)
//   else :  m = key j,   k = query i: A(k, m) = g(S[m][k]),  X(k) = Q[k]                -> dP | dN (M = 2B, K = B; rows >= B go to C1)
//   g(S[j][i]) = (exp(S[j][i] - lse[i]) - [j == i]) * gscale
// Block = 4 waves, 64 x 64 output tile (32 x 32 per wave), K in chunks of 32 through two LDS buffers; the next chunk's global loads
// are in flight while the current chunk's 6 MFMAs per wave run.  LDS images are [row][k] with 80-byte rows (5 x 16 B: the 16 lanes of
// a ds_read_b128 group hit 16 distinct slots): the transposes the operand layout needs happen in the staging stores.
constexpr int GG_T = 64, GG_THREADS = 256, GG_KC = 32, GG_KP = 40;   // 64 x 64 tile, K chunk 32, LDS row pitch 40 bf16
__device__ __forceinline__ void split3(float g, uint16_t &hi, uint16_t &mid, uint16_t &lo) {
    const __bf16 a = (__bf16)g;
    const float r1 = g - (float)a;
    const __bf16 b = (__bf16)r1;
    const __bf16 c = (__bf16)(r1 - (float)b);
    hi = __builtin_bit_cast(uint16_t, a);
    mid = __builtin_bit_cast(uint16_t, b);
    lo = __builtin_bit_cast(uint16_t, c);
}

template <bool FOR_Q>
__global__ __launch_bounds__(GG_THREADS) void inbatch_grad_kernel(const float *__restrict__ S, int ldS, const float *__restrict__ lse,
                                                                 const uint16_t *__restrict__ X0, const uint16_t *__restrict__ X1, int B,
                                                                 int dim, float gscale, const float *__restrict__ gscale_dev,
                                                                 float *__restrict__ C0, float *__restrict__ C1) {
    __shared__ __attribute__((aligned(16))) uint16_t As[2][3][GG_T][GG_KP];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[2][GG_T][GG_KP];
    if (gscale_dev) gscale *= gscale_dev[0];   // upstream gradient read on the device: no host round trip
    const int M = FOR_Q ? B : 2 * B, K = FOR_Q ? 2 * B : B;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;
    const int m0 = blockIdx.y * GG_T, n0 = blockIdx.x * GG_T;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;

    // staging roles.  A, FOR_Q (S rows are keys = k, columns = queries = m, m contiguous): thread -> 4 consecutive m (am) of the k pair
    // (2 ak, 2 ak + 1): two float4 loads, 4 x 3 packed (k, k + 1) dword stores.  A, dK (rows = keys = m, columns = queries = k, k
    // contiguous): thread -> 4 consecutive k (4 ak4) of rows am and am + 32: two float4 loads, 2 x 3 eight-byte stores.
    // X (rows = k, 8 consecutive n per 16-byte load): threads 0..127 -> the k pair (2 bk, 2 bk + 1) x 8 columns: 8 dword stores.
    const int am = FOR_Q ? (tid & 15) * 4 : (tid >> 3), ak = FOR_Q ? (tid >> 4) : (tid & 7);
    const int bk = (tid >> 3) & 15, bn = (tid & 7) * 8;
    // Three chunks of global loads in flight (register sets 0..2): a chunk is ~200 MFMA cycles per wave and less than one workgroup
    // sits on a CU, so a single chunk of lookahead leaves every chunk waiting for an L2 round trip
    constexpr int PF = 3;
    float4 va[PF][2];
    uint4 vb[PF][2];
    float4 lse_m = make_float4(0.f, 0.f, 0.f, 0.f), lse_k[PF];
    if (FOR_Q) {   // lse of the thread's 4 query columns: the same for every chunk
        float t4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = m0 + am + u;
            t4[u] = lse[i < B ? i : B - 1];
        }
        lse_m = make_float4(t4[0], t4[1], t4[2], t4[3]);
    }
    auto fetch = [&](int set, int k0) {
        if (FOR_Q) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                int k = k0 + 2 * ak + r;
                if (k > K - 1) k = K - 1;
                int mc = m0 + am;
                if (mc > ldS - 4) mc = ldS - 4;           // (ldS is a multiple of 4: an aligned vector inside the row pitch)
                va[set][r] = *reinterpret_cast<const float4 *>(S + (int64_t)k * ldS + mc);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                int mrow = m0 + am + 32 * r;
                if (mrow > M - 1) mrow = M - 1;
                int kc = k0 + 4 * ak;
                if (kc > ldS - 4) kc = ldS - 4;
                va[set][r] = *reinterpret_cast<const float4 *>(S + (int64_t)mrow * ldS + kc);
            }
            float t4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = k0 + 4 * ak + u;
                t4[u] = lse[i < B ? i : B - 1];
            }
            lse_k[set] = make_float4(t4[0], t4[1], t4[2], t4[3]);
        }
        {   // (every thread loads -- 128..255 the same rows again -- so that the number of loads in flight is the same on every path:
            // behind a branch hipcc counts vmcnt for the path without the loads and drains the prefetch window)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                int kr = k0 + 2 * bk + r;
                if (kr > K - 1) kr = K - 1;
                int nc = n0 + bn;
                if (nc > dim - 8) nc = dim - 8;
                const uint16_t *row = FOR_Q ? key_row(X0, X1, B, kr, dim) : X0 + (int64_t)kr * dim;
                vb[set][r] = *reinterpret_cast<const uint4 *>(row + nc);
            }
        }
    };
    auto grad = [&](float sv, float lsev, int j, int i) -> float {
        float g = __expf(sv - lsev);
        if (j == i) g -= 1.f;
        return (j < 2 * B && i < B) ? g * gscale : 0.f;   // (a select, not a product: staged positions outside the matrix may hold anything)
    };
    auto stash = [&](int buf, int set, int k0) {
        if (FOR_Q) {
            const float s0[4] = {va[set][0].x, va[set][0].y, va[set][0].z, va[set][0].w}, s1[4] = {va[set][1].x, va[set][1].y, va[set][1].z, va[set][1].w};
            const float ls[4] = {lse_m.x, lse_m.y, lse_m.z, lse_m.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = m0 + am + u, j = k0 + 2 * ak;
                uint16_t p0[3], p1[3];
                split3(grad(s0[u], ls[u], j, i), p0[0], p0[1], p0[2]);
                split3(grad(s1[u], ls[u], j + 1, i), p1[0], p1[1], p1[2]);
#pragma unroll
                for (int part = 0; part < 3; ++part)
                    *reinterpret_cast<uint32_t *>(&As[buf][part][am + u][2 * ak]) = (uint32_t)p0[part] | ((uint32_t)p1[part] << 16);
            }
        } else {
            const float lk[4] = {lse_k[set].x, lse_k[set].y, lse_k[set].z, lse_k[set].w};
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float sv[4] = {va[set][r].x, va[set][r].y, va[set][r].z, va[set][r].w};
                const int j = m0 + am + 32 * r;
                uint16_t p[4][3];
#pragma unroll
                for (int u = 0; u < 4; ++u) split3(grad(sv[u], lk[u], j, k0 + 4 * ak + u), p[u][0], p[u][1], p[u][2]);
#pragma unroll
                for (int part = 0; part < 3; ++part)
                    *reinterpret_cast<uint2 *>(&As[buf][part][am + 32 * r][4 * ak]) =
                        make_uint2((uint32_t)p[0][part] | ((uint32_t)p[1][part] << 16), (uint32_t)p[2][part] | ((uint32_t)p[3][part] << 16));
            }
        }
        if (tid < 128) {
            const uint32_t w0[4] = {vb[set][0].x, vb[set][0].y, vb[set][0].z, vb[set][0].w}, w1[4] = {vb[set][1].x, vb[set][1].y, vb[set][1].z, vb[set][1].w};
            const uint32_t ok0 = (k0 + 2 * bk < K && n0 + bn < dim) ? 0xffffffffu : 0u, ok1 = (k0 + 2 * bk + 1 < K && n0 + bn < dim) ? 0xffffffffu : 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {   // columns bn + 2e (low halves) and bn + 2e + 1 (high halves) of the two k rows
                const uint32_t a = w0[e] & ok0, b = w1[e] & ok1;
                *reinterpret_cast<uint32_t *>(&Bs[buf][bn + 2 * e][2 * bk]) = (a & 0xffffu) | (b << 16);
                *reinterpret_cast<uint32_t *>(&Bs[buf][bn + 2 * e + 1][2 * bk]) = (a >> 16) | (b & 0xffff0000u);
            }
        }
    };
    // chunk c: LDS buffer c % 2, register set c % 3.  Iteration c: fetch chunk c + 3 into the set stash(c) has just freed | MFMAs of
    // chunk c | stash chunk c + 1 | barrier.  Loads beyond K are clamped (never stashed), so neither the prologue nor the loop needs bounds.
#pragma unroll
    for (int u = 0; u < PF; ++u) fetch(u, u * GG_KC);
    stash(0, 0, 0);
    __syncthreads();
    for (int k0 = 0; k0 < K; k0 += PF * GG_KC) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int kc = k0 + u * GG_KC;          // chunk c = kc / 32: buffer (c & 1), set u (k0 / 32 is a multiple of 3)
            if (kc < K) {                           // block-uniform
                const int buf = (kc / GG_KC) & 1;
                fetch(u, kc + PF * GG_KC);
#pragma unroll
                for (int ks = 0; ks < GG_KC / 16; ++ks) {
                    const bf16x8 b = *reinterpret_cast<const bf16x8 *>(&Bs[buf][wn * 32 + l31][ks * 16 + 8 * h]);
#pragma unroll
                    for (int part = 0; part < 3; ++part) {
                        const bf16x8 a = *reinterpret_cast<const bf16x8 *>(&As[buf][part][wm * 32 + l31][ks * 16 + 8 * h]);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                    }
                }
                if (kc + GG_KC < K) stash(buf ^ 1, (u + 1) % PF, kc + GG_KC);
                __syncthreads();
            }
        }
    }
    // C layout of v_mfma_f32_32x32x16: column = lane & 31, register e -> row (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
    const int ncol = n0 + wn * 32 + l31;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int mrow = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (mrow < M && ncol < dim) {
            float *crow = (FOR_Q || mrow < B) ? C0 + (int64_t)mrow * dim : C1 + (int64_t)(mrow - B) * dim;
            crow[ncol] = acc[e];
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
struct InbatchWs {
    float *pm, *pl, *pd, *S;
    int ldS;
    unsigned int *tile_ticket, *loss_ticket;
    double *tile_part;
    size_t zero_bytes;   // tickets: zeroed before every forward
    size_t total;
};
InbatchWs inbatch_ws(void *workspace, int B) {
    InbatchWs w;
    const size_t partial = (((size_t)3 * 64 * B * sizeof(float) + 255) / 256) * 256;
    w.ldS = (B + 3) / 4 * 4;
    const size_t sbytes = (((size_t)2 * B * w.ldS * sizeof(float) + 255) / 256) * 256;
    const int qtiles = (B + 31) / 32;
    w.zero_bytes = (((size_t)(qtiles + 1) * 4 + 255) / 256) * 256;
    const size_t parts = (((size_t)qtiles * 8 + 255) / 256) * 256;
    char *base = (char *)workspace;
    w.pm = (float *)base;
    w.pl = w.pm + (size_t)64 * B;
    w.pd = w.pl + (size_t)64 * B;
    w.S = (float *)(base + partial);
    w.tile_ticket = (unsigned int *)(base + partial + sbytes);
    w.loss_ticket = w.tile_ticket + qtiles;
    w.tile_part = (double *)(base + partial + sbytes + w.zero_bytes);
    w.total = partial + sbytes + w.zero_bytes + parts + 256;   // + 256: the caller's pointer need only be 16-byte aligned
    return w;
}
char *align256(void *p) { return (char *)p + (256 - (uintptr_t)p % 256) % 256; }
}  // namespace

extern "C" size_t ccr_inbatch_ce_workspace_bytes(int B, int dim) {
    (void)dim;
    if (B <= 0) return 0;
    return inbatch_ws(nullptr, B).total;
}

extern "C" int ccr_inbatch_ce_fwd(const uint16_t *Qe, const uint16_t *Pe, const uint16_t *Ne, int B, int dim,
                                  float inv_temperature, float *loss, float *lse, void *workspace, size_t ws_bytes, void *stream) {
    CCR_REQUIRE(Qe && Pe && Ne && loss && lse, "ccr_inbatch_ce_fwd: null pointer");
    CCR_REQUIRE(B >= 1 && dim >= 16 && dim % 16 == 0, "ccr_inbatch_ce_fwd: B=%d dim=%d (dim %% 16 == 0)", B, dim);
    CCR_REQUIRE(((uintptr_t)Qe | (uintptr_t)Pe | (uintptr_t)Ne) % 16 == 0, "ccr_inbatch_ce_fwd: embedding pointers must be 16-byte aligned");
    if (!workspace || ws_bytes < ccr_inbatch_ce_workspace_bytes(B, dim)) {
        set_error("ccr_inbatch_ce_fwd: workspace %zu bytes required, got %zu", ccr_inbatch_ce_workspace_bytes(B, dim), ws_bytes);
        return CCR_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const InbatchWs w = inbatch_ws(align256(workspace), B);
    const int splits = pick_splits(B);
    CCR_HIP_CHECK(hipMemsetAsync(w.tile_ticket, 0, w.zero_bytes, s));
    dim3 grid((B + 31) / 32, splits);
    hipLaunchKernelGGL(inbatch_fwd_kernel, grid, dim3(64), 0, s, Qe, Pe, Ne, B, dim, inv_temperature, splits, w.pm, w.pl, w.pd, w.S, w.ldS,
                       lse, loss, w.tile_ticket, w.loss_ticket, w.tile_part);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
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
    const InbatchWs w = inbatch_ws(align256(workspace), B);   // the FORWARD's workspace: it holds the scaled logits S
    const float gscale = inv_temperature * grad_out / (float)B;
    // dQ[i][:] = sum_j G[j][i] K[j][:];   dK[j][:] = sum_i G[j][i] Q[i][:], rows < B -> dP, the rest -> dN
    const dim3 gq((dim + GG_T - 1) / GG_T, (B + GG_T - 1) / GG_T), gk((dim + GG_T - 1) / GG_T, (2 * B + GG_T - 1) / GG_T);
    hipLaunchKernelGGL((inbatch_grad_kernel<true>), gq, dim3(GG_THREADS), 0, s, w.S, w.ldS, lse, Pe, Ne, B, dim, gscale, grad_out_dev, dQ, dQ);
    CCR_LAUNCH_CHECK();
    hipLaunchKernelGGL((inbatch_grad_kernel<false>), gk, dim3(GG_THREADS), 0, s, w.S, w.ldS, lse, Qe, Qe, B, dim, gscale, grad_out_dev, dP, dN);
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
