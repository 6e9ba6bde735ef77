// ccr_inbatch.hip -- in-batch-negative contrastive loss ("multiple_nrl", bbpr.py:205-212), fwd + bwd.
//
//   logits[i][j] = inv_T * <Q_i, K_j>,  K = [P ; N]  (2B keys),  loss = mean_i( lse_i - logits[i][i] )
//
// The logits tile is the retrieval kernel's MFMA tile (v_mfma_f32_32x32x16_bf16, keys on the
// accumulator rows, queries on the lanes), so a lane owns one query column and the softmax over keys
// is an in-register online reduction.  B x 2B is small (the encoder dominates the step): operands are
// read straight from L2 in fragment layout, no LDS staging.
//   fwd : per (query tile, key split) partial (max, sum, diagonal) -> fixed-order combine -> lse, loss
//   bwd : G[j][i] = (softmax - onehot) * inv_T * grad_out / B  (fp32, recomputed logits, MFMA)
//         dQ = G^T K,  dK = G Q   as fp32 tiled outer-product GEMMs (deterministic, no atomics)
#include "ccr_common.h"

namespace ccr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ const uint16_t *key_row(const uint16_t *P, const uint16_t *N, int B, int j, int dim) {
    return (j < B) ? P + (int64_t)j * dim : N + (int64_t)(j - B) * dim;
}

// MODE 0: partial softmax statistics; MODE 1: gradient matrix G[2B][B]
template <int MODE>
__global__ __launch_bounds__(64) void inbatch_logits_kernel(const uint16_t *__restrict__ Q, const uint16_t *__restrict__ P,
                                                           const uint16_t *__restrict__ N, int B, int dim, float inv_t,
                                                           int splits, float *__restrict__ pm, float *__restrict__ pl,
                                                           float *__restrict__ pd, const float *__restrict__ lse,
                                                           float gscale, float *__restrict__ G) {
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
    const float my_lse = (MODE == 1) ? lse[iq] : 0.f;
    for (int t = t_lo; t < t_hi; ++t) {
        int jr = t * 32 + l31;
        if (jr > 2 * B - 1) jr = 2 * B - 1;
        const uint16_t *krow = key_row(P, N, B, jr, dim) + 8 * h;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        for (int k0 = 0; k0 < dim; k0 += 16) {
            const bf16x8 a = *reinterpret_cast<const bf16x8 *>(krow + k0);
            const bf16x8 b = *reinterpret_cast<const bf16x8 *>(qrow + k0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        if (MODE == 0) {
            float tm = -INFINITY;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int j = t * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const float v = (j < 2 * B) ? acc[e] * inv_t : -INFINITY;
                acc[e] = v;
                tm = fmaxf(tm, v);
                if (j == i) dg = v;
            }
            const float mn = fmaxf(m, tm);
            if (mn > -INFINITY) {
                float sum = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) sum += __expf(acc[e] - mn);
                l = l * __expf(m - mn) + sum;
                m = mn;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int j = t * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (j < 2 * B && i < B) {
                    float g = __expf(acc[e] * inv_t - my_lse);
                    if (j == i) g -= 1.f;
                    G[(int64_t)j * B + i] = g * gscale;
                }
            }
        }
    }
    if (MODE == 0) {
        // the two lane halves hold different key rows of the same query column
        const float m2 = __shfl_xor(m, 32, 64), l2 = __shfl_xor(l, 32, 64), d2 = __shfl_xor(dg, 32, 64);
        const float M = fmaxf(m, m2);
        float L = 0.f;
        if (M > -INFINITY) L = (m > -INFINITY ? l * __expf(m - M) : 0.f) + (m2 > -INFINITY ? l2 * __expf(m2 - M) : 0.f);
        if (h == 0 && i < B) {
            pm[(int64_t)s * B + i] = M;
            pl[(int64_t)s * B + i] = L;
            pd[(int64_t)s * B + i] = fmaxf(dg, d2);
        }
    }
}

// combine the split partials in a fixed order: lse[i], loss = sum_i (lse_i - logit_ii) / B
__global__ __launch_bounds__(256) void inbatch_reduce_kernel(const float *__restrict__ pm, const float *__restrict__ pl,
                                                            const float *__restrict__ pd, int B, int splits,
                                                            float *__restrict__ lse, float *__restrict__ loss) {
    __shared__ double s_sum[256];
    double part = 0.0;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        float M = -INFINITY, dg = -INFINITY;
        for (int s = 0; s < splits; ++s) {
            M = fmaxf(M, pm[(int64_t)s * B + i]);
            dg = fmaxf(dg, pd[(int64_t)s * B + i]);
        }
        float L = 0.f;
        for (int s = 0; s < splits; ++s) {
            const float ms = pm[(int64_t)s * B + i];
            if (ms > -INFINITY) L += pl[(int64_t)s * B + i] * __expf(ms - M);
        }
        const float v = M + __logf(L);
        lse[i] = v;
        part += (double)(v - dg);
    }
    s_sum[threadIdx.x] = part;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) s_sum[threadIdx.x] += s_sum[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(s_sum[0] / (double)B);
}

// C[M][Nc] (fp32) = sum_k A(k, m) * Bm(k, n);  A fp32: A_KMAJOR ? A[k*lda + m] : A[m*lda + k];
// Bm rows are bf16 embedding rows selected by the mode: KEYS -> key_row(P, N, B, k), else Q + k*dim.
// Output row m goes to C0 (m < split) or C1 (m >= split).  64x64 tile, 256 threads x (4x4), K chunk 16.
template <bool A_KMAJOR, bool B_KEYS>
__global__ __launch_bounds__(256) void inbatch_grad_gemm_kernel(const float *__restrict__ A, int lda, const uint16_t *__restrict__ X0,
                                                               const uint16_t *__restrict__ X1, int B, int M, int K, int dim,
                                                               float *__restrict__ C0, float *__restrict__ C1, int split) {
    __shared__ __attribute__((aligned(16))) float As[16][68];
    __shared__ __attribute__((aligned(16))) float Bs[16][68];
    const int tid = threadIdx.x;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tm = tid >> 4, tn = tid & 15;
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        // stage A chunk [16 k][64 m] and B chunk [16 k][64 n]: 1024 elements each, 4 per thread
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int idx = r * 256 + tid;
            int kk, mm;
            if (A_KMAJOR) {
                kk = idx >> 6;
                mm = idx & 63;
            } else {
                mm = idx >> 4;
                kk = idx & 15;
            }
            const int k = k0 + kk, mrow = m0 + mm;
            float v = 0.f;
            if (k < K && mrow < M) v = A_KMAJOR ? A[(int64_t)k * lda + mrow] : A[(int64_t)mrow * lda + k];
            As[kk][mm] = v;
            const int kb = idx >> 6, nn = idx & 63;
            const int kr = k0 + kb, ncol = n0 + nn;
            float w = 0.f;
            if (kr < K && ncol < dim) {
                const uint16_t *row = B_KEYS ? key_row(X0, X1, B, kr, dim) : X0 + (int64_t)kr * dim;
                w = bf16_bits_to_f32(row[ncol]);
            }
            Bs[kb][nn] = w;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float4 av = *reinterpret_cast<const float4 *>(&As[kk][4 * tm]);
            const float4 bv = *reinterpret_cast<const float4 *>(&Bs[kk][4 * tn]);
            const float a4[4] = {av.x, av.y, av.z, av.w}, b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = fmaf(a4[a], b4[b], acc[a][b]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int mrow = m0 + 4 * tm + a;
        if (mrow >= M) continue;
        float *crow = (mrow < split) ? C0 + (int64_t)mrow * dim : C1 + (int64_t)(mrow - split) * dim;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int ncol = n0 + 4 * tn + b;
            if (ncol < dim) crow[ncol] = acc[a][b];
        }
    }
}

static int pick_splits(int B) {
    const int qtiles = (B + 31) / 32, ktiles = (2 * B + 31) / 32;
    int s = 512 / (qtiles > 0 ? qtiles : 1);
    if (s < 1) s = 1;
    if (s > ktiles) s = ktiles;
    if (s > 64) s = 64;
    return s;
}

}  // namespace ccr

using namespace ccr;

extern "C" size_t ccr_inbatch_ce_workspace_bytes(int B, int dim) {
    (void)dim;
    if (B <= 0) return 0;
    const size_t partial = (size_t)3 * 64 * B * sizeof(float);
    const size_t g = (size_t)2 * B * B * sizeof(float);
    return ((partial + 255) / 256) * 256 + g + 256;
}

extern "C" int ccr_inbatch_ce_fwd(const uint16_t *Qe, const uint16_t *Pe, const uint16_t *Ne, int B, int dim,
                                  float inv_temperature, float *loss, float *lse, void *workspace, size_t ws_bytes, void *stream) {
    CCR_REQUIRE(Qe && Pe && Ne && loss && lse, "ccr_inbatch_ce_fwd: null pointer");
    CCR_REQUIRE(B >= 1 && dim >= 16 && dim % 16 == 0, "ccr_inbatch_ce_fwd: B=%d dim=%d (dim %% 16 == 0)", B, dim);
    if (!workspace || ws_bytes < ccr_inbatch_ce_workspace_bytes(B, dim)) {
        set_error("ccr_inbatch_ce_fwd: workspace %zu bytes required, got %zu", ccr_inbatch_ce_workspace_bytes(B, dim), ws_bytes);
        return CCR_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int splits = pick_splits(B);
    float *pm = (float *)workspace, *pl = pm + (size_t)64 * B, *pd = pl + (size_t)64 * B;
    dim3 grid((B + 31) / 32, splits);
    hipLaunchKernelGGL(inbatch_logits_kernel<0>, grid, dim3(64), 0, s, Qe, Pe, Ne, B, dim, inv_temperature, splits, pm, pl, pd,
                       (const float *)nullptr, 0.f, (float *)nullptr);
    CCR_LAUNCH_CHECK();
    hipLaunchKernelGGL(inbatch_reduce_kernel, dim3(1), dim3(256), 0, s, pm, pl, pd, B, splits, lse, loss);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

extern "C" int ccr_inbatch_ce_bwd(const uint16_t *Qe, const uint16_t *Pe, const uint16_t *Ne, const float *lse, int B, int dim,
                                  float inv_temperature, float grad_out, float *dQ, float *dP, float *dN, void *workspace,
                                  size_t ws_bytes, void *stream) {
    CCR_REQUIRE(Qe && Pe && Ne && lse && dQ && dP && dN, "ccr_inbatch_ce_bwd: null pointer");
    CCR_REQUIRE(B >= 1 && dim >= 16 && dim % 16 == 0, "ccr_inbatch_ce_bwd: B=%d dim=%d (dim %% 16 == 0)", B, dim);
    if (!workspace || ws_bytes < ccr_inbatch_ce_workspace_bytes(B, dim)) {
        set_error("ccr_inbatch_ce_bwd: workspace %zu bytes required, got %zu", ccr_inbatch_ce_workspace_bytes(B, dim), ws_bytes);
        return CCR_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t partial = (((size_t)3 * 64 * B * sizeof(float) + 255) / 256) * 256;
    float *G = (float *)((char *)workspace + partial);
    const int splits = pick_splits(B);
    dim3 grid((B + 31) / 32, splits);
    hipLaunchKernelGGL(inbatch_logits_kernel<1>, grid, dim3(64), 0, s, Qe, Pe, Ne, B, dim, inv_temperature, splits,
                       (float *)nullptr, (float *)nullptr, (float *)nullptr, lse, inv_temperature * grad_out / (float)B, G);
    CCR_LAUNCH_CHECK();
    // dQ[i][:] = sum_j G[j][i] K[j][:]      (A k-major: A[k=j][m=i])
    hipLaunchKernelGGL((inbatch_grad_gemm_kernel<true, true>), dim3((dim + 63) / 64, (B + 63) / 64), dim3(256), 0, s, G, B, Pe, Ne,
                       B, B, 2 * B, dim, dQ, dQ, B);
    CCR_LAUNCH_CHECK();
    // dK[j][:] = sum_i G[j][i] Q[i][:]      (A row-major [m=j][k=i]); rows < B -> dP, the rest -> dN
    hipLaunchKernelGGL((inbatch_grad_gemm_kernel<false, false>), dim3((dim + 63) / 64, (2 * B + 63) / 64), dim3(256), 0, s, G, B,
                       Qe, Qe, B, 2 * B, B, dim, dP, dN, B);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}
