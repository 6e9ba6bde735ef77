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
//         dQ = G^T K,  dK = G Q   on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products of the fp32
//         gradient matrix with the bf16 embeddings, fp32 accumulate; LDS-staged 64x64 tiles, deterministic, no atomics)
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
                                                           float gscale, const float *__restrict__ gscale_dev, float *__restrict__ G) {
    const int lane = threadIdx.x;
    if (MODE == 1 && gscale_dev) gscale *= gscale_dev[0];   // upstream gradient read on the device: no host round trip
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

// combine the split partials in a fixed order: lse[i], loss = sum_i (lse_i - logit_ii) / B.
// grid = ceil(B / 256) blocks of 256 queries; every block writes its fp64 partial, the block that draws the last ticket
// adds the partials in block order (deterministic) and writes the loss.  `ticket` is zeroed by the caller's memset.
__global__ __launch_bounds__(256) void inbatch_reduce_kernel(const float *__restrict__ pm, const float *__restrict__ pl,
                                                            const float *__restrict__ pd, int B, int splits,
                                                            float *__restrict__ lse, float *__restrict__ loss,
                                                            double *__restrict__ block_part, unsigned int *__restrict__ ticket) {
    __shared__ double s_sum[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    double part = 0.0;
    if (i < B) {
        float M = -INFINITY, dg = -INFINITY, L = 0.f;
        for (int s0 = 0; s0 < splits; s0 += 16) {   // 3 x 16 independent loads per round
            float vm[16], vl[16], vd[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const bool in = s0 + u < splits;
                const int64_t at = (int64_t)(in ? s0 + u : 0) * B + i;
                vm[u] = in ? pm[at] : -INFINITY;
                vl[u] = in ? pl[at] : 0.f;
                vd[u] = in ? pd[at] : -INFINITY;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {   // online combine in split order
                const float mn = fmaxf(M, vm[u]);
                if (mn > -INFINITY) L = (M > -INFINITY ? L * __expf(M - mn) : 0.f) + (vm[u] > -INFINITY ? vl[u] * __expf(vm[u] - mn) : 0.f);
                M = mn;
                dg = fmaxf(dg, vd[u]);
            }
        }
        const float v = M + __logf(L);
        lse[i] = v;
        part = (double)(v - dg);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        // publish the partial (agent-scope release before the ticket), last arriver acquires and sums in block order
        block_part[blockIdx.x] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            double tot = 0.0;
            for (unsigned int b = 0; b < gridDim.x; ++b) tot += block_part[b];
            loss[0] = (float)(tot / (double)B);
        }
    }
}

// C[M][Nc] (fp32) = sum_k A(k, m) * Bm(k, n) on v_mfma_f32_32x32x2_f32.
//   A fp32: A_KMAJOR ? A[k*lda + m] : A[m*lda + k];  Bm rows are bf16 embedding rows selected by the mode:
//   KEYS -> key_row(P, N, B, k), else Q + k*dim (widened to fp32 exactly).  Output row m goes to C0 (m < split) or C1.
// Block = 4 waves, 64 x 64 output tile (32 x 32 per wave), K walked in chunks of 32 through two LDS buffers; the next
// chunk's global loads are in flight while the current chunk's 16 MFMAs per wave run.  LDS rows are k-major with a
// stride of 68 floats (16-byte aligned rows): the MFMA operand reads (32 consecutive floats per half wave) and both staging patterns
// (m-fastest and k-fastest) are bank-conflict free.
typedef float f32x16v __attribute__((ext_vector_type(16)));
constexpr int GG_T = 64, GG_THREADS = 256, GG_KC = 32, GG_LD = 68;   // 64 x 64 block tile, 4 waves of 32 x 32, K chunk 32
// VEC: 16-byte loads (needs lda % 4 == 0, M % 4 == 0, dim % 8 == 0, 16-byte aligned bases): 3 loads per thread and chunk
// instead of 16 scalar ones.
template <bool A_KMAJOR, bool B_KEYS, bool VEC>
__global__ __launch_bounds__(GG_THREADS) void inbatch_grad_gemm_kernel(const float *__restrict__ A, int lda, const uint16_t *__restrict__ X0,
                                                               const uint16_t *__restrict__ X1, int B, int M, int K, int dim,
                                                               float *__restrict__ C0, float *__restrict__ C1, int split) {
    __shared__ __attribute__((aligned(16))) float As[2][GG_KC][GG_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][GG_KC][GG_LD];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;
    const int m0 = blockIdx.y * GG_T, n0 = blockIdx.x * GG_T;
    f32x16v acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;

    // Raw load results are only touched AFTER the MFMAs of the current chunk and out-of-range elements are zeroed
    // arithmetically there (addresses are clamped): behind a select LLVM sinks each load into the select's branch and
    // waits for it separately, one dependent round trip per element.
    constexpr int NA = VEC ? 2 : 8;
    float4 va[VEC ? 2 : 1];
    uint4 vb;
    float ra[VEC ? 1 : 8];
    uint32_t rb[VEC ? 1 : 8];
    auto fetch = [&](int k0) {
        if constexpr (VEC) {
#pragma unroll
            for (int r = 0; r < NA; ++r) {
                const int v = r * GG_THREADS + tid;   // 512 float4 of the A chunk
                const int kk = A_KMAJOR ? (v >> 4) : ((v & 7) * 4), mm = A_KMAJOR ? ((v & 15) * 4) : (v >> 3);
                const int k = k0 + kk, mrow = m0 + mm;
                // K and M are multiples of 4 here, so a 4-vector is entirely inside or entirely outside the matrix
                const int kc = A_KMAJOR ? (k < K ? k : K - 1) : (k < K ? k : K - 4);
                const int mc = A_KMAJOR ? (mrow < M ? mrow : M - 4) : (mrow < M ? mrow : M - 1);
                va[r] = *reinterpret_cast<const float4 *>(A_KMAJOR ? A + (int64_t)kc * lda + mc : A + (int64_t)mc * lda + kc);
            }
            const int kb = tid >> 3, n8 = (tid & 7) * 8;   // 256 x 16 bytes of the B chunk
            const int kr = k0 + kb, ncol = n0 + n8;
            const int krc = kr < K ? kr : K - 1, ncc = ncol + 7 < dim ? ncol : dim - 8;
            const uint16_t *row = B_KEYS ? key_row(X0, X1, B, krc, dim) : X0 + (int64_t)krc * dim;
            vb = *reinterpret_cast<const uint4 *>(row + ncc);
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int idx = r * GG_THREADS + tid;
                const int kk = A_KMAJOR ? (idx >> 6) : (idx & 31), mm = A_KMAJOR ? (idx & 63) : (idx >> 5);
                const int k = k0 + kk, mrow = m0 + mm;
                const int kc = k < K ? k : K - 1, mc = mrow < M ? mrow : M - 1;
                ra[r] = A_KMAJOR ? A[(int64_t)kc * lda + mc] : A[(int64_t)mc * lda + kc];
                const int kr = k0 + (idx >> 6), ncol = n0 + (idx & 63);
                const int krc = kr < K ? kr : K - 1, ncc = ncol < dim ? ncol : dim - 1;
                const uint16_t *row = B_KEYS ? key_row(X0, X1, B, krc, dim) : X0 + (int64_t)krc * dim;
                rb[r] = row[ncc];
            }
        }
    };
    auto stash = [&](int buf, int k0) {
        if constexpr (VEC) {
#pragma unroll
            for (int r = 0; r < NA; ++r) {
                const int v = r * GG_THREADS + tid;
                const int kk = A_KMAJOR ? (v >> 4) : ((v & 7) * 4), mm = A_KMAJOR ? ((v & 15) * 4) : (v >> 3);
                const float e4[4] = {va[r].x, va[r].y, va[r].z, va[r].w};
                if (A_KMAJOR) {   // 4 consecutive m of one k row; a row/vector is entirely in or out (M % 4 == 0)
                    const float ok = (k0 + kk < K && m0 + mm < M) ? 1.f : 0.f;
                    *reinterpret_cast<float4 *>(&As[buf][kk][mm]) = make_float4(e4[0] * ok, e4[1] * ok, e4[2] * ok, e4[3] * ok);
                } else {          // 4 consecutive k of one m row
                    const float ok = (k0 + kk < K && m0 + mm < M) ? 1.f : 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) As[buf][kk + j][mm] = e4[j] * ok;
                }
            }
            const int kb = tid >> 3, n8 = (tid & 7) * 8;
            const uint32_t ok = (k0 + kb < K && n0 + n8 < dim) ? 0xffffffffu : 0u;
            const uint32_t w[4] = {vb.x, vb.y, vb.z, vb.w};
            float f[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[2 * e] = __uint_as_float((w[e] << 16) & ok);
                f[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u & ok);
            }
            *reinterpret_cast<float4 *>(&Bs[buf][kb][n8]) = make_float4(f[0], f[1], f[2], f[3]);
            *reinterpret_cast<float4 *>(&Bs[buf][kb][n8 + 4]) = make_float4(f[4], f[5], f[6], f[7]);
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int idx = r * GG_THREADS + tid;
                const int kk = A_KMAJOR ? (idx >> 6) : (idx & 31), mm = A_KMAJOR ? (idx & 63) : (idx >> 5);
                As[buf][kk][mm] = ra[r] * ((k0 + kk < K && m0 + mm < M) ? 1.f : 0.f);
                Bs[buf][idx >> 6][idx & 63] =
                    __uint_as_float((rb[r] << 16) & ((k0 + (idx >> 6) < K && n0 + (idx & 63) < dim) ? 0xffffffffu : 0u));
            }
        }
    };
    fetch(0);
    stash(0, 0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += GG_KC) {
        const bool more = k0 + GG_KC < K;
        if (more) fetch(k0 + GG_KC);                   // global loads of the next chunk stay in flight over the MFMAs
#pragma unroll
        for (int ks = 0; ks < GG_KC / 2; ++ks) {
            const float a = As[buf][2 * ks + h][wm * 32 + l31];
            const float b = Bs[buf][2 * ks + h][wn * 32 + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (more) stash(buf ^ 1, k0 + GG_KC);
        __syncthreads();
        buf ^= 1;
    }
    // C layout of v_mfma_f32_32x32x2: column = lane & 31, register e -> row (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
    const int ncol = n0 + wn * 32 + l31;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int mrow = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (mrow < M && ncol < dim) {
            float *crow = (mrow < split) ? C0 + (int64_t)mrow * dim : C1 + (int64_t)(mrow - split) * dim;
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

extern "C" size_t ccr_inbatch_ce_workspace_bytes(int B, int dim) {
    (void)dim;
    if (B <= 0) return 0;
    const size_t partial = (size_t)3 * 64 * B * sizeof(float);
    const size_t g = (size_t)2 * B * B * sizeof(float);
    const size_t red = (size_t)((B + 255) / 256) * sizeof(double) + 256;   // block partials + ticket of the loss reduction
    return ((partial + 255) / 256) * 256 + g + ((red + 255) / 256) * 256 + 256;
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
                       (const float *)nullptr, 0.f, (const float *)nullptr, (float *)nullptr);
    CCR_LAUNCH_CHECK();
    const size_t partial = (((size_t)3 * 64 * B * sizeof(float) + 255) / 256) * 256;
    char *red = (char *)workspace + partial + (size_t)2 * B * B * sizeof(float);
    red += (256 - (uintptr_t)red % 256) % 256;
    unsigned int *ticket = (unsigned int *)red;
    double *block_part = (double *)(red + 64);
    CCR_HIP_CHECK(hipMemsetAsync(ticket, 0, 64, s));
    hipLaunchKernelGGL(inbatch_reduce_kernel, dim3((B + 255) / 256), dim3(256), 0, s, pm, pl, pd, B, splits, lse, loss, block_part,
                       ticket);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

static int inbatch_bwd_impl(const uint16_t *Qe, const uint16_t *Pe, const uint16_t *Ne, const float *lse, int B, int dim,
                            float inv_temperature, float grad_out, const float *grad_out_dev, float *dQ, float *dP, float *dN,
                            void *workspace, size_t ws_bytes, void *stream) {
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
                       (float *)nullptr, (float *)nullptr, (float *)nullptr, lse, inv_temperature * grad_out / (float)B, grad_out_dev, G);
    CCR_LAUNCH_CHECK();
    // dQ[i][:] = sum_j G[j][i] K[j][:]      (A k-major: A[k=j][m=i])
    // dK[j][:] = sum_i G[j][i] Q[i][:]      (A row-major [m=j][k=i]); rows < B -> dP, the rest -> dN
    const bool vec = (B % 4 == 0) && B >= 4 && (dim % 8 == 0) && dim >= 8 && ((uintptr_t)Qe % 16 == 0) && ((uintptr_t)Pe % 16 == 0) &&
                     ((uintptr_t)Ne % 16 == 0) && ((uintptr_t)G % 16 == 0);
    const dim3 gq((dim + GG_T - 1) / GG_T, (B + GG_T - 1) / GG_T), gk((dim + GG_T - 1) / GG_T, (2 * B + GG_T - 1) / GG_T);
    if (vec) {
        hipLaunchKernelGGL((inbatch_grad_gemm_kernel<true, true, true>), gq, dim3(GG_THREADS), 0, s, G, B, Pe, Ne, B, B, 2 * B, dim, dQ, dQ, B);
        CCR_LAUNCH_CHECK();
        hipLaunchKernelGGL((inbatch_grad_gemm_kernel<false, false, true>), gk, dim3(GG_THREADS), 0, s, G, B, Qe, Qe, B, 2 * B, B, dim, dP, dN, B);
    } else {
        hipLaunchKernelGGL((inbatch_grad_gemm_kernel<true, true, false>), gq, dim3(GG_THREADS), 0, s, G, B, Pe, Ne, B, B, 2 * B, dim, dQ, dQ, B);
        CCR_LAUNCH_CHECK();
        hipLaunchKernelGGL((inbatch_grad_gemm_kernel<false, false, false>), gk, dim3(GG_THREADS), 0, s, G, B, Qe, Qe, B, 2 * B, B, dim, dP, dN, B);
    }
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
