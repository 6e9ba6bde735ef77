// ccr_pack.hip -- fp32 -> bf16 pack of encoder outputs (+ L2 normalise, + fused masked mean pooling).
//
// Pure HBM streaming kernels: 6 B per element (4 read + 2 written); the mean-pool variant reads
// L * dim * sizeof(hidden) per row.  16 B per lane loads, 16 B per lane stores, one wave per row
// where a row reduction is needed (fixed reduction order, mirrored by oracle/ccr_oracle.c).
#include <stdlib.h>

#include "ccr_common.h"

namespace ccr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4n __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x4 cvt4(float4 v) {
    bf16x4 r;
    r[0] = (__bf16)v.x;
    r[1] = (__bf16)v.y;
    r[2] = (__bf16)v.z;
    r[3] = (__bf16)v.w;
    return r;
}

// ---------------------------------------------------------------- plain elementwise pack
// n8 = number of 8-element groups; tail handled by the scalar kernel below.
template <int UNROLL>
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float4 *__restrict__ src, bf16x8 *__restrict__ dst,
                                                       int64_t n8) {
    // one block = UNROLL x 256 output vectors: the 2 * UNROLL loads of a thread are issued before the first conversion,
    // consecutive lanes touch consecutive 32-byte segments
    const int64_t base = (int64_t)blockIdx.x * (256 * UNROLL) + threadIdx.x;
    float4 a[UNROLL], b[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const int64_t i = base + u * 256;
        const int64_t ic = i < n8 ? i : n8 - 1;
        a[u] = src[2 * ic];
        b[u] = src[2 * ic + 1];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const int64_t i = base + u * 256;
        const bf16x4 lo = cvt4(a[u]), hi = cvt4(b[u]);
        if (i < n8) dst[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

__global__ void pack_bf16_tail_kernel(const float *__restrict__ src, __bf16 *__restrict__ dst, int64_t begin,
                                      int64_t n) {
    int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (__bf16)src[i];
}


// ---------------------------------------------------------------- pack + norm bound of every PACKED row
// One wave per row, 8 consecutive floats per lane per step (two 16-B loads -> one 16-B store), fp32 sum of
// squares of the bf16-ROUNDED values, wave reduce, one 4-byte store per row: bounds[r] >= ||packed row r|| (inflated by 2^-11
// for the fp32 summation; a NaN / Inf row gives a NaN / Inf bound).  The index reduces the bounds to one per 256-row tile;
// they are used only inside the filter's error margins.
template <int ROWS>
__global__ __launch_bounds__(256) void pack_rows_bound_kernel(const float *__restrict__ src, __bf16 *__restrict__ dst,
                                                             int64_t rows, int dim, float *__restrict__ bounds) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int n8 = dim >> 3;
    // ROWS rows per iteration: 2*ROWS 16-byte loads in flight per lane (pure HBM stream, read once -> nontemporal)
    for (int64_t r = ROWS * wave; r < rows; r += ROWS * nwaves) {
        float ss[ROWS];
#pragma unroll
        for (int j = 0; j < ROWS; ++j) ss[j] = 0.f;
        for (int c = lane; c < n8; c += 64) {
            f32x4n a[ROWS], b[ROWS];
#pragma unroll
            for (int j = 0; j < ROWS; ++j) {
                const int64_t rr = (r + j < rows) ? r + j : rows - 1;
                const f32x4n *x = reinterpret_cast<const f32x4n *>(src + rr * dim);
                a[j] = __builtin_nontemporal_load(x + 2 * c);
                b[j] = __builtin_nontemporal_load(x + 2 * c + 1);
            }
#pragma unroll
            for (int j = 0; j < ROWS; ++j) {
                const bf16x4 lo = cvt4(make_float4(a[j][0], a[j][1], a[j][2], a[j][3]));
                const bf16x4 hi = cvt4(make_float4(b[j][0], b[j][1], b[j][2], b[j][3]));
                const bf16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                if (r + j < rows) __builtin_nontemporal_store(v, reinterpret_cast<bf16x8 *>(dst + (r + j) * dim) + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)v[e];
                    ss[j] = fmaf(f, f, ss[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            float t = ss[j];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off, 64);
            if (lane == 0 && r + j < rows) bounds[r + j] = sqrtf(t) * 1.00048828125f;
        }
    }
}

// ---------------------------------------------------------------- row-wise: norms / normalise + pack
// One wave per row.  Lane l owns float4 chunks c with c % 64 == l; per-lane fp64 partial sum in
// increasing index, then the butterfly p += shfl_xor(p, off) for off = 32..1 (every lane ends
// with the same fp64 sum of squares).  This order is part of the canonical definition.
__device__ __forceinline__ double wave_sumsq(const float *__restrict__ x, int dim, int lane) {
    double p = 0.0;
    const int nchunk = dim >> 2;
    for (int c = lane; c < nchunk; c += 64) {
        float4 v = reinterpret_cast<const float4 *>(x)[c];
        p = p + (double)v.x * (double)v.x;
        p = p + (double)v.y * (double)v.y;
        p = p + (double)v.z * (double)v.z;
        p = p + (double)v.w * (double)v.w;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) p = p + __shfl_xor(p, off, 64);
    return p;
}

__global__ __launch_bounds__(256) void pack_rows_kernel(const float *__restrict__ src, __bf16 *__restrict__ dst,
                                                       float *__restrict__ norms, int64_t rows, int dim,
                                                       int normalize, float *__restrict__ bounds) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int nchunk = dim >> 2;
    for (int64_t r = wave; r < rows; r += nwaves) {
        const float *x = src + r * dim;
        const double ss = wave_sumsq(x, dim, lane);
        const double nrm = sqrt(ss);
        if (norms && lane == 0) norms[r] = (float)nrm;
        const double den = nrm > 1e-12 ? nrm : 1e-12;
        bf16x4 *y = reinterpret_cast<bf16x4 *>(dst + r * dim);
        for (int c = lane; c < nchunk; c += 64) {
            float4 v = reinterpret_cast<const float4 *>(x)[c];
            if (normalize) {
                v.x = (float)((double)v.x / den);
                v.y = (float)((double)v.y / den);
                v.z = (float)((double)v.z / den);
                v.w = (float)((double)v.w / den);
            }
            y[c] = cvt4(v);
        }
        // upper bound of the packed row's norm: bf16 rounding moves each element by at most 2^-8 relative.  One plain store
        // per row (round 2 first did an atomicMax per row on ONE word: 2.7 M atomics on a single address took 30 ms at the NQ
        // shape -- the whole normalising pack ran at 0.4 TB/s)
        if (bounds && lane == 0) bounds[r] = (float)((normalize ? nrm / den : nrm) * 1.004);
    }
}

// ---------------------------------------------------------------- rows of ANY width -> zero-padded rows of a multiple of 8
// One wave per row, element-wise loads (source rows of an odd width are not 16-byte aligned).  The sum of squares keeps the
// canonical order of wave_sumsq: element i belongs to "chunk" i / 4, lane (i / 4) % 64, accumulated in increasing index, then
// the butterfly -- for dim % 4 == 0 the same bits as pack_rows_kernel, and appending zeros to a row changes nothing (p + 0.0 = p).
__global__ __launch_bounds__(256) void pack_rows_padded_kernel(const float *__restrict__ src, int src_dim, __bf16 *__restrict__ dst,
                                                              int dst_dim, float *__restrict__ norms, int64_t rows, int normalize,
                                                              float *__restrict__ bounds) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int nchunk = (dst_dim + 3) >> 2;
    for (int64_t r = wave; r < rows; r += nwaves) {
        const float *x = src + r * src_dim;
        double p = 0.0;
        for (int c = lane; c < nchunk; c += 64) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * c + e;
                const double v = i < src_dim ? (double)x[i] : 0.0;
                p = p + v * v;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) p = p + __shfl_xor(p, off, 64);
        const double nrm = sqrt(p);
        if (norms && lane == 0) norms[r] = (float)nrm;
        const double den = nrm > 1e-12 ? nrm : 1e-12;
        __bf16 *y = dst + r * dst_dim;
        for (int c = lane; c < nchunk; c += 64) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * c + e;
                if (i >= dst_dim) break;
                float v = i < src_dim ? x[i] : 0.f;
                if (normalize && i < src_dim) v = (float)((double)v / den);
                y[i] = (__bf16)v;
            }
        }
        if (bounds && lane == 0) bounds[r] = (float)((normalize ? nrm / den : nrm) * 1.004);
    }
}

// ---------------------------------------------------------------- fused masked mean pooling + pack
// One workgroup of `dim/4` (<=256) threads... generalised: thread t owns float4 chunk columns
// t, t+blockDim, ...; loop over tokens l ascending (fp32 adds in token order = the oracle's order);
// divide by the mask count; optional normalise needs the row norm -> block reduction in fp64
// (NOT the canonical wave order: normalised mean-pool output is tolerance-checked, not bit-checked).
template <typename T>
__device__ __forceinline__ float4 load4(const T *p);
template <>
__device__ __forceinline__ float4 load4<float>(const float *p) {
    return *reinterpret_cast<const float4 *>(p);
}
template <>
__device__ __forceinline__ float4 load4<_Float16>(const _Float16 *p) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    h4 v = *reinterpret_cast<const h4 *>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
template <>
__device__ __forceinline__ float4 load4<__bf16>(const __bf16 *p) {
    bf16x4 v = *reinterpret_cast<const bf16x4 *>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}

template <typename T>
__global__ __launch_bounds__(256) void meanpool_pack_kernel(const T *__restrict__ hidden,
                                                           const int64_t *__restrict__ mask,
                                                           __bf16 *__restrict__ dst_bf16,
                                                           float *__restrict__ dst_f32, int L, int dim,
                                                           int normalize, const int64_t *__restrict__ dst_rows,
                                                           float *__restrict__ bounds,
                                                           const int32_t *__restrict__ seq_start,
                                                           const int32_t *__restrict__ seq_len) {
    __shared__ double red[4];
    __shared__ float redf[4];
    const int b = blockIdx.x;
    // packed token array (seq_start != NULL): sequence b is rows seq_start[b] .. + seq_len[b] - 1 of hidden [T][dim], every one a
    // real token; the sum runs over them in the same order as over the unmasked positions of the padded form: same bits
    const bool packed = seq_start != nullptr;
    if (packed) L = seq_len[b] > 0 ? seq_len[b] : 0;
    const int64_t first_row = packed ? (int64_t)seq_start[b] : (int64_t)b * L;
    const int64_t ob = dst_rows ? dst_rows[b] : (int64_t)b;   // destination row (length-sorted batches scatter back)
    const int tid = threadIdx.x;
    const int nchunk = dim >> 2;
    const int64_t *m = packed ? nullptr : mask + (int64_t)b * L;
    // pass 1: pooled fp32 value per owned chunk, kept in registers (<= 4 chunks per thread for dim <= 4096).
    // Every thread walks the whole mask row anyway, so it counts the tokens itself (no serial pre-pass, no barrier);
    // 8 token positions are loaded unconditionally and masked with an AND (a masked value becomes +0.0, and x + (+0.0) == x
    // bit for bit because the running sum starts at +0.0 and can never be -0.0): the loads of a group are all in flight.
    float4 acc[4];
    int nown = 0;
    double ss = 0.0;
    float inv_is_div = 1.f;
    for (int c = tid; c < nchunk && nown < 4; c += blockDim.x, ++nown) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        const T *h = hidden + first_row * dim + 4 * c;
        long long cnt = 0;
        for (int l0 = 0; l0 < L; l0 += 8) {
            uint32_t keep[8];
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int l = l0 + u < L ? l0 + u : L - 1;
                const long long mv = packed ? 1 : m[l];
                cnt += (l0 + u < L) ? mv : 0;            // the reference divides by mask.sum(1) (item_tower.py:145)
                keep[u] = (l0 + u < L && mv != 0) ? 0xffffffffu : 0u;
                v[u] = load4<T>(h + (int64_t)l * dim);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a.x += __uint_as_float(__float_as_uint(v[u].x) & keep[u]);
                a.y += __uint_as_float(__float_as_uint(v[u].y) & keep[u]);
                a.z += __uint_as_float(__float_as_uint(v[u].z) & keep[u]);
                a.w += __uint_as_float(__float_as_uint(v[u].w) & keep[u]);
            }
        }
        inv_is_div = (float)(int)cnt;
        a.x /= inv_is_div;
        a.y /= inv_is_div;
        a.z /= inv_is_div;
        a.w /= inv_is_div;
        acc[nown] = a;
        ss += (double)a.x * a.x + (double)a.y * a.y + (double)a.z * a.z + (double)a.w * a.w;
    }
    double den = 1.0;
    if (normalize) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
        if ((tid & 63) == 0) red[tid >> 6] = ss;
        __syncthreads();
        double tot = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += red[w];
        double nrm = sqrt(tot);
        den = nrm > 1e-12 ? nrm : 1e-12;
    }
    nown = 0;
    float pss = 0.f;   // squared norm of the PACKED row (what the index's filter margin needs a bound of)
    for (int c = tid; c < nchunk && nown < 4; c += blockDim.x, ++nown) {
        float4 a = acc[nown];
        if (dst_f32) reinterpret_cast<float4 *>(dst_f32 + ob * dim)[c] = a;  // un-normalised pooled row
        if (dst_bf16) {
            if (normalize) {
                a.x = (float)((double)a.x / den);
                a.y = (float)((double)a.y / den);
                a.z = (float)((double)a.z / den);
                a.w = (float)((double)a.w / den);
            }
            const bf16x4 v = cvt4(a);
            reinterpret_cast<bf16x4 *>(dst_bf16 + ob * dim)[c] = v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float f = (float)v[e];
                pss = fmaf(f, f, pss);
            }
        }
    }
    if (bounds && dst_bf16) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) pss += __shfl_xor(pss, off, 64);
        if ((tid & 63) == 0) redf[tid >> 6] = pss;
        __syncthreads();
        if (tid == 0) {
            float tot = 0.f;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += redf[w];
            bounds[ob] = sqrtf(tot) * 1.0001f;   // norm bound of the packed row, at its destination row
        }
    }
}

// Backward of the masked mean pooling: d hidden[b][l][:] = mask[b][l] ? grad[b][:] / count_b : 0 (item_tower.py:141-146 under
// autograd).  One workgroup per batch row: the count once, then a stream of stores over the token axis.
template <typename T>
__device__ __forceinline__ void store4(T *p, float4 v);
template <>
__device__ __forceinline__ void store4<float>(float *p, float4 v) {
    *reinterpret_cast<float4 *>(p) = v;
}
template <>
__device__ __forceinline__ void store4<_Float16>(_Float16 *p, float4 v) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    h4 o = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    *reinterpret_cast<h4 *>(p) = o;
}
template <>
__device__ __forceinline__ void store4<__bf16>(__bf16 *p, float4 v) {
    *reinterpret_cast<bf16x4 *>(p) = cvt4(v);
}

template <typename T>
__global__ __launch_bounds__(256) void meanpool_bwd_kernel(const float *__restrict__ grad, const int64_t *__restrict__ mask,
                                                          T *__restrict__ dhidden, int L, int dim) {
    __shared__ long long s_cnt[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t *m = mask + (int64_t)b * L;
    long long c = 0;
    for (int l = tid; l < L; l += blockDim.x) c += m[l];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((tid & 63) == 0) s_cnt[tid >> 6] = c;
    __syncthreads();
    long long cnt = 0;
    for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) cnt += s_cnt[w];
    const float den = (float)(int)cnt;
    const int nchunk = dim >> 2;
    for (int ch = tid; ch < nchunk; ch += blockDim.x) {
        float4 g = reinterpret_cast<const float4 *>(grad + (int64_t)b * dim)[ch];
        g.x /= den;
        g.y /= den;
        g.z /= den;
        g.w /= den;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        T *o = dhidden + (int64_t)b * L * dim + 4 * ch;
        for (int l = 0; l < L; ++l) store4<T>(o + (int64_t)l * dim, m[l] != 0 ? g : z);
    }
}

}  // namespace ccr

using namespace ccr;

extern "C" int ccr_meanpool_bwd(const float *grad, const int64_t *mask, void *dhidden, int hidden_dtype, int B, int L, int dim,
                                void *stream) {
    CCR_REQUIRE(grad && mask && dhidden, "ccr_meanpool_bwd: null pointer");
    CCR_REQUIRE(B >= 0 && L > 0 && dim > 0 && dim % 4 == 0, "ccr_meanpool_bwd: bad shape B=%d L=%d dim=%d (dim %% 4 == 0)", B, L, dim);
    if (B == 0) return CCR_OK;
    hipStream_t s = (hipStream_t)stream;
    int threads = ((dim / 4 + 63) / 64) * 64;
    if (threads > 256) threads = 256;
    switch (hidden_dtype) {
        case CCR_DTYPE_F32:
            hipLaunchKernelGGL(meanpool_bwd_kernel<float>, dim3(B), dim3(threads), 0, s, grad, mask, (float *)dhidden, L, dim);
            break;
        case CCR_DTYPE_F16:
            hipLaunchKernelGGL(meanpool_bwd_kernel<_Float16>, dim3(B), dim3(threads), 0, s, grad, mask, (_Float16 *)dhidden, L, dim);
            break;
        case CCR_DTYPE_BF16:
            hipLaunchKernelGGL(meanpool_bwd_kernel<__bf16>, dim3(B), dim3(threads), 0, s, grad, mask, (__bf16 *)dhidden, L, dim);
            break;
        default:
            set_error("ccr_meanpool_bwd: unknown hidden_dtype %d", hidden_dtype);
            return CCR_ERR_INVALID;
    }
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

extern "C" int ccr_pack_bf16_ex(const float *src, uint16_t *dst, float *norms, float *row_norm_bounds, int64_t rows, int dim,
                                int normalize, void *stream) {
    CCR_REQUIRE(src && dst, "ccr_pack_bf16: null pointer");
    CCR_REQUIRE(rows >= 0 && dim > 0, "ccr_pack_bf16: bad shape rows=%lld dim=%d", (long long)rows, dim);
    if (rows == 0) return CCR_OK;
    hipStream_t s = (hipStream_t)stream;
    float *bounds = row_norm_bounds;
    if (!normalize && !norms) {
        CCR_REQUIRE(((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0), "ccr_pack_bf16: buffers must be 16-byte aligned");
        if (bounds) {
            CCR_REQUIRE(dim % 8 == 0, "ccr_pack_bf16: row_norm_bounds needs dim %% 8 == 0 (dim=%d)", dim);
            static int rows_per = -1;
            if (rows_per < 0) {
                const char *e = getenv("CCR_PACK_ROWS");
                rows_per = e ? atoi(e) : 1;
            }
            int64_t blocks = (rows + 4 * rows_per - 1) / (4 * rows_per);
            if (blocks > 131072) blocks = 131072;   // measured best at the NQ shape: 16 K - 128 K blocks, one row per wave and trip
            __bf16 *dp = reinterpret_cast<__bf16 *>(dst);
            if (rows_per == 1)
                hipLaunchKernelGGL(pack_rows_bound_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, src, dp, rows, dim, bounds);
            else if (rows_per == 2)
                hipLaunchKernelGGL(pack_rows_bound_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, src, dp, rows, dim, bounds);
            else
                hipLaunchKernelGGL(pack_rows_bound_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, src, dp, rows, dim, bounds);
            CCR_LAUNCH_CHECK();
            return CCR_OK;
        }
        const int64_t n = rows * dim;
        const int64_t n8 = n / 8;
        if (n8 > 0) {
            // one output vector per thread, no loop: many short-lived blocks stream faster than a capped grid-stride loop
            // (measured at the NQ shape: 6.0 TB/s vs 5.5 with 8 vectors per thread and 4.7 with a 4 096-block grid-stride)
            const int64_t blocks = (n8 + 255) / 256;
            hipLaunchKernelGGL(pack_bf16_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const float4 *>(src),
                               reinterpret_cast<bf16x8 *>(dst), n8);
            CCR_LAUNCH_CHECK();
        }
        if (n8 * 8 < n) {
            hipLaunchKernelGGL(pack_bf16_tail_kernel, dim3(1), dim3(64), 0, s, src, reinterpret_cast<__bf16 *>(dst),
                               n8 * 8, n);
            CCR_LAUNCH_CHECK();
        }
        return CCR_OK;
    }
    CCR_REQUIRE(dim % 4 == 0, "ccr_pack_bf16: dim %% 4 != 0 (dim=%d) with normalize/norms", dim);
    CCR_REQUIRE(((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 8 == 0), "ccr_pack_bf16: buffers must be 16-byte aligned");
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 65536) blocks = 65536;   // short-lived waves (a few rows each) stream faster than a 2 048-block grid-stride loop
    hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, reinterpret_cast<__bf16 *>(dst),
                       norms, rows, dim, normalize, bounds);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

extern "C" int ccr_pack_bf16_padded(const float *src, int64_t rows, int src_dim, uint16_t *dst, int dst_dim, float *norms,
                                    float *row_norm_bounds, int normalize, void *stream) {
    CCR_REQUIRE(src && dst, "ccr_pack_bf16_padded: null pointer");
    CCR_REQUIRE(rows >= 0 && src_dim > 0 && dst_dim >= src_dim && dst_dim % 8 == 0 && dst_dim - src_dim < 8,
                "ccr_pack_bf16_padded: bad shape rows=%lld src_dim=%d dst_dim=%d (dst_dim = src_dim rounded up to a multiple of 8)",
                (long long)rows, src_dim, dst_dim);
    if (rows == 0) return CCR_OK;
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(pack_rows_padded_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, src_dim,
                       reinterpret_cast<__bf16 *>(dst), dst_dim, norms, rows, normalize, row_norm_bounds);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

extern "C" int ccr_pack_bf16(const float *src, uint16_t *dst, float *norms, int64_t rows, int dim, int normalize,
                             void *stream) {
    return ccr_pack_bf16_ex(src, dst, norms, nullptr, rows, dim, normalize, stream);
}

extern "C" int ccr_meanpool_pack_bf16_ex(const void *hidden, int hidden_dtype, const int64_t *mask, uint16_t *dst_bf16,
                                         float *dst_f32, const int64_t *dst_rows, float *row_norm_bounds, int B, int L, int dim,
                                         int normalize, void *stream) {
    CCR_REQUIRE(hidden && mask && (dst_bf16 || dst_f32), "ccr_meanpool_pack_bf16: null pointer");
    CCR_REQUIRE(B >= 0 && L > 0 && dim > 0 && dim % 4 == 0 && dim <= 4096,
                "ccr_meanpool_pack_bf16: bad shape B=%d L=%d dim=%d (dim %% 4 == 0, dim <= 4096)", B, L, dim);
    if (B == 0) return CCR_OK;
    hipStream_t s = (hipStream_t)stream;
    int threads = ((dim / 4 + 63) / 64) * 64;
    if (threads > 256) threads = 256;
    __bf16 *db = reinterpret_cast<__bf16 *>(dst_bf16);
    float *mb = row_norm_bounds;
    switch (hidden_dtype) {
        case CCR_DTYPE_F32:
            hipLaunchKernelGGL(meanpool_pack_kernel<float>, dim3(B), dim3(threads), 0, s, (const float *)hidden, mask, db,
                               dst_f32, L, dim, normalize, dst_rows, mb, (const int32_t *)nullptr, (const int32_t *)nullptr);
            break;
        case CCR_DTYPE_F16:
            hipLaunchKernelGGL(meanpool_pack_kernel<_Float16>, dim3(B), dim3(threads), 0, s, (const _Float16 *)hidden,
                               mask, db, dst_f32, L, dim, normalize, dst_rows, mb, (const int32_t *)nullptr, (const int32_t *)nullptr);
            break;
        case CCR_DTYPE_BF16:
            hipLaunchKernelGGL(meanpool_pack_kernel<__bf16>, dim3(B), dim3(threads), 0, s, (const __bf16 *)hidden, mask,
                               db, dst_f32, L, dim, normalize, dst_rows, mb, (const int32_t *)nullptr, (const int32_t *)nullptr);
            break;
        default:
            set_error("ccr_meanpool_pack_bf16: unknown hidden_dtype %d", hidden_dtype);
            return CCR_ERR_INVALID;
    }
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

extern "C" int ccr_meanpool_pack_bf16_packed(const void *hidden, int hidden_dtype, const int32_t *seq_start, const int32_t *seq_len,
                                             uint16_t *dst_bf16, float *dst_f32, const int64_t *dst_rows, float *row_norm_bounds,
                                             int n_seq, int dim, int normalize, void *stream) {
    CCR_REQUIRE(hidden && seq_start && seq_len && (dst_bf16 || dst_f32), "ccr_meanpool_pack_bf16_packed: null pointer");
    CCR_REQUIRE(n_seq >= 0 && dim > 0 && dim % 4 == 0 && dim <= 4096,
                "ccr_meanpool_pack_bf16_packed: bad shape n_seq=%d dim=%d (dim %% 4 == 0, dim <= 4096)", n_seq, dim);
    if (n_seq == 0) return CCR_OK;
    hipStream_t s = (hipStream_t)stream;
    int threads = ((dim / 4 + 63) / 64) * 64;
    if (threads > 256) threads = 256;
    __bf16 *db = reinterpret_cast<__bf16 *>(dst_bf16);
    switch (hidden_dtype) {
        case CCR_DTYPE_F32:
            hipLaunchKernelGGL(meanpool_pack_kernel<float>, dim3(n_seq), dim3(threads), 0, s, (const float *)hidden, nullptr, db,
                               dst_f32, 0, dim, normalize, dst_rows, row_norm_bounds, seq_start, seq_len);
            break;
        case CCR_DTYPE_F16:
            hipLaunchKernelGGL(meanpool_pack_kernel<_Float16>, dim3(n_seq), dim3(threads), 0, s, (const _Float16 *)hidden, nullptr,
                               db, dst_f32, 0, dim, normalize, dst_rows, row_norm_bounds, seq_start, seq_len);
            break;
        case CCR_DTYPE_BF16:
            hipLaunchKernelGGL(meanpool_pack_kernel<__bf16>, dim3(n_seq), dim3(threads), 0, s, (const __bf16 *)hidden, nullptr, db,
                               dst_f32, 0, dim, normalize, dst_rows, row_norm_bounds, seq_start, seq_len);
            break;
        default:
            set_error("ccr_meanpool_pack_bf16_packed: unknown hidden_dtype %d", hidden_dtype);
            return CCR_ERR_INVALID;
    }
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

extern "C" int ccr_meanpool_pack_bf16(const void *hidden, int hidden_dtype, const int64_t *mask, uint16_t *dst_bf16,
                                      float *dst_f32, int B, int L, int dim, int normalize, void *stream) {
    return ccr_meanpool_pack_bf16_ex(hidden, hidden_dtype, mask, dst_bf16, dst_f32, nullptr, nullptr, B, L, dim, normalize,
                                     stream);
}
