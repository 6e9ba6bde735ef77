// ccr_dense.hip -- exact brute-force path: canonical (fp64-ordered) score rows + exact radix top-k.
//
// Used (a) for corpora too small for the sampled-threshold fused path, (b) as the fallback for
// queries the fused path flags (candidate overflow, mass ties), (c) by tests as a second
// implementation of the canonical definition.  No margins anywhere: scores are canonical and
// keys are unique, so the result is the canonical order by construction.
#include "ccr_index.h"
#include "ccr_topk_device.h"

namespace ccr {

constexpr int DT = 64;   // docs per tile
constexpr int QT = 64;   // queries per tile
constexpr int DK = 32;   // k elements per LDS stage
constexpr int LDP = 68;  // padded leading dimension (floats)

// scores[qi][j] = canonical(Q[qsel(qi)], D[j]);  qlist == nullptr -> qsel(qi) = q_begin + qi.
// grid = (ceil(n_rows/DT), ceil(nq_chunk/QT)), block = 256 (thread = 4 queries x 4 docs).
__global__ __launch_bounds__(256) void dense_scores_kernel(const uint16_t *__restrict__ D, int64_t n_rows, int dim,
                                                          const uint16_t *__restrict__ Q, const uint32_t *__restrict__ qlist,
                                                          int q_begin, int nq_chunk, const uint32_t *__restrict__ count_dev,
                                                          float *__restrict__ out) {
    if (count_dev) {   // on-stream fallback chunk: only the first *count_dev - q_begin list entries exist
        const int have = (int)*count_dev - q_begin;
        if (have < nq_chunk) nq_chunk = have;
        if (nq_chunk <= (int)blockIdx.y * QT) return;
    }
    __shared__ __attribute__((aligned(16))) float Qs[DK][LDP];
    __shared__ __attribute__((aligned(16))) float Ds[DK][LDP];
    const int tid = threadIdx.x;
    const int64_t d0 = (int64_t)blockIdx.x * DT;
    const int q0 = blockIdx.y * QT;
    // staging role: one 16-byte chunk (8 bf16) of one row per operand
    const int srow = tid >> 2, schunk = tid & 3;
    int64_t drow = d0 + srow;
    if (drow > n_rows - 1) drow = n_rows - 1;
    int qi = q0 + srow;
    if (qi > nq_chunk - 1) qi = nq_chunk - 1;
    const int qrow = qlist ? (int)(qlist[qi] & ~FLAG_DENSE) : q_begin + qi;
    const uint16_t *dsrc = D + drow * dim + schunk * 8;
    const uint16_t *qsrc = Q + (int64_t)qrow * dim + schunk * 8;
    // compute role
    const int tq = tid & 15, td = tid >> 4;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;

    for (int k0 = 0; k0 < dim; k0 += DK) {
        // zero-fill past dim: fma(0, 0, acc) == acc exactly, so a ragged last stage stays canonical
        uint4 dv = make_uint4(0, 0, 0, 0), qv = make_uint4(0, 0, 0, 0);
        if (k0 + schunk * 8 < dim) {
            dv = *reinterpret_cast<const uint4 *>(dsrc + k0);
            qv = *reinterpret_cast<const uint4 *>(qsrc + k0);
        }
        __syncthreads();  // previous stage fully consumed
        const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w};
        const uint32_t qw[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            Ds[schunk * 8 + 2 * e][srow] = __uint_as_float(dw[e] << 16);
            Ds[schunk * 8 + 2 * e + 1][srow] = __uint_as_float(dw[e] & 0xffff0000u);
            Qs[schunk * 8 + 2 * e][srow] = __uint_as_float(qw[e] << 16);
            Qs[schunk * 8 + 2 * e + 1][srow] = __uint_as_float(qw[e] & 0xffff0000u);
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < DK; ++kk) {
            const float4 qf = *reinterpret_cast<const float4 *>(&Qs[kk][4 * tq]);
            const float4 df = *reinterpret_cast<const float4 *>(&Ds[kk][4 * td]);
            const double qd[4] = {(double)qf.x, (double)qf.y, (double)qf.z, (double)qf.w};
            const double dd[4] = {(double)df.x, (double)df.y, (double)df.z, (double)df.w};
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = fma(qd[a], dd[b], acc[a][b]);
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int qo = q0 + 4 * tq + a;
        if (qo >= nq_chunk) continue;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int64_t j = d0 + 4 * td + b;
            if (j < n_rows) out[(int64_t)qo * n_rows + j] = (float)acc[a][b];
        }
    }
}

// Exact top-k of each score row.  grid = nq_chunk, block = 256, dyn LDS = pow2_ceil(k) * 8 bytes.
// Output row = out_rows ? out_rows[qi] : q_begin + qi.
template <bool AGG>
__global__ __launch_bounds__(256) void dense_select_kernel(const float *__restrict__ scores, int64_t n_rows, int k,
                                                          const uint32_t *__restrict__ out_rows, int q_begin,
                                                          const uint32_t *__restrict__ count_dev,
                                                          int64_t id_offset, float *__restrict__ out_scores,
                                                          int64_t *__restrict__ out_ids) {
    if (count_dev && (int)blockIdx.x >= (int)*count_dev - q_begin) return;
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_keys[];
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_ctl[4];
    __shared__ int s_wave_eq[4];
    __shared__ int s_gt_pos;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int qi = blockIdx.x;
    const float *row = scores + (int64_t)qi * n_rows;
    const int kp2 = pow2_ceil(k);

    uint32_t kth;
    int need_eq;
    block_radix_select<AGG>(
        [&](int64_t i, bool &skip) -> uint32_t {
            (void)skip;
            return f32_orderable(row[i]);
        },
        n_rows, k, s_hist, s_ctl, kth, need_eq);
    const int cnt_gt = k - need_eq;

    for (int i = tid; i < kp2; i += blockDim.x) s_keys[i] = 0ull;
    if (tid == 0) s_gt_pos = 0;
    __syncthreads();

    // Each wave owns a contiguous quarter of the row so that tie order == index order.
    const int64_t seg = (n_rows + 3) / 4;
    const int64_t lo = seg * wv;
    const int64_t hi = (lo + seg < n_rows) ? lo + seg : n_rows;
    int my_eq = 0;
    for (int64_t base = lo; base < hi; base += 64) {
        const int64_t j = base + lane;
        const bool eq = (j < hi) && (f32_orderable(row[j]) == kth);
        my_eq += __popcll(__ballot(eq));
    }
    if (lane == 0) s_wave_eq[wv] = my_eq;
    __syncthreads();
    int eq_base = 0;
    for (int w = 0; w < wv; ++w) eq_base += s_wave_eq[w];

    int eq_run = eq_base;  // wave-uniform running rank of ties
    for (int64_t base = lo; base < hi; base += 64) {
        const int64_t j = base + lane;
        uint32_t o = 0;
        bool in = j < hi;
        if (in) o = f32_orderable(row[j]);
        const bool gt = in && (o > kth);
        const bool eq = in && (o == kth);
        if (gt) {
            const int p = atomicAdd(&s_gt_pos, 1);
            s_keys[p] = ((unsigned long long)o << 32) | (unsigned long long)(~(uint32_t)j);
        }
        const unsigned long long em = __ballot(eq);
        if (eq) {
            const int rank = eq_run + __popcll(em & ((1ull << lane) - 1ull));
            if (rank < need_eq) s_keys[cnt_gt + rank] = ((unsigned long long)o << 32) | (unsigned long long)(~(uint32_t)j);
        }
        eq_run += __popcll(em);
    }
    __syncthreads();
    block_bitonic_sort_desc(s_keys, kp2);

    const int64_t orow = out_rows ? (int64_t)(out_rows[qi] & ~FLAG_DENSE) : (int64_t)(q_begin + qi);
    for (int i = tid; i < k; i += blockDim.x) {
        const unsigned long long key = s_keys[i];
        out_scores[orow * k + i] = key_score(key);
        out_ids[orow * k + i] = id_offset + (int64_t)key_idx(key);
    }
}

int launch_dense_scores(const uint16_t *D, int64_t n_rows, int dim, const uint16_t *Q, const uint32_t *qlist,
                        int q_begin, int nq_chunk, const uint32_t *count_dev, float *out, hipStream_t s) {
    dim3 grid((unsigned)((n_rows + DT - 1) / DT), (unsigned)((nq_chunk + QT - 1) / QT));
    hipLaunchKernelGGL(dense_scores_kernel, grid, dim3(256), 0, s, D, n_rows, dim, Q, qlist, q_begin, nq_chunk, count_dev, out);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

int launch_dense_select(const float *scores, int64_t n_rows, int k, const uint32_t *out_rows, int q_begin, int nq_chunk,
                        const uint32_t *count_dev, int64_t id_offset, float *out_scores, int64_t *out_ids, hipStream_t s,
                        bool aggregate) {
    // aggregate: wave-aggregated histogram updates in the radix passes.  Inner-product score rows sit in a handful of top-byte
    // bins (2.8 x faster per NQ-sized row: 0.65 vs 1.8 ms per query); BM25 score rows do not gain (24 k vs 29 k queries/s), so
    // that caller keeps plain atomics.
    const size_t lds = (size_t)pow2_ceil(k) * 8;
    if (aggregate)
        hipLaunchKernelGGL(dense_select_kernel<true>, dim3(nq_chunk), dim3(256), lds, s, scores, n_rows, k, out_rows, q_begin, count_dev,
                           id_offset, out_scores, out_ids);
    else
        hipLaunchKernelGGL(dense_select_kernel<false>, dim3(nq_chunk), dim3(256), lds, s, scores, n_rows, k, out_rows, q_begin, count_dev,
                           id_offset, out_scores, out_ids);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

}  // namespace ccr
