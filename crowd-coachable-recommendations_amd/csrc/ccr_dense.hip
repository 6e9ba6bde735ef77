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
                                                          int64_t *__restrict__ out_ids, const uint32_t *__restrict__ in_rows,
                                                          bool in_rows_compact) {
    // in_rows: block b serves score row in_rows[b] of the chunk (in_rows_compact: score row b -- the listed rows were scored into
    // consecutive rows) and writes output row q_begin + in_rows[b]; with count_dev, blocks beyond *count_dev exit (the listed-rows
    // form: the BM25 filter's redo list)
    if (in_rows) {
        if (count_dev && blockIdx.x >= *count_dev) return;
    } else if (count_dev && (int)blockIdx.x >= (int)*count_dev - q_begin) {
        return;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_keys[];
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_ctl[4];
    __shared__ int s_wave_eq[4];
    __shared__ int s_gt_pos;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int qi = in_rows ? (int)in_rows[blockIdx.x] : (int)blockIdx.x;
    const float *row = scores + (int64_t)(in_rows_compact ? (int)blockIdx.x : qi) * n_rows;
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
        store_id(out_ids, orow * k + i, id_offset, key_idx(key));
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
                        bool aggregate, const uint32_t *in_rows, bool in_rows_compact) {
    // aggregate: wave-aggregated histogram updates in the radix passes.  Inner-product score rows sit in a handful of top-byte
    // bins (2.8 x faster per NQ-sized row: 0.65 vs 1.8 ms per query); BM25 score rows do not gain (24 k vs 29 k queries/s), so
    // that caller keeps plain atomics.
    const size_t lds = (size_t)pow2_ceil(k) * 8;
    if (aggregate)
        hipLaunchKernelGGL(dense_select_kernel<true>, dim3(nq_chunk), dim3(256), lds, s, scores, n_rows, k, out_rows, q_begin, count_dev,
                           id_offset, out_scores, out_ids, in_rows, in_rows_compact);
    else
        hipLaunchKernelGGL(dense_select_kernel<false>, dim3(nq_chunk), dim3(256), lds, s, scores, n_rows, k, out_rows, q_begin, count_dev,
                           id_offset, out_scores, out_ids, in_rows, in_rows_compact);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}


// ---------------------------------------------------------------------------------------------
// Exact top-k from MFMA score rows + error margins (the fast form of this file's path for inner-product scores).
// scores: [chunk][pitch] fp32 MFMA scores (EPI_STORE of the fused kernel) of the chunk's queries against ALL rows.  Row j of
// tile t = j / 256 has its canonical score inside [m_j - c nt, m_j + c ||d_j||] (c = gamma ||q||), so: L = the
// k-th largest m_j - c nt is a lower bound of the k-th largest canonical score; every row whose upper bound reaches L is
// re-scored canonically (fp64 ordered) and the k best of those, by (score desc, row asc), are the exact result.  A query
// with more than `cap` such rows (mass ties) or non-finite margins is flagged for the fp64 path.
// grid = chunk, block = 1024, dyn LDS = dim * 2 (query row) + cap * 8 (keys).
__global__ __launch_bounds__(1024) void margin_select_kernel(const float *__restrict__ scores, int64_t pitch, int64_t n_rows, int k, int dim,
                                                            const uint16_t *__restrict__ Q, const uint16_t *__restrict__ D,
                                                            const float *__restrict__ tile_norm, const float *__restrict__ row_norm,
                                                            const uint32_t *__restrict__ dmax_bits, float gamma, int cap,
                                                            const float *__restrict__ hint,
                                                            const uint32_t *__restrict__ out_rows, int q_begin, int64_t id_offset,
                                                            float *__restrict__ out_scores, int64_t *__restrict__ out_ids,
                                                            uint32_t *__restrict__ flag_count, uint32_t *__restrict__ flag_list) {
    extern __shared__ __attribute__((aligned(16))) char sm_ms[];
    uint16_t *s_q = reinterpret_cast<uint16_t *>(sm_ms);
    unsigned long long *s_keys = reinterpret_cast<unsigned long long *>(sm_ms + (((size_t)dim * 2 + 15) & ~(size_t)15));
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_ctl[4];
    __shared__ float s_red[16];
    __shared__ uint32_t s_n;
    const int tid = threadIdx.x;
    const int qi = blockIdx.x;
    const float *row = scores + (int64_t)qi * pitch;
    const uint16_t *qrow = Q + (int64_t)qi * dim;   // the chunk's query rows are contiguous (gathered by the caller if need be)
    const int64_t orow = out_rows ? (int64_t)(out_rows[qi] & ~FLAG_DENSE) : (int64_t)(q_begin + qi);

    float ss = 0.f;
    for (int c = tid; c < dim / 8; c += blockDim.x) {
        const uint4 v = reinterpret_cast<const uint4 *>(qrow)[c];
        reinterpret_cast<uint4 *>(s_q)[c] = v;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float lo = __uint_as_float(w[e] << 16), hi = __uint_as_float(w[e] & 0xffff0000u);
            ss = fmaf(lo, lo, ss);
            ss = fmaf(hi, hi, ss);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if ((tid & 63) == 0) s_red[tid >> 6] = ss;
    if (tid == 0) s_n = 0;
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += s_red[w];
    const float c = gamma * (sqrtf(tot) * 1.001f) * 1.001f;
    const float dmax = __uint_as_float(*dmax_bits);
    auto give_up = [&]() {
        if (tid == 0) flag_list[atomicAdd(flag_count, 1u)] = (uint32_t)orow | FLAG_DENSE;
    };
    if (!(dmax < INFINITY) || !(c < INFINITY)) {   // NaN / Inf embeddings: the fp64 path ranks them by its own rule
        give_up();
        return;
    }
    uint32_t kth;
    int need_eq;
    bool done = false;
    // With a valid lower bound tau of the k-th largest score (the fused search's threshold of a flagged query) ONE scan of the
    // row is enough: the rows whose upper bound reaches tau are collected (their lower bounds with them), L is selected among
    // those in LDS and the list is cut down to the rows that reach L.  Without a hint, or if that list overflows: four radix
    // passes + one collecting scan over the row.
    const float tau = hint ? hint[qi] : -INFINITY;
    if (tau > -INFINITY) {
        for (int64_t i = tid; i < n_rows; i += blockDim.x) {
            const float m = row[i], cn = c * tile_norm[i / TILE_DOCS];
            if (m + cn >= tau) {
                const uint32_t p = atomicAdd(&s_n, 1u);
                if (p < (uint32_t)cap) s_keys[p] = ((unsigned long long)f32_orderable(m - cn) << 32) | (unsigned long long)i;
            }
        }
        __syncthreads();
        const int n0 = (int)s_n;
        __syncthreads();
        if (n0 >= k && n0 <= cap) {
            block_radix_select<false>(
                [&](int64_t i, bool &skip) -> uint32_t {
                    (void)skip;
                    return (uint32_t)(s_keys[i] >> 32);
                },
                (int64_t)n0, k, s_hist, s_ctl, kth, need_eq);
            const float low = orderable_to_f32(kth);
            // cut the list down to the rows whose own upper bound reaches L (in place: survivors are re-appended at the front
            // after everybody has read its entries)
            unsigned long long mine[8];
            int nm = 0;
            for (int i = tid; i < n0; i += blockDim.x) {
                const unsigned long long e = s_keys[i];
                const uint32_t r = (uint32_t)e;
                if (fmaf(c, row_norm[r], row[r]) >= low && nm < 8) mine[nm++] = (unsigned long long)r;
            }
            __syncthreads();
            if (tid == 0) s_n = 0;
            __syncthreads();
            for (int j = 0; j < nm; ++j) s_keys[atomicAdd(&s_n, 1u)] = mine[j];
            __syncthreads();
            done = true;
        } else {
            if (tid == 0) s_n = 0;
            __syncthreads();
        }
    }
    if (!done) {
        block_radix_select<true>(
            [&](int64_t i, bool &skip) -> uint32_t {
                (void)skip;
                return f32_orderable(fmaf(-c, tile_norm[i / TILE_DOCS], row[i]));
            },
            n_rows, k, s_hist, s_ctl, kth, need_eq);
        const float low = orderable_to_f32(kth);
        for (int64_t i = tid; i < n_rows; i += blockDim.x) {
            const float m = row[i];
            // the tile's bound first (a coherent table lookup); the row's own norm only for the few that pass it
            if (fmaf(c, tile_norm[i / TILE_DOCS], m) >= low && fmaf(c, row_norm[i], m) >= low) {
                const uint32_t p = atomicAdd(&s_n, 1u);
                if (p < (uint32_t)cap) s_keys[p] = (unsigned long long)i;
            }
        }
        __syncthreads();
    }
    const int n = (int)s_n;
    if (n > cap || n < k) {   // (n < k cannot happen with finite margins: the k rows that define L pass their own test)
        give_up();
        return;
    }
    const int np2 = pow2_ceil(n);
    for (int i = tid; i < np2; i += blockDim.x) {
        unsigned long long key = 0ull;
        if (i < n) {
            const uint32_t r = (uint32_t)s_keys[i];
            key = make_key(canonical_dot(s_q, D + (int64_t)r * dim, dim), r);
        }
        s_keys[i] = key;
    }
    block_bitonic_sort_desc(s_keys, np2);
    for (int i = tid; i < k; i += blockDim.x) {
        const unsigned long long key = s_keys[i];
        out_scores[orow * k + i] = key_score(key);
        store_id(out_ids, orow * k + i, id_offset, key_idx(key));
    }
}

constexpr int MARGIN_SELECT_CAP = 8192;   // rows re-scored per query at most (64 KiB of keys)
static_assert(MARGIN_SELECT_CAP <= 8 * 1024, "margin_select_kernel: a thread keeps at most 8 list entries while the list is rebuilt in place");

int launch_margin_select(const float *scores, int64_t pitch, int64_t n_rows, int k, int dim, const uint16_t *Q, const uint16_t *D,
                         const float *tile_norm, const float *row_norm, const uint32_t *dmax_bits, const float *hint, const uint32_t *out_rows, int q_begin,
                         int nq_chunk, int64_t id_offset, float *out_scores, int64_t *out_ids, uint32_t *flag_count, uint32_t *flag_list,
                         hipStream_t s) {
    const size_t lds = (((size_t)dim * 2 + 15) & ~(size_t)15) + (size_t)MARGIN_SELECT_CAP * 8;
    const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&margin_select_kernel), 96 * 1024);
    if (rc != CCR_OK) return rc;
    hipLaunchKernelGGL(margin_select_kernel, dim3(nq_chunk), dim3(1024), lds, s, scores, pitch, n_rows, k, dim, Q, D, tile_norm, row_norm,
                       dmax_bits, mfma_gamma(dim), MARGIN_SELECT_CAP, hint, out_rows, q_begin, id_offset, out_scores, out_ids, flag_count, flag_list);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

}  // namespace ccr
