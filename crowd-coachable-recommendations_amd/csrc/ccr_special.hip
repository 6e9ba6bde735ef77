// ccr_special.hip -- searches whose scores are modified on a sparse set of (query, column) cells:
//   * ccr_search_blocked       per-query blocked ids of any length (scores[block_ind] = -1e6,
//                              scripts/ms_marco_eval.py:224-227)
//   * ccr_search_sparse_prior  low-rank score + sparse prior (bbpr.transform(D) + D.prior_score,
//                              src/ccrec/models/bbpr.py:592-595)
// Both use the same routing: a query whose special cells are few (k + len <= min(n_rows, MAX_K)) over-fetches
// k + len rows through the fused MFMA search and merges; a query with a longer list takes the exact dense path
// with the special columns overwritten in the score row before the selection.
#include <string.h>

#include <algorithm>
#include <vector>

#include "ccr_index.h"
#include "ccr_topk_device.h"

extern "C" int ccr_search(ccr_index *ix, const uint16_t *Q_bf16, int n_q, int k, float *out_scores, int64_t *out_ids,
                          void *workspace, size_t ws_bytes, int flags, void *stream);
extern "C" size_t ccr_search_workspace_bytes(const ccr_index *ix, int n_q, int k);

namespace ccr {

constexpr int MAX_PRIOR_PER_ROW = 4096;

__device__ __forceinline__ unsigned long long f64_orderable(double d) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

// scratch[qi][id - id_lo] = value for every listed id (inside the shard) of query qlist[qi].  grid = nq_chunk.
__global__ __launch_bounds__(256) void mask_rows_kernel(float *__restrict__ scratch, int64_t n_rows,
                                                       const uint32_t *__restrict__ qlist, const int64_t *__restrict__ ptr,
                                                       const int64_t *__restrict__ idx, int64_t id_lo, float value) {
    const int q = (int)qlist[blockIdx.x];
    float *row = scratch + (int64_t)blockIdx.x * n_rows;
    for (int64_t e = ptr[q] + threadIdx.x; e < ptr[q + 1]; e += blockDim.x) {
        const int64_t j = idx[e] - id_lo;
        if (j >= 0 && j < n_rows) row[j] = value;
    }
}

// block_dict post-filter on an over-fetched canonical list (the kernel behind ccr_apply_block and the short route of
// ccr_search_blocked).  Queries with k_out + len > kf belong to the dense route and are skipped.  Blocked ids outside
// [id_lo, id_hi) (other shards) are never emitted.  grid = n_q, block = 256.
__global__ __launch_bounds__(256) void apply_block_kernel(const float *__restrict__ in_scores, const int64_t *__restrict__ in_ids,
                                                         int k_in, const int64_t *__restrict__ block_ptr,
                                                         const int64_t *__restrict__ block_idx, int64_t id_lo, int64_t id_hi,
                                                         float *__restrict__ out_scores, int64_t *__restrict__ out_ids, int k_out,
                                                         int kf) {
    __shared__ int s_wave[4];
    __shared__ int s_base;
    const int q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t b0 = block_ptr[q], b1 = block_ptr[q + 1];
    if ((int64_t)k_out + (b1 - b0) > (int64_t)kf) return;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < k_in; c0 += blockDim.x) {
        const int i = c0 + tid;
        bool keep = false;
        float s = 0.f;
        int64_t id = 0;
        if (i < k_in) {
            s = in_scores[(int64_t)q * k_in + i];
            id = in_ids[(int64_t)q * k_in + i];
            int64_t lo = b0, hi = b1;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (block_idx[mid] < id)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            keep = !(lo < b1 && block_idx[lo] == id);
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_wave[wv] = __popcll(m);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wv; ++w) off += s_wave[w];
        const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
        if (keep && pos < k_out) {
            out_scores[(int64_t)q * k_out + pos] = s;
            out_ids[(int64_t)q * k_out + pos] = id;
        }
        __syncthreads();
        if (tid == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
        if (s_base >= k_out) return;
    }
    // fewer unblocked rows than k_out: the blocked ids of this shard follow, ascending, at -1e6 (ms_marco_eval.py:227)
    for (int64_t c0 = b0; c0 < b1; c0 += blockDim.x) {
        const int64_t e = c0 + tid;
        int64_t id = 0;
        bool keep = false;
        if (e < b1) {
            id = block_idx[e];
            keep = id >= id_lo && id < id_hi;
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_wave[wv] = __popcll(m);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wv; ++w) off += s_wave[w];
        const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
        if (keep && pos < k_out) {
            out_scores[(int64_t)q * k_out + pos] = -1e6f;
            out_ids[(int64_t)q * k_out + pos] = id;
        }
        __syncthreads();
        if (tid == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
        if (s_base >= k_out) return;
    }
}

// Lp[e] = canonical score of (query of entry e, column idx[e]) for every prior entry inside the shard (0 outside).
__global__ __launch_bounds__(256) void prior_rescore_kernel(const int64_t *__restrict__ ptr, int n_q, const int64_t *__restrict__ idx,
                                                           int64_t nnz, const uint16_t *__restrict__ Q,
                                                           const uint16_t *__restrict__ D, int dim, int64_t id_lo,
                                                           int64_t n_rows, float *__restrict__ Lp) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    int lo = 0, hi = n_q;   // last q with ptr[q] <= e
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (ptr[mid] <= e)
            lo = mid;
        else
            hi = mid - 1;
    }
    const int64_t j = idx[e] - id_lo;
    Lp[e] = (j >= 0 && j < n_rows) ? canonical_dot(Q + (int64_t)lo * dim, D + j * dim, dim) : 0.f;
}

// Merge of (A) a canonical list of the low-rank score with the prior columns removed and (B) the prior columns with
// final = (double) L + prior, by (final desc, id asc).  grid = queries of this route, block = 256.
// dyn LDS: [k_in int64 ids][k_in fp32 scores][mp2_max u64 keys][mp2_max u32 positions]
__global__ __launch_bounds__(256) void apply_prior_kernel(const float *__restrict__ in_scores, const int64_t *__restrict__ in_ids,
                                                         int k_in, const uint32_t *__restrict__ qmap,
                                                         const int64_t *__restrict__ ptr, const int64_t *__restrict__ idx,
                                                         const double *__restrict__ val, const float *__restrict__ Lp,
                                                         int64_t id_lo, int64_t id_hi, double *__restrict__ out_scores,
                                                         int64_t *__restrict__ out_ids, int k_out, int kf, int mp2_max) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    int64_t *iA = reinterpret_cast<int64_t *>(sm);
    float *sA = reinterpret_cast<float *>(sm + (size_t)k_in * 8);
    unsigned long long *keyB = reinterpret_cast<unsigned long long *>(sm + (((size_t)k_in * 12 + 15) & ~(size_t)15));
    uint32_t *posB = reinterpret_cast<uint32_t *>(keyB + mp2_max);
    __shared__ int s_wave[4];
    __shared__ int s_base;
    __shared__ int s_mb;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.x;
    const int q = qmap ? (int)qmap[b] : b;
    const int64_t b0 = ptr[q], b1 = ptr[q + 1];
    const int m = (int)(b1 - b0);
    if (!qmap && (int64_t)k_out + m > (int64_t)kf) return;   // the dense route's query
    if (tid == 0) {
        s_base = 0;
        s_mb = 0;
    }
    __syncthreads();
    // (A) drop the prior columns from the in-list; only the first k_out survivors can reach the output
    for (int c0 = 0; c0 < k_in; c0 += blockDim.x) {
        const int i = c0 + tid;
        bool keep = false;
        float s = 0.f;
        int64_t id = 0;
        if (i < k_in) {
            s = in_scores[(int64_t)b * k_in + i];
            id = in_ids[(int64_t)b * k_in + i];
            int64_t lo = b0, hi = b1;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (idx[mid] < id)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            keep = !(lo < b1 && idx[lo] == id);
        }
        const unsigned long long mk = __ballot(keep);
        if (lane == 0) s_wave[wv] = __popcll(mk);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wv; ++w) off += s_wave[w];
        const int pos = off + __popcll(mk & ((1ull << lane) - 1ull));
        if (keep && pos < k_out) {
            sA[pos] = s;
            iA[pos] = id;
        }
        __syncthreads();
        if (tid == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
        if (s_base >= k_out) break;
    }
    const int nA = s_base < k_out ? s_base : k_out;
    // (B) finals of the prior columns of this shard, sorted by (final desc, position asc = id asc)
    int mp2 = 1;
    while (mp2 < m) mp2 <<= 1;
    int mine = 0;
    for (int e = tid; e < mp2; e += blockDim.x) {
        unsigned long long key = 0ull;
        uint32_t pos = 0xffffffffu;
        if (e < m) {
            const int64_t id = idx[b0 + e];
            if (id >= id_lo && id < id_hi) {
                const double f = ((double)Lp[b0 + e] + val[b0 + e]) + 0.0;   // + 0.0: -0.0 ranks as +0.0
                key = f64_orderable(f);
                pos = (uint32_t)e;
                ++mine;
            }
        }
        keyB[e] = key;
        posB[e] = pos;
    }
    if (mine) atomicAdd(&s_mb, mine);
    for (int size = 2; size <= mp2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = tid; i < (mp2 >> 1); i += blockDim.x) {
                const int p0 = 2 * i - (i & (stride - 1)), p1 = p0 + stride;
                const unsigned long long ka = keyB[p0], kb = keyB[p1];
                const uint32_t pa = posB[p0], pb = posB[p1];
                const bool a_first = (ka > kb) || (ka == kb && pa < pb);   // a precedes b in the wanted order
                const bool desc = ((p0 & size) == 0);
                if (a_first != desc) {
                    keyB[p0] = kb;
                    keyB[p1] = ka;
                    posB[p0] = pb;
                    posB[p1] = pa;
                }
            }
        }
    }
    __syncthreads();
    const int mB = s_mb;   // the entries of other shards (key 0, position 0xffffffff) sorted behind the valid ones
    // ranks by counting: both lists are sorted by the same strict order
    for (int p = tid; p < nA; p += blockDim.x) {
        const unsigned long long ka = f64_orderable((double)sA[p]);
        const int64_t ida = iA[p];
        int lo = 0, hi = mB;   // number of B elements that precede A[p]
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const unsigned long long kb = keyB[mid];
            if (kb > ka || (kb == ka && idx[b0 + posB[mid]] < ida))
                lo = mid + 1;
            else
                hi = mid;
        }
        const int rank = p + lo;
        if (rank < k_out) {
            out_scores[(int64_t)q * k_out + rank] = (double)sA[p];
            out_ids[(int64_t)q * k_out + rank] = ida;
        }
    }
    for (int t = tid; t < mB; t += blockDim.x) {
        const unsigned long long kb = keyB[t];
        const int64_t e = b0 + posB[t];
        const int64_t idb = idx[e];
        int lo = 0, hi = nA;   // number of A elements that precede B[t]
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const unsigned long long ka = f64_orderable((double)sA[mid]);
            if (ka > kb || (ka == kb && iA[mid] < idb))
                lo = mid + 1;
            else
                hi = mid;
        }
        const int rank = t + lo;
        if (rank < k_out) {
            out_scores[(int64_t)q * k_out + rank] = ((double)Lp[e] + val[e]) + 0.0;
            out_ids[(int64_t)q * k_out + rank] = idb;
        }
    }
}

// out[c] = sum over rows of X[r][c] in fp64.  One workgroup per 64 columns (8 chunks of 16 bytes = one 128-byte line of
// each row): 32 row lanes x 8 chunk lanes; the row lanes are combined in a fixed order (deterministic).
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const uint16_t *__restrict__ X, int64_t rows, int dim,
                                                         double *__restrict__ out) {
    __shared__ double s_part[32][8][8];
    const int tid = threadIdx.x;
    const int cl = tid & 7, rl = tid >> 3;
    const int c0 = (blockIdx.x * 8 + cl) * 8;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c0 < dim) {
        for (int64_t r = rl; r < rows; r += 32) {
            const uint4 v = *reinterpret_cast<const uint4 *>(X + r * dim + c0);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[2 * e] += (double)__uint_as_float(w[e] << 16);
                acc[2 * e + 1] += (double)__uint_as_float(w[e] & 0xffff0000u);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) s_part[rl][cl][e] = acc[e];
    __syncthreads();
    if (tid < 64) {
        const int c = tid >> 3, e = tid & 7;
        double t = 0.0;
        for (int r = 0; r < 32; ++r) t += s_part[r][c][e];
        const int col = (blockIdx.x * 8 + c) * 8 + e;
        if (col < dim) out[col] = t;
    }
}

// ---------------------------------------------------------------------------------------------- routing
struct SpecialPlan {
    int kf;          // longest list the fused route can fetch: min(n_rows, MAX_K)
    int k_in;        // rows fetched by the fused route (0: no query takes it)
    int max_len;     // longest special list of any query
    std::vector<uint32_t> long_list;   // queries of the dense route
    size_t search_bytes, off_ls, off_li, off_ptr, off_qlist, off_dense, off_lp, off_ts, off_ti, total;
};

static int plan_special(const ccr_index *ix, int n_q, int k, const int64_t *ptr_host, bool prior, SpecialPlan &sp) {
    sp.kf = (int)std::min<int64_t>(ix->n_rows, MAX_K);
    int max_short = 0;
    sp.max_len = 0;
    sp.long_list.clear();
    for (int q = 0; q < n_q; ++q) {
        const int64_t len = ptr_host[q + 1] - ptr_host[q];
        CCR_REQUIRE(len >= 0, "special search: the row pointer decreases at query %d", q);
        CCR_REQUIRE(!prior || len <= MAX_PRIOR_PER_ROW, "ccr_search_sparse_prior: %lld prior entries in row %d (at most %d)",
                    (long long)len, q, MAX_PRIOR_PER_ROW);
        CCR_REQUIRE(len <= ((int64_t)1 << 31) - 1 - k, "special search: list of query %d too long", q);
        sp.max_len = std::max<int>(sp.max_len, (int)len);
        if ((int64_t)k + len <= sp.kf)
            max_short = std::max<int>(max_short, (int)len);
        else
            sp.long_list.push_back((uint32_t)q);
    }
    const int n_long = (int)sp.long_list.size();
    sp.k_in = n_long < n_q ? k + max_short : 0;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off = (off + bytes + 255) / 256 * 256;
        return o;
    };
    sp.search_bytes = sp.k_in ? ccr_search_workspace_bytes(ix, n_q, sp.k_in) : 0;
    take(sp.search_bytes);
    sp.off_ls = take((size_t)n_q * sp.k_in * 4);
    sp.off_li = take((size_t)n_q * sp.k_in * 8);
    sp.off_ptr = take((size_t)(n_q + 1) * 8);
    sp.off_qlist = take((size_t)std::max(1, n_long) * 4);
    sp.off_dense = take(n_long ? (size_t)FALLBACK_ROWS * ix->n_rows * 4 : 0);
    sp.off_lp = take(prior ? (size_t)std::max<int64_t>(1, ptr_host[n_q]) * 4 : 0);
    sp.off_ts = take(prior ? (size_t)n_long * k * 4 : 0);
    sp.off_ti = take(prior ? (size_t)n_long * k * 8 : 0);
    sp.total = off + 256;
    return CCR_OK;
}

static int check_special_args(const ccr_index *ix, const uint16_t *Q, int n_q, int k, const int64_t *ptr_host, const void *idx,
                              const void *out_s, const void *out_i, const char *what) {
    CCR_REQUIRE(ix && Q && ptr_host && out_s && out_i, "%s: null pointer", what);
    CCR_REQUIRE(n_q >= 0, "%s: n_q=%d", what, n_q);
    CCR_REQUIRE(k >= 1 && k <= MAX_K && (int64_t)k <= ix->n_rows, "%s: k=%d must be in [1, min(n_rows=%lld, %d)]", what, k,
                (long long)ix->n_rows, MAX_K);
    CCR_REQUIRE(ptr_host[0] == 0, "%s: row pointer must start at 0", what);
    CCR_REQUIRE(idx || ptr_host[n_q] == 0, "%s: null index array", what);
    return CCR_OK;
}

}  // namespace ccr

using namespace ccr;

extern "C" size_t ccr_search_blocked_workspace_bytes(const ccr_index *ix, int n_q, int k, const int64_t *block_ptr_host) {
    if (!ix || n_q <= 0 || k <= 0 || !block_ptr_host) return 0;
    SpecialPlan sp;
    if (plan_special(ix, n_q, k, block_ptr_host, false, sp) != CCR_OK) return 0;
    return sp.total;
}

extern "C" int ccr_search_blocked(ccr_index *ix, const uint16_t *Q_bf16, int n_q, int k, const int64_t *block_ptr_host,
                                  const int64_t *block_idx, float *out_scores, int64_t *out_ids, void *workspace,
                                  size_t ws_bytes, int flags, void *stream) {
    int rc = check_special_args(ix, Q_bf16, n_q, k, block_ptr_host, block_idx, out_scores, out_ids, "ccr_search_blocked");
    if (rc != CCR_OK) return rc;
    if (n_q == 0) return CCR_OK;
    flags &= ~CCR_SEARCH_ASYNC;
    SpecialPlan sp;
    rc = plan_special(ix, n_q, k, block_ptr_host, false, sp);
    if (rc != CCR_OK) return rc;
    if (!workspace || ws_bytes < sp.total || (uintptr_t)workspace % 256 != 0) {
        set_error("ccr_search_blocked: workspace %zu bytes (256-byte aligned) required, got %zu at %p", sp.total, ws_bytes, workspace);
        return CCR_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    if (sp.max_len == 0)   // nothing blocked anywhere
        return ccr_search(ix, Q_bf16, n_q, k, out_scores, out_ids, ws, sp.search_bytes, flags, stream);
    int64_t *d_ptr = (int64_t *)(ws + sp.off_ptr);
    CCR_HIP_CHECK(hipMemcpyAsync(d_ptr, block_ptr_host, (size_t)(n_q + 1) * 8, hipMemcpyHostToDevice, s));
    const int64_t id_lo = ix->offset, id_hi = ix->offset + ix->n_rows;
    const int n_long = (int)sp.long_list.size();
    if (sp.k_in) {   // fused route: over-fetch, drop the blocked ids
        float *ls = (float *)(ws + sp.off_ls);
        int64_t *li = (int64_t *)(ws + sp.off_li);
        rc = ccr_search(ix, Q_bf16, n_q, sp.k_in, ls, li, ws, sp.search_bytes, flags, stream);
        if (rc != CCR_OK) return rc;
        hipLaunchKernelGGL(apply_block_kernel, dim3(n_q), dim3(256), 0, s, ls, li, sp.k_in, d_ptr, block_idx, id_lo, id_hi,
                           out_scores, out_ids, k, sp.kf);
        CCR_LAUNCH_CHECK();
    }
    if (n_long) {   // dense route: blocked columns become -1e6 in the score row, then the exact selection
        uint32_t *d_ql = (uint32_t *)(ws + sp.off_qlist);
        float *scratch = (float *)(ws + sp.off_dense);
        CCR_HIP_CHECK(hipMemcpyAsync(d_ql, sp.long_list.data(), (size_t)n_long * 4, hipMemcpyHostToDevice, s));
        for (int lo = 0; lo < n_long; lo += FALLBACK_ROWS) {
            const int mq = std::min(FALLBACK_ROWS, n_long - lo);
            rc = launch_dense_scores(ix->D, ix->n_rows, ix->dim, Q_bf16, d_ql + lo, 0, mq, nullptr, scratch, s);
            if (rc != CCR_OK) return rc;
            hipLaunchKernelGGL(mask_rows_kernel, dim3(mq), dim3(256), 0, s, scratch, ix->n_rows, d_ql + lo, d_ptr, block_idx, id_lo,
                               -1e6f);
            CCR_LAUNCH_CHECK();
            rc = launch_dense_select(scratch, ix->n_rows, k, d_ql + lo, 0, mq, nullptr, ix->offset, out_scores, out_ids, s);
            if (rc != CCR_OK) return rc;
        }
        CCR_HIP_CHECK(hipStreamSynchronize(s));   // sp.long_list (host) was the source of an asynchronous copy
    }
    return CCR_OK;
}

extern "C" size_t ccr_search_sparse_prior_workspace_bytes(const ccr_index *ix, int n_q, int k, const int64_t *prior_ptr_host) {
    if (!ix || n_q <= 0 || k <= 0 || !prior_ptr_host) return 0;
    SpecialPlan sp;
    if (plan_special(ix, n_q, k, prior_ptr_host, true, sp) != CCR_OK) return 0;
    return sp.total;
}

extern "C" int ccr_search_sparse_prior(ccr_index *ix, const uint16_t *Q_bf16, int n_q, int k, const int64_t *prior_ptr_host,
                                       const int64_t *prior_idx, const double *prior_val, double *out_scores, int64_t *out_ids,
                                       void *workspace, size_t ws_bytes, int flags, void *stream) {
    int rc = check_special_args(ix, Q_bf16, n_q, k, prior_ptr_host, prior_idx, out_scores, out_ids, "ccr_search_sparse_prior");
    if (rc != CCR_OK) return rc;
    CCR_REQUIRE(prior_val || prior_ptr_host[n_q] == 0, "ccr_search_sparse_prior: null value array");
    if (n_q == 0) return CCR_OK;
    flags &= ~CCR_SEARCH_ASYNC;
    SpecialPlan sp;
    rc = plan_special(ix, n_q, k, prior_ptr_host, true, sp);
    if (rc != CCR_OK) return rc;
    if (!workspace || ws_bytes < sp.total || (uintptr_t)workspace % 256 != 0) {
        set_error("ccr_search_sparse_prior: workspace %zu bytes (256-byte aligned) required, got %zu at %p", sp.total, ws_bytes,
                  workspace);
        return CCR_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    int64_t *d_ptr = (int64_t *)(ws + sp.off_ptr);
    float *Lp = (float *)(ws + sp.off_lp);
    const int64_t nnz = prior_ptr_host[n_q];
    CCR_HIP_CHECK(hipMemcpyAsync(d_ptr, prior_ptr_host, (size_t)(n_q + 1) * 8, hipMemcpyHostToDevice, s));
    const int64_t id_lo = ix->offset, id_hi = ix->offset + ix->n_rows;
    if (nnz > 0) {
        hipLaunchKernelGGL(prior_rescore_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, s, d_ptr, n_q, prior_idx, nnz,
                           Q_bf16, ix->D, ix->dim, id_lo, ix->n_rows, Lp);
        CCR_LAUNCH_CHECK();
    }
    int mp2_max = 1;
    while (mp2_max < sp.max_len) mp2_max <<= 1;
    auto merge = [&](const float *ls, const int64_t *li, int k_in, const uint32_t *qmap, int n_blocks) -> int {
        const size_t lds = (((size_t)k_in * 12 + 15) & ~(size_t)15) + (size_t)mp2_max * 12;
        if (lds > 48 * 1024) {   // 4 096-entry lists: 96 KiB; the opt-in is set once per device
            const int rc_lds = ensure_dynamic_lds(reinterpret_cast<const void *>(&apply_prior_kernel), 100 * 1024);
            if (rc_lds != CCR_OK) return rc_lds;
        }
        hipLaunchKernelGGL(apply_prior_kernel, dim3(n_blocks), dim3(256), lds, s, ls, li, k_in, qmap, d_ptr, prior_idx, prior_val, Lp,
                           id_lo, id_hi, out_scores, out_ids, k, sp.kf, mp2_max);
        CCR_LAUNCH_CHECK();
        return CCR_OK;
    };
    const int n_long = (int)sp.long_list.size();
    if (sp.k_in) {
        float *ls = (float *)(ws + sp.off_ls);
        int64_t *li = (int64_t *)(ws + sp.off_li);
        rc = ccr_search(ix, Q_bf16, n_q, sp.k_in, ls, li, ws, sp.search_bytes, flags, stream);
        if (rc != CCR_OK) return rc;
        rc = merge(ls, li, sp.k_in, nullptr, n_q);
        if (rc != CCR_OK) return rc;
    }
    if (n_long) {   // dense route: the prior columns leave the score row (-inf), the exact top-k of the rest is merged with them
        uint32_t *d_ql = (uint32_t *)(ws + sp.off_qlist);
        float *scratch = (float *)(ws + sp.off_dense);
        float *ts = (float *)(ws + sp.off_ts);
        int64_t *ti = (int64_t *)(ws + sp.off_ti);
        CCR_HIP_CHECK(hipMemcpyAsync(d_ql, sp.long_list.data(), (size_t)n_long * 4, hipMemcpyHostToDevice, s));
        for (int lo = 0; lo < n_long; lo += FALLBACK_ROWS) {
            const int mq = std::min(FALLBACK_ROWS, n_long - lo);
            rc = launch_dense_scores(ix->D, ix->n_rows, ix->dim, Q_bf16, d_ql + lo, 0, mq, nullptr, scratch, s);
            if (rc != CCR_OK) return rc;
            hipLaunchKernelGGL(mask_rows_kernel, dim3(mq), dim3(256), 0, s, scratch, ix->n_rows, d_ql + lo, d_ptr, prior_idx, id_lo,
                               -INFINITY);
            CCR_LAUNCH_CHECK();
            rc = launch_dense_select(scratch, ix->n_rows, k, nullptr, lo, mq, nullptr, ix->offset, ts, ti, s);
            if (rc != CCR_OK) return rc;
        }
        rc = merge(ts, ti, k, d_ql, n_long);
        if (rc != CCR_OK) return rc;
    }
    CCR_HIP_CHECK(hipStreamSynchronize(s));   // host-side sources of the asynchronous copies go out of scope
    return CCR_OK;
}

extern "C" int ccr_apply_block(const float *in_scores, const int64_t *in_ids, int n_q, int k_in, const int64_t *block_ptr,
                               const int64_t *block_idx, int64_t n_rows_total, float *out_scores, int64_t *out_ids, int k_out,
                               void *stream) {
    CCR_REQUIRE(in_scores && in_ids && block_ptr && out_scores && out_ids, "ccr_apply_block: null pointer");
    CCR_REQUIRE(n_q >= 0 && k_in >= 1 && k_out >= 1 && k_out <= k_in, "ccr_apply_block: bad shape n_q=%d k_in=%d k_out=%d", n_q,
                k_in, k_out);
    (void)n_rows_total;
    if (n_q == 0) return CCR_OK;
    hipLaunchKernelGGL(apply_block_kernel, dim3(n_q), dim3(256), 0, (hipStream_t)stream, in_scores, in_ids, k_in, block_ptr,
                       block_idx, (int64_t)INT64_MIN, (int64_t)INT64_MAX, out_scores, out_ids, k_out, INT32_MAX);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

extern "C" int ccr_colsum_bf16(const uint16_t *X, int64_t rows, int dim, double *out, void *stream) {
    CCR_REQUIRE(X && out, "ccr_colsum_bf16: null pointer");
    CCR_REQUIRE(rows >= 0 && dim >= 8 && dim % 8 == 0, "ccr_colsum_bf16: bad shape rows=%lld dim=%d (dim %% 8 == 0)", (long long)rows, dim);
    CCR_REQUIRE((uintptr_t)X % 16 == 0, "ccr_colsum_bf16: pointer must be 16-byte aligned");
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3((unsigned)((dim / 8 + 7) / 8)), dim3(256), 0, (hipStream_t)stream, X, rows, dim, out);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}
