// ccr_merge.hip -- shard merge after the RCCL all-gather, and the block_dict post-filter.
#include "ccr_common.h"

namespace ccr {

__device__ __forceinline__ bool precedes(float sa, int64_t ia, float sb, int64_t ib) {
    return (sa > sb) || (sa == sb && ia < ib);
}

// Rank-by-counting merge: element p of list r has global rank p + sum over the other lists of the
// number of their elements that precede it (binary search; every list is already in canonical
// order and ids are unique across shards, so the order is strict and ranks are a permutation).
// grid = (ceil(R*k/256), n_q), block = 256.
__global__ __launch_bounds__(256) void merge_topk_kernel(const float *__restrict__ scores, const int64_t *__restrict__ ids,
                                                        int64_t rs_scores, int64_t rs_ids, int R, int n_q, int k,
                                                        float *__restrict__ out_scores, int64_t *__restrict__ out_ids) {
    // list (r, q) starts at scores[r * rs_scores + q * k] / ids[r * rs_ids + q * k] (rank strides in elements)
    const int q = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= R * k || q >= n_q) return;
    const int r = e / k, p = e - r * k;
    const float s = scores[r * rs_scores + (int64_t)q * k + p];
    const int64_t id = ids[r * rs_ids + (int64_t)q * k + p];
    int rank = p;
    for (int o = 0; o < R && rank < k; ++o) {
        if (o == r) continue;
        const float *os = scores + o * rs_scores + (int64_t)q * k;
        const int64_t *oi = ids + o * rs_ids + (int64_t)q * k;
        int lo = 0, hi = k;  // first position whose element does NOT precede (s, id)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (precedes(os[mid], oi[mid], s, id))
                lo = mid + 1;
            else
                hi = mid;
        }
        rank += lo;
    }
    if (rank < k) {
        out_scores[(int64_t)q * k + rank] = s;
        out_ids[(int64_t)q * k + rank] = id;
    }
}

// Same merge with the R lists of a query staged in LDS first (coalesced loads, then every binary search probes LDS):
// one workgroup per query.  The global-memory version above pays ~R * log2(k) dependent L2 round trips per element
// (199 us at R = 8, n_q = 3 452, k = 100).  dyn LDS = R * k * 12 bytes; used while that fits.
__global__ __launch_bounds__(1024) void merge_topk_lds_kernel(const float *__restrict__ scores, const int64_t *__restrict__ ids,
                                                            int64_t rs_scores, int64_t rs_ids, int R, int k,
                                                            float *__restrict__ out_scores, int64_t *__restrict__ out_ids) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int n = R * k;
    int64_t *s_id = reinterpret_cast<int64_t *>(sm);
    float *s_sc = reinterpret_cast<float *>(sm + (size_t)n * 8);
    const int q = blockIdx.x;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int r = e / k, p = e - r * k;
        s_sc[e] = scores[r * rs_scores + (int64_t)q * k + p];
        s_id[e] = ids[r * rs_ids + (int64_t)q * k + p];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int r = e / k, p = e - r * k;
        const float s = s_sc[e];
        const int64_t id = s_id[e];
        int rank = p;
        for (int o = 0; o < R && rank < k; ++o) {
            if (o == r) continue;
            const float *os = s_sc + o * k;
            const int64_t *oi = s_id + o * k;
            int lo = 0, hi = k;  // first position whose element does NOT precede (s, id)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const float so = os[mid];
                if (so > s || (so == s && oi[mid] < id))   // the id is only read on an exact score tie
                    lo = mid + 1;
                else
                    hi = mid;
            }
            rank += lo;
        }
        if (rank < k) {
            out_scores[(int64_t)q * k + rank] = s;
            out_ids[(int64_t)q * k + rank] = id;
        }
    }
}

// block_dict post-filter on an over-fetched canonical list.  grid = n_q, block = 256.
__global__ __launch_bounds__(256) void apply_block_kernel(const float *__restrict__ in_scores, const int64_t *__restrict__ in_ids,
                                                         int k_in, const int64_t *__restrict__ block_ptr,
                                                         const int64_t *__restrict__ block_idx, float *__restrict__ out_scores,
                                                         int64_t *__restrict__ out_ids, int k_out) {
    __shared__ int s_wave[4];
    __shared__ int s_base;
    const int q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t b0 = block_ptr[q], b1 = block_ptr[q + 1];
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
    // fewer unblocked rows than k_out: the blocked ids follow, ascending, at -1e6 (ms_marco_eval.py:227)
    const int have = s_base;
    for (int64_t j = tid; j < b1 - b0 && have + j < k_out; j += blockDim.x) {
        out_scores[(int64_t)q * k_out + have + j] = -1e6f;
        out_ids[(int64_t)q * k_out + have + j] = block_idx[b0 + j];
    }
}

}  // namespace ccr

using namespace ccr;

extern "C" int ccr_merge_topk_strided(const float *scores, const int64_t *ids, int64_t score_rank_stride, int64_t id_rank_stride,
                                      int R, int n_q, int k, float *out_scores, int64_t *out_ids, void *stream) {
    CCR_REQUIRE(scores && ids && out_scores && out_ids, "ccr_merge_topk: null pointer");
    CCR_REQUIRE(R >= 1 && n_q >= 0 && k >= 1 && k <= MAX_K, "ccr_merge_topk: bad shape R=%d n_q=%d k=%d", R, n_q, k);
    CCR_REQUIRE(score_rank_stride >= (int64_t)n_q * k && id_rank_stride >= (int64_t)n_q * k,
                "ccr_merge_topk: rank strides %lld / %lld shorter than one [n_q, k] list", (long long)score_rank_stride,
                (long long)id_rank_stride);
    if (n_q == 0) return CCR_OK;
    const size_t lds = (size_t)R * k * 12;
    if (lds <= 96 * 1024) {
        if (lds > 48 * 1024)
            CCR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&merge_topk_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int threads = R * k >= 2048 ? 1024 : 256;
        hipLaunchKernelGGL(merge_topk_lds_kernel, dim3((unsigned)n_q), dim3(threads), lds, (hipStream_t)stream, scores, ids, score_rank_stride,
                           id_rank_stride, R, k, out_scores, out_ids);
        CCR_LAUNCH_CHECK();
        return CCR_OK;
    }
    dim3 grid((unsigned)((R * k + 255) / 256), (unsigned)n_q);
    hipLaunchKernelGGL(merge_topk_kernel, grid, dim3(256), 0, (hipStream_t)stream, scores, ids, score_rank_stride, id_rank_stride, R,
                       n_q, k, out_scores, out_ids);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

extern "C" int ccr_merge_topk(const float *scores, const int64_t *ids, int R, int n_q, int k, float *out_scores,
                              int64_t *out_ids, void *stream) {
    return ccr_merge_topk_strided(scores, ids, (int64_t)n_q * k, (int64_t)n_q * k, R, n_q, k, out_scores, out_ids, stream);
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
                       block_idx, out_scores, out_ids, k_out);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}
