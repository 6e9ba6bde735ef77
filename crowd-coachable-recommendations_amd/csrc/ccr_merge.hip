// ccr_merge.hip -- shard merge after the RCCL all-gather (the block_dict post-filter lives in ccr_special.hip).
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
