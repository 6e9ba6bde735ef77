// ccr_metrics.hip -- retrieval quality from the [Q, k] id tensor without building python dicts.
// Replaces: EvaluateRetrieval.evaluate_custom(qrels, ranking_profile, [1,5,10,100], metric="mrr") as called at
// scripts/al_0_rank.py:130-133 (BEIR's mrr: per query, reciprocal rank of the first relevant hit within k,
// summed over queries, divided by the number of queries, rounded to 5 decimals by the caller) plus Recall@k.
#include "ccr_common.h"

namespace ccr {

// One thread per query.  qrel_idx[qrel_ptr[q] .. qrel_ptr[q+1]) = relevant ids of query q, ascending.
// out_rr[q][j] = 1 / rank of the first relevant id within k_values[j] (0 if none);
// out_hits[q][j] = number of relevant ids within the first k_values[j] ranks.
__global__ __launch_bounds__(256) void rank_metrics_kernel(const int64_t *__restrict__ ids, int n_q, int k,
                                                          const int64_t *__restrict__ qrel_ptr,
                                                          const int64_t *__restrict__ qrel_idx,
                                                          const int32_t *__restrict__ k_values, int nk,
                                                          float *__restrict__ out_rr, int32_t *__restrict__ out_hits) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_q) return;
    const int64_t b0 = qrel_ptr[q], b1 = qrel_ptr[q + 1];
    int first = -1;
    for (int j = 0; j < nk; ++j) {
        out_rr[(int64_t)q * nk + j] = 0.f;
        out_hits[(int64_t)q * nk + j] = 0;
    }
    for (int r = 0; r < k; ++r) {
        const int64_t id = ids[(int64_t)q * k + r];
        int64_t lo = b0, hi = b1;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (qrel_idx[mid] < id)
                lo = mid + 1;
            else
                hi = mid;
        }
        if (lo < b1 && qrel_idx[lo] == id) {
            if (first < 0) first = r;
            for (int j = 0; j < nk; ++j)
                if (r < k_values[j]) out_hits[(int64_t)q * nk + j] += 1;
        }
    }
    if (first >= 0)
        for (int j = 0; j < nk; ++j)
            if (first < k_values[j]) out_rr[(int64_t)q * nk + j] = 1.f / (float)(first + 1);
}

}  // namespace ccr

using namespace ccr;

extern "C" int ccr_rank_metrics(const int64_t *ids, int n_q, int k, const int64_t *qrel_ptr, const int64_t *qrel_idx,
                                const int32_t *k_values, int n_k, float *out_rr, int32_t *out_hits, void *stream) {
    CCR_REQUIRE(ids && qrel_ptr && qrel_idx && k_values && out_rr && out_hits, "ccr_rank_metrics: null pointer");
    CCR_REQUIRE(n_q >= 0 && k >= 1 && n_k >= 1 && n_k <= 16, "ccr_rank_metrics: bad shape n_q=%d k=%d n_k=%d", n_q, k, n_k);
    if (n_q == 0) return CCR_OK;
    hipLaunchKernelGGL(rank_metrics_kernel, dim3((n_q + 255) / 256), dim3(256), 0, (hipStream_t)stream, ids, n_q, k, qrel_ptr,
                       qrel_idx, k_values, n_k, out_rr, out_hits);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}
