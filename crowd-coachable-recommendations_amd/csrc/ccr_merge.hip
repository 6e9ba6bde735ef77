// ccr_merge.hip -- shard merge after the RCCL all-gather (the block_dict post-filter lives in ccr_special.hip).
#include <string.h>

#include "ccr_index.h"

namespace ccr {

__device__ __forceinline__ bool precedes(float sa, int64_t ia, float sb, int64_t ib) {
    return (sa > sb) || (sa == sb && ia < ib);
}

// ---- merge of R sorted lists staged in LDS (s_sc / s_id: [R][k]), keeping the k best.
// Only k of the R k staged elements end up in the result -- on average k / R per list -- so the ranks of the others are never
// computed.  Phase 1: per list r the CUT c_r = number of its elements whose global rank is < k, by bisection on the position
// (the rank p + sum over the other lists of the elements that precede (r, p) is monotone in p); the R (R - 1) binary searches of a
// bisection step are spread over the workgroup's threads.  Phase 2: element (r, p < c_r) goes to rank p + sum_o #{elements of list o BEFORE ITS CUT
// that precede it} -- an element behind its list's cut has rank >= k and precedes nothing that is kept.  Keys (score desc, id asc)
// are distinct (ids are unique across shards, padding slots carry distinct ids), so the ranks are a permutation and sum c_r = k.
// R = 8, 3 452 queries (tools/bench_merge.py): k = 100 66 -> 39 us, k = 1001 1 006 -> 442 us (the full rank-by-counting did
// R k (R - 1) log k LDS probes per query; it stays for R k < 600, where the bisection's block barriers cost more than they save).
__device__ __forceinline__ int lds_lower_bound(const float *os, const int64_t *oi, int n, float s, int64_t id) {
    int lo = 0, hi = n;   // first position whose element does NOT precede (s, id)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const float so = os[mid];
        if (so > s || (so == s && oi[mid] < id))   // the id is only read on an exact score tie
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// (lists of kl elements each, the ko best kept: kl == ko for the full-list exchange, kl < ko for the short-list exchange, whose caller
//  reads the cuts back from s_cut afterwards -- a list consumed to its end may have had more to give)
__device__ __forceinline__ void merge_lists_lds(const float *s_sc, const int64_t *s_id, int R, int kl, int ko, int *s_cut, int *s_lohi, int *s_cnt,
                                                int64_t out_base, float *__restrict__ out_scores, int64_t *__restrict__ out_ids) {
    const int tid = threadIdx.x;
    const int k = kl;   // list stride and bisection range
    // ---- phase 1: cuts (s_lohi[2r] / [2r+1] = the bisection interval of list r, s_cnt[r] = rank accumulator of its probe).
    // The whole workgroup takes part (uniform trip count, block barriers): the R (R - 1) searches of a step are spread over its threads.
    for (int r = tid; r < R; r += blockDim.x) {
        s_lohi[2 * r] = 0;
        s_lohi[2 * r + 1] = k;
        s_cnt[r] = 0;
    }
    __syncthreads();
    int steps = 1;
    while ((1 << steps) < k + 1) ++steps;
    for (int it = 0; it < steps; ++it) {
        for (int pr = tid; pr < R * R; pr += blockDim.x) {   // pair (r, o): how many elements of list o precede (r, mid_r)
            const int r = pr / R, o = pr - r * R;
            const int lo = s_lohi[2 * r], hi = s_lohi[2 * r + 1];
            if (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const int c = (o == r) ? mid : lds_lower_bound(s_sc + o * k, s_id + o * k, k, s_sc[r * k + mid], s_id[r * k + mid]);
                atomicAdd(&s_cnt[r], c);
            }
        }
        __syncthreads();
        for (int r = tid; r < R; r += blockDim.x) {
            const int lo = s_lohi[2 * r], hi = s_lohi[2 * r + 1];
            if (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s_cnt[r] < ko)
                    s_lohi[2 * r] = mid + 1;
                else
                    s_lohi[2 * r + 1] = mid;
            }
            s_cnt[r] = 0;
        }
        __syncthreads();
    }
    for (int r = tid; r < R; r += blockDim.x) s_cut[r] = s_lohi[2 * r];
    __syncthreads();
    // ---- phase 2: the kept elements (sum of the cuts = k) find their ranks among the kept prefixes
    for (int t = tid; t < ko; t += blockDim.x) {
        int r = 0, p = t;
        while (r < R - 1 && p >= s_cut[r]) {
            p -= s_cut[r];
            ++r;
        }
        if (p >= s_cut[r]) continue;   // (cannot happen: the cuts add up to k)
        const float s = s_sc[r * k + p];
        const int64_t id = s_id[r * k + p];
        int rank = p;
        for (int o = 0; o < R; ++o)
            if (o != r) rank += lds_lower_bound(s_sc + o * k, s_id + o * k, s_cut[o], s, id);
        if (rank < ko) {
            out_scores[out_base + rank] = s;
            out_ids[out_base + rank] = id;
        }
    }
}

// Few short lists (R k < 600: R = 2 or 4 at k = 100): the bisection's block barriers cost more than ranking all R k elements.
__device__ __forceinline__ void merge_lists_lds_all(const float *s_sc, const int64_t *s_id, int R, int k, int64_t out_base,
                                                    float *__restrict__ out_scores, int64_t *__restrict__ out_ids) {
    const int n = R * k;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int r = e / k, p = e - r * k;
        const float s = s_sc[e];
        const int64_t id = s_id[e];
        int rank = p;
        for (int o = 0; o < R && rank < k; ++o)
            if (o != r) rank += lds_lower_bound(s_sc + o * k, s_id + o * k, k, s, id);
        if (rank < k) {
            out_scores[out_base + rank] = s;
            out_ids[out_base + rank] = id;
        }
    }
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
    __shared__ int s_cut[64], s_lohi[128], s_cnt[64];
    if (R * k < 600)
        merge_lists_lds_all(s_sc, s_id, R, k, (int64_t)q * k, out_scores, out_ids);
    else
        merge_lists_lds(s_sc, s_id, R, k, k, s_cut, s_lohi, s_cnt, (int64_t)q * k, out_scores, out_ids);
}

}  // namespace ccr
namespace ccr {

// ---- packed shard messages (include/ccr_retrieval.h: header | scores [n_q][k] fp32 | rows [n_q][k] u32 local)
__host__ __device__ __forceinline__ size_t shard_rows_at(int n_q, int k) { return (sizeof(ccr_shard_header) + (size_t)n_q * k * 4 + 15) / 16 * 16; }

__global__ void shard_header_kernel(ccr_shard_header h, ccr_shard_header *dst) { *dst = h; }

int launch_shard_header(const ccr_shard_header &h, void *message, hipStream_t s) {
    hipLaunchKernelGGL(shard_header_kernel, dim3(1), dim3(1), 0, s, h, reinterpret_cast<ccr_shard_header *>(message));
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

// entry (r, q, p) of the gathered messages as (score, global id); padding slots are (-inf, a distinct id above every real one)
struct ShardSrc {
    const char *base;
    int64_t stride;
    size_t rows_at;
    int k;
};
__device__ __forceinline__ void shard_entry(const ShardSrc &m, int r, int q, int p, float &s, int64_t &id) {
    const char *msg = m.base + (int64_t)r * m.stride;
    const ccr_shard_header *h = reinterpret_cast<const ccr_shard_header *>(msg);
    if (p < (int)h->k_valid) {
        s = reinterpret_cast<const float *>(msg + sizeof(ccr_shard_header))[(int64_t)q * m.k + p];
        id = h->row_offset + (int64_t) reinterpret_cast<const uint32_t *>(msg + m.rows_at)[(int64_t)q * m.k + p];
    } else {
        s = -INFINITY;
        id = INT64_MAX - ((int64_t)r * m.k + p);
    }
}

// one workgroup per query, the R lists staged in LDS (R * k * 12 bytes), rank by counting as above
__global__ __launch_bounds__(1024) void merge_messages_lds_kernel(ShardSrc m, int R, float *__restrict__ out_scores,
                                                                int64_t *__restrict__ out_ids) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int k = m.k, n = R * k;
    int64_t *s_id = reinterpret_cast<int64_t *>(sm);
    float *s_sc = reinterpret_cast<float *>(sm + (size_t)n * 8);
    const int q = blockIdx.x;
    __shared__ int s_valid[64];
    __shared__ int64_t s_off[64];
    for (int r = threadIdx.x; r < R; r += blockDim.x) {   // the R headers once per workgroup, not once per staged element
        const ccr_shard_header *h = reinterpret_cast<const ccr_shard_header *>(m.base + (int64_t)r * m.stride);
        s_valid[r] = (int)h->k_valid;
        s_off[r] = h->row_offset;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int r = e / k, p = e - r * k;
        const char *msg = m.base + (int64_t)r * m.stride;
        if (p < s_valid[r]) {
            s_sc[e] = reinterpret_cast<const float *>(msg + sizeof(ccr_shard_header))[(int64_t)q * k + p];
            s_id[e] = s_off[r] + (int64_t) reinterpret_cast<const uint32_t *>(msg + m.rows_at)[(int64_t)q * k + p];
        } else {   // padding slot: ranks last, distinct id
            s_sc[e] = -INFINITY;
            s_id[e] = INT64_MAX - ((int64_t)r * k + p);
        }
    }
    __syncthreads();
    __shared__ int s_cut[64], s_lohi[128], s_cnt[64];
    if (R * k < 600)
        merge_lists_lds_all(s_sc, s_id, R, k, (int64_t)q * k, out_scores, out_ids);
    else
        merge_lists_lds(s_sc, s_id, R, k, k, s_cut, s_lohi, s_cnt, (int64_t)q * k, out_scores, out_ids);
}

// Short-list exchange: every rank sent only the kl < ko best entries of each query (kl ~ ko / R + 6 sigma of the share a shard of
// exchangeable rows holds of a global top-ko), the merge keeps ko of the R kl.  A shard's list is its EXACT canonical top-kl, so every
// row it did not send ranks behind its last entry: if that entry is NOT among the kept ko (cut_r < kl), no unsent row of the shard
// is either, and the merged list is the global top-ko, bit for bit.  A list consumed to its end (cut_r == kl) whose shard holds more
// rows may have had more to give: the query is marked in flags[] (and counted) and the caller repeats it with full lists.
__global__ __launch_bounds__(1024) void merge_short_lists_kernel(ShardSrc m, int R, int ko, float *__restrict__ out_scores,
                                                               int64_t *__restrict__ out_ids, uint32_t *__restrict__ flags,
                                                               uint32_t *__restrict__ n_flagged) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int kl = m.k, n = R * kl;
    int64_t *s_id = reinterpret_cast<int64_t *>(sm);
    float *s_sc = reinterpret_cast<float *>(sm + (size_t)n * 8);
    const int q = blockIdx.x;
    __shared__ int s_valid[64], s_more[64];
    __shared__ int64_t s_off[64];
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        const ccr_shard_header *h = reinterpret_cast<const ccr_shard_header *>(m.base + (int64_t)r * m.stride);
        s_valid[r] = (int)h->k_valid;
        s_off[r] = h->row_offset;
        s_more[r] = (h->n_rows > (int64_t)h->k_valid) ? 1 : 0;   // the shard holds rows it did not send
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int r = e / kl, p = e - r * kl;
        const char *msg = m.base + (int64_t)r * m.stride;
        if (p < s_valid[r]) {
            s_sc[e] = reinterpret_cast<const float *>(msg + sizeof(ccr_shard_header))[(int64_t)q * kl + p];
            s_id[e] = s_off[r] + (int64_t) reinterpret_cast<const uint32_t *>(msg + m.rows_at)[(int64_t)q * kl + p];
        } else {   // padding slot: ranks last, distinct id
            s_sc[e] = -INFINITY;
            s_id[e] = INT64_MAX - ((int64_t)r * kl + p);
        }
    }
    __syncthreads();
    __shared__ int s_cut[64], s_lohi[128], s_cnt[64];
    merge_lists_lds(s_sc, s_id, R, kl, ko, s_cut, s_lohi, s_cnt, (int64_t)q * ko, out_scores, out_ids);
    // (s_cut was complete before phase 2 of the merge began; nothing writes it afterwards)
    if (threadIdx.x == 0) {
        int bad = 0;
        for (int r = 0; r < R; ++r) bad |= (s_cut[r] >= s_valid[r] && s_more[r]) ? 1 : 0;   // every REAL entry of list r was kept
        flags[q] = (uint32_t)bad;
        if (bad) atomicAdd(n_flagged, 1u);
    }
}

// lists too long for the LDS: every probe decodes its entry from the messages.  grid = (ceil(R*k/256), n_q)
__global__ __launch_bounds__(256) void merge_messages_kernel(ShardSrc m, int R, int n_q, float *__restrict__ out_scores,
                                                           int64_t *__restrict__ out_ids) {
    const int k = m.k, q = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= R * k || q >= n_q) return;
    const int r = e / k, p = e - r * k;
    float s;
    int64_t id;
    shard_entry(m, r, q, p, s, id);
    int rank = p;
    for (int o = 0; o < R && rank < k; ++o) {
        if (o == r) continue;
        int lo = 0, hi = k;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            float so;
            int64_t io;
            shard_entry(m, o, q, mid, so, io);
            if (precedes(so, io, s, id))
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

// message from ordinary results: scores [n_q][k_valid], ids int64 global -> u32 local rows; slots [k_valid, k) zeroed
__global__ __launch_bounds__(256) void shard_fill_kernel(char *msg, size_t rows_at, int n_q, int k, int k_valid, const float *__restrict__ scores,
                                                        const int64_t *__restrict__ ids, int64_t row_offset) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)n_q * k) return;
    const int q = (int)(e / k), p = (int)(e - (int64_t)q * k);
    float s = 0.0f;
    uint32_t row = 0u;
    if (p < k_valid) {
        s = scores[(int64_t)q * k_valid + p];
        row = (uint32_t)(ids[(int64_t)q * k_valid + p] - row_offset);
    }
    reinterpret_cast<float *>(msg + sizeof(ccr_shard_header))[e] = s;
    reinterpret_cast<uint32_t *>(msg + rows_at)[e] = row;
}

}  // namespace ccr

using namespace ccr;

extern "C" int ccr_shard_message_fill(void *message, int n_q, int k, int k_valid, const float *scores, const int64_t *ids, int64_t row_offset,
                                      int64_t n_rows, void *stream) {
    CCR_REQUIRE(message && (uintptr_t)message % 16 == 0, "ccr_shard_message_fill: message must be a 16-byte aligned device pointer");
    CCR_REQUIRE(n_q >= 0 && k >= 1 && k <= MAX_K && k_valid >= 0 && k_valid <= k, "ccr_shard_message_fill: bad shape n_q=%d k=%d k_valid=%d", n_q, k,
                k_valid);
    CCR_REQUIRE(n_q == 0 || k_valid == 0 || (scores && ids), "ccr_shard_message_fill: null results");
    ccr_shard_header h;
    memset(&h, 0, sizeof(h));
    h.magic = CCR_SHARD_MAGIC;
    h.k_valid = (uint32_t)k_valid;
    h.row_offset = row_offset;
    h.n_rows = n_rows;
    int rc = launch_shard_header(h, message, (hipStream_t)stream);
    if (rc != CCR_OK || n_q == 0) return rc;
    const int64_t n = (int64_t)n_q * k;
    hipLaunchKernelGGL(shard_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (char *)message,
                       shard_rows_at(n_q, k), n_q, k, k_valid, scores, ids, row_offset);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

extern "C" int ccr_merge_shard_messages(const void *messages, int64_t message_stride_bytes, int R, int n_q, int k, float *out_scores,
                                        int64_t *out_ids, void *stream) {
    CCR_REQUIRE(messages && out_scores && out_ids, "ccr_merge_shard_messages: null pointer");
    CCR_REQUIRE(R >= 1 && R <= 64 && n_q >= 0 && k >= 1 && k <= MAX_K, "ccr_merge_shard_messages: bad shape R=%d n_q=%d k=%d", R, n_q, k);
    const size_t rows_at = shard_rows_at(n_q, k);
    CCR_REQUIRE((uintptr_t)messages % 16 == 0 && message_stride_bytes % 16 == 0 &&
                    (size_t)message_stride_bytes >= rows_at + (size_t)n_q * k * 4,
                "ccr_merge_shard_messages: message stride %lld shorter than one message or not 16-byte aligned", (long long)message_stride_bytes);
    if (n_q == 0) return CCR_OK;
    ShardSrc m = {reinterpret_cast<const char *>(messages), message_stride_bytes, rows_at, k};
    const size_t lds = (size_t)R * k * 12;
    if (lds <= 96 * 1024) {
        if (lds > 48 * 1024) {
            const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&merge_messages_lds_kernel), 96 * 1024);
            if (rc != CCR_OK) return rc;
        }
        const int threads = R * k >= 2048 ? 1024 : 256;
        hipLaunchKernelGGL(merge_messages_lds_kernel, dim3((unsigned)n_q), dim3(threads), lds, (hipStream_t)stream, m, R, out_scores, out_ids);
        CCR_LAUNCH_CHECK();
        return CCR_OK;
    }
    dim3 grid((unsigned)((R * k + 255) / 256), (unsigned)n_q);
    hipLaunchKernelGGL(merge_messages_kernel, grid, dim3(256), 0, (hipStream_t)stream, m, R, n_q, out_scores, out_ids);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

extern "C" int ccr_merge_short_lists(const void *messages, int64_t message_stride_bytes, int R, int n_q, int k_list, int k_out,
                                     float *out_scores, int64_t *out_ids, uint32_t *flags, uint32_t *n_flagged, void *stream) {
    CCR_REQUIRE(messages && out_scores && out_ids && flags && n_flagged, "ccr_merge_short_lists: null pointer");
    CCR_REQUIRE(R >= 1 && R <= 64 && n_q >= 0 && k_list >= 1 && k_out >= 1 && k_out <= MAX_K && k_list <= k_out,
                "ccr_merge_short_lists: bad shape R=%d n_q=%d k_list=%d k_out=%d", R, n_q, k_list, k_out);
    CCR_REQUIRE((int64_t)R * k_list >= k_out, "ccr_merge_short_lists: R * k_list = %lld entries cannot fill k_out = %d", (long long)R * k_list, k_out);
    const size_t rows_at = shard_rows_at(n_q, k_list);
    CCR_REQUIRE((uintptr_t)messages % 16 == 0 && message_stride_bytes % 16 == 0 &&
                    (size_t)message_stride_bytes >= rows_at + (size_t)n_q * k_list * 4,
                "ccr_merge_short_lists: message stride %lld shorter than one message or not 16-byte aligned", (long long)message_stride_bytes);
    const size_t lds = (size_t)R * k_list * 12;
    CCR_REQUIRE(lds <= 96 * 1024, "ccr_merge_short_lists: R * k_list * 12 = %zu bytes of lists do not fit the LDS (use the full-list exchange)", lds);
    hipStream_t s = (hipStream_t)stream;
    CCR_HIP_CHECK(hipMemsetAsync(n_flagged, 0, 4, s));
    if (n_q == 0) return CCR_OK;
    if (lds > 48 * 1024) {
        const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&merge_short_lists_kernel), 96 * 1024);
        if (rc != CCR_OK) return rc;
    }
    ShardSrc m = {reinterpret_cast<const char *>(messages), message_stride_bytes, rows_at, k_list};
    const int threads = (R * k_list >= 2048 || k_out >= 1024) ? 1024 : 256;
    hipLaunchKernelGGL(merge_short_lists_kernel, dim3((unsigned)n_q), dim3(threads), lds, s, m, R, k_out, out_scores, out_ids, flags, n_flagged);
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

extern "C" int ccr_merge_topk_strided(const float *scores, const int64_t *ids, int64_t score_rank_stride, int64_t id_rank_stride,
                                      int R, int n_q, int k, float *out_scores, int64_t *out_ids, void *stream) {
    CCR_REQUIRE(scores && ids && out_scores && out_ids, "ccr_merge_topk: null pointer");
    CCR_REQUIRE(R >= 1 && n_q >= 0 && k >= 1 && k <= MAX_K, "ccr_merge_topk: bad shape R=%d n_q=%d k=%d", R, n_q, k);
    CCR_REQUIRE(score_rank_stride >= (int64_t)n_q * k && id_rank_stride >= (int64_t)n_q * k,
                "ccr_merge_topk: rank strides %lld / %lld shorter than one [n_q, k] list", (long long)score_rank_stride,
                (long long)id_rank_stride);
    if (n_q == 0) return CCR_OK;
    const size_t lds = (size_t)R * k * 12;
    if (lds <= 96 * 1024 && R <= 64) {   // (the LDS merge keeps one cut per list in a 64-entry table)
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
