// ccr_bm25.hip -- BM25 as a sparse scorer on the device (the lexical leg of the candidate builder: the reference
// scores every query against the whole corpus with scipy on the host, scripts/bm_25.py:31-52, then sorts all N
// scores per query, scripts/ms_marco_eval.py:165-186).
//
// Index = the term-major postings (CSC of the count matrix): indptr[n_terms + 1], doc ids, term counts, plus the
// per-document length factor K_d = k1 * (1 - b + b * len_d / avdl).  HBM-bound integer/float streaming work:
//   score(q, d) = sum over the query's distinct terms t, ascending term id, of
//                 (tf(d,t) * idf_t) * (k1 + 1) / (tf(d,t) + K_d)          fp64, one rounding per operation
//   result      = (float) score  ->  exact top-k, order (score desc, doc index asc)
// Determinism without atomics: queries are processed in batches with a dense fp64 accumulator [batch][n_docs]; round r
// adds the postings of every query's r-th term (one launch per round: within a round a (query, doc) cell is touched at
// most once, and the rounds run in stream order = ascending term order).  The accumulator is converted to fp32 and
// re-zeroed in one pass, and the exact dense selection of the retrieval path picks the top-k.
#include <algorithm>
#include <vector>

#include "ccr_common.h"

namespace ccr {
int launch_dense_select(const float *scores, int64_t n_rows, int k, const uint32_t *out_rows, int q_begin, int nq_chunk,
                        const uint32_t *count_dev, int64_t id_offset, float *out_scores, int64_t *out_ids, hipStream_t s,
                        bool aggregate);

struct Bm25Round {       // one (query row of the batch, term) pair of a round
    int64_t begin, end;  // posting range
    double idf;
    int row;             // accumulator row
    int pad;
};

// grid = (chunks, pairs of this round), block = 256: block (c, p) adds postings [begin + c*CHUNK, ...) of pair p.
constexpr int BM25_CHUNK = 256 * 8;
__global__ __launch_bounds__(256) void bm25_round_kernel(const Bm25Round *__restrict__ pairs, const int32_t *__restrict__ doc_ids,
                                                        const float *__restrict__ tf, const double *__restrict__ doc_k,
                                                        double k1p1, int64_t n_docs, double *__restrict__ acc) {
    const Bm25Round pr = pairs[blockIdx.y];
    const int64_t lo = pr.begin + (int64_t)blockIdx.x * BM25_CHUNK;
    if (lo >= pr.end) return;
    const int64_t hi = lo + BM25_CHUNK < pr.end ? lo + BM25_CHUNK : pr.end;
    double *row = acc + (int64_t)pr.row * n_docs;
    // the documents of one posting list are distinct, so the 8 read-modify-writes of a thread are independent: all ids and
    // counts first, then both gathers, then the stores (written as a plain loop the accumulator store of one posting and
    // the accumulator load of the next may alias, and every posting pays two dependent memory round trips)
    constexpr int PER = BM25_CHUNK / 256;
    int32_t d[PER];
    double f[PER], kd[PER], old[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int64_t i = lo + threadIdx.x + u * 256;
        const int64_t ic = i < hi ? i : hi - 1;
        d[u] = doc_ids[ic];
        f[u] = (double)tf[ic];
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        kd[u] = doc_k[d[u]];
        old[u] = row[d[u]];
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int64_t i = lo + threadIdx.x + u * 256;
        const double numer = (f[u] * pr.idf) * k1p1;
        const double denom = f[u] + kd[u];
        if (i < hi) row[d[u]] = old[u] + numer / denom;
    }
}

// fp64 accumulator -> fp32 score rows; the accumulator is left zeroed for the next batch.
__global__ __launch_bounds__(256) void bm25_finish_kernel(double *__restrict__ acc, float *__restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        out[i] = (float)acc[i];
        acc[i] = 0.0;
    }
}
}  // namespace ccr

using namespace ccr;

struct ccr_bm25_index {
    std::vector<int64_t> indptr;   // host copy: posting ranges size the launches
    const int32_t *doc_ids;        // device, borrowed
    const float *tf;               // device, borrowed
    const double *doc_k;           // device, borrowed
    int64_t n_terms, n_docs;
    double k1;
};

extern "C" int ccr_bm25_index_create(const int64_t *indptr_host, const int32_t *doc_ids, const float *tf, const double *doc_k,
                                     int64_t n_terms, int64_t n_docs, double k1, ccr_bm25_index **out) {
    CCR_REQUIRE(indptr_host && doc_ids && tf && doc_k && out, "ccr_bm25_index_create: null pointer");
    CCR_REQUIRE(n_terms >= 1 && n_docs >= 1 && n_docs < ((int64_t)1 << 31), "ccr_bm25_index_create: bad shape terms=%lld docs=%lld",
                (long long)n_terms, (long long)n_docs);
    for (int64_t t = 0; t < n_terms; ++t)
        CCR_REQUIRE(indptr_host[t] <= indptr_host[t + 1], "ccr_bm25_index_create: indptr not monotone at term %lld", (long long)t);
    ccr_bm25_index *ix = new ccr_bm25_index();
    ix->indptr.assign(indptr_host, indptr_host + n_terms + 1);
    ix->doc_ids = doc_ids;
    ix->tf = tf;
    ix->doc_k = doc_k;
    ix->n_terms = n_terms;
    ix->n_docs = n_docs;
    ix->k1 = k1;
    *out = ix;
    return CCR_OK;
}

extern "C" int ccr_bm25_index_destroy(ccr_bm25_index *ix) {
    delete ix;
    return CCR_OK;
}

static int bm25_batch_rows(const ccr_bm25_index *ix, int n_q) {
    const int64_t per_row = ix->n_docs * 12;                       // fp64 accumulator + fp32 score row
    int64_t rows = ((int64_t)8 << 30) / per_row;                   // ~8 GiB of score rows per batch
    rows = std::min<int64_t>(std::max<int64_t>(rows, 1), 256);
    return (int)std::min<int64_t>(rows, n_q);
}

extern "C" size_t ccr_bm25_search_workspace_bytes(const ccr_bm25_index *ix, int n_q, int max_terms_per_query) {
    if (!ix || n_q <= 0 || max_terms_per_query < 0) return 0;
    const int rows = bm25_batch_rows(ix, n_q);
    return (size_t)rows * ix->n_docs * 12 + (size_t)rows * (size_t)std::max(1, max_terms_per_query) * sizeof(Bm25Round) + 1024;
}

extern "C" int ccr_bm25_search(const ccr_bm25_index *ix, const int64_t *q_ptr_host, const int32_t *q_terms_host,
                               const double *q_idf_host, int n_q, int k, float *out_scores, int64_t *out_ids, void *workspace,
                               size_t ws_bytes, void *stream) {
    CCR_REQUIRE(ix && q_ptr_host && out_scores && out_ids, "ccr_bm25_search: null pointer");
    CCR_REQUIRE(n_q >= 0 && k >= 1 && k <= MAX_K && (int64_t)k <= ix->n_docs, "ccr_bm25_search: k=%d must be in [1, min(n_docs, %d)]", k,
                MAX_K);
    if (n_q == 0) return CCR_OK;
    int max_terms = 0;
    for (int q = 0; q < n_q; ++q) {
        const int64_t a = q_ptr_host[q], b = q_ptr_host[q + 1];
        CCR_REQUIRE(a <= b, "ccr_bm25_search: q_ptr not monotone at query %d", q);
        max_terms = std::max<int>(max_terms, (int)(b - a));
        for (int64_t i = a; i < b; ++i) {
            CCR_REQUIRE(q_terms_host && q_idf_host, "ccr_bm25_search: null term arrays");
            CCR_REQUIRE(q_terms_host[i] >= 0 && q_terms_host[i] < ix->n_terms, "ccr_bm25_search: term id %d out of range", q_terms_host[i]);
            CCR_REQUIRE(i == a || q_terms_host[i] > q_terms_host[i - 1], "ccr_bm25_search: terms of query %d not strictly ascending", q);
        }
    }
    const size_t need = ccr_bm25_search_workspace_bytes(ix, n_q, max_terms);
    if (!workspace || ws_bytes < need || (uintptr_t)workspace % 256 != 0) {
        set_error("ccr_bm25_search: workspace %zu bytes (256-byte aligned) required, got %zu at %p", need, ws_bytes, workspace);
        return CCR_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int rows = bm25_batch_rows(ix, n_q);
    char *ws = (char *)workspace;
    double *acc = (double *)ws;
    float *scores = (float *)(ws + (size_t)rows * ix->n_docs * 8);
    Bm25Round *d_pairs = (Bm25Round *)(ws + (size_t)rows * ix->n_docs * 12);
    const int64_t cells = (int64_t)rows * ix->n_docs;
    CCR_HIP_CHECK(hipMemsetAsync(acc, 0, (size_t)cells * 8, s));
    // One pair table per batch (all rounds back to back), uploaded once; the host tables stay alive until the
    // synchronisation at the end of the call.
    struct Round {
        size_t first, count;
        int64_t longest;
    };
    std::vector<std::vector<Bm25Round>> tables;
    tables.reserve((size_t)(n_q + rows - 1) / rows);
    for (int q0 = 0; q0 < n_q; q0 += rows) {
        const int m = std::min(rows, n_q - q0);
        int rounds = 0;
        for (int q = q0; q < q0 + m; ++q) rounds = std::max<int>(rounds, (int)(q_ptr_host[q + 1] - q_ptr_host[q]));
        tables.emplace_back();
        std::vector<Bm25Round> &table = tables.back();
        std::vector<Round> plan;
        for (int r = 0; r < rounds; ++r) {
            Round rd = {table.size(), 0, 0};
            for (int q = q0; q < q0 + m; ++q) {
                const int64_t i = q_ptr_host[q] + r;
                if (i >= q_ptr_host[q + 1]) continue;
                const int32_t t = q_terms_host[i];
                Bm25Round pr;
                pr.begin = ix->indptr[t];
                pr.end = ix->indptr[t + 1];
                pr.idf = q_idf_host[i];
                pr.row = q - q0;
                pr.pad = 0;
                if (pr.end > pr.begin) {
                    table.push_back(pr);
                    rd.longest = std::max(rd.longest, pr.end - pr.begin);
                }
            }
            rd.count = table.size() - rd.first;
            if (rd.count) plan.push_back(rd);
        }
        if (!table.empty())   // stream-ordered behind the previous batch's kernels, which read the same device region
            CCR_HIP_CHECK(hipMemcpyAsync(d_pairs, table.data(), table.size() * sizeof(Bm25Round), hipMemcpyHostToDevice, s));
        for (const Round &rd : plan) {
            dim3 grid((unsigned)((rd.longest + BM25_CHUNK - 1) / BM25_CHUNK), (unsigned)rd.count);
            hipLaunchKernelGGL(bm25_round_kernel, grid, dim3(256), 0, s, d_pairs + rd.first, ix->doc_ids, ix->tf, ix->doc_k,
                               ix->k1 + 1.0, ix->n_docs, acc);
            CCR_LAUNCH_CHECK();
        }
        const int64_t n = (int64_t)m * ix->n_docs;
        hipLaunchKernelGGL(bm25_finish_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, (int64_t)1 << 30)), dim3(256), 0, s, acc, scores, n);   // one cell per thread: streams faster than a capped grid-stride loop
        CCR_LAUNCH_CHECK();
        const int rc = launch_dense_select(scores, ix->n_docs, k, nullptr, q0, m, nullptr, 0, out_scores, out_ids, s, false);
        if (rc != CCR_OK) return rc;
    }
    CCR_HIP_CHECK(hipStreamSynchronize(s));
    return CCR_OK;
}
