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
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include <math.h>

#include "ccr_common.h"
#include "ccr_topk_device.h"

namespace ccr {
int launch_dense_select(const float *scores, int64_t n_rows, int k, const uint32_t *out_rows, int q_begin, int nq_chunk,
                        const uint32_t *count_dev, int64_t id_offset, float *out_scores, int64_t *out_ids, hipStream_t s,
                        bool aggregate, const uint32_t *in_rows, bool in_rows_compact);
int ensure_dynamic_lds(const void *kernel, size_t lds);

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

// ---------------------------------------------------------------------------------------------------------------------------
// The document-tile scorer (r4): the same sums without the fp64 rows in HBM.  A WAVE owns (query row, run of consecutive document
// tiles); the fp64 accumulators of one tile live in the wave's slice of LDS; for every tile the wave walks the query's terms in
// ascending order (the order of the adds of a cell = the round order above, so the bits are the same) and, per term, streams the
// slice of the posting list that falls into the tile: lane r keeps term r's cursor and the next document at the cursor, so a term
// with no posting in the tile costs nothing, and a posting list is read once per query from front to back (documents ascend inside
// a term; the cursor of a run's first tile comes from a lane-parallel binary search).  A finished tile leaves as fp32 scores:
// HBM traffic per posting 8 B (+ the K_d gather, L2-resident: 8 B x documents) instead of 32, per (query, document) cell 4 B
// instead of 28.  No barrier, no atomic on the data path (one ticket per run of tiles).
struct Bm25Term {        // one distinct term of a query row
    int64_t begin, end;  // posting range
    double idf;
};
constexpr int BM25_TILE_WAVES = 4;
constexpr int BM25_TILE_GROUP = 4;           // terms whose first steps are in flight together
constexpr int BM25_MAX_TILE_TERMS = 256;     // lane r <-> terms r, 64 + r, 128 + r, 192 + r; longer queries take the round kernels

__device__ __forceinline__ int64_t readlane64(int64_t v, int l) {
    const int lo = __builtin_amdgcn_readlane((int)(uint32_t)v, l), hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

// What a finished tile becomes (r5).  STORE: fp32 scores of every document into the row's score row (the exact dense selection reads
// them: small corpora, k beyond the candidate lists, and the rows the filter could not finish).  SAMPLE: a "run" is ONE piece of T
// documents out of every `run_stride_docs`; its scores go to a compact [rows][n_runs * T] sample the threshold kernel ranks.  FILTER:
// the tile is compared against the row's tau while it is still in the wave's registers and only the documents that pass leave the
// chip, as {score bits, document} records of the row's candidate list -- ONE list reservation per wave and tile that has a hit (about
// five hits per tile at k = 1001 of 500 k documents) -- so the [rows][n_docs] fp32 score rows (written once and read once by a separate
// collect pass in r4: 8 of the call's 13.6 GB of fabric traffic) never exist.
enum { BM25_STORE = 0, BM25_SAMPLE = 1, BM25_FILTER = 2 };
constexpr int BM25_LIST_CAP = 16384;         // candidate records per row (128 KiB of keys in the top-k kernel)

// grid = any, block = 256 (four independent waves).  T documents per tile, U 64-posting chunks per step, run_tiles tiles per ticket.
// row_map (STORE only, may be null): the launch scores table rows row_map[0 .. n_rows) into score rows 0 .. n_rows (the redo list).
// TABLE: the index holds the FINISHED contribution of every posting (ccr_bm25_index_set_idf: (tf idf_t)(k1 + 1) / (tf + K_d) evaluated once, by
// the same fp64 operations in the same order, so the same bits) and the queries use the index's idf: a posting is 4 + 8 coalesced bytes and
// one add -- no K_d gather (64 scattered 8-byte requests per wave and step), no fp64 division (~30 instructions per posting).
// G: lane r keeps the cursors of terms r, 64 + r, ... 64 (G - 1) + r of the row (G = 1: queries of up to 64 distinct terms; G = 4: up to 256 --
// product descriptions as queries, the reference's prime_pantry set-up: scripts/ms_marco_eval.py:56-57); the groups are walked in order, so
// the adds of a cell keep the ascending term order.
template <int T, int U, int MODE, bool TABLE, int G>
__global__ __launch_bounds__(256) void bm25_tile_kernel(const Bm25Term *__restrict__ terms, const int32_t *__restrict__ row_ptr, int n_rows,
                                                       const int32_t *__restrict__ doc_ids, const float *__restrict__ tf,
                                                       const double *__restrict__ doc_k, const double *__restrict__ contrib, double k1p1,
                                                       int64_t n_docs, int run_tiles, int n_runs,
                                                       int64_t run_stride_docs, uint32_t *__restrict__ ticket, float *__restrict__ scores,
                                                       const uint32_t *__restrict__ row_map, const float *__restrict__ tau,
                                                       uint2 *__restrict__ list, uint32_t *__restrict__ list_cnt, uint32_t *__restrict__ odd_cnt) {
    __shared__ double s_acc[BM25_TILE_WAVES][T];
    const int lane = threadIdx.x & 63;
    double *acc = s_acc[threadIdx.x >> 6];
#pragma unroll
    for (int j = 0; j < T / 64; ++j) acc[j * 64 + lane] = 0.0;
    const uint32_t n_items = (uint32_t)n_runs * (uint32_t)n_rows;
    for (;;) {
        uint32_t item = 0;
        if (lane == 0) item = atomicAdd(ticket, 1u);
        item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item);
        if (item >= n_items) break;   // every wave reaches this: the ticket only grows
        const int row = (int)(item % (uint32_t)n_rows), run = (int)(item / (uint32_t)n_rows);
        const int trow = row_map ? (int)row_map[row] : row;   // the row of the term table
        const int t0 = row_ptr[trow], nt = row_ptr[trow + 1] - t0;
        int64_t cur[G], end[G];
        double idf[G];
        int nd[G];   // the next document of the lane's term of group gq
        const int64_t run_base = (int64_t)run * run_stride_docs;
        const int64_t run_end = run_base + (int64_t)run_tiles * T < n_docs ? run_base + (int64_t)run_tiles * T : n_docs;
#pragma unroll
        for (int gq = 0; gq < G; ++gq) {
            cur[gq] = 0, end[gq] = 0, idf[gq] = 0.0;
            if (64 * gq + lane < nt) {
                const Bm25Term t = terms[t0 + 64 * gq + lane];
                cur[gq] = t.begin, end[gq] = t.end, idf[gq] = t.idf;
            }
            if (run > 0) {   // first posting of every term at or behind the run's first document
                int64_t lo = cur[gq], hi = end[gq];
                while (__ballot(lo < hi) != 0ull) {
                    const int64_t mid = lo + ((hi - lo) >> 1);
                    if (lo < hi) {
                        if ((int64_t)doc_ids[mid] < run_base) lo = mid + 1;
                        else hi = mid;
                    }
                }
                cur[gq] = lo;
            }
            nd[gq] = cur[gq] < end[gq] ? doc_ids[cur[gq]] : 0x7fffffff;
        }
        float *out_row = MODE == BM25_SAMPLE ? scores + ((int64_t)row * n_runs + run) * T : scores + (int64_t)row * n_docs;
        float thr = 0.f;
        bool odd = false;
        if (MODE == BM25_FILTER) thr = tau[row];
        for (int64_t tile_base = run_base; tile_base < run_end; tile_base += T) {
            const int tile_end = (int)(tile_base + T < n_docs ? tile_base + T : n_docs);
            // The terms with a posting in this tile, ascending (wave-uniform), in groups of NB: the first steps of a group's terms (ids +
            // values of up to 64 U postings each) are fetched TOGETHER, then consumed in term order -- one global round trip per group
            // instead of one per term (at five waves per SIMD the chain "load -> LDS add -> count -> next term's load", ~8 links per tile,
            // was the whole kernel).  A slot without a term repeats slot 0's addresses: every wave issues the same number of loads, so
            // the wait in front of slot b is a plain count.  The values are pinned behind the loads: left alone, hipcc sinks the load of
            // a value into the `if (in)` that uses it -- a second dependent round trip per step.
            constexpr int NB = BM25_TILE_GROUP;
            // (addresses = a wave-uniform 64-bit base + a 32-bit lane offset: the loads take the scalar-base form, no 64-bit lane arithmetic)
            auto fetch = [&](int32_t(&dd)[U], double(&cc)[U], float(&ff)[U], int64_t c0, int64_t e0) {
                const int last = (int)(e0 - c0 < (int64_t)(64 * U) ? e0 - c0 : (int64_t)(64 * U)) - 1;   // >= 0: c0 < e0
                const int32_t *pd = doc_ids + c0;
                const double *pc = contrib + c0;
                const float *pf = tf + c0;
#pragma unroll
                for (int u = 0; u < U; ++u) {   // clamped: every load is valid and unconditional
                    const int o = lane + 64 * u < last ? lane + 64 * u : last;
                    dd[u] = pd[o];
                    if (TABLE) cc[u] = pc[o];
                    else ff[u] = pf[o];
                }
            };
#pragma unroll
            for (int gq = 0; gq < G; ++gq) {
            unsigned long long act = __ballot(nd[gq] < tile_end);
            while (act != 0ull) {
                int32_t d[NB][U];
                double cv[NB][U];
                float f[NB][U];
                int rr[NB];
                int64_t cc[NB], ee[NB];
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    rr[b] = -1;
                    if (b > 0) cc[b] = cc[0], ee[b] = ee[0];
                    if (act != 0ull) {
                        rr[b] = __builtin_amdgcn_readfirstlane(__builtin_ctzll(act));
                        act &= act - 1ull;
                        cc[b] = readlane64(cur[gq], rr[b]), ee[b] = readlane64(end[gq], rr[b]);
                    }
                    fetch(d[b], cv[b], f[b], cc[b], ee[b]);
                }
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if (rr[b] < 0) break;
                    const int r = rr[b];
                    int64_t c = cc[b];
                    const int64_t e = ee[b];
                    const double w = __longlong_as_double(readlane64(__double_as_longlong(idf[gq]), r));
                    int cnt;
                    for (;;) {
                        cnt = 0;
                        const int have = (int)(e - c < (int64_t)(64 * U) ? e - c : (int64_t)(64 * U));   // real postings of this step
                        if (TABLE) {
#pragma unroll
                            for (int u = 0; u < U; ++u) asm volatile("" : "+v"(cv[b][u]));
#pragma unroll
                            for (int u = 0; u < U; ++u) {
                                const bool in = lane + 64 * u < have && d[b][u] < tile_end;
                                if (in) {
                                    const int x = d[b][u] - (int)tile_base;
                                    acc[x] = acc[x] + cv[b][u];
                                }
                                cnt += __popcll(__ballot(in));
                            }
                        } else {
                            double kd[U];
#pragma unroll
                            for (int u = 0; u < U; ++u) kd[u] = doc_k[d[b][u]];
#pragma unroll
                            for (int u = 0; u < U; ++u) {
                                const bool in = lane + 64 * u < have && d[b][u] < tile_end;
                                const double fd = (double)f[b][u];
                                const double numer = (fd * w) * k1p1;
                                const double denom = fd + kd[u];
                                if (in) {
                                    const int x = d[b][u] - (int)tile_base;
                                    acc[x] = acc[x] + numer / denom;
                                }
                                cnt += __popcll(__ballot(in));
                            }
                        }
                        c += cnt;
                        if (cnt != 64 * U) break;
                        fetch(d[b], cv[b], f[b], c, e);   // a list that fills the whole step goes on (c < e: a full step took real postings only)
                    }
                    // the postings are in document order, so the `cnt` taken ones are a prefix of the step and element `cnt` is the next one
                    int next = 0x7fffffff;
                    if (c < e) {
                        const int l = cnt & 63, u_sel = cnt >> 6;
                        next = __builtin_amdgcn_readlane(d[b][0], l);
#pragma unroll
                        for (int u = 1; u < U; ++u) {
                            const int v = __builtin_amdgcn_readlane(d[b][u], l);
                            next = u_sel == u ? v : next;
                        }
                    }
                    if (lane == r) cur[gq] = c, nd[gq] = next;
                }
            }
            }   // term groups
            if (MODE == BM25_FILTER) {
                // the tile against tau: the same test as the r4 collect pass (score >= tau, and > 0 when tau <= 0: a BM25 row is mostly
                // exact zeros).  Two walks over the tile's LDS image -- count, then write -- instead of 16 live scores per lane: the
                // registers of the second form cost a wave of occupancy per SIMD (117 against 72 VGPRs)
                int hits = 0;
#pragma unroll 4
                for (int j = 0; j < T / 64; ++j) {
                    const int x = j * 64 + lane;
                    const float v = (float)acc[x];
                    const bool in = tile_base + x < n_docs;
                    odd = odd || (in && (v < 0.f || v != v));
                    hits += __popcll(__ballot(in && v >= thr && (thr > 0.f || v > 0.f)));
                }
                if (hits != 0) {   // wave-uniform
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&list_cnt[row], (uint32_t)hits);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    uint2 *out = list + (int64_t)row * BM25_LIST_CAP;
#pragma unroll 4
                    for (int j = 0; j < T / 64; ++j) {
                        const int x = j * 64 + lane;
                        const float v = (float)acc[x];
                        acc[x] = 0.0;
                        const bool pass = tile_base + x < n_docs && v >= thr && (thr > 0.f || v > 0.f);
                        const unsigned long long m = __ballot(pass);
                        const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                        if (pass && at < (uint32_t)BM25_LIST_CAP) out[at] = make_uint2(__float_as_uint(v), (uint32_t)(tile_base + x));
                        base += (uint32_t)__popcll(m);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < T / 64; ++j) acc[j * 64 + lane] = 0.0;
                }
            } else {
#pragma unroll
                for (int j = 0; j < T / 64; ++j) {
                    const int x = j * 64 + lane;
                    const double v = acc[x];
                    acc[x] = 0.0;
                    if (MODE == BM25_SAMPLE) out_row[x] = tile_base + x < n_docs ? (float)v : -INFINITY;   // one tile per run
                    else if (tile_base + x < n_docs) out_row[tile_base + x] = (float)v;
                }
            }
        }
        if (MODE == BM25_FILTER && __ballot(odd) != 0ull && lane == 0) atomicAdd(&odd_cnt[row], 1u);
    }
}

// contrib[p] = the posting's finished contribution under the index's idf: the expression of the scorers, operation for operation.
// grid = (chunks, 1), block = 256; term_of[p] is found by a binary search of indptr (device copy) per posting.
__global__ __launch_bounds__(256) void bm25_contrib_kernel(const int64_t *__restrict__ indptr, int64_t n_terms, int64_t nnz,
                                                          const int32_t *__restrict__ doc_ids, const float *__restrict__ tf,
                                                          const double *__restrict__ doc_k, const double *__restrict__ idf, double k1p1,
                                                          double *__restrict__ contrib) {
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < nnz; p += (int64_t)gridDim.x * 256) {
        int64_t lo = 0, hi = n_terms;   // the last term t with indptr[t] <= p
        while (hi - lo > 1) {
            const int64_t mid = lo + ((hi - lo) >> 1);
            if (indptr[mid] <= p) lo = mid;
            else hi = mid;
        }
        const double w = idf[lo];
        const double fd = (double)tf[p];
        const double numer = (fd * w) * k1p1;
        const double denom = fd + doc_k[doc_ids[p]];
        contrib[p] = numer / denom;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Batched selection of the top-k of a batch of rows (r4; r5: fused into the tile scorer).  The exact dense selection (one 256-thread
// workgroup per query walking its whole row through four radix passes, every zero score landing in ONE histogram bin) took 70 %
// of a BM25 search: 5.2 ms per batch of 256 queries x 500 k documents.  Instead, the retrieval path's estimate-and-verify filter:
//   sample    : bm25_tile_kernel<.., BM25_SAMPLE> scores one 1 024-document piece out of every 64 of each row (a wave per (row, piece))
//               into a compact [rows][pieces * 1024] sample -- 1/64 of the scoring work, but a pass of dependent round trips (a binary
//               search per term, then ids -> K_d -> LDS per term): 256-document pieces, four times as many items, took 0.80 ms against
//               4.6 ms for the whole filter pass;
//   threshold : tau_row = the r-th largest score of the row's sample, r = the rank for which fewer than k documents pass with
//               probability < 1e-7 (r = 41 at k = 1001: ~2 600 pass);
//   filter    : bm25_tile_kernel<.., BM25_FILTER> scores every tile and appends the documents with score >= tau (and > 0 when
//               tau <= 0: a BM25 row is mostly exact zeros) to the row's candidate list straight from the tile's registers;
//   top-k     : per row, if k <= candidates <= capacity the k best of the list by (score desc, document asc) ARE the row's top-k
//               (everything >= tau is in the list and at least k documents are >= tau): bitonic sort of 64-bit keys in LDS.
//               Otherwise (an estimate that came out too high, a query whose terms match fewer than k documents, a flooded list)
//               the row is put on a list; the listed rows are scored again with their fp32 rows stored (BM25_STORE) and go through
//               the exact dense selection.
// Exact for every input; the order rule is the same everywhere.
constexpr int BM25_SAMPLE_PIECE = 1024;      // documents per sampled piece (the SAMPLE instantiation's tile)
constexpr int BM25_SAMPLE_EVERY = 64;        // one piece out of this many
constexpr int BM25_SAMPLE_MAX = 16384;       // sampled scores the threshold kernel holds in LDS

// grid = rows, block = 256.  tau[row] = the rank-th largest of the row's n sampled scores (sample rows are contiguous; pieces that
// reach beyond the corpus were filled with -inf).  Also zeroes the row's list counter.
__global__ __launch_bounds__(256) void bm25_threshold_kernel(const float *__restrict__ sample, int n, int rank, float *__restrict__ tau,
                                                            uint32_t *__restrict__ list_cnt) {
    extern __shared__ __attribute__((aligned(16))) float s_val[];
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_ctl[4];
    const int tid = threadIdx.x;
    const float *row = sample + (int64_t)blockIdx.x * n;
    for (int i = tid; i < n; i += 256) s_val[i] = row[i];
    if (tid == 0) list_cnt[blockIdx.x] = 0u;
    __syncthreads();
    uint32_t kth = 0;
    int need_eq = 0;
    block_radix_select<true>(
        [&](int64_t i, bool &skip) -> uint32_t {
            (void)skip;
            return f32_orderable(s_val[i]);
        },
        n, rank, s_hist, s_ctl, kth, need_eq);
    if (tid == 0) tau[blockIdx.x] = orderable_to_f32(kth);
}

// grid = rows, block = NT, dyn LDS = 12 bytes x the class's key capacity.  Two launches share the rows: the SMALL class (lists of up to
// BM25_TOPK_SMALL records, 48 KiB of LDS, three 512-thread workgroups per CU -- nearly every row: ~2 600 records are expected at k = 1001)
// and the rest (up to BM25_LIST_CAP, 128 KiB + , one 1024-thread workgroup per CU); a block whose row belongs to the other class
// exits.  Rows that cannot be finished here are appended to redo_list.
constexpr int BM25_TOPK_SMALL = 4096;
template <int NT, bool SMALL>
__global__ __launch_bounds__(NT) void bm25_topk_kernel(const uint2 *__restrict__ list, const uint32_t *__restrict__ list_cnt,
                                                      const uint32_t *__restrict__ odd_cnt, const float *__restrict__ tau, int64_t n_docs, int k,
                                                      int q_begin, int n_rows, float *__restrict__ out_scores, int64_t *__restrict__ out_ids,
                                                      uint32_t *__restrict__ redo_cnt, uint32_t *__restrict__ redo_list) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_keys[];
    const int tid = threadIdx.x;
    // SMALL: a block per row.  Large: a few blocks walk all rows and take the large ones (2 000 blocks of 128 KiB that only exit cost 70 us)
    for (int r = blockIdx.x; r < n_rows; r += gridDim.x) {
    const uint32_t n = list_cnt[r];
    if ((n <= (uint32_t)BM25_TOPK_SMALL) != SMALL) continue;   // block-uniform: the other launch's row
    __syncthreads();   // the previous row's keys are done with
    // Fewer than k documents passed.  With tau <= 0 and no negative / NaN score in the row the list holds EVERY non-zero document, all
    // others score exactly zero, and the row's top-k is the sorted list followed by the k - n lowest-numbered documents outside it
    // (a query whose terms match fewer than k documents: common for rare terms).  Anything else -- an estimate that came out too high,
    // a flooded list -- goes to the exact dense selection.
    const bool zero_fill = n < (uint32_t)k && tau[r] <= 0.f && odd_cnt[r] == 0u;
    if ((n < (uint32_t)k && !zero_fill) || n > (uint32_t)BM25_LIST_CAP) {   // block-uniform
        if (tid == 0) redo_list[atomicAdd(redo_cnt, 1u)] = (uint32_t)r;
        continue;
    }
    const int np2 = n ? pow2_ceil((int)n) : 1;
    const uint2 *src = list + (int64_t)r * BM25_LIST_CAP;
    for (int i = tid; i < np2; i += NT) {
        unsigned long long key = 0ull;
        if (i < (int)n) {
            const uint2 e = src[i];
            key = make_key(__uint_as_float(e.x), e.y);
        }
        s_keys[i] = key;
    }
    block_bitonic_sort_desc(s_keys, np2);
    const int64_t orow = (int64_t)(q_begin + r);
    const int head = n < (uint32_t)k ? (int)n : k;
    for (int i = tid; i < head; i += NT) {
        const unsigned long long key = s_keys[i];
        out_scores[orow * k + i] = key_score(key);
        out_ids[orow * k + i] = (int64_t)key_idx(key);
    }
    if (zero_fill) {
        // the j-th lowest document that is NOT in the list (j = 0 .. k - n - 1) is document j + (number of listed documents <= it): at most
        // n < k listed documents lie below it, so thread j walks candidates j, j + 1, ... and counts the listed ones below by scanning the
        // list (n < k <= 8 192 entries in LDS; this path serves a few rare-term queries)
        __syncthreads();
        unsigned int *s_doc = reinterpret_cast<unsigned int *>(s_keys + np2);     // the listed documents, unsorted
        for (int i = tid; i < (int)n; i += NT) s_doc[i] = key_idx(s_keys[i]);
        __syncthreads();
        for (int j = tid; j < k - (int)n; j += NT) {
            // fixed point of d = j + #{listed < = d}: monotone, converges in at most n + 1 rounds (usually 1 - 2)
            unsigned int d = (unsigned int)j;
            for (;;) {
                unsigned int below = 0;
                for (int i = 0; i < (int)n; ++i) below += s_doc[i] <= d ? 1u : 0u;
                const unsigned int nd = (unsigned int)j + below;
                if (nd == d) break;
                d = nd;
            }
            out_scores[orow * k + n + j] = 0.f;
            out_ids[orow * k + n + j] = (int64_t)d;
        }
    }
    }
}

// Index validation (once, at ccr_bm25_index_create): the tile scorer walks every posting list with a cursor, so the documents of a
// term must ascend strictly; both scorers gather doc_k[doc].  mark[i] = 1 where a term's list starts.
__global__ __launch_bounds__(256) void bm25_mark_starts_kernel(const int64_t *__restrict__ indptr, int64_t n_terms, int64_t nnz,
                                                              uint8_t *__restrict__ mark) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < n_terms && indptr[t] < nnz) mark[indptr[t]] = 1;
}
__global__ __launch_bounds__(256) void bm25_check_postings_kernel(const int32_t *__restrict__ doc_ids, const uint8_t *__restrict__ mark,
                                                                 int64_t nnz, int64_t n_docs, unsigned long long *__restrict__ bad) {
    // bad[0] = postings with a document outside [0, n_docs), bad[1] = neighbours inside a term that do not ascend, bad[2] = first such position + 1
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * 256) {
        const int32_t d = doc_ids[i];
        if (d < 0 || (int64_t)d >= n_docs) atomicAdd(&bad[0], 1ull);
        if (i + 1 < nnz && !mark[i + 1] && doc_ids[i + 1] <= d) {
            atomicAdd(&bad[1], 1ull);
            atomicMin(&bad[2], (unsigned long long)i + 1ull);
        }
    }
}

// rank of the sample whose value at most k documents fail to reach with probability < 1e-7 (the planner's rule for estimated
// thresholds, ccr_api.hip: Wilson-Hilferty lower quantile of Gamma(r) >= k x sample fraction; at least 40)
static int bm25_sample_rank(int k, double fs) {
    const double need = (double)k * fs;
    int r = 40;
    for (; r < 1 << 20; ++r) {
        const double a = 1.0 / (9.0 * r), t = 1.0 - a - 5.2 * sqrt(a);
        if (t > 0.0 && (double)r * t * t * t >= need) break;
    }
    return r;
}

}  // namespace ccr

using namespace ccr;

struct ccr_bm25_index {
    std::vector<int64_t> indptr;   // host copy: posting ranges size the launches
    const int32_t *doc_ids;        // device, borrowed
    const float *tf;               // device, borrowed
    const double *doc_k;           // device, borrowed
    const double *contrib;         // device, borrowed: finished contributions under `idf` (ccr_bm25_index_set_idf), or null
    std::vector<double> idf;       // host copy of the idf the table was built with
    int64_t n_terms, n_docs;
    double k1;
    int num_cu;
    int run_tiles_knob;            // CCR_BM25_RUN_TILES: 0 = planner's choice, else tiles per ticket (tuning)
    mutable int64_t stats[4];      // of the last search: path (0 rounds + stored rows, 1 tile scorer + stored rows, 2 fused filter; + 4: contribution table), rows redone, batches, sample rank
    int redo_rows_knob;            // CCR_BM25_REDO_ROWS: rows of the fused path's redo area (tests: forces several redo chunks)
    int tile_cfg;                  // CCR_BM25_TILE: -1 = round kernels only (the A/B knob), 0 = default tile shape (1024 documents, 128-posting steps), 1 = 1024/64, 2 = 512/128, 3 = 1024/256
};

extern "C" int ccr_bm25_index_create(const int64_t *indptr_host, const int32_t *doc_ids, const float *tf, const double *doc_k,
                                     int64_t n_terms, int64_t n_docs, double k1, ccr_bm25_index **out) {
    CCR_REQUIRE(indptr_host && doc_ids && tf && doc_k && out, "ccr_bm25_index_create: null pointer");
    CCR_REQUIRE(n_terms >= 1 && n_docs >= 1 && n_docs < ((int64_t)1 << 31), "ccr_bm25_index_create: bad shape terms=%lld docs=%lld",
                (long long)n_terms, (long long)n_docs);
    for (int64_t t = 0; t < n_terms; ++t)
        CCR_REQUIRE(indptr_host[t] <= indptr_host[t + 1], "ccr_bm25_index_create: indptr not monotone at term %lld", (long long)t);
    int dev = 0, num_cu = 0;
    CCR_HIP_CHECK(hipGetDevice(&dev));
    CCR_HIP_CHECK(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
    const int64_t nnz = indptr_host[n_terms];
    CCR_REQUIRE(indptr_host[0] == 0 && nnz >= 0, "ccr_bm25_index_create: indptr must start at 0");
    if (nnz > 0) {   // one pass over the postings (a temporary of nnz bytes; index creation is not on the search path)
        char *tmp = nullptr;
        const size_t ptr_bytes = ((size_t)(n_terms + 1) * 8 + 255) / 256 * 256, mark_bytes = ((size_t)nnz + 255) / 256 * 256;
        CCR_HIP_CHECK(hipMalloc((void **)&tmp, ptr_bytes + mark_bytes + 64));
        int64_t *d_indptr = (int64_t *)tmp;
        uint8_t *mark = (uint8_t *)(tmp + ptr_bytes);
        unsigned long long *bad = (unsigned long long *)(tmp + ptr_bytes + mark_bytes);
        unsigned long long h_bad[3] = {0ull, 0ull, ~0ull};
        hipError_t e = hipMemcpy(d_indptr, indptr_host, (size_t)(n_terms + 1) * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemset(mark, 0, mark_bytes);
        if (e == hipSuccess) e = hipMemcpy(bad, h_bad, sizeof(h_bad), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(bm25_mark_starts_kernel, dim3((unsigned)((n_terms + 255) / 256)), dim3(256), 0, 0, d_indptr, n_terms, nnz, mark);
            hipLaunchKernelGGL(bm25_check_postings_kernel, dim3((unsigned)std::min<int64_t>((nnz + 255) / 256, 65536)), dim3(256), 0, 0, doc_ids, mark, nnz,
                               n_docs, bad);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpy(h_bad, bad, sizeof(h_bad), hipMemcpyDeviceToHost);
        (void)hipFree(tmp);
        if (e != hipSuccess) {
            set_error("ccr_bm25_index_create: validating the postings failed: %s", hipGetErrorString(e));
            return CCR_ERR_HIP;
        }
        CCR_REQUIRE(h_bad[0] == 0, "ccr_bm25_index_create: %llu postings name a document outside [0, %lld)", h_bad[0], (long long)n_docs);
        CCR_REQUIRE(h_bad[1] == 0, "ccr_bm25_index_create: documents must ascend strictly inside a term (%llu violations, first at posting %llu)",
                    h_bad[1], h_bad[2] - 1ull);
    }
    ccr_bm25_index *ix = new ccr_bm25_index();
    ix->indptr.assign(indptr_host, indptr_host + n_terms + 1);
    ix->doc_ids = doc_ids;
    ix->tf = tf;
    ix->doc_k = doc_k;
    ix->contrib = nullptr;
    ix->n_terms = n_terms;
    ix->n_docs = n_docs;
    ix->k1 = k1;
    ix->num_cu = num_cu > 0 ? num_cu : 256;
    const char *cfg = getenv("CCR_BM25_TILE");
    ix->tile_cfg = cfg ? atoi(cfg) : 0;
    const char *rt = getenv("CCR_BM25_RUN_TILES");
    ix->run_tiles_knob = rt ? atoi(rt) : 0;
    const char *rr = getenv("CCR_BM25_REDO_ROWS");
    ix->redo_rows_knob = rr ? atoi(rr) : 0;
    ix->stats[0] = ix->stats[1] = ix->stats[2] = ix->stats[3] = 0;
    *out = ix;
    return CCR_OK;
}

extern "C" int ccr_bm25_index_set_idf(ccr_bm25_index *ix, const double *idf_host, double *contrib, void *stream) {
    CCR_REQUIRE(ix && idf_host && contrib, "ccr_bm25_index_set_idf: null pointer");
    const int64_t nnz = ix->indptr[ix->n_terms];
    ix->contrib = nullptr;
    ix->idf.assign(idf_host, idf_host + ix->n_terms);
    if (nnz == 0) return CCR_OK;
    hipStream_t s = (hipStream_t)stream;
    char *tmp = nullptr;
    const size_t ptr_bytes = ((size_t)(ix->n_terms + 1) * 8 + 255) / 256 * 256;
    CCR_HIP_CHECK(hipMalloc((void **)&tmp, ptr_bytes + (size_t)ix->n_terms * 8));
    hipError_t e = hipMemcpyAsync(tmp, ix->indptr.data(), (size_t)(ix->n_terms + 1) * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(tmp + ptr_bytes, idf_host, (size_t)ix->n_terms * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(bm25_contrib_kernel, dim3((unsigned)std::min<int64_t>((nnz + 255) / 256, 1 << 20)), dim3(256), 0, s, (const int64_t *)tmp,
                           ix->n_terms, nnz, ix->doc_ids, ix->tf, ix->doc_k, (const double *)(tmp + ptr_bytes), ix->k1 + 1.0, contrib);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(tmp);
    if (e != hipSuccess) {
        set_error("ccr_bm25_index_set_idf: %s", hipGetErrorString(e));
        return CCR_ERR_HIP;
    }
    const char *off = getenv("CCR_BM25_TABLE");
    if (!(off && atoi(off) == 0)) ix->contrib = contrib;   // CCR_BM25_TABLE=0: build it, never use it (the A/B knob)
    return CCR_OK;
}

extern "C" int ccr_bm25_search_last_stats(const ccr_bm25_index *ix, int64_t *out4) {
    CCR_REQUIRE(ix && out4, "ccr_bm25_search_last_stats: null pointer");
    for (int i = 0; i < 4; ++i) out4[i] = ix->stats[i];
    return CCR_OK;
}

extern "C" int ccr_bm25_index_destroy(ccr_bm25_index *ix) {
    delete ix;
    return CCR_OK;
}

namespace {
// The sample of the estimate-and-verify selection: one BM25_SAMPLE_PIECE-document piece out of every `every`.
struct Bm25Sample {
    int64_t every;
    int n_pieces, rank;
    bool ok;     // the corpus is large enough for the sampled filter at this k
};
Bm25Sample bm25_sample_plan(const ccr_bm25_index *ix, int k) {
    Bm25Sample S;
    const int64_t all_pieces = (ix->n_docs + BM25_SAMPLE_PIECE - 1) / BM25_SAMPLE_PIECE;
    S.every = BM25_SAMPLE_EVERY;
    while ((all_pieces + S.every - 1) / S.every * BM25_SAMPLE_PIECE > BM25_SAMPLE_MAX) S.every *= 2;
    S.n_pieces = (int)((all_pieces + S.every - 1) / S.every);
    const double fs = (double)S.n_pieces * BM25_SAMPLE_PIECE / (double)ix->n_docs;
    S.rank = bm25_sample_rank(k, fs);
    // the sampled filter needs a sample that holds several times the rank; smaller corpora keep the exact dense selection
    S.ok = getenv("CCR_BM25_DENSE_SELECT") == nullptr && (int64_t)S.n_pieces * BM25_SAMPLE_PIECE >= 8 * (int64_t)S.rank &&
           (double)S.rank / fs * 2.0 + 1024.0 <= (double)BM25_LIST_CAP && k <= BM25_LIST_CAP / 2;
    return S;
}

// Workspace of one search.
//   fused (tile scorer + sampled filter, the default from ~30 k documents up): no score rows at all -- per row of the batch (up to
//     4 096) a 128-KiB candidate list and a sample row, plus fp32 score rows for `redo_rows` rows at a time (the rows the filter could
//     not finish; ~1 GiB);
//   stored (small corpora, k beyond the candidate lists, CCR_BM25_DENSE_SELECT): fp32 score rows of the whole batch (~8 GiB of them),
//     exact dense selection of every row; the round kernels (queries of more than 256 distinct terms, CCR_BM25_TILE=-1) keep their fp64
//     accumulator rows beside them (up to 256 rows).
struct Bm25Layout {
    bool tile, fused;
    int rows, redo_rows;
    Bm25Sample sample;
    size_t scores_off, table_off, sel_off, sample_off, total;
};
Bm25Layout bm25_layout(const ccr_bm25_index *ix, int n_q, int max_terms, int k) {
    Bm25Layout L;
    L.tile = ix->tile_cfg >= 0 && max_terms <= BM25_MAX_TILE_TERMS;
    L.sample = bm25_sample_plan(ix, k > 0 ? k : 1);
    L.fused = L.tile && k > 0 && L.sample.ok;
    const size_t terms = (size_t)std::max(1, max_terms);
    if (L.fused) {
        L.rows = std::min(n_q, 4096);
        L.redo_rows = (int)std::min<int64_t>(std::max<int64_t>(((int64_t)1 << 30) / (ix->n_docs * 4), 4), L.rows);
        if (ix->redo_rows_knob > 0) L.redo_rows = std::min(ix->redo_rows_knob, L.rows);
        L.scores_off = 0;
        L.table_off = ((size_t)L.redo_rows * ix->n_docs * 4 + 255) / 256 * 256;
    } else {
        const int64_t per_row = ix->n_docs * (L.tile ? 4 : 12);
        int64_t rows = ((int64_t)8 << 30) / per_row;                   // ~8 GiB of score rows per batch
        rows = std::min<int64_t>(std::max<int64_t>(rows, 1), L.tile ? 4096 : 256);
        L.rows = (int)std::min<int64_t>(rows, n_q);
        L.redo_rows = 0;
        L.scores_off = L.tile ? 0 : (size_t)L.rows * ix->n_docs * 8;
        L.table_off = (L.scores_off + (size_t)L.rows * ix->n_docs * 4 + 255) / 256 * 256;
    }
    const size_t table = L.tile ? (size_t)L.rows * terms * sizeof(Bm25Term) + ((size_t)L.rows + 1) * 4 : (size_t)L.rows * terms * sizeof(Bm25Round);
    L.sel_off = (L.table_off + table + 255) / 256 * 256;
    // the fused selection: candidate lists, thresholds, counters, redo list, then the sample rows
    const size_t sel = L.fused ? (size_t)L.rows * BM25_LIST_CAP * 8 + (size_t)L.rows * 20 + 256 * 4 : 256;
    L.sample_off = (L.sel_off + sel + 255) / 256 * 256;
    L.total = L.sample_off + (L.fused ? (size_t)L.rows * L.sample.n_pieces * BM25_SAMPLE_PIECE * 4 : 0);
    return L;
}

struct Bm25TileArgs {
    const Bm25Term *terms;
    const int32_t *row_ptr;
    int m;                      // rows of this launch
    uint32_t *ticket;
    float *scores;              // STORE: [m][n_docs]; SAMPLE: [m][n_pieces * T]
    const uint32_t *row_map;    // STORE of listed rows
    const float *tau;           // FILTER
    uint2 *list;
    uint32_t *list_cnt, *odd_cnt;
    int64_t sample_every;       // SAMPLE: one piece of T documents out of this many
    int sample_pieces;
    bool table;                 // the queries use the index's idf: finished contributions instead of tf / K_d
    int max_terms;              // of the batch: up to 64 -> one cursor group per lane, up to 256 -> four
};

template <int T, int U, int MODE>
int launch_bm25_tile(const ccr_bm25_index *ix, const Bm25TileArgs &a, hipStream_t s) {
    const int64_t n_tiles = (ix->n_docs + T - 1) / T;
    const int m = a.m;
    const int wgs_per_cu = std::max(1, std::min(8, (int)(160 * 1024 / (BM25_TILE_WAVES * T * 8))));
    const int64_t waves = (int64_t)ix->num_cu * wgs_per_cu * BM25_TILE_WAVES;
    int64_t runs, stride;
    int run_tiles;
    if (MODE == BM25_SAMPLE) {
        runs = a.sample_pieces, run_tiles = 1, stride = a.sample_every * T;
    } else {
        // Tiles per ticket.  Consecutive tickets are consecutive query rows on the SAME run of documents, so the waves in flight read the
        // same slices of the common terms' posting lists: short runs keep those slices (and their doc_k) inside the XCDs' L2 -- 4 to 12 tiles
        // per ticket measured 10 % faster end to end than 24 or 48 (L2 hit rate of the kernel 81 % at 24), 2 slower again (a binary search
        // per term and ticket).  About 24 tickets per wave, at most 12 tiles, at least 4 where that still leaves two tickets per wave.
        runs = std::min<int64_t>(std::max<int64_t>((24 * waves + m - 1) / m, 1), n_tiles);
        run_tiles = (int)std::min<int64_t>((n_tiles + runs - 1) / runs, 12);
        if (run_tiles < 4 && (n_tiles / 4) * (int64_t)m >= 2 * waves) run_tiles = 4;
        if (ix->run_tiles_knob > 0) run_tiles = (int)std::min<int64_t>(ix->run_tiles_knob, n_tiles);
        runs = (n_tiles + run_tiles - 1) / run_tiles;
        stride = (int64_t)run_tiles * T;
    }
    const int64_t items = runs * m;
    const unsigned grid = (unsigned)std::min<int64_t>((items + BM25_TILE_WAVES - 1) / BM25_TILE_WAVES, (int64_t)ix->num_cu * wgs_per_cu);
#define CCR_BM25_LAUNCH(TAB, GG)                                                                                                                  \
    hipLaunchKernelGGL((bm25_tile_kernel<T, U, MODE, TAB, GG>), dim3(grid), dim3(64 * BM25_TILE_WAVES), 0, s, a.terms, a.row_ptr, m, ix->doc_ids, \
                       ix->tf, ix->doc_k, ix->contrib, ix->k1 + 1.0, ix->n_docs, run_tiles, (int)runs, stride, a.ticket, a.scores, a.row_map,     \
                       a.tau, a.list, a.list_cnt, a.odd_cnt)
    if (a.max_terms <= 64) {
        if (a.table) CCR_BM25_LAUNCH(true, 1);
        else CCR_BM25_LAUNCH(false, 1);
    } else {
        if (a.table) CCR_BM25_LAUNCH(true, 4);
        else CCR_BM25_LAUNCH(false, 4);
    }
#undef CCR_BM25_LAUNCH
    CCR_LAUNCH_CHECK();
    return CCR_OK;
}

template <int MODE>
int launch_bm25_tile_cfg(const ccr_bm25_index *ix, const Bm25TileArgs &a, hipStream_t s) {
    switch (ix->tile_cfg) {   // measured at 500 k documents x 2 000 queries (tools/exp_bm25_tile.py): 1024/2 4.7 ms, 512/2 the same, wider steps slower
        case 1: return launch_bm25_tile<1024, 1, MODE>(ix, a, s);
        case 2: return launch_bm25_tile<512, 2, MODE>(ix, a, s);
        case 3: return launch_bm25_tile<1024, 4, MODE>(ix, a, s);
        case 4: return launch_bm25_tile<512, 4, MODE>(ix, a, s);
        default: return launch_bm25_tile<1024, 2, MODE>(ix, a, s);
    }
}
}  // namespace

extern "C" size_t ccr_bm25_search_workspace_bytes_k(const ccr_bm25_index *ix, int n_q, int max_terms_per_query, int k) {
    if (!ix || n_q <= 0 || max_terms_per_query < 0 || k < 1) return 0;
    return bm25_layout(ix, n_q, max_terms_per_query, k).total;
}

// k unknown: the larger of the two layouts (a workspace of this size serves every k)
extern "C" size_t ccr_bm25_search_workspace_bytes(const ccr_bm25_index *ix, int n_q, int max_terms_per_query) {
    if (!ix || n_q <= 0 || max_terms_per_query < 0) return 0;
    return std::max(bm25_layout(ix, n_q, max_terms_per_query, 0).total, bm25_layout(ix, n_q, max_terms_per_query, 1).total);
}

extern "C" int ccr_bm25_search(const ccr_bm25_index *ix, const int64_t *q_ptr_host, const int32_t *q_terms_host,
                               const double *q_idf_host, int n_q, int k, float *out_scores, int64_t *out_ids, void *workspace,
                               size_t ws_bytes, void *stream) {
    CCR_REQUIRE(ix && q_ptr_host && out_scores && out_ids, "ccr_bm25_search: null pointer");
    CCR_REQUIRE(n_q >= 0 && k >= 1 && k <= MAX_K && (int64_t)k <= ix->n_docs, "ccr_bm25_search: k=%d must be in [1, min(n_docs, %d)]", k,
                MAX_K);
    if (n_q == 0) return CCR_OK;
    int max_terms = 0;
    bool use_table = ix->contrib != nullptr;   // ... and every query weight is the index's idf of that term, bit for bit
    for (int q = 0; q < n_q; ++q) {
        const int64_t a = q_ptr_host[q], b = q_ptr_host[q + 1];
        CCR_REQUIRE(a <= b, "ccr_bm25_search: q_ptr not monotone at query %d", q);
        max_terms = std::max<int>(max_terms, (int)(b - a));
        for (int64_t i = a; i < b; ++i) {
            CCR_REQUIRE(q_terms_host && q_idf_host, "ccr_bm25_search: null term arrays");
            CCR_REQUIRE(q_terms_host[i] >= 0 && q_terms_host[i] < ix->n_terms, "ccr_bm25_search: term id %d out of range", q_terms_host[i]);
            CCR_REQUIRE(i == a || q_terms_host[i] > q_terms_host[i - 1], "ccr_bm25_search: terms of query %d not strictly ascending", q);
            if (use_table && memcmp(&q_idf_host[i], &ix->idf[q_terms_host[i]], 8) != 0) use_table = false;
        }
    }
    const Bm25Layout L = bm25_layout(ix, n_q, max_terms, k);
    if (!workspace || ws_bytes < L.total || (uintptr_t)workspace % 256 != 0) {
        set_error("ccr_bm25_search: workspace %zu bytes (256-byte aligned) required, got %zu at %p", L.total, ws_bytes, workspace);
        return CCR_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int rows = L.rows;
    ix->stats[0] = (L.fused ? 2 : (L.tile ? 1 : 0)) + (L.tile && use_table ? 4 : 0), ix->stats[1] = 0, ix->stats[2] = (n_q + rows - 1) / rows, ix->stats[3] = L.fused ? L.sample.rank : 0;
    char *ws = (char *)workspace;
    double *acc = (double *)ws;                                  // round kernels only
    float *scores = (float *)(ws + L.scores_off);                // stored: [rows][n_docs]; fused: [redo_rows][n_docs]
    Bm25Round *d_pairs = (Bm25Round *)(ws + L.table_off);        // round kernels
    Bm25Term *d_terms = (Bm25Term *)(ws + L.table_off);          // tile scorer: the batch's terms, then its row pointers
    char *sel = ws + L.sel_off;
    uint2 *cand_list = (uint2 *)sel;                             // fused only from here on
    float *tau = (float *)(sel + (size_t)rows * BM25_LIST_CAP * 8);
    uint32_t *list_cnt = (uint32_t *)(tau + rows);
    uint32_t *odd_cnt = list_cnt + rows;           // [rows]: waves that saw a negative / NaN score
    uint32_t *ctl = L.fused ? odd_cnt + rows : (uint32_t *)sel;   // [0] redo count, [1..3] tickets of the sample / filter / store launches
    uint32_t *redo_list = ctl + 4;                 // [rows]
    float *sample = (float *)(ws + L.sample_off);
    const Bm25Sample &S = L.sample;
    if (L.fused) {
        const int rc1 = ensure_dynamic_lds(reinterpret_cast<const void *>(&bm25_topk_kernel<1024, false>), (size_t)BM25_LIST_CAP * 8);
        if (rc1 != CCR_OK) return rc1;
        const int rc3 = ensure_dynamic_lds(reinterpret_cast<const void *>(&bm25_topk_kernel<512, true>), (size_t)BM25_TOPK_SMALL * 12);
        if (rc3 != CCR_OK) return rc3;
        // (the kernel's static LDS -- histogram + control words, ~1 KiB -- sits on top of the dynamic part: opt in with head room)
        const int rc2 = ensure_dynamic_lds(reinterpret_cast<const void *>(&bm25_threshold_kernel), (size_t)BM25_SAMPLE_MAX * 4 + 2048);
        if (rc2 != CCR_OK) return rc2;
    }
    if (!L.tile) CCR_HIP_CHECK(hipMemsetAsync(acc, 0, (size_t)rows * ix->n_docs * 8, s));
    // One table per batch, uploaded once; the host tables stay alive until the synchronisation at the end of the batch.
    struct Round {
        size_t first, count;
        int64_t longest;
    };
    std::vector<std::vector<Bm25Round>> tables;
    std::vector<std::vector<char>> tile_tables;
    tables.reserve((size_t)(n_q + rows - 1) / rows);
    tile_tables.reserve((size_t)(n_q + rows - 1) / rows);
    for (int q0 = 0; q0 < n_q; q0 += rows) {
        const int m = std::min(rows, n_q - q0);
        if (L.fused) CCR_HIP_CHECK(hipMemsetAsync(odd_cnt, 0, (size_t)rows * 4 + 16, s));   // odd counts + the redo count and the tickets behind them
        else CCR_HIP_CHECK(hipMemsetAsync(ctl, 0, 16, s));
        if (L.tile) {
            // [terms of row 0 | row 1 | ...] with empty posting lists dropped, then int32 row pointers (16-byte aligned behind the terms)
            const size_t n_terms_batch = (size_t)(q_ptr_host[q0 + m] - q_ptr_host[q0]);
            const size_t ptr_off = n_terms_batch * sizeof(Bm25Term);
            tile_tables.emplace_back(ptr_off + ((size_t)m + 1) * 4);
            std::vector<char> &blob = tile_tables.back();
            Bm25Term *tt = reinterpret_cast<Bm25Term *>(blob.data());
            int32_t *rp = reinterpret_cast<int32_t *>(blob.data() + ptr_off);
            int32_t n = 0;
            for (int q = q0; q < q0 + m; ++q) {
                rp[q - q0] = n;
                for (int64_t i = q_ptr_host[q]; i < q_ptr_host[q + 1]; ++i) {
                    const int32_t t = q_terms_host[i];
                    if (ix->indptr[t + 1] > ix->indptr[t]) tt[n++] = Bm25Term{ix->indptr[t], ix->indptr[t + 1], q_idf_host[i]};
                }
            }
            rp[m] = n;
            CCR_HIP_CHECK(hipMemcpyAsync(d_terms, blob.data(), blob.size(), hipMemcpyHostToDevice, s));
            const int32_t *d_row_ptr = reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(d_terms) + ptr_off);
            Bm25TileArgs a = {d_terms, d_row_ptr, m, ctl + 1, scores, nullptr, tau, cand_list, list_cnt, odd_cnt, S.every, S.n_pieces, use_table, max_terms};
            if (L.fused) {
                // sample pieces -> tau per row -> every tile against tau -> sort the lists
                a.scores = sample;
                int rc = launch_bm25_tile<BM25_SAMPLE_PIECE, 2, BM25_SAMPLE>(ix, a, s);   // (this instantiation whatever CCR_BM25_TILE says)
                if (rc != CCR_OK) return rc;
                const int n_sample = S.n_pieces * BM25_SAMPLE_PIECE;
                hipLaunchKernelGGL(bm25_threshold_kernel, dim3((unsigned)m), dim3(256), (size_t)n_sample * 4, s, sample, n_sample, S.rank, tau, list_cnt);
                CCR_LAUNCH_CHECK();
                a.scores = nullptr, a.ticket = ctl + 2;
                rc = launch_bm25_tile_cfg<BM25_FILTER>(ix, a, s);
                if (rc != CCR_OK) return rc;
                hipLaunchKernelGGL((bm25_topk_kernel<512, true>), dim3((unsigned)m), dim3(512), (size_t)BM25_TOPK_SMALL * 12, s, cand_list, list_cnt, odd_cnt,
                                   tau, ix->n_docs, k, q0, m, out_scores, out_ids, ctl, redo_list);
                CCR_LAUNCH_CHECK();
                hipLaunchKernelGGL((bm25_topk_kernel<1024, false>), dim3((unsigned)std::min(m, 128)), dim3(1024), (size_t)BM25_LIST_CAP * 8, s, cand_list,
                                   list_cnt, odd_cnt, tau, ix->n_docs, k, q0, m, out_scores, out_ids, ctl, redo_list);
                CCR_LAUNCH_CHECK();
                // the rows the filter could not finish (usually none or a few rare-term queries): scored again with their rows stored,
                // `redo_rows` at a time, exact dense selection
                uint32_t n_redo = 0;
                CCR_HIP_CHECK(hipMemcpyAsync(&n_redo, ctl, 4, hipMemcpyDeviceToHost, s));
                CCR_HIP_CHECK(hipStreamSynchronize(s));
                ix->stats[1] += n_redo;
                for (uint32_t r0 = 0; r0 < n_redo; r0 += (uint32_t)L.redo_rows) {
                    const int mr = (int)std::min<uint32_t>((uint32_t)L.redo_rows, n_redo - r0);
                    CCR_HIP_CHECK(hipMemsetAsync(ctl + 3, 0, 4, s));
                    Bm25TileArgs b = a;
                    b.m = mr, b.ticket = ctl + 3, b.scores = scores, b.row_map = redo_list + r0;
                    rc = launch_bm25_tile_cfg<BM25_STORE>(ix, b, s);
                    if (rc != CCR_OK) return rc;
                    rc = launch_dense_select(scores, ix->n_docs, k, nullptr, q0, mr, nullptr, 0, out_scores, out_ids, s, false, redo_list + r0, true);
                    if (rc != CCR_OK) return rc;
                }
                continue;
            }
            const int rc = launch_bm25_tile_cfg<BM25_STORE>(ix, a, s);
            if (rc != CCR_OK) return rc;
        } else {
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
        }
        const int rc = launch_dense_select(scores, ix->n_docs, k, nullptr, q0, m, nullptr, 0, out_scores, out_ids, s, false, nullptr, false);
        if (rc != CCR_OK) return rc;
    }
    CCR_HIP_CHECK(hipStreamSynchronize(s));
    return CCR_OK;
}
