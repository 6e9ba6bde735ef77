// Host-only check of the search planner (ccr::make_plan in libccr_hip.so): no GPU call.  Built and run by
// tests/test_cpu_host.py::test_planner_invariants_on_cpu.  Prints one line per shape; exits non-zero on a violated invariant.
#include <stdio.h>
#include <string.h>

#include "ccr_index.h"
#include "ccr_narrow.h"

using namespace ccr;

static int fail(const char *what, long long n, int d, int nq, int k) {
    printf("FAIL %s at n=%lld dim=%d nq=%d k=%d\n", what, n, d, nq, k);
    return 1;
}

int main() {
    Knobs kn;
    memset(&kn, 0, sizeof(kn));
    kn.progressive = 1;
    kn.max_phases = 3;
    kn.mfma16 = -1;
    kn.stagger = 1;
    kn.optimistic = -1;   // the planner's choice, as in production
    kn.wide = -1;         // 256 x 384 main-pass tiles where the planner prices them in
    const long long rows[] = {300, 9862, 70000, 335184, 1105228, 2681468, 8841823, 6250000};
    const int dims[] = {64, 768, 1024};
    const int nqs[] = {1, 40, 300, 3452, 6980, 10000};
    const int ks[] = {1, 10, 100, 300, 1001, 4096};
    int bad = 0, fused = 0, total = 0;
    for (long long n : rows)
        for (int d : dims)
            for (int nq : nqs)
                for (int k : ks) {
                    if (k > n) continue;
                    const Plan p = make_plan(n, d, nq, k, CCR_SEARCH_DEFAULT, 256, kn);
                    ++total;
                    if (p.total == 0) bad += fail("empty workspace", n, d, nq, k);
                    if (!p.fused) continue;
                    ++fused;
                    // work items of the MAIN pass: blocks of p.tile_q queries (256, or 384 with the wide tile) in p.main_qgroups groups
                    const int nrc = NUM_XCD / p.main_qgroups, qb_per = p.main_qblocks / p.main_qgroups, per_x = p.grid / NUM_XCD;
                    const int items = p.ranges / nrc * qb_per;
                    if (p.nq_pad % TILE_Q || p.nq_pad < nq || p.qblocks * TILE_Q != p.nq_pad) bad += fail("query padding", n, d, nq, k);
                    if (p.qblocks % p.qgroups || NUM_XCD % p.qgroups) bad += fail("query groups", n, d, nq, k);
                    if (p.tile_q != TILE_Q && p.tile_q != WIDE_Q) bad += fail("main-pass tile", n, d, nq, k);
                    if (p.main_qblocks != (nq + p.tile_q - 1) / p.tile_q || (long long)p.main_qblocks * p.tile_q > p.nq_pad)
                        bad += fail("main-pass query blocks (every column of every block needs its counters)", n, d, nq, k);
                    if (p.main_qblocks % p.main_qgroups || NUM_XCD % p.main_qgroups) bad += fail("main-pass query groups", n, d, nq, k);
                    if (p.tile_q == WIDE_Q && (d % 32 || !p.mfma16 || p.narrow)) bad += fail("wide tile where its kernel does not apply", n, d, nq, k);
                    // the planner's choices this round measured: nine 384-blocks at NQ, 256-blocks where the block count cannot be grouped
                    if (d == 768 && nq == 3452 && p.tile_q != WIDE_Q) bad += fail("NQ batch not on the wide tile", n, d, nq, k);
                    if (d == 768 && nq == 6980 && p.tile_q != TILE_Q) bad += fail("19 ungroupable 384-blocks chosen", n, d, nq, k);
                    if (nq == 300 && d % 32 == 0 && !p.narrow && p.mfma16 && p.tile_q != WIDE_Q) bad += fail("300 queries not in one 384-block", n, d, nq, k);
                    if (p.ranges % NUM_XCD || p.ranges < NUM_XCD) bad += fail("ranges not a multiple of the XCD count", n, d, nq, k);
                    if (p.sublists != (p.mfma16 ? 8 : 4)) bad += fail("sublists", n, d, nq, k);
                    // 2 048 sub-lists is what the select stage's wide form and the threshold update walk; above 1 024 only at large k or
                    // when one or two query blocks would otherwise leave workgroups without a work item
                    if (p.ranges * p.sublists > 2048) bad += fail("too many sub-lists for the select stage", n, d, nq, k);
                    if (p.ranges * p.sublists > 1024 && p.rescore_cap <= 512 && (1024 / p.sublists) * p.main_qblocks >= p.grid)
                        bad += fail("more than 1 024 sub-lists without need", n, d, nq, k);
                    if (p.item_a < 0 || p.item_b < 0 || (p.item_b && p.item_b <= p.item_a) || p.item_a > items || p.item_b > items)
                        bad += fail("phase ends", n, d, nq, k);
                    if (p.item_a && p.item_a % per_x) bad += fail("phase A is not whole rounds", n, d, nq, k);
                    if (p.item_b && p.item_b % per_x) bad += fail("phase B is not whole rounds", n, d, nq, k);
                    if (p.sample_tiles * GROUPS_PER_TILE < k) bad += fail("sample smaller than k groups", n, d, nq, k);
                    // estimated thresholds: one launch, the rank well inside the sampled groups, never for small k (where the
                    // conservative bound + re-tightening passes fewer rows)
                    if (p.opt_rank && (p.item_a || p.item_b)) bad += fail("estimated thresholds with phases", n, d, nq, k);
                    if (p.opt_rank && (p.opt_rank < 40 || (long long)p.opt_rank * 4 > (long long)p.sample_tiles * GROUPS_PER_TILE))
                        bad += fail("estimated-threshold rank", n, d, nq, k);
                    if (p.opt_rank && (long long)p.opt_rank * p.tiles / p.sample_tiles < k) bad += fail("estimated thresholds pass fewer than k rows", n, d, nq, k);
                    if ((long long)p.sample_tiles * p.sample_stride > p.full_tiles + p.sample_stride) bad += fail("sample beyond the shard", n, d, nq, k);
                    // candidate segments: ascending, capacities hold one whole tile per sub-list, area inside the workspace
                    const CandLayout &L = p.cand;
                    if (L.nseg < 1 || L.nseg > 3) bad += fail("segments", n, d, nq, k);
                    long long end = 0;
                    int prev = 0;
                    for (int g = 0; g < L.nseg; ++g) {
                        const int seg_end = g + 1 < L.nseg ? L.seg_end[g] : p.ranges;
                        if (seg_end <= prev) bad += fail("segment order", n, d, nq, k);
                        if (L.cap[g] < TILE_DOCS / p.sublists + 16 || L.cap[g] % 4) bad += fail("sub-list capacity", n, d, nq, k);
                        if (L.base[g] < end) bad += fail("segments overlap", n, d, nq, k);
                        end = L.base[g] + (long long)(seg_end - prev) * p.nq_pad * p.sublists * L.cap[g];
                        prev = seg_end;
                    }
                    if (p.off_cand + (size_t)end * 8 > p.off_flag) bad += fail("candidate area overruns the flag area", n, d, nq, k);
                    if (!(p.off_qnorm < p.off_thr && p.off_thr < p.off_gmax && p.off_gmax < p.off_cnt && p.off_cnt < p.off_cand &&
                          p.off_cand < p.off_flag && p.off_flag < p.off_dense && p.off_dense < p.off_retry && p.off_retry < p.total))
                        bad += fail("workspace layout order", n, d, nq, k);
                    if ((n == 2681468 && d == 768 && nq == 3452 && (k == 100 || k == 1001)) || (n == 8841823 && d == 768 && nq == 6980 && k == 100) ||
                        (n == 6250000 && d == 1024 && nq == 10000 && k == 1001))
                        printf("n=%lld nq=%d k=%d: ranges %d x %d sub-lists, items/XCD-set %d = %.2f rounds, phases end at %d / %d, estimated-threshold rank %d, sample %d tiles, caps %d/%d/%d, "
                               "workspace %.2f GB\n", n, nq, k, p.ranges, p.sublists, items, (double)items / per_x, p.item_a, p.item_b, p.opt_rank, p.sample_tiles,
                               L.cap[0], L.cap[1], L.cap[2], (double)p.total / 1e9);
                }
    // small batches: the first main pass is the streaming kernel (ccr_narrow.hip) -- one launch, two sub-lists per query inside the
    // candidate area, chosen exactly when the query rows fit its LDS image; the retry layout stays the tile kernels'
    kn.narrow = -1;
    kn.narrow_nt = 1;
    int narrow = 0;
    for (long long n : rows)
        for (int d : dims)
            for (int nq : {1, 16, 17, 33, 40, 64, 65, 300})
                for (int k : ks) {
                    if (k > n) continue;
                    const Plan p = make_plan(n, d, nq, k, CCR_SEARCH_DEFAULT, 256, kn);
                    if (!p.fused) {
                        if (p.narrow) bad += fail("narrow without the fused path", n, d, nq, k);
                        continue;
                    }
                    const int nqt = nq <= 16 ? 1 : (nq <= 32 ? 2 : 4);
                    const bool want = nq <= NARROW_MAX_Q && narrow_lds_bytes(nqt, d) <= (size_t)160 * 1024;
                    if ((p.narrow != 0) != want || (want && p.narrow != nqt)) bad += fail("narrow choice", n, d, nq, k);
                    if (!p.narrow) {
                        if (p.first_nsub != p.ranges * p.sublists || p.first_sp != p.sublists || memcmp(&p.first_lay, &p.cand, sizeof(CandLayout)))
                            bad += fail("first-pass layout of a tile plan", n, d, nq, k);
                        continue;
                    }
                    ++narrow;
                    if (p.item_a || p.item_b) bad += fail("narrow plan with phases", n, d, nq, k);
                    if (p.first_nsub != NARROW_SUBLISTS || p.first_sp != NARROW_SUBLISTS || p.first_lay.nseg != 1) bad += fail("narrow layout", n, d, nq, k);
                    const long long capn = p.first_lay.cap[0];
                    if (capn % 4 || capn < 4096 || p.off_cand + (size_t)capn * NARROW_SUBLISTS * nq * 8 > p.off_flag) bad += fail("narrow capacity", n, d, nq, k);
                    if ((size_t)nq * NARROW_SUBLISTS * 4 > p.off_cand - p.off_cnt) bad += fail("narrow counters", n, d, nq, k);
                    if (p.ranges % NUM_XCD || p.ranges * p.sublists > 2048) bad += fail("retry layout of a narrow plan", n, d, nq, k);
                }
    printf("%d narrow plans\n", narrow);
    printf("%d plans (%d fused), %d violations\n", total, fused, bad);
    return bad ? 1 : 0;
}
