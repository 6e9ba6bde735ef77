// ccr_api.hip -- host side of the C ABI: index object, planner, ccr_search orchestration.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <math.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "ccr_index.h"
#include "ccr_narrow.h"
#include "ccr_topk_device.h"

namespace ccr {

// ------------------------------------------------------------------ error text (thread-local)
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#ifndef CCR_QDIRECT_DEFAULT
#define CCR_QDIRECT_DEFAULT 0   // 16x16x32 main pass: query fragments through the LDS ring (0) or straight from global memory (1 / 3 / 4 / 5)
#endif
// ------------------------------------------------------------------ planner
// Query-block groups per XCD set: the smallest divisor of qblocks among {1,2,4,8} that keeps one XCD's
// query rows (qblocks / groups blocks of 256 rows) within ~3 MiB of its 4-MiB L2.
static int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

Knobs read_knobs() {
    Knobs kn;
    kn.qgroups = env_int("CCR_QGROUPS", 0);
    kn.progressive = env_int("CCR_PROGRESSIVE", 1);
    kn.max_phases = env_int("CCR_PHASES", 3);
    kn.mfma16 = env_int("CCR_MFMA16", -1);
    kn.sample_div = env_int("CCR_SAMPLE_DIV", 0);
    kn.gemm_dbg = env_int("CCR_GEMM_DBG", 0);
    kn.stagger = env_int("CCR_GEMM_STAGGER", 1);
    kn.qdirect = env_int("CCR_QDIRECT", CCR_QDIRECT_DEFAULT);
    kn.max_lists = env_int("CCR_MAX_LISTS", 0);
    kn.ranges = env_int("CCR_RANGES", 0);
    kn.item_swap = env_int("CCR_ITEM_SWAP", 0);
    kn.optimistic = env_int("CCR_OPTIMISTIC", -1);
    kn.opt_rank = env_int("CCR_OPT_RANK", 0);
    kn.narrow = env_int("CCR_NARROW", -1);
    kn.wide = env_int("CCR_WIDE", -1);
    kn.narrow_nt = env_int("CCR_NARROW_NT", 0);
    kn.narrow_grid = env_int("CCR_NARROW_GRID", 0);
    kn.narrow_groups = env_int("CCR_NARROW_GROUPS", NARROW_MAX_GROUPS);
    return kn;
}

static int pick_qgroups(int qblocks, int dim, const Knobs &kn, int tile_q = TILE_Q) {
    {
        const int v = kn.qgroups;
        if ((v == 1 || v == 2 || v == 4 || v == 8) && qblocks % v == 0) return v;
    }
    const size_t block_bytes = (size_t)tile_q * dim * 2;
    int best = 1;
    for (int gq = 1; gq <= NUM_XCD; gq *= 2) {
        if (qblocks % gq) continue;
        best = gq;
        if ((size_t)(qblocks / gq) * block_bytes <= (size_t)3 << 20) break;
    }
    return best;
}

static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// Range count of the SAMPLE pass: its few tiles (1/16 ... 1/64 of the corpus) are cut into (range, query block) items like the main
// pass's, and with so few tiles per item the rounding decides the time: 82 sample tiles x 14 query blocks in 24 ranges are 42 items per XCD
// set = two rounds of 4- and 3-tile items, in 16 ranges one round of 6- and 5-tile items (1/8 NQ shard: 0.145 -> 0.115 ms).  The kernel's
// static assignment is simulated exactly (workgroup j of an XCD set takes items j, j + per_x, ...; item = range-row * blocks + block).
static int best_sample_ranges(int64_t n_vt, int qblocks, int qgroups, int grid) {
    const int nrc = NUM_XCD / qgroups, qb_per = qblocks / qgroups, per_x = std::max(1, grid / NUM_XCD);
    int best_r = NUM_XCD;
    double best = 1e300;
    const int64_t r_max = round_up(std::min<int64_t>(n_vt, 1024), NUM_XCD);
    std::vector<double> load((size_t)per_x);
    for (int64_t R = NUM_XCD; R <= r_max; R += NUM_XCD) {
        double worst = 0.0;
        for (int rc = 0; rc < nrc; ++rc) {   // the XCD sets of one query group differ only in their range class
            std::fill(load.begin(), load.end(), 0.0);
            const int64_t items = R / nrc * qb_per;
            for (int64_t item = 0; item < items; ++item) {
                const int64_t r = rc + nrc * (item / qb_per);
                const int64_t ntile = (n_vt - r + R - 1) / R;
                if (ntile > 0) load[(size_t)(item % per_x)] += (double)ntile + 0.15;   // + pipeline fill per item
            }
            worst = std::max(worst, *std::max_element(load.begin(), load.end()));
        }
        if (worst < best - 1e-9) {
            best = worst;
            best_r = (int)R;
        }
    }
    return best_r;
}

constexpr size_t DENSE_SCRATCH_TARGET = (size_t)1 << 30;  // ~1 GiB of score rows per dense chunk
#ifndef CCR_MFMA16_DEFAULT
#define CCR_MFMA16_DEFAULT 1
#endif

// ------------------------------------------------------------------ main-pass planner
// Work items of the main pass are (range, query block); range r owns the tiles r, r + R, ...  An XCD set of per_x
// workgroups walks (R / nrc) * qb_per items round-robin, so R (a multiple of 8, near 6 items per workgroup, items of at
// least 8 tiles) is chosen to make that count sit just below a multiple of per_x -- NQ: 128 ranges give exactly 7 items per
// workgroup where 112 left the last round 1/8 full.  The launches: phase A = one round of items with the sample
// thresholds, a re-tightening (threshold_update_kernel), optionally a second re-tightening after whole rounds, the rest.
//
// Every candidate (sample size, R, phase split) is priced in TILE UNITS of one workgroup (one 256 x 256 x dim tile ~ 22 us):
//   makespan of each launch    simulated static item assignment (+ 0.15 tile of pipeline fill per item)
//   sample pass                its tiles spread over the grid + the threshold kernel's passes over the group maxima
//   select stage               0.1 per range (it walks R x sublists sub-lists per query)
//   surviving candidates       0.014 each (filter hit path + select), a phase that covers the corpus fraction f with
//                              thresholds taken from a fraction g seen before lets through k * f / g rows per query
//                              (measured: 3 280 from a 1/32 sample alone, 1 180 with one re-tightening at k = 100)
//   launch boundary            3 per extra phase
// The per-query terms were measured at 3 452 queries (14 query blocks) and scale with the query count.
struct MainPassChoice {
    int64_t sample, ranges;
    int item_a, item_b;   // phase ends in work items of one XCD set (0 = absent)
    int opt_rank;         // > 0: ESTIMATED thresholds (the opt_rank-th largest sampled group maximum), single launch, verified by the select
};

// Estimated ("optimistic") thresholds.  The conservative threshold -- the k-th largest group maximum of a sample of the corpus
// fraction fs -- lets k / fs rows per query through the first phase, so large k needs a big sample (1/16 at k = 1001: 1/16 of the
// corpus scored twice) and two re-tightenings.  But the filter does not need a BOUND: any threshold tau works if at least k rows
// turn out to pass it with their lower bounds (then the k-th largest exact score is >= tau and every row that could reach it was
// recorded) -- which the select stage checks (L >= tau; a query that fails is retried under a valid bound).  So tau is set to
// the r-th largest sampled group maximum: about r / fs rows pass over the WHOLE corpus in ONE launch.  Given r, the pass count is
// (r / fs) x Gamma(r) / r: fewer than k rows pass iff Gamma(r) < k fs.  r is the smallest rank whose 1e-7 lower quantile
// (Wilson-Hilferty: r (1 - 1/(9r) - 5.2 sqrt(1/(9r)))^3) reaches k fs, and at least 40 -- k = 1001 on a 1/64 sample: r = 41,
// 2 700 rows pass.  Why the floor: a corpus in topical order can fool the estimate when the sample holds a query's whole
// cluster tile; with r <= 16 that ONE tile's 16 group maxima set tau (measured with r = 13 at k = 100 on the topically sorted
// bench corpus: one query in 3 452 fails the check, and its retry costs 5 ms of an 18-ms step), with r >= 40 at least three
// tiles contribute (k = 1001, same corpus: nobody fails).  With the floor the planner keeps the conservative plan at small k
// (2 560 rows would pass at k = 100 against 1 296), where the phased plan is cheap anyway.  Exact either way.
static int optimistic_rank(int k, int64_t sample, int64_t tiles, const Knobs &kn) {
    if (kn.opt_rank > 0) return kn.opt_rank;   // tests: a small rank makes most queries fail the check
    const double need = (double)k * (double)sample / (double)tiles;
    int r = 40;
    for (; r < 1 << 20; ++r) {
        const double a = 1.0 / (9.0 * r), t = 1.0 - a - 5.2 * sqrt(a);
        if (t > 0.0 && (double)r * t * t * t >= need) break;
    }
    return r;
}

static MainPassChoice choose_main_pass(const Plan &p, int k, int64_t sample_a, int64_t sample_b, int64_t sample_c, const Knobs &kn,
                                       bool single_only = false) {
    const int nrc = NUM_XCD / p.main_qgroups, qb_per = p.main_qblocks / p.main_qgroups, per_x = p.grid / NUM_XCD;
    // a 256 x 384 tile in units of the 256 x 256 tile the other terms were measured in (1.5x the multiply-adds at ~1.09x the rate)
    const double tile_cost = p.tile_q == WIDE_Q ? 1.37 : 1.0;
    // the select stage walks ranges x sublists sub-lists per query: 1 024 with its 256-thread form, 2 048 with the 1 024-thread
    // form large k uses anyway (rescore_cap > 512)
    int64_t max_lists = p.rescore_cap > 512 ? 2048 : 1024;
    // one or two query blocks: 128 ranges x 8 sub-lists would leave half of the workgroups without a work item (n_q = 1: main pass
    // 1.34 ms instead of 0.7) -- the select stage's wide form walks 2 048 sub-lists, and its cost does not matter for so few queries
    if ((max_lists / p.sublists) * p.main_qblocks < p.grid) max_lists = 2048;
    if (kn.max_lists >= 64 && kn.max_lists < max_lists) max_lists = kn.max_lists;
    const int64_t r_hi = std::min<int64_t>(max_lists / p.sublists, round_up(std::max<int64_t>(1, p.tiles / 8), NUM_XCD));
    const int64_t target = std::min<int64_t>(r_hi, round_up(std::max<int64_t>(NUM_XCD, (int64_t)p.grid * 6 / p.main_qblocks), NUM_XCD));
    const bool prog_on = kn.progressive != 0 && !single_only;   // 0: single launch (the streaming kernel of small batches has no phases)
    const int max_phases = kn.max_phases;        // 2: at most one re-tightening
    const double qscale = (double)p.nq_pad / 3584.0;
    const double select_per_range = 0.1 * qscale, hit_w = 0.014 * qscale, phase_w = 3.0;
    // 0.11 ms = 4.9 units of threshold kernel for 328 sample tiles x 3 584 queries
    // (the threshold kernel's time follows the number of sampled groups, not the number of queries, below ~3 500 queries: it runs
    // nq_pad / 16 workgroups of four latency-bound passes -- 0.21 ms for 655 sample tiles at n_q = 1)
    auto sample_cost = [&](int64_t smp) -> double {
        return (double)((smp * p.qblocks + p.grid - 1) / p.grid) + 0.015 * (double)smp * std::max(1.0, qscale);
    };

    MainPassChoice best_choice = {sample_a, target, 0, 0, 0};
    double best = 1e300;
    for (int pass = 0; pass < 3; ++pass) {
        const int64_t smp = pass == 0 ? sample_a : (pass == 1 ? sample_b : sample_c);
        if ((pass >= 1 && smp == sample_a) || (pass == 2 && smp == sample_b)) continue;
        const double fs = (double)smp / (double)p.tiles;
        const double smp_cost = sample_cost(smp);
        auto survivors = [&](double fa, double fb) -> double {   // fa, fb: corpus fractions of phases A and B1 (0 = absent)
            if (fa <= 0.0) return (double)k / fs;
            if (fb <= 0.0) return (double)k * (fa / fs + (1.0 - fa) / fa);
            return (double)k * (fa / fs + fb / fa + (1.0 - fa - fb) / (fa + fb));
        };
        // Every XCD set holds the same items = (R / classes) * blocks_per_group, all of ceil(tiles / R) tiles at most, dealt
        // round-robin to its per_x workgroups: a launch over n items takes ceil(n / per_x) rounds.  Phases end at ITEM
        // indices, so phase A is exactly one full round (and the middle phase whole rounds) for any R.
        // R stays a multiple of the XCD count (not just of the range classes): the retry pass of flagged queries re-uses the
        // ranges with its own, possibly different, query grouping
        const int64_t r_lo = std::max<int64_t>(NUM_XCD, target / 2 / NUM_XCD * NUM_XCD);
        for (int64_t R = r_lo; R <= std::min(r_hi, target * 2); R += NUM_XCD) {
            if (kn.ranges > 0 && R != std::min<int64_t>(r_hi, std::max<int64_t>(r_lo, round_up(kn.ranges, NUM_XCD)))) continue;
            const int64_t items = R / nrc * qb_per;
            const double item_cost = ((double)((p.tiles + R - 1) / R) + 0.15) * tile_cost;   // + pipeline fill per item
            auto rounds = [&](int64_t n) { return (double)((n + per_x - 1) / per_x); };
            const double common = smp_cost + select_per_range * (double)R + 1e-3 * std::abs((double)(R - target));
            const double single = rounds(items) * item_cost + hit_w * survivors(0.0, 0.0) + common;
            if (single < best && kn.optimistic != 1) {
                best = single;
                best_choice = {smp, R, 0, 0, 0};
            }
            if (kn.optimistic != 0 && smp * GROUPS_PER_TILE >= 4 * 40) {   // estimated thresholds: one launch, ~rank / fs rows pass
                const int rank = optimistic_rank(k, smp, p.tiles, kn);
                if ((int64_t)rank * 4 <= smp * GROUPS_PER_TILE) {
                    const double pass = std::max((double)rank / fs, (double)k);
                    const double opt = rounds(items) * item_cost + hit_w * pass + common - (kn.optimistic == 1 ? 1e9 : 0.0);
                    if (opt < best) {
                        best = opt;
                        best_choice = {smp, R, 0, 0, rank};
                    }
                }
            }
            if (!(prog_on && items > per_x && (double)p.tiles / (double)R >= 8.0)) continue;
            // the thresholds after phase A come from the ranges it completed (+ the started one for part of the queries)
            const double fa = (double)per_x / (double)items;
            const double two = (1.0 + rounds(items - per_x)) * item_cost + phase_w + hit_w * survivors(fa, 0.0) + common;
            if (two < best) {
                best = two;
                best_choice = {smp, R, per_x, 0, 0};
            }
            if (max_phases < 3) continue;
            for (int m = 1; m <= 3; ++m) {   // a second re-tightening after m more rounds
                const int64_t ib = (int64_t)per_x * (1 + m);
                if (ib >= items) break;
                const double fb = (double)m * per_x / (double)items;
                const double three = (1.0 + m + rounds(items - ib)) * item_cost + 2.0 * phase_w + hit_w * survivors(fa, fb) + common;
                if (three < best) {
                    best = three;
                    best_choice = {smp, R, per_x, (int)ib, 0};
                }
            }
        }
    }
    return best_choice;
}

static Plan make_plan_for(int64_t n_rows, int dim, int n_q, int k, int flags, int num_cu, const Knobs &kn, int mfma16) {
    Plan p;
    memset(&p, 0, sizeof(p));
    p.nq_pad = (int)round_up(n_q, TILE_Q);
    p.qblocks = p.nq_pad / TILE_Q;
    p.tile_q = TILE_Q;
    p.main_qblocks = p.qblocks;
    p.main_qgroups = 1;
    p.tiles = (n_rows + TILE_DOCS - 1) / TILE_DOCS;
    p.full_tiles = n_rows / TILE_DOCS;
    p.grid = std::max(NUM_XCD, num_cu / NUM_XCD * NUM_XCD);
    {
        p.mfma16 = mfma16;   // main pass on v_mfma_f32_16x16x32_bf16 (8 sub-lists per (range, query)) or 32x32x16 (4)
        p.sublists = p.mfma16 ? 8 : 4;
    }
    p.rescore_cap = std::min(8192, std::max(256, 2 * pow2_ceil(k)));

    // sample pass: group maxima of 16 rows; need comfortably more groups than k
    const int64_t min_sample = (2 * (int64_t)k + GROUPS_PER_TILE - 1) / GROUPS_PER_TILE;
    int64_t sample_div = 32;   // fraction of the tiles scored by the threshold pass; the planner also prices 1/64
    bool sample_div_forced = false;   // CCR_SAMPLE_DIV pins it
    {
        if (kn.sample_div >= 4 && kn.sample_div <= 256) {
            sample_div = kn.sample_div;
            sample_div_forced = true;
        }
    }
    auto sample_for = [&](int64_t div) { return std::max<int64_t>({(p.tiles + div - 1) / div, min_sample, 4}); };
    int64_t sample = sample_for(sample_div);
    // any dim the index accepts (a multiple of 8): the fused kernels zero-fill the last 32-element K sub-stage when dim % 32 != 0
    bool fused = (dim % 8 == 0) && dim >= 8 && (sample * 4 <= p.full_tiles) && k <= MAX_K;
    if ((flags & CCR_SEARCH_FORCE_FUSED) && (dim % 8 == 0) && p.full_tiles >= 1) {
        // honour the request where at all possible: the sample may be the whole corpus
        if (!fused) sample = std::min<int64_t>(std::max<int64_t>(sample, 1), p.full_tiles);
        fused = sample * GROUPS_PER_TILE >= k;
    }
    if (flags & CCR_SEARCH_FORCE_DENSE) fused = false;
    p.fused = fused ? 1 : 0;

    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = (size_t)round_up((int64_t)(off + bytes), 256);
        return o;
    };
    if (p.fused) {
        p.qgroups = pick_qgroups(p.qblocks, dim, kn);
        // the planner prices three sample sizes: 1/32 of the tiles, 1/64 (cheaper pass, twice the survivors of phase A) and 1/16
        // (large k: phase A's survivors cost more than the second half of the sample)
        const bool pinned = sample_div_forced || (flags & CCR_SEARCH_FORCE_FUSED);
        const int64_t sample_alt = pinned ? sample : sample_for(2 * sample_div);
        int64_t sample_big = pinned ? sample : sample_for(sample_div / 2);
        if (sample_big * 4 > p.full_tiles) sample_big = sample;
        // small batches: the first main pass is the streaming kernel (ccr_narrow.hip) when the query rows fit its LDS image
        // (65 .. 128 queries: two groups of <= 64 on paired workgroups that walk the same rows)
        int nqt = n_q <= 16 ? 1 : (n_q <= 32 ? 2 : 4);
        int ngroups = n_q <= NARROW_MAX_Q ? 1 : 2;
        // 65 .. 96 queries: ONE group of six query tiles where their rows fit the LDS (dim <= 768) -- every corpus row is pulled once
        if (n_q > NARROW_MAX_Q && n_q <= 96 && kn.narrow_groups >= 2 && dim % TILE_K == 0 && narrow_lds_bytes(6, dim) <= (size_t)160 * 1024 &&
            env_int("CCR_NARROW_WIDE", 1) != 0)
            nqt = 6, ngroups = 1;
        const bool narrow = kn.narrow != 0 && n_q <= NARROW_MAX_Q * std::min(std::max(kn.narrow_groups, 1), NARROW_MAX_GROUPS) && dim % TILE_K == 0 &&
                            narrow_lds_bytes(nqt, dim) <= (size_t)160 * 1024 && (ngroups == 1 || num_cu >= 16);
        p.narrow = narrow ? nqt : 0;
        p.narrow_groups = narrow ? ngroups : 1;
        // the main pass's tile: 256 x 384 (gemm_topk16w_kernel: -17 % bytes per flop through the L1 miss path that bounds the 256 x 256
        // kernel) where it pays.  Measured at the NQ corpus (profiles/r06_wide_*.txt), per padded multiply-add against the 256 x 256
        // kernel: 0.92-0.96 while the query rows one XCD walks stay L2-resident (300 and 1 100 queries: main pass -31 % / -14 %, most
        // of it the padding: one block instead of two, three instead of five), 0.97 at 5 MiB per XCD (3 452 queries: nine blocks =
        // 3 456 columns against fourteen = 3 584: -6.5 %), and SLOWER beyond -- 4 096 queries = eleven blocks, 6 980 = nineteen: a prime
        // block count cannot be split over the XCDs, every XCD streams all the query rows past its 4-MiB L2 (L2 hit rate 0.71 against
        // 0.91, +3.4 % main pass) while sixteen / twenty-eight 256-blocks split into groups that fit.
        p.tile_q = TILE_Q;
        p.main_qblocks = p.qblocks;
        if (!narrow && p.mfma16 && dim % 32 == 0 && kn.wide != 0) {
            const int wb = (n_q + WIDE_Q - 1) / WIDE_Q;
            const double per_xcd = (double)(wb / pick_qgroups(wb, dim, kn, WIDE_Q)) * WIDE_Q * dim * 2.0;   // query bytes one XCD walks
            const double per_flop = per_xcd <= 3.2 * 1048576.0 ? 0.95 : (per_xcd <= 5.5 * 1048576.0 ? 0.97 : 1.04);
            if (kn.wide == 1 || (double)wb * WIDE_Q * per_flop < (double)p.qblocks * TILE_Q) {
                p.tile_q = WIDE_Q;
                p.main_qblocks = wb;
                p.nq_pad = (int)round_up(std::max<int64_t>(n_q, (int64_t)wb * WIDE_Q), TILE_Q);   // the counters of every block's columns exist
                p.qblocks = p.nq_pad / TILE_Q;
                p.qgroups = pick_qgroups(p.qblocks, dim, kn);
            }
        }
        p.main_qgroups = pick_qgroups(p.main_qblocks, dim, kn, p.tile_q);
        const MainPassChoice choice = choose_main_pass(p, k, sample, sample_alt, sample_big, kn, narrow);
        sample = choice.sample;
        const int64_t R = choice.ranges;
        p.sample_tiles = (int)sample;
        p.sample_stride = std::max<int64_t>(1, p.full_tiles / sample);
        p.sample_ranges = best_sample_ranges(sample, p.qblocks, p.qgroups, p.grid);
        p.ranges = (int)R;
        p.item_a = choice.item_a;
        p.item_b = choice.item_b;
        p.opt_rank = choice.opt_rank;
        // ranges that hold items of a phase: the candidate segments (a range started in phase A keeps phase A's capacity)
        const int nrc_p = NUM_XCD / p.main_qgroups, qb_per_p = p.main_qblocks / p.main_qgroups;
        const int64_t items_p = R / nrc_p * qb_per_p;
        const int64_t RA = p.item_a ? std::min<int64_t>(R, (int64_t)nrc_p * ((p.item_a + qb_per_p - 1) / qb_per_p)) : 0;
        const int64_t RB = p.item_b ? std::min<int64_t>(R, (int64_t)nrc_p * ((p.item_b + qb_per_p - 1) / qb_per_p)) : 0;
        const double ratio = (double)p.tiles / (double)sample;
        const double expect = (double)k * ratio * 1.3 + 64.0;  // survivors per query under the sample thresholds alone
        // Sub-list capacities, one per phase (segment).  A phase whose thresholds were taken from a corpus fraction g lets
        // k / g rows per query through, spread over R * sublists sub-lists.  Two noise terms on top of that mean:
        //   the threshold itself is the k-th order statistic of what was seen: its pass rate is Gamma(k)-distributed, an
        //     upper quantile of it is (k + 4.75 sqrt(k) + 8) / k times the mean (15 x at k = 1, 1.6 x at k = 100, 1.16 x at
        //     k = 1000) -- times 1.25 for uneven data;
        //   the count of one sub-list is Poisson around its share: + 6 sigma + 16.
        // An overflowing sub-list only flags its query for the exact path, so the model needs to be safe, not a bound.
        const double fs = 1.0 / ratio, nlists = (double)R * p.sublists;
        const double xk = 1.25 * ((double)k + 4.75 * sqrt((double)k) + 8.0) / (double)k;
        // Allowance: one WHOLE tile may pass for a query (a corpus in topical order: all 256 rows of a tile belong to the
        // query's cluster); that is TILE_DOCS / sublists rows for one sub-list on top of its regular share.
        const int64_t tile_rows = TILE_DOCS / p.sublists;
        // estimated thresholds: about opt_rank / fs rows pass, Gamma(opt_rank)-distributed
        const double xr = p.opt_rank ? 1.25 * ((double)p.opt_rank + 4.75 * sqrt((double)p.opt_rank) + 8.0) / (double)p.opt_rank : 0.0;
        auto cap_for = [&](double g) -> int {
            const double mean = p.opt_rank ? xr * std::max((double)p.opt_rank / fs, (double)k) / nlists : xk * (double)k / (g * nlists);
            const int64_t c = (int64_t)(mean + 6.0 * sqrt(mean)) + 16 + tile_rows;   // the model's share PLUS one whole tile
            return (int)round_up(std::min<int64_t>(std::max<int64_t>(c, 16), 8192), 4);
        };
        // corpus fractions the re-tightenings have SEEN: the ranges completed by then
        const double fa = p.item_a ? (double)(p.item_a / qb_per_p) * nrc_p / (double)R : 0.0;
        const double fb = p.item_b ? std::max(0.0, (double)(p.item_b / qb_per_p) * nrc_p / (double)R - fa) : 0.0;
        (void)items_p;
        // the re-tightening selects among at most `compact` of the candidates found so far (launch_threshold_update): when
        // phase A is expected to leave more than that, the bound is the k-th of a subset and passes proportionally more
        const double upd_compact = std::min(32768.0, std::max(4096.0, 8.0 * pow2_ceil(k)));
        auto seen = [&](double g, double found) { return found > upd_compact ? g * upd_compact / found : g; };
        const double found_a = (double)k * fa / fs;
        CandLayout &L = p.cand;
        memset(&L, 0, sizeof(L));
        const int seg_end_v[3] = {RA ? (int)RA : (int)R, RB > RA ? (int)RB : (int)R, (int)R};
        int64_t recs = 0;
        L.nseg = 0;
        int prev_end = 0;
        for (int g = 0; g < 3; ++g) {
            L.seg_end[g] = INT32_MAX;
            L.cap[g] = 16;
            L.base[g] = recs;
        }
        for (int g = 0; g < 3; ++g) {
            if (seg_end_v[g] <= prev_end) continue;
            const int sg = L.nseg++;
            double frac_seen = fs;                                  // phase A: the sample
            if (sg == 1) frac_seen = fa >= 2.0 * fs ? seen(fa, found_a) : fs;
            if (sg == 2) frac_seen = fa >= 2.0 * fs ? seen(fa + fb, found_a + (double)k * fb / fa) : fs;
            L.seg_end[sg] = seg_end_v[g];
            L.cap[sg] = cap_for(std::min(1.0, std::max(frac_seen, fs)));
            L.base[sg] = recs;
            recs += (int64_t)(seg_end_v[g] - prev_end) * p.nq_pad * p.sublists * L.cap[sg];
            recs = round_up(recs, 32);
            prev_end = seg_end_v[g];
        }
        L.seg_end[L.nseg - 1] = INT32_MAX;   // the last segment is open-ended (sub-list lookups never fall off the table)
        p.cap = L.cap[0];
        // survivors per query reaching the select stage: ~12 k after the progressive re-tightening, `expect` without it
        // (estimated thresholds leave fewer candidates, but the select's staged re-score borrows this area: about 1.15 k rows are
        // re-scored per query and a 64-byte slice of each needs 80 bytes = 10 entries -- a smaller area sends the re-score down its
        // unstaged path: measured 2.41 instead of 2.06 ms at k = 1001)
        const double want_opt = std::max(xr * std::max((double)p.opt_rank / fs, (double)k) + 512.0, k > 256 ? 18.0 * k : 0.0);   // (18 entries per re-scored row: the 128-byte slices, where the LDS budget allows)
        p.select_compact = select_compact_entries(dim, p.ranges * p.sublists, p.rescore_cap,
                                                  (int64_t)((p.opt_rank ? want_opt : (RA ? 16.0 * k + 512.0 : expect)) * 1.25));
        // what the FIRST main pass fills: the tile kernels' (range, query) cells -- or, for the streaming kernel, one cell of two
        // sub-lists per query with whatever the candidate area gives each of them (at least 8 x the model's expectation)
        p.first_nsub = p.ranges * p.sublists;
        p.first_sp = p.sublists;
        p.first_lay = p.cand;
        if (p.narrow) {
            const double pass = p.opt_rank ? xr * std::max((double)p.opt_rank / fs, (double)k) : expect;
            int64_t capn = std::max<int64_t>(recs / ((int64_t)NARROW_SUBLISTS * std::max(1, n_q)), (int64_t)(8.0 * pass) + 4096);
            capn = std::min<int64_t>(capn, (int64_t)1 << 22) / 4 * 4;
            recs = std::max<int64_t>(recs, capn * NARROW_SUBLISTS * n_q);
            CandLayout &N = p.first_lay;
            memset(&N, 0, sizeof(N));
            N.nseg = 1;
            N.seg_end[0] = N.seg_end[1] = N.seg_end[2] = INT32_MAX;
            N.cap[0] = N.cap[1] = N.cap[2] = (int)capn;
            p.first_nsub = NARROW_SUBLISTS;
            p.first_sp = NARROW_SUBLISTS;
        }
        p.off_qnorm = take((size_t)p.nq_pad * 4);
        p.off_thr = take((size_t)p.nq_pad * 4 * 2);  // thr then delta
        p.off_gmax = take((size_t)p.sample_tiles * GROUPS_PER_TILE * p.nq_pad * 4);
        p.off_cnt = take((size_t)p.ranges * p.nq_pad * p.sublists * 4);
        p.off_cand = take((size_t)recs * 8);
        p.off_flag = take(64 + (size_t)n_q * 4);
        // the k best lower bounds per query between two re-tightenings (three-phase plans only)
        p.off_top = take(p.item_b ? ((size_t)p.nq_pad + (size_t)n_q * k) * 4 : 0);
        // estimated thresholds: room for the conservative bounds (+ margin coefficients) a failing search computes after the fact
        p.off_safe = take(p.opt_rank ? (size_t)p.nq_pad * 8 : 0);
        p.dense_rows_per_chunk = FALLBACK_ROWS;
        p.off_dense = take((size_t)FALLBACK_ROWS * n_rows * 4);
        // retry pass of flagged queries: compact query rows, thresholds + margins, the two lists, counts, second flag area
        p.off_retry = take((size_t)p.nq_pad * dim * 2 + (size_t)p.nq_pad * 8 + (size_t)n_q * 16 + 64 + 64 + (size_t)n_q * 4 + 256 * 10);
    } else {
        int64_t rows = (int64_t)(DENSE_SCRATCH_TARGET / ((size_t)n_rows * 4));
        rows = std::min<int64_t>(std::max<int64_t>(rows, 1), n_q);
        if (rows >= 64) rows = rows / 64 * 64;
        p.dense_rows_per_chunk = rows;
        p.off_flag = take(64 + (size_t)n_q * 4);   // queries the margin select hands to the fp64 path
        p.off_dense = take((size_t)rows * (n_rows + 3) * 4);
    }
    p.total = off;
    return p;
}

// Which main-pass kernel: CCR_MFMA16 pins it.  Otherwise the 16x16x32 kernel (the chip holds a higher clock on that MFMA shape)
// whenever its eight sub-lists per (range, query) do not force FEWER ranges than the 32x32x16 plan wants: up to k = 512 always
// (measured), above that when the four-sub-list plan's range count times 8 stays within the select stage's 2 048
// sub-lists (config-4 shard, k = 1000, 62 ranges: main pass 111 vs 120 ms, select 8.9 vs 9.4 ms).
Plan make_plan(int64_t n_rows, int dim, int n_q, int k, int flags, int num_cu, const Knobs &kn) {
    if (kn.mfma16 >= 0) return make_plan_for(n_rows, dim, n_q, k, flags, num_cu, kn, kn.mfma16 ? 1 : 0);
    if (!CCR_MFMA16_DEFAULT) return make_plan_for(n_rows, dim, n_q, k, flags, num_cu, kn, 0);
    if (k <= 512) return make_plan_for(n_rows, dim, n_q, k, flags, num_cu, kn, 1);
    const Plan p32 = make_plan_for(n_rows, dim, n_q, k, flags, num_cu, kn, 0);
    if (p32.fused && p32.ranges * 8 <= 2048) return make_plan_for(n_rows, dim, n_q, k, flags, num_cu, kn, 1);
    return p32;
}

}  // namespace ccr

using namespace ccr;

// Per-device slab of 256-byte slots for the indices' small device state: building an index per active-learning
// step must not cost a hipMalloc / hipFree pair (hipFree synchronises the device).
namespace {
constexpr int SLAB_SLOTS = 1024, SLAB_SLOT_BYTES = 256, MAX_DEVICES = 64;
struct Slab {
    char *base = nullptr;
    char *host = nullptr;   // pinned host twin (one 64-byte line per slot): where an asynchronous search leaves its flag count
    std::vector<int> free_slots;
};
constexpr int SLAB_HOST_BYTES = 64;
std::mutex g_slab_mutex;
Slab g_slabs[MAX_DEVICES];

int slab_take(int device, uint32_t **out, volatile uint32_t **host_out) {
    std::lock_guard<std::mutex> lock(g_slab_mutex);
    CCR_REQUIRE(device >= 0 && device < MAX_DEVICES, "device ordinal %d out of range", device);
    Slab &sl = g_slabs[device];
    if (!sl.base) {
        CCR_HIP_CHECK(hipMalloc((void **)&sl.base, (size_t)SLAB_SLOTS * SLAB_SLOT_BYTES));
        if (hipHostMalloc((void **)&sl.host, (size_t)SLAB_SLOTS * SLAB_HOST_BYTES, hipHostMallocDefault) != hipSuccess) {
            (void)hipFree(sl.base);
            sl.base = nullptr;
            set_error("ccr_index_create: hipHostMalloc of the pinned flag lines failed");
            return CCR_ERR_HIP;
        }
        for (int i = SLAB_SLOTS - 1; i >= 0; --i) sl.free_slots.push_back(i);
    }
    if (sl.free_slots.empty()) {
        set_error("ccr_index_create: more than %d live indices on device %d", SLAB_SLOTS, device);
        return CCR_ERR_INVALID;
    }
    *out = reinterpret_cast<uint32_t *>(sl.base + (size_t)sl.free_slots.back() * SLAB_SLOT_BYTES);
    *host_out = reinterpret_cast<volatile uint32_t *>(sl.host + (size_t)sl.free_slots.back() * SLAB_HOST_BYTES);
    sl.free_slots.pop_back();
    return CCR_OK;
}

void slab_give(int device, uint32_t *p) {
    if (!p || device < 0 || device >= MAX_DEVICES) return;
    std::lock_guard<std::mutex> lock(g_slab_mutex);
    Slab &sl = g_slabs[device];
    sl.free_slots.push_back((int)((reinterpret_cast<char *>(p) - sl.base) / SLAB_SLOT_BYTES));
}
}  // namespace

extern "C" const char *ccr_last_error(void) { return g_err; }
extern "C" int ccr_version(void) { return 100; }

extern "C" int ccr_index_destroy(ccr_index *ix);

// Per-device cache of the indices' tile-norm arrays (4 B per 256 corpus rows): an index built per step gets the block the
// previous one gave back instead of a hipMalloc / hipFree pair.
namespace {
struct CachedBlock {
    void *p;
    size_t bytes;
};
std::vector<CachedBlock> g_blocks[MAX_DEVICES];
constexpr size_t BLOCK_CACHE_ENTRIES = 8;

int block_take(int device, size_t bytes, void **out, size_t *got) {
    CCR_REQUIRE(device >= 0 && device < MAX_DEVICES, "device ordinal %d out of range", device);
    bytes = (bytes + 4095) / 4096 * 4096;
    {
        std::lock_guard<std::mutex> lock(g_slab_mutex);
        auto &v = g_blocks[device];
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i].bytes >= bytes && v[i].bytes <= 2 * bytes) {
                *out = v[i].p;
                *got = v[i].bytes;
                v.erase(v.begin() + i);
                return CCR_OK;
            }
    }
    CCR_HIP_CHECK(hipMalloc(out, bytes));
    *got = bytes;
    return CCR_OK;
}

void block_give(int device, void *p, size_t bytes) {
    if (!p) return;
    void *drop = nullptr;
    if (device >= 0 && device < MAX_DEVICES && bytes <= ((size_t)64 << 20)) {   // larger blocks are not worth pinning
        std::lock_guard<std::mutex> lock(g_slab_mutex);
        auto &v = g_blocks[device];
        v.push_back({p, bytes});
        if (v.size() > BLOCK_CACHE_ENTRIES) {   // oldest out
            drop = v.front().p;
            v.erase(v.begin());
        }
    } else {
        drop = p;
    }
    if (drop) (void)hipFree(drop);
}
}  // namespace

// row_bounds (device, n_rows floats, or null): upper bounds of the packed rows' norms as ccr_pack_bf16_ex wrote them (borrowed for
// the life of the index: the select stage reads them per candidate); without them the index makes its own pass over the shard.
static int index_create_impl(const uint16_t *D_bf16, int64_t n_rows, int dim, int64_t global_row_offset, const float *row_bounds,
                             void *stream, ccr_index **out) {
    CCR_REQUIRE(D_bf16 && out, "ccr_index_create: null pointer");
    CCR_REQUIRE(n_rows >= 1 && n_rows < ((int64_t)1 << 32), "ccr_index_create: n_rows=%lld out of range [1, 2^32)",
                (long long)n_rows);
    CCR_REQUIRE(dim >= 8 && dim % 8 == 0 && dim <= 4096, "ccr_index_create: dim=%d must be a multiple of 8 in [8, 4096]", dim);
    CCR_REQUIRE((uintptr_t)D_bf16 % 16 == 0, "ccr_index_create: corpus pointer must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    ccr_index *ix = new ccr_index();
    memset(ix, 0, sizeof(*ix));
    ix->D = D_bf16;
    ix->n_rows = n_rows;
    ix->dim = dim;
    ix->offset = ix->id_out = global_row_offset;
    ix->knobs = read_knobs();
    // on any failure below the partially built index is released before returning
    auto build = [&]() -> int {
        CCR_HIP_CHECK(hipGetDevice(&ix->device));
        CCR_HIP_CHECK(hipDeviceGetAttribute(&ix->num_cu, hipDeviceAttributeMultiprocessorCount, ix->device));
        int rc = slab_take(ix->device, &ix->dmax_bits, &ix->host_flags);
        if (rc != CCR_OK) return rc;
        const int64_t tiles = (n_rows + TILE_DOCS - 1) / TILE_DOCS;
        void *tn = nullptr;
        rc = block_take(ix->device, (size_t)tiles * 4, &tn, &ix->tile_bytes);
        if (rc != CCR_OK) return rc;
        ix->tile_norm = (float *)tn;
        uint32_t *tile_bits = reinterpret_cast<uint32_t *>(ix->tile_norm);   // non-negative floats: bit patterns order as unsigned
        if (row_bounds) {   // no pass over the shard, no synchronisation (max_tile_norm_kernel WRITES the maximum: no memset)
            ix->row_norm = row_bounds;
            rc = launch_tile_norms(row_bounds, n_rows, tile_bits, ix->dmax_bits, s);
            if (rc != CCR_OK) return rc;
        } else {
            void *rn = nullptr;
            rc = block_take(ix->device, (size_t)n_rows * 4, &rn, &ix->row_bytes);
            if (rc != CCR_OK) return rc;
            ix->row_norm = ix->row_norm_own = (float *)rn;
            CCR_HIP_CHECK(hipMemsetAsync(ix->dmax_bits, 0, 4, s));   // (this path max-accumulates with atomics)
            CCR_HIP_CHECK(hipMemsetAsync(tile_bits, 0, (size_t)tiles * 4, s));
            rc = launch_row_norms_bf16(D_bf16, n_rows, dim, ix->row_norm_own, ix->dmax_bits, tile_bits, s);
            if (rc != CCR_OK) return rc;
            CCR_HIP_CHECK(hipStreamSynchronize(s));
        }
        return CCR_OK;
    };
    const int rc = build();
    if (rc != CCR_OK) {
        ccr_index_destroy(ix);
        return rc;
    }
    *out = ix;
    return CCR_OK;
}

extern "C" int ccr_index_create(const uint16_t *D_bf16, int64_t n_rows, int dim, int64_t global_row_offset, void *stream,
                                ccr_index **out) {
    return index_create_impl(D_bf16, n_rows, dim, global_row_offset, nullptr, stream, out);
}

extern "C" int ccr_index_create_with_norms(const uint16_t *D_bf16, int64_t n_rows, int dim, int64_t global_row_offset,
                                           const float *row_norm_bounds, void *stream, ccr_index **out) {
    CCR_REQUIRE(row_norm_bounds, "ccr_index_create_with_norms: null row_norm_bounds");
    return index_create_impl(D_bf16, n_rows, dim, global_row_offset, row_norm_bounds, stream, out);
}

extern "C" int ccr_index_destroy(ccr_index *ix) {
    if (!ix) return CCR_OK;
    // an asynchronous search the caller never finished: its kernels still read the index's arrays -- wait for THAT search (not
    // for the stream) before the blocks go back to the cache; queries it flagged beyond the on-stream chunk stay un-redone
    if (ix->pending.active && ix->have_events) (void)hipEventSynchronize(ix->ev[7]);
    ix->pending.active = false;
    // a slot / block handed back may be rewritten by the next index's create on ITS stream; the caller destroys an index only
    // after the work that uses it has completed (the same contract as for the borrowed corpus)
    slab_give(ix->device, ix->dmax_bits);
    block_give(ix->device, ix->tile_norm, ix->tile_bytes);
    block_give(ix->device, ix->row_norm_own, ix->row_bytes);
    if (ix->have_events)
        for (int i = 0; i < 8; ++i)
            if (ix->ev[i]) (void)hipEventDestroy(ix->ev[i]);
    delete ix;
    return CCR_OK;
}

extern "C" int64_t ccr_index_rows(const ccr_index *ix) { return ix ? ix->n_rows : -1; }
extern "C" int ccr_index_dim(const ccr_index *ix) { return ix ? ix->dim : -1; }

extern "C" size_t ccr_search_workspace_bytes(const ccr_index *ix, int n_q, int k) {
    if (!ix || n_q <= 0 || k <= 0) return 0;
    // the larger of the three plans, so that flags may be chosen at call time
    Plan a = make_plan(ix->n_rows, ix->dim, n_q, k, CCR_SEARCH_DEFAULT, ix->num_cu, ix->knobs);
    Plan b = make_plan(ix->n_rows, ix->dim, n_q, k, CCR_SEARCH_FORCE_DENSE, ix->num_cu, ix->knobs);
    Plan c = make_plan(ix->n_rows, ix->dim, n_q, k, CCR_SEARCH_FORCE_FUSED, ix->num_cu, ix->knobs);
    return std::max({a.total, b.total, c.total}) + 256;
}

extern "C" int ccr_search_stream_wait_main_pass(const ccr_index *ix, void *stream) {
    CCR_REQUIRE(ix, "ccr_search_stream_wait_main_pass: null index");
    if (!ix->main_pass_recorded) return CCR_OK;   // dense path / no search yet: nothing to order against
    CCR_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, ix->ev[4], 0));
    return CCR_OK;
}

extern "C" int ccr_search_last_stats(const ccr_index *ix, ccr_search_stats *stats) {
    CCR_REQUIRE(ix && stats, "ccr_search_last_stats: null pointer");
    CCR_REQUIRE(!ix->pending.active, "ccr_search_last_stats: an asynchronous search is pending (call ccr_search_finish first)");
    *stats = ix->stats;
    return CCR_OK;
}

// Exact dense path for n queries: the rows d_qlist[lo .. lo + n) (or q_begin + lo ... when d_qlist is null), results to the
// same rows of the outputs.
static int dense_for_list(const ccr_index *ix, const uint16_t *Q, const uint32_t *d_qlist, int q_begin, int n, int k,
                          float *scratch, int64_t rows_per_chunk, float *out_scores, int64_t *out_ids, hipStream_t s) {
    for (int lo = 0; lo < n; lo += (int)rows_per_chunk) {
        const int m = std::min<int64_t>(rows_per_chunk, n - lo);
        int rc = launch_dense_scores(ix->D, ix->n_rows, ix->dim, Q, d_qlist ? d_qlist + lo : nullptr, q_begin + lo, m, nullptr,
                                     scratch, s);
        if (rc != CCR_OK) return rc;
        rc = launch_dense_select(scratch, ix->n_rows, k, d_qlist ? d_qlist + lo : nullptr, q_begin + lo, m, nullptr, ix->id_out,
                                 out_scores, out_ids, s);
        if (rc != CCR_OK) return rc;
    }
    return CCR_OK;
}

// The same for inner-product scores the fast way: MFMA score rows of the chunk (EPI_STORE of the fused kernel) + margin select
// (ccr_dense.hip).  Queries are the rows Qc[0 .. n) -- contiguous: the caller gathers a list first -- results go to rows
// out_rows[i] (or q_begin + i).  Queries the margin select cannot finish are appended to flag_list (flag_count is NOT reset here).
static bool margin_path_ok(const ccr_index *ix, int k) { return ix->dim % 8 == 0 && k <= MAX_K; }

static int margin_for_rows(const ccr_index *ix, const uint16_t *Qc, const float *hint, const uint32_t *out_rows, int q_begin, int n, int k, float *scratch,
                           size_t scratch_bytes, float *out_scores, int64_t *out_ids, uint32_t *flag_count, uint32_t *flag_list,
                           hipStream_t s) {
    const int64_t pitch = round_up(ix->n_rows, 4);
    int64_t chunk = (int64_t)(scratch_bytes / ((size_t)pitch * 4));   // score rows the scratch holds: any number of query blocks per launch
    CCR_REQUIRE(chunk >= 1, "margin path: no room for one score row");
    if (chunk > TILE_Q) chunk = chunk / TILE_Q * TILE_Q;
    const int grid = std::max(NUM_XCD, ix->num_cu / NUM_XCD * NUM_XCD);
    for (int lo = 0; lo < n; lo += (int)chunk) {
        const int m = std::min<int64_t>(chunk, n - lo);
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.D = ix->D;
        g.n_rows = ix->n_rows;
        g.dim = ix->dim;
        g.Q = Qc + (int64_t)lo * ix->dim;
        g.n_q = m;
        g.nq_pad = (int)round_up(m, TILE_Q);
        g.qblocks = g.nq_pad / TILE_Q;
        g.n_vt = (ix->n_rows + TILE_DOCS - 1) / TILE_DOCS;
        g.tile_stride = 1;
        g.ranges = (int)round_up(std::min<int64_t>(std::max<int64_t>(1, g.n_vt / 4), 1024), NUM_XCD);   // ~4 tiles per work item
        g.qgroups = 1;
        g.item_begin = 0;
        g.item_end = INT32_MAX;
        g.store = scratch;
        g.store_pitch = pitch;
        g.stagger = 1;
        g.qdirect = ix->knobs.qdirect;
        int rc = (ix->knobs.mfma16 >= 0 ? ix->knobs.mfma16 : CCR_MFMA16_DEFAULT) ? launch_gemm16_store(g, grid, s) : launch_gemm_store(g, grid, s);
        if (rc != CCR_OK) return rc;
        rc = launch_margin_select(scratch, pitch, ix->n_rows, k, ix->dim, g.Q, ix->D, ix->tile_norm, ix->row_norm, ix->dmax_bits,
                                  hint ? hint + lo : nullptr, out_rows ? out_rows + lo : nullptr, q_begin + lo, m, ix->id_out, out_scores, out_ids, flag_count, flag_list, s);
        if (rc != CCR_OK) return rc;
    }
    return CCR_OK;
}

// The main pass: one launch per phase (work items [begin, end) of every XCD set), the thresholds re-tightened from the candidates
// of the ranges completed so far between two launches when `retighten` is set.
static int run_main_pass(const ccr_index *ix, const Plan &p, GemmArgs gm, uint2 *cand, uint32_t *cnt, float *thr, const float *cq,
                         uint32_t *top, int n_q, int k, bool retighten, hipStream_t s) {
    const int nrc = NUM_XCD / gm.qgroups, qb_per = gm.qblocks / gm.qgroups;
    const int items = p.ranges / nrc * qb_per;
    const int bounds[4] = {0, retighten ? p.item_a : 0, retighten ? p.item_b : 0, items};
    gm.cand = cand;
    gm.lay = p.cand;
    gm.cq = cq;
    gm.tile_norm = ix->tile_norm;
    int done = 0, updates = 0;
    int prev_full = 0, prev_part = 0, prev_blocks = 0;
    for (int ph = 1; ph < 4; ++ph) {
        if (bounds[ph] <= done) continue;
        if (done > 0) {
            const int rl_full = done / qb_per, part = done % qb_per;   // complete range rows; blocks done of the started one
            const int full = rl_full * nrc * p.sublists, partial = part ? (rl_full + 1) * nrc * p.sublists : 0;
            const bool more = bounds[ph] < items && ph < 3;   // another re-tightening follows this phase
            const int rc = launch_threshold_update(cand, cnt, full, partial, part, prev_full, prev_part, prev_blocks, qb_per, p.sublists, n_q,
                                                   p.nq_pad, p.cand, k, cq, ix->tile_norm, top, updates > 0, more, thr, s, p.tile_q);
            if (rc != CCR_OK) return rc;
            prev_full = full;
            prev_part = partial;
            prev_blocks = part;
            ++updates;
        }
        gm.item_begin = done;
        gm.item_end = bounds[ph];
        const int rc = p.tile_q == WIDE_Q ? launch_gemm16w_filter(gm, p.grid, s)
                                          : (p.mfma16 ? launch_gemm16_filter(gm, p.grid, s) : launch_gemm_filter(gm, p.grid, s));
        if (rc != CCR_OK) return rc;
        done = bounds[ph];
    }
    return CCR_OK;
}

// Completion of a fused search: read how many queries the select stage flagged and re-do them.
//   * a query whose candidates were dropped (sub-list overflow: a corpus in topical order floods the lists of the query's
//     own cluster) is RETRIED on the fused path: its truncated lists still hold real rows, so the k-th largest of them is a
//     valid -- and now nearly final -- threshold; the flagged queries are compacted and one more main pass + select runs
//     for them alone (about one corpus pass per 256 flagged queries instead of 1.8 ms of fp64 scoring per query);
//   * a query the selection cannot finish (mass ties around the cut) and a query flagged again by the retry take the exact
//     dense path.
// Synchronises the stream.
static int search_complete(ccr_index *ix) {
    auto &pd = ix->pending;
    const Plan &p = ix->plan;
    hipStream_t s = pd.stream;
    char *ws = pd.ws;
    const bool was_async = pd.active;
    pd.active = false;
    struct {
        uint32_t nflag, pad;
        unsigned long long ncand;
    } host;
    if (was_async) {
        // the search itself copied its flag line to pinned host memory and recorded ev[7] behind it: wait for THAT event, not for
        // the stream -- work enqueued after the search (the next step's pack and search, the exchange) is not waited for
        CCR_HIP_CHECK(hipEventSynchronize(ix->ev[7]));
        host.nflag = ix->host_flags[0];
        host.ncand = *reinterpret_cast<volatile unsigned long long *>(ix->host_flags + 2);
    } else {
        CCR_HIP_CHECK(hipMemcpyAsync(&host, ws + p.off_flag, sizeof(host), hipMemcpyDeviceToHost, s));
        CCR_HIP_CHECK(hipStreamSynchronize(s));
    }
    {
        float a = 0, b = 0;
        CCR_HIP_CHECK(hipEventElapsedTime(&ix->stats.ms_sample, ix->ev[1], ix->ev[2]));
        CCR_HIP_CHECK(hipEventElapsedTime(&a, ix->ev[0], ix->ev[1]));
        CCR_HIP_CHECK(hipEventElapsedTime(&b, ix->ev[2], ix->ev[3]));
        ix->stats.ms_threshold = a + b;
        CCR_HIP_CHECK(hipEventElapsedTime(&ix->stats.ms_main, ix->ev[3], ix->ev[4]));
        CCR_HIP_CHECK(hipEventElapsedTime(&ix->stats.ms_select, ix->ev[4], ix->ev[5]));
        CCR_HIP_CHECK(hipEventElapsedTime(&ix->stats.ms_total, ix->ev[0], ix->ev[5]));
    }
    ix->stats.path = 1;
    ix->stats.n_fallback = (int32_t)host.nflag;
    ix->stats.sample_tiles = p.sample_tiles;
    ix->stats.ranges = p.narrow ? 1 : p.ranges;                      // (streaming main pass of a small batch: one range, two sub-lists per query)
    ix->stats.cap = p.narrow ? p.first_lay.cap[0] : p.cap;
    ix->stats.sublists = p.narrow ? p.first_sp : p.sublists;
    ix->stats.main_launches = 1 + (p.item_a ? 1 : 0) + (p.item_b ? 1 : 0);
    ix->stats.opt_rank = p.opt_rank;
    ix->stats.main_tile_queries = p.narrow ? 0 : p.tile_q;
    ix->stats.n_candidates = (int64_t)host.ncand;
    const int begin = 0;
    ix->stats.n_dense = 0;
    if (host.nflag == 0) return CCR_OK;

    const int n_q = pd.n_q, k = pd.k;
    uint32_t *flag_list = (uint32_t *)(ws + p.off_flag + 64);
    float *thr = (float *)(ws + p.off_thr);
    float *delta = thr + p.nq_pad;   // cq: margin coefficients gamma ||q|| (the per-tile margin is cq * tile norm)
    uint32_t *cnt = (uint32_t *)(ws + p.off_cnt);
    uint2 *cand = (uint2 *)(ws + p.off_cand);
    float *dense_scratch = (float *)(ws + p.off_dense);
    // retry area: [Q2 nq_pad x dim bf16][thr2 nq_pad][delta2 nq_pad][retry_list n_q][dense_list n_q][counts 64 B][flag2 64 B + n_q]
    char *ra = ws + p.off_retry;
    auto carve = [&](size_t bytes) {
        char *o = ra;
        ra += (bytes + 255) / 256 * 256;
        return o;
    };
    uint16_t *Q2 = (uint16_t *)carve((size_t)p.nq_pad * ix->dim * 2);
    float *thr2 = (float *)carve((size_t)p.nq_pad * 4);
    float *delta2 = (float *)carve((size_t)p.nq_pad * 4);
    uint32_t *retry_list = (uint32_t *)carve((size_t)n_q * 4);
    uint32_t *dense_list = (uint32_t *)carve((size_t)n_q * 4);
    uint32_t *counts = (uint32_t *)carve(64);
    uint32_t *flag2 = (uint32_t *)carve(64 + (size_t)n_q * 4);

    uint32_t *list_b = (uint32_t *)carve((size_t)n_q * 4);   // the rounds of a group ping-pong between these two lists
    uint32_t *list_c = (uint32_t *)carve((size_t)n_q * 4);
    int rc = CCR_OK;
    if (p.opt_rank) {
        // Estimated thresholds: a query for which fewer than k rows passed has no list to take a bound from (the select left
        // thr = -inf and FLAG_DENSE).  The sample's group maxima are still in the workspace: the CONSERVATIVE threshold (the k-th
        // largest, a valid bound) is computed now -- only on this path -- and those queries join the retry under it.
        float *thr_safe = (float *)(ws + p.off_safe);
        rc = launch_threshold((const float *)(ws + p.off_gmax), (int64_t)p.sample_tiles * GROUPS_PER_TILE, n_q, p.nq_pad, k,
                              (const float *)(ws + p.off_qnorm), ix->dmax_bits, ix->dim, ix->tile_norm, p.sample_stride, thr_safe,
                              thr_safe + p.nq_pad, s);
        if (rc != CCR_OK) return rc;
        rc = launch_underfilled_to_retry(flag_list, begin, (int)host.nflag, thr_safe, thr, s);
        if (rc != CCR_OK) return rc;
    }
    rc = launch_partition_flags(flag_list, begin, (int)host.nflag, retry_list, dense_list, counts, s);
    if (rc != CCR_OK) return rc;
    uint32_t hc[2] = {0, 0};
    CCR_HIP_CHECK(hipMemcpyAsync(hc, counts, 8, hipMemcpyDeviceToHost, s));
    CCR_HIP_CHECK(hipStreamSynchronize(s));
    const int n_retry = (int)hc[0];
    int n_dense = (int)hc[1];
    int n_again = 0;   // queries the retry could not finish (they join the dense list)
    const int64_t area_recs = (int64_t)((p.off_flag - p.off_cand) / 8);   // records the candidate area holds
    const int nsub_all = p.ranges * p.sublists;
    auto to_dense = [&](const uint32_t *list, int n) -> int {
        CCR_HIP_CHECK(hipMemcpyAsync(dense_list + n_dense, list, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        n_dense += n;
        n_again += n;
        return CCR_OK;
    };
    if (n_retry > 0 && nsub_all > 2048) {
        rc = to_dense(retry_list, n_retry);
        if (rc != CCR_OK) return rc;
    } else if (n_retry > 0) {
        // thresholds re-tightened from everything the first attempt recorded (truncated lists included) -- for every flagged
        // query at once, before the candidate area is reused
        rc = launch_threshold_update(cand, cnt, p.first_nsub, 0, 0, 0, 0, 0, 1, p.first_sp, n_q, p.nq_pad, p.first_lay, k, delta, ix->tile_norm, nullptr, false,
                                     false, thr, s);
        if (rc != CCR_OK) return rc;
        // The flagged queries are retried in GROUPS that get the whole candidate area to themselves: the fewer queries share
        // it, the larger every sub-list.  A group is as large as still leaves four times the first attempt's capacity (when
        // most of a batch is flagged -- the lists were flooded, not unlucky -- one group per query block: about one corpus
        // pass of ONE block per 256 queries instead of 0.65 ms of fp64 scoring per query).
        const int64_t want_cap = std::min<int64_t>(8192, 4 * (int64_t)p.cap);
        int64_t group = area_recs / ((int64_t)nsub_all * want_cap) / TILE_Q * TILE_Q;
        group = std::max<int64_t>(TILE_Q, std::min<int64_t>(group, round_up(n_retry, TILE_Q)));
        for (int g0 = 0; g0 < n_retry; g0 += (int)group) {
            const uint32_t *cur = retry_list + g0;
            int n_cur = std::min<int>((int)group, n_retry - g0);
            const uint32_t *prev_list = nullptr;
            CandLayout prev_lay = p.cand;
            int prev_n = 0, prev_pad = 0;
            for (int round = 0; round < 3 && n_cur > 0; ++round) {
                if (round > 0) {   // re-tighten from the previous round's (truncated) lists of this group
                    rc = launch_threshold_update(cand, cnt, nsub_all, 0, 0, 0, 0, 0, 1, p.sublists, prev_n, prev_pad, prev_lay, k, delta2, ix->tile_norm,
                                                 nullptr, false, false, thr2, s);
                    if (rc != CCR_OK) return rc;
                    rc = launch_scatter_thresholds(prev_list, prev_n, thr2, thr, s);
                    if (rc != CCR_OK) return rc;
                }
                rc = launch_gather_queries(pd.Q, ix->dim, cur, n_cur, thr, delta, Q2, thr2, delta2, s);
                if (rc != CCR_OK) return rc;
                const int pad2 = (int)round_up(n_cur, TILE_Q);
                int64_t cap2 = area_recs / ((int64_t)nsub_all * pad2);
                cap2 = std::min<int64_t>(8192, cap2 / 4 * 4);
                if (cap2 < 16) break;
                CandLayout lay2;
                memset(&lay2, 0, sizeof(lay2));
                lay2.nseg = 1;
                lay2.seg_end[0] = lay2.seg_end[1] = lay2.seg_end[2] = INT32_MAX;
                lay2.cap[0] = lay2.cap[1] = lay2.cap[2] = (int)cap2;
                CCR_HIP_CHECK(hipMemsetAsync(cnt, 0, (size_t)nsub_all * pad2 * 4, s));
                CCR_HIP_CHECK(hipMemsetAsync(flag2, 0, 64, s));
                GemmArgs g;
                memset(&g, 0, sizeof(g));
                g.D = ix->D;
                g.n_rows = ix->n_rows;
                g.dim = ix->dim;
                g.Q = Q2;
                g.n_q = n_cur;
                g.nq_pad = pad2;
                g.qblocks = pad2 / TILE_Q;
                g.qgroups = pick_qgroups(g.qblocks, ix->dim, ix->knobs);
                g.stagger = ix->knobs.stagger;
                g.qdirect = ix->knobs.qdirect;
                g.n_vt = p.tiles;
                g.tile_stride = 1;
                g.ranges = p.ranges;
                g.thr = thr2;
                g.cq = delta2;
                g.tile_norm = ix->tile_norm;
                g.cnt = cnt;
                g.cand = cand;
                g.lay = lay2;
                g.item_begin = 0;
                g.item_end = INT32_MAX;
                rc = p.mfma16 ? launch_gemm16_filter(g, p.grid, s) : launch_gemm_filter(g, p.grid, s);
                if (rc != CCR_OK) return rc;
                rc = launch_select_rescore(cand, cnt, nsub_all, p.sublists, n_cur, pad2, lay2, k, p.rescore_cap, p.select_compact, ix->n_rows,
                                           thr2, delta2, ix->tile_norm, ix->row_norm, ix->dmax_bits, Q2, ix->D, ix->dim, ix->id_out, pd.out_scores, pd.out_ids, flag2,
                                           flag2 + 16, nullptr, cur, s);
                if (rc != CCR_OK) return rc;
                ix->stats.n_retried += n_cur;
                uint32_t again = 0;
                CCR_HIP_CHECK(hipMemcpyAsync(&again, flag2, 4, hipMemcpyDeviceToHost, s));
                CCR_HIP_CHECK(hipStreamSynchronize(s));
                prev_list = cur;
                prev_lay = lay2;
                prev_n = n_cur;
                prev_pad = pad2;
                if (again == 0) {
                    n_cur = 0;
                    break;
                }
                uint32_t *next = (cur == list_b) ? list_c : list_b;
                rc = launch_partition_flags(flag2 + 16, 0, (int)again, next, dense_list + n_dense, counts, s);
                if (rc != CCR_OK) return rc;
                CCR_HIP_CHECK(hipMemcpyAsync(hc, counts, 8, hipMemcpyDeviceToHost, s));
                CCR_HIP_CHECK(hipStreamSynchronize(s));
                n_dense += (int)hc[1];
                cur = next;
                n_cur = (int)hc[0];
                if (n_cur >= prev_n) break;   // no progress (every retried query overflowed again): the dense path takes them
            }
            if (n_cur > 0) {   // still flagged after the rounds of this group
                rc = to_dense(cur, n_cur);
                if (rc != CCR_OK) return rc;
            }
        }
    }
    // Many queries for the dense path: its 64 x 64 score tiles are a quarter full with the 16 reserved rows (2.2 instead of
    // 0.65 ms per NQ query), and the candidate area is free by now -- score 64 (or more) queries per chunk in there.
    int64_t chunk = p.dense_rows_per_chunk;
    {
        const int64_t fit = (int64_t)((p.off_flag - p.off_cand) / ((size_t)ix->n_rows * 4)) / 64 * 64;
        if (n_dense > chunk && fit >= 64) {
            chunk = std::min<int64_t>(fit, 256);
            dense_scratch = (float *)cand;
        }
    }
    if (n_dense > 0 && margin_path_ok(ix, k)) {
        // MFMA score rows + margin select first (0.9 ms of GEMM per 256 queries + one row scan each, against 0.65 ms of fp64 scoring
        // per query); only what that cannot finish -- more than 8 192 rows inside the margin, non-finite embeddings -- is left
        size_t room = (size_t)p.dense_rows_per_chunk * ix->n_rows * 4;
        float *scr = (float *)(ws + p.off_dense);
        if ((size_t)(p.off_flag - p.off_cand) > room) {   // the candidate area is free by now
            room = (size_t)(p.off_flag - p.off_cand);
            scr = (float *)cand;
        }
        CCR_HIP_CHECK(hipMemsetAsync(flag2, 0, 64, s));
        for (int lo = 0; lo < n_dense; lo += p.nq_pad) {   // Q2 holds nq_pad gathered query rows
            const int m = std::min(p.nq_pad, n_dense - lo);
            rc = launch_gather_queries(pd.Q, ix->dim, dense_list + lo, m, thr, delta, Q2, thr2, delta2, s);
            if (rc != CCR_OK) return rc;
            // (thr2: the gathered thresholds of these queries -- valid lower bounds of their k-th largest scores)
            rc = margin_for_rows(ix, Q2, thr2, dense_list + lo, 0, m, k, scr, room, pd.out_scores, pd.out_ids, flag2, flag2 + 16, s);
            if (rc != CCR_OK) return rc;
        }
        uint32_t left = 0;
        CCR_HIP_CHECK(hipMemcpyAsync(&left, flag2, 4, hipMemcpyDeviceToHost, s));
        CCR_HIP_CHECK(hipStreamSynchronize(s));
        if (left > 0) {
            rc = dense_for_list(ix, pd.Q, flag2 + 16, 0, (int)left, k, dense_scratch, chunk, pd.out_scores, pd.out_ids, s);
            if (rc != CCR_OK) return rc;
        }
    } else if (n_dense > 0) {
        rc = dense_for_list(ix, pd.Q, dense_list, 0, n_dense, k, dense_scratch, chunk, pd.out_scores, pd.out_ids, s);
        if (rc != CCR_OK) return rc;
    }
    ix->stats.n_dense += n_dense;   // (n_again of them after a retry)
    (void)n_again;
    CCR_HIP_CHECK(hipEventRecord(ix->ev[6], s));
    CCR_HIP_CHECK(hipStreamSynchronize(s));
    CCR_HIP_CHECK(hipEventElapsedTime(&ix->stats.ms_fallback, ix->ev[5], ix->ev[6]));
    ix->stats.ms_total += ix->stats.ms_fallback;
    return CCR_OK;
}

extern "C" int ccr_search_finish(ccr_index *ix) {
    CCR_REQUIRE(ix, "ccr_search_finish: null index");
    if (!ix->pending.active) return CCR_OK;
    return search_complete(ix);
}

// id_out: what the result ids are (ix->offset: int64 global ids; ID_LOCAL_U32: u32 local rows of a shard message).
// flagged_out (device, may be null; zero on entry): an ASYNCHRONOUS search leaves the select stage's flag count there, on the stream.
static int search_impl(ccr_index *ix, const uint16_t *Q_bf16, int n_q, int k, float *out_scores, int64_t *out_ids, int64_t id_out,
                       uint32_t *flagged_out, void *workspace, size_t ws_bytes, int flags, void *stream) {
    CCR_REQUIRE(ix && Q_bf16 && out_scores && out_ids, "ccr_search: null pointer");
    CCR_REQUIRE(n_q >= 0, "ccr_search: n_q=%d", n_q);
    CCR_REQUIRE(k >= 1 && k <= MAX_K && (int64_t)k <= ix->n_rows, "ccr_search: k=%d must be in [1, min(n_rows=%lld, %d)]", k,
                (long long)ix->n_rows, MAX_K);
    CCR_REQUIRE((uintptr_t)Q_bf16 % 16 == 0, "ccr_search: query pointer must be 16-byte aligned");
    CCR_REQUIRE(!ix->pending.active, "ccr_search: an asynchronous search is pending on this index (call ccr_search_finish first)");
    memset(&ix->stats, 0, sizeof(ix->stats));
    ix->main_pass_recorded = false;
    if (n_q == 0) return CCR_OK;
    hipStream_t s = (hipStream_t)stream;
    ix->id_out = id_out;
    const bool async = (flags & CCR_SEARCH_ASYNC) != 0;
    const int plan_flags = flags & ~CCR_SEARCH_ASYNC;
    if (!(ix->plan_nq == n_q && ix->plan_k == k && ix->plan_flags == plan_flags)) {
        ix->plan = make_plan(ix->n_rows, ix->dim, n_q, k, plan_flags, ix->num_cu, ix->knobs);
        ix->plan_nq = n_q;
        ix->plan_k = k;
        ix->plan_flags = plan_flags;
    }
    const Plan p = ix->plan;
    if (!workspace || ws_bytes < p.total || (uintptr_t)workspace % 256 != 0) {
        set_error("ccr_search: workspace %zu bytes (256-byte aligned) required, got %zu at %p", p.total, ws_bytes, workspace);
        return CCR_ERR_WORKSPACE;
    }
    char *ws = (char *)workspace;
    float *dense_scratch = (float *)(ws + p.off_dense);

    if (!ix->have_events) {   // phase-boundary events of the statistics, created on first use
        for (int i = 0; i < 8; ++i) CCR_HIP_CHECK(hipEventCreate(&ix->ev[i]));
        ix->have_events = true;
    }
    CCR_HIP_CHECK(hipEventRecord(ix->ev[0], s));
    if (!p.fused) {
        ix->stats.path = 0;
        int rc0 = CCR_OK;
        if (!async && !(flags & CCR_SEARCH_FORCE_DENSE) && margin_path_ok(ix, k)) {
            // MFMA score rows + margin select; what it cannot finish (mass ties, non-finite embeddings) takes the fp64 path
            uint32_t *fc = (uint32_t *)(ws + p.off_flag), *fl = (uint32_t *)(ws + p.off_flag + 64);
            CCR_HIP_CHECK(hipMemsetAsync(fc, 0, 64, s));
            rc0 = margin_for_rows(ix, Q_bf16, nullptr, nullptr, 0, n_q, k, dense_scratch, (size_t)p.dense_rows_per_chunk * (ix->n_rows + 3) * 4,
                                  out_scores, out_ids, fc, fl, s);
            if (rc0 != CCR_OK) return rc0;
            uint32_t nf = 0;
            CCR_HIP_CHECK(hipMemcpyAsync(&nf, fc, 4, hipMemcpyDeviceToHost, s));
            CCR_HIP_CHECK(hipStreamSynchronize(s));
            ix->stats.n_fallback = ix->stats.n_dense = (int32_t)nf;
            if (nf > 0) rc0 = dense_for_list(ix, Q_bf16, fl, 0, (int)nf, k, dense_scratch, p.dense_rows_per_chunk, out_scores, out_ids, s);
        } else {
            rc0 = dense_for_list(ix, Q_bf16, nullptr, 0, n_q, k, dense_scratch, p.dense_rows_per_chunk, out_scores, out_ids, s);
        }
        if (rc0 != CCR_OK) return rc0;
        CCR_HIP_CHECK(hipEventRecord(ix->ev[6], s));
        if (async) return CCR_OK;   // nothing to complete: the dense path has no flagged queries (no timing statistics either)
        CCR_HIP_CHECK(hipStreamSynchronize(s));
        CCR_HIP_CHECK(hipEventElapsedTime(&ix->stats.ms_total, ix->ev[0], ix->ev[6]));
        ix->stats.ms_fallback = ix->stats.ms_total;
        return CCR_OK;
    }

    float *qnorm = (float *)(ws + p.off_qnorm);
    float *thr = (float *)(ws + p.off_thr);
    float *delta = thr + p.nq_pad;
    float *gmax = (float *)(ws + p.off_gmax);
    uint32_t *cnt = (uint32_t *)(ws + p.off_cnt);
    uint2 *cand = (uint2 *)(ws + p.off_cand);
    uint32_t *flag_count = (uint32_t *)(ws + p.off_flag);
    unsigned long long *stat_cand = (unsigned long long *)(ws + p.off_flag + 8);
    uint32_t *flag_list = (uint32_t *)(ws + p.off_flag + 64);

    CCR_HIP_CHECK(hipMemsetAsync(flag_count, 0, 64, s));
    // every (range, query block) item with at least one tile writes its counters at its end (padded query columns included);
    // only a plan with more ranges than tiles (forced fused searches of tiny corpora) leaves counters unwritten
    if ((int64_t)p.ranges > p.tiles) CCR_HIP_CHECK(hipMemsetAsync(cnt, 0, (size_t)p.ranges * p.nq_pad * p.sublists * 4, s));
    int rc = launch_row_norms_bf16(Q_bf16, n_q, ix->dim, qnorm, nullptr, nullptr, s);
    if (rc != CCR_OK) return rc;

    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.D = ix->D;
    g.n_rows = ix->n_rows;
    g.dim = ix->dim;
    g.Q = Q_bf16;
    g.n_q = n_q;
    g.nq_pad = p.nq_pad;
    g.qblocks = p.qblocks;
    g.qgroups = p.qgroups;
#ifdef CCR_DIAGNOSTICS
    g.dbg = ix->knobs.gemm_dbg;
    g.dbg_pitch = env_int("CCR_DBG_PITCH", ix->dim);
    g.dbg_alloc_rows = getenv("CCR_DBG_ALLOC_ROWS") ? atoll(getenv("CCR_DBG_ALLOC_ROWS")) : 0;
    g.dbg_alloc_q = env_int("CCR_DBG_ALLOC_Q", 0);
#endif
    g.stagger = ix->knobs.stagger;
    g.qdirect = ix->knobs.qdirect;

    // sample pass -> group maxima -> thresholds
    GemmArgs gs = g;
    gs.n_vt = p.sample_tiles;
    gs.tile_stride = p.sample_stride;
    gs.ranges = p.sample_ranges;
    gs.gmax = gmax;
    gs.item_begin = 0;
    gs.item_end = INT32_MAX;
    CCR_HIP_CHECK(hipEventRecord(ix->ev[1], s));
    rc = launch_gemm_gmax(gs, p.grid, s);
    if (rc != CCR_OK) return rc;
    CCR_HIP_CHECK(hipEventRecord(ix->ev[2], s));
    // (small batches: the streaming pass's sub-list counters are cleared by the threshold launch)
    rc = launch_threshold(gmax, (int64_t)p.sample_tiles * GROUPS_PER_TILE, n_q, p.nq_pad, p.opt_rank ? p.opt_rank : k, qnorm, ix->dmax_bits,
                          ix->dim, ix->tile_norm, p.sample_stride, thr, delta, s, p.narrow ? cnt : nullptr, p.narrow ? NARROW_SUBLISTS : 0);
    if (rc != CCR_OK) return rc;

    // main pass -> candidates
    GemmArgs gm = g;
    gm.n_vt = p.tiles;
    gm.tile_stride = 1;
    gm.ranges = p.ranges;
    gm.thr = thr;
    gm.cnt = cnt;
    gm.qblocks = p.main_qblocks;   // (blocks of p.tile_q queries; the sample pass above walks blocks of TILE_Q)
    gm.qgroups = p.main_qgroups;
    gm.item_swap = (ix->knobs.item_swap && !p.item_a) ? 1 : 0;   // the phases' "ranges completed so far" needs the default order
    unsigned long long *stamps = nullptr;   // CCR_GEMM_DBG=16: in-kernel cycle stamps of the main pass (diagnostic build only)
#ifdef CCR_DIAGNOSTICS
    const bool want_stamps = ix->knobs.gemm_dbg == 16 || ix->knobs.gemm_dbg == 144;
#else
    const bool want_stamps = false;
#endif
    if (want_stamps) {
        CCR_HIP_CHECK(hipMalloc((void **)&stamps, (size_t)p.grid * 64 * 8));
        CCR_HIP_CHECK(hipMemsetAsync(stamps, 0, (size_t)p.grid * 64 * 8, s));
        gm.store = reinterpret_cast<float *>(stamps);
    }
    CCR_HIP_CHECK(hipEventRecord(ix->ev[3], s));
    if (p.narrow) {
        // small batch: the corpus is STREAMED past query rows resident in LDS (ccr_narrow.hip); two atomically filled sub-lists per query
        // (their counters were cleared by launch_threshold)
        NarrowArgs na;
        memset(&na, 0, sizeof(na));
        na.D = ix->D;
        na.n_rows = ix->n_rows;
        na.dim = ix->dim;
        na.Q = Q_bf16;
        na.n_q = n_q;
        na.q_stride = narrow_query_stride(ix->dim);
        na.thr = thr;
        na.cq = delta;
        na.tile_norm = ix->tile_norm;
        na.cand = cand;
        na.cnt = cnt;
        na.cap = p.first_lay.cap[0];
        na.groups = p.narrow_groups;
        // one workgroup (8 waves x 12 KiB of loads in flight) per CU: measured at NQ 6.2 / 6.2 / 6.0 TB/s of corpus bytes at n_q = 1 / 16 / 64;
        // two or four per CU -- the 16- and 32-query images would fit -- stream no faster (6.2 / 5.9 TB/s at 1 / 16)
        int ngrid = ix->knobs.narrow_grid > 0 ? ix->knobs.narrow_grid : ix->num_cu;
        const int64_t blocks = (ix->n_rows + 16 * NARROW_WAVES - 1) / (16 * NARROW_WAVES);
        if ((int64_t)ngrid > blocks) ngrid = (int)std::max<int64_t>(1, blocks);
        if (p.narrow_groups == 2) ngrid = std::max(16, ngrid / 16 * 16);   // whole sets of eight (b, b + 8) pairs; a pair shares its row stream
        rc = launch_narrow_filter(na, p.narrow, ngrid, ix->knobs.narrow_nt != 0, s);
    } else {
        rc = run_main_pass(ix, p, gm, cand, cnt, thr, delta, p.item_b ? (uint32_t *)(ws + p.off_top) : nullptr, n_q, k, true, s);
    }
    if (rc != CCR_OK) return rc;
    CCR_HIP_CHECK(hipEventRecord(ix->ev[4], s));
    ix->main_pass_recorded = true;
    if (want_stamps) {
        std::vector<unsigned long long> h((size_t)p.grid * 64);
        CCR_HIP_CHECK(hipMemcpyAsync(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost, s));
        CCR_HIP_CHECK(hipStreamSynchronize(s));
        static const char *names[8] = {"barrierB+loop", "epi_hits", "dma_wait", "lds_reads+dma_issue", "barrierA", "mfma", "epi_trees|lgkm_before_A", "mfma_drain"};
        for (int grp = 0; grp < 2; ++grp) {
            double sum[8] = {0};
            int n = 0;
            for (int b = 0; b < p.grid; ++b)
                for (int w = grp * 4; w < grp * 4 + 4; ++w, ++n)
                    for (int i = 0; i < 8; ++i) sum[i] += (double)h[((size_t)b * 8 + w) * 8 + i];
            double tot = 0;
            for (int i = 0; i < 8; ++i) tot += sum[i];
            fprintf(stderr, "[ccr stamps] waves %d-%d: total %.0f cycles/wave;", grp * 4, grp * 4 + 3, tot / n);
            for (int i = 0; i < 8; ++i) fprintf(stderr, " %s %.1f%%", names[i], 100.0 * sum[i] / tot);
            fprintf(stderr, "\n");
        }
        (void)hipFree(stamps);
    }

#ifdef CCR_DIAGNOSTICS
    if (ix->knobs.gemm_dbg != 0 && ix->knobs.gemm_dbg != 16) {
        // timing-only ablation of the main pass: its candidates are meaningless, so nothing is selected or re-done -- the outputs are
        // zeroed and the main pass's time by the library's own events goes to stderr (diagnostic library only)
        CCR_HIP_CHECK(hipMemsetAsync(out_scores, 0, (size_t)n_q * k * 4, s));
        CCR_HIP_CHECK(hipStreamSynchronize(s));
        float ms = 0.f;
        CCR_HIP_CHECK(hipEventElapsedTime(&ms, ix->ev[3], ix->ev[4]));
        fprintf(stderr, "[ccr diag] CCR_GEMM_DBG=%d main pass %.4f ms (%d launches)\n", ix->knobs.gemm_dbg, ms, p.item_a ? 3 : 1);
        ix->pending.active = false;
        return CCR_OK;
    }
#endif
    rc = launch_select_rescore(cand, cnt, p.first_nsub, p.first_sp, n_q, p.nq_pad, p.first_lay, k, p.rescore_cap, p.select_compact, ix->n_rows, thr, delta,
                               ix->tile_norm, ix->row_norm, ix->dmax_bits, Q_bf16, ix->D, ix->dim,
                               ix->id_out, out_scores, out_ids, flag_count, flag_list, stat_cand, nullptr, s);
    if (rc != CCR_OK) return rc;
    CCR_HIP_CHECK(hipEventRecord(ix->ev[5], s));

    ix->pending.Q = Q_bf16;
    ix->pending.n_q = n_q;
    ix->pending.k = k;
    ix->pending.out_scores = out_scores;
    ix->pending.out_ids = out_ids;
    ix->pending.ws = ws;
    ix->pending.stream = s;
    if (async) {
        // The host does not learn the flag count here, and nothing is re-done on the stream: flagged queries (rare: a sub-list
        // overflow, mass ties, an estimated threshold that failed its check) are completed by ccr_search_finish(), which has the
        // retry pass and the margin path at its disposal (2-3 ms for a handful of NQ queries; an unconditional on-stream chunk of
        // the fp64 path cost 35 ms as soon as ONE query was flagged, and two no-op launches per search when none was).
        // the shard message's header learns the count on the stream: the exchange can be enqueued without the host knowing it
        if (flagged_out) CCR_HIP_CHECK(hipMemcpyAsync(flagged_out, flag_count, 4, hipMemcpyDeviceToDevice, s));
        CCR_HIP_CHECK(hipMemcpyAsync((void *)ix->host_flags, flag_count, 16, hipMemcpyDeviceToHost, s));   // pinned: stays asynchronous
        CCR_HIP_CHECK(hipEventRecord(ix->ev[7], s));
        ix->pending.active = true;
        return CCR_OK;
    }
    ix->pending.active = false;
    return search_complete(ix);   // (synchronous form: every flagged query has been re-done, the header's zero stands)
}

extern "C" int ccr_search(ccr_index *ix, const uint16_t *Q_bf16, int n_q, int k, float *out_scores, int64_t *out_ids,
                          void *workspace, size_t ws_bytes, int flags, void *stream) {
    CCR_REQUIRE(ix, "ccr_search: null index");
    return search_impl(ix, Q_bf16, n_q, k, out_scores, out_ids, ix->offset, nullptr, workspace, ws_bytes, flags, stream);
}

// ------------------------------------------------------------------ packed shard message (row-sharded multi-GPU search)
static inline size_t shard_rows_at(int n_q, int k) { return (sizeof(ccr_shard_header) + (size_t)n_q * k * 4 + 15) / 16 * 16; }

extern "C" size_t ccr_shard_message_bytes(int n_q, int k) {
    if (n_q < 0 || k < 1) return 0;
    return (shard_rows_at(n_q, k) + (size_t)n_q * k * 4 + 15) / 16 * 16;
}

extern "C" int ccr_search_shard(ccr_index *ix, const uint16_t *Q_bf16, int n_q, int k, void *message, void *workspace, size_t ws_bytes,
                                int flags, void *stream) {
    CCR_REQUIRE(ix && message, "ccr_search_shard: null pointer");
    CCR_REQUIRE((uintptr_t)message % 16 == 0, "ccr_search_shard: message must be 16-byte aligned");
    CCR_REQUIRE(n_q >= 0 && k >= 1, "ccr_search_shard: bad shape n_q=%d k=%d", n_q, k);
    hipStream_t s = (hipStream_t)stream;
    ccr_shard_header h;
    memset(&h, 0, sizeof(h));
    h.magic = CCR_SHARD_MAGIC;
    h.k_valid = (uint32_t)k;
    h.n_covered = 0u;   // (an asynchronous search completes no flagged query by itself: ccr_search_finish does)
    h.row_offset = ix->offset;
    h.n_rows = ix->n_rows;
    int rc = launch_shard_header(h, message, s);   // by value through a kernel: no host buffer, no synchronisation
    if (rc != CCR_OK) return rc;
    if (n_q == 0) return CCR_OK;
    char *m = (char *)message;
    uint32_t *flagged = reinterpret_cast<uint32_t *>(m + offsetof(ccr_shard_header, n_flagged));
    return search_impl(ix, Q_bf16, n_q, k, reinterpret_cast<float *>(m + sizeof(ccr_shard_header)),
                       reinterpret_cast<int64_t *>(m + shard_rows_at(n_q, k)), ID_LOCAL_U32, flagged, workspace, ws_bytes, flags, stream);
}

// ------------------------------------------------------------------ dense score matrix
extern "C" int ccr_scores(const ccr_index *ix, const uint16_t *Q_bf16, int n_q, int mode, float *out, void *stream) {
    CCR_REQUIRE(ix && Q_bf16 && out && n_q > 0, "ccr_scores: bad argument");
    CCR_REQUIRE((uintptr_t)Q_bf16 % 16 == 0, "ccr_scores: query pointer must be 16-byte aligned");
    if (mode == CCR_SCORES_CANONICAL)
        return launch_dense_scores(ix->D, ix->n_rows, ix->dim, Q_bf16, nullptr, 0, n_q, nullptr, out, (hipStream_t)stream);
    CCR_REQUIRE(mode == CCR_SCORES_MFMA, "ccr_scores: unknown mode %d", mode);
    CCR_REQUIRE(ix->dim % 8 == 0, "ccr_scores: CCR_SCORES_MFMA needs dim %% 8 == 0 (dim=%d)", ix->dim);
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.D = ix->D;
    g.n_rows = ix->n_rows;
    g.dim = ix->dim;
    g.Q = Q_bf16;
    g.n_q = n_q;
    g.nq_pad = (int)round_up(n_q, TILE_Q);
    g.qblocks = g.nq_pad / TILE_Q;
    g.n_vt = (ix->n_rows + TILE_DOCS - 1) / TILE_DOCS;
    g.tile_stride = 1;
    g.ranges = (int)round_up(std::min<int64_t>(64, g.n_vt), NUM_XCD);
    g.qgroups = 1;
    g.item_begin = 0;
    g.item_end = INT32_MAX;
    g.store = out;
    g.stagger = 1;
    g.qdirect = ix->knobs.qdirect;
    const int grid = std::max(NUM_XCD, ix->num_cu / NUM_XCD * NUM_XCD);
    if (ix->knobs.mfma16 >= 0 ? ix->knobs.mfma16 : CCR_MFMA16_DEFAULT) return launch_gemm16_store(g, grid, (hipStream_t)stream);
    return launch_gemm_store(g, grid, (hipStream_t)stream);
}
