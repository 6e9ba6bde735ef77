#!/usr/bin/env python3
"""L2-locality A/B of the main pass with hardware counters (guide rule 28: streamed data served from L2
rather than from beyond it).  Runs ON THE GPU BOX:  python3 tools/exp_locality.py [outfile] [--quick]

For every variant of the work mapping (environment knobs read by the library at index creation):
  * one plain `bench.py` run                               -> main-pass ms per step (HIP events), step time
  * three separate `rocprofv3 --kernel-trace --pmc` passes  -> FETCH_SIZE (x2 on gfx950, KiB), WRITE_SIZE + TCC_HIT/MISS,
                                                              GRBM_GUI_ACTIVE (/ 8 XCDs / kernel duration = effective clock)
Counters are never combined with sys/hip traces; the profiled program is `python3 bench.py` itself.  This driver makes no
GPU call.  Variants: query-group width (CCR_QGROUPS: how many XCDs share a set of query blocks), single launch
(CCR_PROGRESSIVE=0, so that the mappings are comparable) and the swapped item order (CCR_ITEM_SWAP=1: the co-resident
workgroups of an XCD share the QUERY BLOCK instead of the corpus range)."""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "locality")
PASSES = [["FETCH_SIZE"], ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"], ["GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES"]]


def is_main_pass(name):
    """EPI_FILTER instantiations of the tile kernels: gemm_topk_kernel<0, ..>, gemm_topk16_kernel<0, ..>, and the 256 x 384 form."""
    return "gemm_topk16w_kernel" in name or ("gemm_topk" in name and ("<0," in name.replace(" ", "") or "<(ccr::Epi)0" in name))


def run_plain(env, args, tag):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--cpu-queries", "0", "--no-secondary"] + args,
                       capture_output=True, text=True, env=env, timeout=600)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not line:
        print(f"[{tag}] plain run failed: {r.stderr[-800:]}", flush=True)
        return None
    return json.loads(line[-1])


def run_pmc(env, args, tag, ctrs, i):
    d = os.path.join(OUT, f"{tag}_pass{i}")
    subprocess.run(["rm", "-rf", d])
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + ctrs + ["--output-format", "csv", "-d", d, "--", "python3", os.path.join(ROOT, "bench.py"),
                                                             "--steps", "1", "--warmup", "1", "--cpu-queries", "0", "--no-secondary"] + args
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(env, TMPDIR="/tmp"), cwd="/tmp", timeout=600)
    if r.returncode != 0:
        print(f"[{tag}] pmc pass {i} failed: {r.stderr[-600:]}", flush=True)
        return {}, None
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    per = defaultdict(lambda: defaultdict(float))   # dispatch -> counter -> value
    names = {}
    for f in cc:
        for row in csv.DictReader(open(f)):
            per[row["Dispatch_Id"]][row["Counter_Name"]] += float(row["Counter_Value"])
            names[row["Dispatch_Id"]] = row["Kernel_Name"]
    tot = defaultdict(float)
    n_disp = 0
    for did, c in per.items():
        if is_main_pass(names[did]):
            n_disp += 1
            for k, v in c.items():
                tot[k] += v
    dur = 0.0
    for f in kt:
        for row in csv.DictReader(open(f)):
            nm = row["Kernel_Name"]
            if is_main_pass(nm):
                dur += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
    subprocess.run(["rm", "-rf", d])
    # two steps were profiled (warm-up + timed): per step
    return {k: v / 2.0 for k, v in tot.items()}, (dur / 2.0 if dur else None)


def main():
    out_file = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else os.path.join(ROOT, "gpurun_out", "r03_locality.json")
    quick = "--quick" in sys.argv
    os.makedirs(OUT, exist_ok=True)
    nq, q16 = [], ["--queries", "4096"]
    variants = [
        ("nq_default", {}, nq),
        ("nq_tile256", {"CCR_WIDE": "0"}, nq),            # round 6: the 256 x 256 kernel beside the planner's 256 x 384 choice
        ("nq_single_qg2", {"CCR_PROGRESSIVE": "0", "CCR_QGROUPS": "2"}, nq),
        ("nq_single_qg1", {"CCR_PROGRESSIVE": "0", "CCR_QGROUPS": "1"}, nq),
        ("nq_single_qg2_swap", {"CCR_PROGRESSIVE": "0", "CCR_QGROUPS": "2", "CCR_ITEM_SWAP": "1"}, nq),
        ("nq_single_qg1_swap", {"CCR_PROGRESSIVE": "0", "CCR_QGROUPS": "1", "CCR_ITEM_SWAP": "1"}, nq),
        ("q4096_default", {}, q16),
        ("q4096_single_qg1", {"CCR_PROGRESSIVE": "0", "CCR_QGROUPS": "1"}, q16),
        ("q4096_single_qg2", {"CCR_PROGRESSIVE": "0", "CCR_QGROUPS": "2"}, q16),
        ("q4096_single_qg4", {"CCR_PROGRESSIVE": "0", "CCR_QGROUPS": "4"}, q16),
        ("q4096_single_qg8", {"CCR_PROGRESSIVE": "0", "CCR_QGROUPS": "8"}, q16),
        ("q4096_single_qg8_swap", {"CCR_PROGRESSIVE": "0", "CCR_QGROUPS": "8", "CCR_ITEM_SWAP": "1"}, q16),
        # the secondaries of the bench line (their `traffic` was null in round 2)
        ("msmarco_scale", {}, ["--rows", "8841823", "--queries", "6980"]),
        ("nq_k1001", {}, ["--k", "1001"]),
        ("nq_shard8", {}, ["--rows", "335184"]),
        ("msmarco_shard8", {}, ["--rows", "1105228", "--queries", "6980"]),
    ]
    if quick:
        variants = variants[:2]
    for a in sys.argv[1:]:                       # --slice=lo:hi selects variants (one gpurun call has a time limit)
        if a.startswith("--slice="):
            lo, hi = a.split("=")[1].split(":")
            variants = variants[int(lo):int(hi)]
    # (gpurun_out/ does not travel to the GPU box: a sliced run continues from the committed summary of the same name under profiles/)
    seed = os.path.join(ROOT, "profiles", os.path.basename(out_file))
    results = json.load(open(out_file)) if os.path.isfile(out_file) else (json.load(open(seed)) if os.path.isfile(seed) else {})
    for tag, extra, args in variants:
        env = dict(os.environ, **extra)
        rec = {"env": extra, "bench_args": args}
        plain = run_plain(env, args, tag)
        if plain:
            rec["ms_per_step"] = plain["ms_per_step"]
            rec["main_pass_ms"] = plain["roofline"]["main_pass_ms_per_step"]
            rec["frac_of_mfma_peak"] = plain["roofline"]["frac"]
            rec["launches_per_step"] = plain["roofline"]["launches_per_step"]
            rec["algorithmic_corpus_bytes"] = plain["config"]["corpus_rows"] * plain["config"]["dim"] * 2
            rec["search_stats"] = {k: plain["search_stats"][k] for k in ("ranges", "sublists", "sample_tiles", "candidates_per_query")}
        ctr, dur_under_pmc = {}, None
        for i, c in enumerate(PASSES):
            got, dur = run_pmc(env, args, tag, c, i)
            ctr.update(got)
            if c[0] == "GRBM_GUI_ACTIVE":
                dur_under_pmc = dur
        if "FETCH_SIZE" in ctr:
            rec["fabric_read_bytes"] = ctr["FETCH_SIZE"] * 1024 * 2     # gfx950: wide coalesced reads are tallied at half size
            rec["hbm_write_bytes"] = ctr.get("WRITE_SIZE", 0.0) * 1024
            rec["traffic_bytes"] = rec["fabric_read_bytes"] + rec["hbm_write_bytes"]
            if "algorithmic_corpus_bytes" in rec:
                rec["traffic_over_algorithmic"] = round(rec["traffic_bytes"] / rec["algorithmic_corpus_bytes"], 2)
        if ctr.get("TCC_HIT_sum"):
            rec["l2_hit_rate"] = round(ctr["TCC_HIT_sum"] / (ctr["TCC_HIT_sum"] + ctr["TCC_MISS_sum"]), 4)
        if ctr.get("GRBM_GUI_ACTIVE") and dur_under_pmc:
            rec["main_pass_ms_under_pmc"] = round(dur_under_pmc, 3)
            rec["effective_clock_ghz"] = round(ctr["GRBM_GUI_ACTIVE"] / 8.0 / (dur_under_pmc * 1e-3) / 1e9, 3)
            if ctr.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                rec["mfma_busy_frac"] = round((ctr["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (ctr["GRBM_GUI_ACTIVE"] / 8.0), 4)
        results[tag] = rec
        print(tag, json.dumps(rec), flush=True)
        json.dump(results, open(out_file, "w"), indent=1)
    print("written", out_file)


if __name__ == "__main__":
    main()
