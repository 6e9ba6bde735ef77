#!/bin/bash
# PMC passes over the main pass of a single-block search (tools/one_small_batch.py <n_q>): HBM bytes (FETCH_SIZE, doubled per the guide's
# gfx950 correction), busy cycles, MFMA busy, wave cycles.  bash tools/pmc_small_batch.sh <n_q> [env assignments...]  -> gpurun_out/pmc_small_<n_q>.txt
set -u
NQ=${1:-1}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_small; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | tr ' ' '_')
  timeout -k 10 200 env "$@" rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/$tag -- python3 $ROOT/tools/one_small_batch.py $NQ > $OUT/$tag.log 2>&1 || tail -3 $OUT/$tag.log
done
python3 - $OUT $NQ <<'PY'
import csv, glob, sys, collections
out, nq = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "narrow_filter" in k or "gemm_topk16_kernel<0" in k or "gemm_topk_kernel<0" in k:
            name = "narrow_filter_kernel" if "narrow" in k else "gemm_topk(16)_kernel<EPI_FILTER>"
            tot[name][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(name, r["Counter_Name"])] += 1
for name, c in tot.items():
    print(f"n_q {nq}: {name}")
    for cn, v in sorted(c.items()):
        n = cnt[(name, cn)]
        print(f"   {cn:28s} per launch {v / n:14.4g}   (launches {n})")
    if "FETCH_SIZE" in c:
        n = cnt[(name, "FETCH_SIZE")]
        print(f"   HBM read per launch = FETCH_SIZE x 2 KiB (gfx950 correction) = {c['FETCH_SIZE'] / n * 2 * 1024 / 1e9:.3f} GB   (algorithmic: corpus once = 4.119 GB)")
PY
