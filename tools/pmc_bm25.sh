#!/bin/bash
# PMC passes over the BM25 kernels of tools/exp_bm25_tile.py --cfgs 0 (500 k documents x 2 000 queries x top-1001): bytes fetched from beyond
# the L2 (FETCH_SIZE, doubled per the guide's gfx950 correction), bytes written, L2 hits / misses.  bash tools/pmc_bm25.sh -> gpurun_out/pmc_bm25.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_bm25; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/$tag -- python3 $ROOT/tools/exp_bm25_tile.py --cfgs 0 > $OUT/$tag.log 2>&1 || tail -3 $OUT/$tag.log
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "bm25_" in k:
            name = k.split("(")[0].replace("void ", "").replace("ccr::", "")
            tot[name][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(name, r["Counter_Name"])] += 1
for name, c in sorted(tot.items()):
    print(name)
    for cn, v in sorted(c.items()):
        n = cnt[(name, cn)]
        print(f"   {cn:18s} per launch {v / n:14.4g}   (launches {n})")
    if "FETCH_SIZE" in c:
        print(f"   read from beyond the L2 per launch = FETCH_SIZE x 2 KiB = {c['FETCH_SIZE'] / cnt[(name, 'FETCH_SIZE')] * 2 * 1024 / 1e9:.3f} GB")
    if "WRITE_SIZE" in c:
        print(f"   written per launch = WRITE_SIZE x 1 KiB = {c['WRITE_SIZE'] / cnt[(name, 'WRITE_SIZE')] * 1024 / 1e9:.3f} GB")
    if "TCC_HIT_sum" in c:
        h, m = c["TCC_HIT_sum"], c["TCC_MISS_sum"]
        print(f"   L2 hit rate {h / (h + m):.3f}")
PY
