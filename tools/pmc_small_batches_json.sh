#!/bin/bash
# Fabric traffic of the MAIN PASS of small-batch searches (n_q = 1 .. 512) for bench.py's secondary.small_batches[*].roofline.traffic:
# two separate --pmc passes per batch size (FETCH_SIZE; WRITE_SIZE) over tools/one_small_batch.py, kernel trace only.
#   bash tools/pmc_small_batches_json.sh -> gpurun_out/r06_small_batches_pmc.json   (copy to profiles/)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_small_json; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for NQ in 1 16 64 96 128 256 512; do
  for pass in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/nq${NQ}_$pass -- python3 $ROOT/tools/one_small_batch.py $NQ > $OUT/nq${NQ}_$pass.log 2>&1 || { echo "nq $NQ $pass failed"; tail -3 $OUT/nq${NQ}_$pass.log; }
  done
  echo "nq $NQ done"
done
python3 - $OUT $ROOT/gpurun_out/r06_small_batches_pmc.json <<'PY'
import csv, glob, json, os, re, sys, collections
out, dst = sys.argv[1], sys.argv[2]
res = {}
for d in sorted(glob.glob(out + "/nq*_FETCH_SIZE")):
    nq = re.search(r"nq(\d+)_", d).group(1)
    rec = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        tot, n, name = 0.0, set(), None
        for f in glob.glob(d.replace("FETCH_SIZE", ctr) + "/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if r["Counter_Name"] == ctr and ("narrow_filter" in k or "gemm_topk16_kernel<0" in k or "gemm_topk_kernel<0" in k):
                    tot += float(r["Counter_Value"]); n.add(r["Dispatch_Id"]); name = k.split("(")[0]
        searches = max(1, len(n))
        rec[ctr] = tot
        rec["kernel"] = name
        rec["launches_" + ctr] = len(n)
    # tools/one_small_batch.py runs its searches back to back: per search = total / searches it ran (its main pass is one launch up to 128
    # queries; the tile kernels' plan may use several launches per search: the script prints how many searches it ran)
    log = open(os.path.join(out, f"nq{nq}_FETCH_SIZE.log")).read()
    m = re.search(r"searches (\d+)", log)
    n_search = int(m.group(1)) if m else None
    if n_search:
        rec["searches"] = n_search
        rec["fabric_bytes_main_pass"] = (rec["FETCH_SIZE"] * 2 * 1024 + rec["WRITE_SIZE"] * 1024) / n_search   # gfx950: FETCH_SIZE x 2 (guide, HBM section)
    res[nq] = rec
json.dump({"note": "tools/pmc_small_batches_json.sh: fabric bytes (FETCH_SIZE x 2 KiB + WRITE_SIZE KiB, separate passes) of the main-pass launches of ONE search", **res}, open(dst, "w"), indent=1)
print(json.dumps(res, indent=1))
PY
