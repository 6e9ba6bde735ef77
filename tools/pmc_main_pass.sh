#!/bin/bash
# What the main pass's skeleton waits on: SQ counters of gemm_topk16_kernel<EPI_FILTER> over the default bench command at
# NQ (2 steps), separate --pmc passes, kernel trace only.  bash tools/pmc_main_pass.sh -> gpurun_out/pmc_main/summary.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_main; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $OUT/sq_counters.txt
i=0
for pass in "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES" \
            "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  ok=""
  for c in $pass; do grep -qx "$c" $OUT/sq_counters.txt && ok="$ok $c"; done
  [ -z "$ok" ] && continue
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ok --output-format csv -d $OUT/pass$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-queries 0 --no-secondary > $OUT/pass$i.log 2>&1 || { echo "pass $i ($ok) failed"; tail -3 $OUT/pass$i.log; }
done
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(out + "/pass*/*/*counter_collection.csv"):
    seen = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_topk16_kernel<0" in k:
            seen[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, c), v in seen.items():
        tot[c] += v; cnt[c] += 1
print("gemm_topk16_kernel<EPI_FILTER>, NQ 2,681,468 x 768 x 3,452 queries (3 launches per step); per launch:")
for c in sorted(tot):
    print(f"   {c:28s} {tot[c] / cnt[c]:16.6g}   (launches {cnt[c]})")
def ratio(a, b, label):
    if a in tot and b in tot and tot[b]:
        print(f"   {label:60s} {tot[a] / cnt[a] / (tot[b] / cnt[b]):8.3f}")
ratio("SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES", "LDS issue stall / wave cycles")
ratio("SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES", "LDS instructions active / wave cycles")
ratio("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "issue stall (any) / wave cycles")
ratio("SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "parked on s_waitcnt / barrier / wave cycles")
ratio("SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES", "issuing / wave cycles")
ratio("SQ_INST_CYCLES_VMEM", "SQ_WAVE_CYCLES", "VMEM instruction cycles / wave cycles")
ratio("SQ_ACTIVE_INST_VMEM", "SQ_WAVE_CYCLES", "VMEM active / wave cycles")
ratio("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "MFMA busy / SQ busy cycles")
PY
