#!/bin/bash
# What binds attention_kernel<f16> at the corpus encode's shape (482 sequences of 129-136 tokens x 12 heads = 65 K tokens per launch): separate
# --pmc passes over tools/one_attention.py (ATT_DTYPE=fp16), kernel trace only.  bash tools/pmc_attention.sh -> gpurun_out/pmc_att/summary.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_att; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ATT_DTYPE=fp16
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tools/one_attention.py > $OUT/stats.log 2>&1 || tail -3 $OUT/stats.log
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "SQ_WAVES SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pass$i -- python3 $ROOT/tools/one_attention.py > $OUT/pass$i.log 2>&1 || { echo "pass $i ($pass) failed"; tail -2 $OUT/pass$i.log; }
done
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + "/stats/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "attention_kernel" in r["Name"]:
            print(f"attention_kernel<f16>, 482 sequences of 129-136 tokens x 12 heads: {float(r['AverageNs']) / 1e3:.1f} us per launch (min {float(r['MinNs']) / 1e3:.1f}, {r['Calls']} launches)")
tot = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(out + "/pass*/*/*counter_collection.csv"):
    seen = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "attention_kernel" in r["Kernel_Name"]:
            seen[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, c), v in seen.items():
        tot[c] += v; cnt[c] += 1
for c in sorted(tot):
    print(f"   {c:28s} per launch {tot[c] / cnt[c]:16.6g}   (launches {cnt[c]})")
avg = lambda c: tot[c] / cnt[c] if cnt[c] else float("nan")
print(f"   read from beyond the L2 = FETCH_SIZE x 2 KiB = {avg('FETCH_SIZE') * 2048 / 1e6:.1f} MB, written = WRITE_SIZE x 1 KiB = {avg('WRITE_SIZE') * 1024 / 1e6:.1f} MB  (algorithmic: 302 MB of Q | K | V read, 101 MB written)")
print(f"   L2 hit rate {avg('TCC_HIT_sum') / (avg('TCC_HIT_sum') + avg('TCC_MISS_sum')):.3f}")
for a, label in (("SQ_WAIT_ANY", "parked (s_waitcnt / barrier)"), ("SQ_WAIT_INST_ANY", "stalled at issue"), ("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_WAIT_INST_LDS", "LDS issue stall"), ("SQ_ACTIVE_INST_LDS", "LDS active")):
    print(f"   {label:32s} / wave cycles = {avg(a) / avg('SQ_WAVE_CYCLES'):.3f}")
PY
