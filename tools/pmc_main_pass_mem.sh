#!/bin/bash
# Which unit of the vector-memory pipeline the main pass saturates (round 6): TA / TCP / TD / address-translation counters of
# gemm_topk16_kernel<EPI_FILTER> over the default bench command at NQ (2 steps), separate --pmc passes, kernel trace only.
#   bash tools/pmc_main_pass_mem.sh -> gpurun_out/pmc_mem/summary.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_mem; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "\b[A-Z][A-Za-z_0-9]*\b" | sort -u > $OUT/counters.txt
i=0
for pass in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD" \
            "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" "TA_FLAT_READ_LDS_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum" \
            "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_BUSY_avr" \
            "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
            "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum" "TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
            "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
            "TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum" "TD_LOAD_WAVEFRONT_sum TD_SPI_STALL_sum" \
            "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  ok=""
  for c in $pass; do grep -qx "$c" $OUT/counters.txt && ok="$ok $c"; done
  [ -z "$ok" ] && { echo "pass $i: none of ($pass) exists"; continue; }
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ok --output-format csv -d $OUT/pass$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-queries 0 --no-secondary > $OUT/pass$i.log 2>&1 || { echo "pass $i ($ok) failed"; tail -3 $OUT/pass$i.log; }
  echo "pass $i done: $ok"
done
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float); cnt = collections.Counter()
dur = []
for f in glob.glob(out + "/pass*/*/*counter_collection.csv"):
    seen = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_topk16_kernel<0" in k or "gemm_topk16w_kernel<0" in k:
            seen[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, c), v in seen.items():
        tot[c] += v; cnt[c] += 1
for f in glob.glob(out + "/pass*/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "gemm_topk16_kernel<0" in r["Kernel_Name"] or "gemm_topk16w_kernel<0" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
print("main pass (gemm_topk16_kernel<EPI_FILTER> or, with the planner's default at NQ, gemm_topk16w_kernel: ONE launch per step), NQ 2,681,468 x 768 x 3,452 queries; "
      "counters summed over the chip, average per LAUNCH:")
if dur:
    print(f"   launch duration under the profiler: {sum(dur) / len(dur):.1f} us")
for c in sorted(tot):
    print(f"   {c:42s} {tot[c] / cnt[c]:16.6g}   (launches {cnt[c]})")
PY
