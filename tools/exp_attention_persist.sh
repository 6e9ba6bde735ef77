#!/bin/bash
# A/B of the persistent form of attention_kernel (CCR_ATT_PERSIST = workgroups per CU; 0 = one workgroup per item) at the encoder's shapes:
# kernel time from rocprofv3 --kernel-trace --stats over tools/one_attention.py <n_seq> <L> <shortest> <heads>, fp16.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/att_persist; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp ATT_DTYPE=fp16
for shape in "482 136 129 12" "320 200 180 12" "128 512 400 12" "1400 48 30 12"; do
  for p in 0 1 2 3; do
    export CCR_ATT_PERSIST=$p
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -- python3 $ROOT/tools/one_attention.py $shape > $OUT/log 2>&1 || tail -2 $OUT/log
    python3 - "$shape" $p $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[3] + "/s/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "attention_kernel" in r["Name"]:
            print(f"shape {sys.argv[1]:18s} CCR_ATT_PERSIST={sys.argv[2]}: {float(r['AverageNs']) / 1e3:7.1f} us (min {float(r['MinNs']) / 1e3:.1f})", flush=True)
PY
    rm -rf $OUT/s
  done
done
