#!/bin/bash
# Norm outliers: a few corpus rows scaled by S (the filter margins use the largest row norm of the shard).
# Usage (GPU box): bash tools/exp_outlier.sh > gpurun_out/outlier.txt
for S in 1 10 30 100 1000; do
  echo "== outlier scale $S"
  CCR_BENCH_OUTLIER=$S timeout -k 10 300 python bench.py --data outlier --steps 3 --warmup 1 --no-secondary --cpu-queries 0 2>gpurun_out/outlier_err_$S.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['phases_ms'], d['search_stats'])"
done
