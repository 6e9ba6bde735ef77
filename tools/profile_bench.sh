#!/bin/bash
# rocprofv3 --kernel-trace --stats over the default bench command (on the GPU box): bash tools/profile_bench.sh <outdir> [bench args...]
# Copies the kernel-stats and kernel-trace CSVs next to the bench's JSON line in <outdir>.
set -u
OUT=${1:-gpurun_out/prof}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/raw" -- \
    python3 "$ROOT/bench.py" --cpu-queries 0 --no-secondary "$@" > "$ROOT/$OUT/bench.json" 2> "$ROOT/$OUT/bench.err" || { echo "profile run failed"; tail -5 "$ROOT/$OUT/bench.err"; exit 1; }
find "$ROOT/$OUT/raw" -name '*kernel_stats.csv' -exec cp {} "$ROOT/$OUT/kernel_stats.csv" \;
find "$ROOT/$OUT/raw" -name '*kernel_trace.csv' -exec cp {} "$ROOT/$OUT/kernel_trace.csv" \;
rm -rf "$ROOT/$OUT/raw"
tail -1 "$ROOT/$OUT/bench.json"
head -8 "$ROOT/$OUT/kernel_stats.csv"
