#!/bin/bash
# rocprofv3 --kernel-trace --stats over the encoder forward (on the GPU box): bash tools/profile_encode.sh <outdir> [bench_encode args...]
# Default: 20 000 synthetic passages through the length-sorted encoder on the library's layer kernels; add `--fused off` for the torch modules.
set -u
OUT=${1:-gpurun_out/encprof}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/raw" -- \
    python3 "$ROOT/tools/bench_encode.py" --texts 20000 --dist passages --modes sorted "$@" > "$ROOT/$OUT/out.txt" 2> "$ROOT/$OUT/err.txt" || { echo "profile run failed"; tail -5 "$ROOT/$OUT/err.txt"; exit 1; }
find "$ROOT/$OUT/raw" -name '*kernel_stats.csv' -exec cp {} "$ROOT/$OUT/kernel_stats.csv" \;
rm -rf "$ROOT/$OUT/raw"
grep '"mode"' "$ROOT/$OUT/out.txt" | cut -c1-400
head -8 "$ROOT/$OUT/kernel_stats.csv"
