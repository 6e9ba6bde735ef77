#!/bin/bash
# Separate rocprofv3 --pmc passes over one bench step (counters never combined with sys/hip traces).
# Usage (on the GPU box): bash tools/pmc_profile.sh <outdir> [bench args...]
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "FETCH_SIZE" \
            "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$ROOT/$OUT/pass$i" -- \
      python3 "$ROOT/bench.py" --steps 1 --warmup 1 --cpu-queries 0 --no-secondary "$@" > "$ROOT/$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$ROOT/$OUT/pass$i.log"; }
  echo "pass $i done: $CTRS"
done
python3 "$ROOT/tools/pmc_summary.py" "$ROOT/$OUT" > "$ROOT/$OUT/summary.json"
cat "$ROOT/$OUT/summary.json"
