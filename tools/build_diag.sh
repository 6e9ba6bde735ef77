#!/bin/bash
# Builds the DIAGNOSTIC library (timing-only ablations of the main pass, WRONG results under CCR_GEMM_DBG != 0) into
# crowd-coachable-recommendations_amd/lib_diag/ -- never loaded by the product: tools/exp_main_pass_ablation.py points CCR_LIB_PATH at it.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/crowd-coachable-recommendations_amd/csrc
OUT=$ROOT/crowd-coachable-recommendations_amd/lib_diag
mkdir -p "$OUT"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$SRC -Wall -Wno-unused-function -DCCR_DIAGNOSTICS"
pids=()
for f in ccr_api ccr_pack ccr_dense ccr_fused ccr_merge ccr_inbatch ccr_metrics ccr_bm25 ccr_special ccr_encoder ccr_narrow; do
  if [ ! -f "$OUT/$f.o" ] || [ "$SRC/$f.hip" -nt "$OUT/$f.o" ] || [ -n "$(find "$SRC" -name '*.h' -newer "$OUT/$f.o")" ]; then
    /opt/rocm/bin/hipcc $FLAGS -c "$SRC/$f.hip" -o "$OUT/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libccr_hip.so" "$OUT"/*.o
echo "built $OUT/libccr_hip.so"
