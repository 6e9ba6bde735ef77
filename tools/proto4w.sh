#!/bin/bash
# Builds the timing-only four-wave main-pass experiments (csrc/experimental/*.hip) into lib/libproto_<name>[<mode>].so.
#   bash tools/proto4w.sh                     -> gemm4w_proto (256x256 tile) and gemm4w384_proto (256x384 tile)
#   PROTO_MODE=1|2 bash tools/proto4w.sh      -> DMA-only variants of gemm4w_proto
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/crowd-coachable-recommendations_amd/csrc
mkdir -p $ROOT/crowd-coachable-recommendations_amd/lib
for name in gemm4w_proto gemm4w384_proto; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$SRC -Wall -Wno-unused-function -shared \
      -DPROTO_MODE=${PROTO_MODE:-0} -o $ROOT/crowd-coachable-recommendations_amd/lib/libproto_${name}${PROTO_MODE:-}.so $SRC/experimental/$name.hip
done
