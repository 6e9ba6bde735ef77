#!/bin/bash
# Builds the timing-only four-wave main-pass experiment (csrc/experimental/gemm4w_proto.hip) into lib/libproto4w.so.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/crowd-coachable-recommendations_amd/csrc
mkdir -p $ROOT/crowd-coachable-recommendations_amd/lib
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$SRC -Wall -Wno-unused-function -shared -DPROTO_MODE=${PROTO_MODE:-0} \
    -o $ROOT/crowd-coachable-recommendations_amd/lib/libproto4w${PROTO_MODE:-}.so $SRC/experimental/gemm4w_proto.hip
