#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
bash $ROOT/tools/pmc_attention.sh > /dev/null 2>&1; cat $ROOT/gpurun_out/pmc_att/summary.txt
bash $ROOT/tools/profile_encode.sh gpurun_out/encprof_fp16 --autocast fp16 2>&1 | tail -12
