#!/bin/bash
# round-5 profile collection on the GPU box: bash tools/profile_r5.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
bash $ROOT/tools/pmc_main_pass.sh > /dev/null 2>&1; tail -20 $ROOT/gpurun_out/pmc_main/summary.txt
bash $ROOT/tools/pmc_bm25.sh > /dev/null 2>&1; tail -25 $ROOT/gpurun_out/pmc_bm25/summary.txt
