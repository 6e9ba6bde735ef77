#!/bin/bash
# one PMC pass with the given counters over one bench step; prints the main-pass kernel's values
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcq; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pass1 -- python3 $ROOT/bench.py --steps 1 --warmup 1 --cpu-queries 0 > $OUT/log 2>&1 || tail -5 $OUT/log
python3 $ROOT/tools/pmc_summary.py $OUT | python3 -c "
import json,sys
d=json.load(sys.stdin)['kernels']
for k,v in d.items():
    if 'gemm_topk' in k and v.get('dispatches',0)>=1:
        print(k, {a:(round(b/1e6,2) if isinstance(b,float) else b) for a,b in v.items()}, '(x1e6)')
"
