#!/bin/bash
# Same-box A/B of this tree against the build round 3 started from (a `git worktree` of commit b324bab under _r2/, built in place):
# alternating runs of both bench.py on one GPU.  bash tools/ab_vs_round2.sh > gpurun_out/r03_ab_vs_round2.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
run() {  # tree label args...
  local tree=$1 label=$2; shift 2
  python3 $tree/bench.py --steps 10 --warmup 3 --no-secondary --cpu-queries 0 "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); p=r['phases_ms']
print('$label', '$*' or 'NQ k=100', 'ms/step', r['ms_per_step'], 'main', p['main_pass'], 'sample+thr', round(p['sample_pass']+p['threshold'],3), 'select', p['select_rescore'], 'fallback', p['fallback'], 'flagged', r['search_stats']['n_fallback'])"
}
for args in "" "--k 1001" "--k 300" "--rows 335184" "--rows 1105228 --queries 6980" "--rows 8841823 --queries 6980" "--data sorted --k 1001" "--data sorted" "--data clustered --k 1001" "--rows 6250000 --dim 1024 --queries 10000 --k 1000"; do
  for rep in 1 2; do
    run $ROOT/_r2 "round2" $args
    run $ROOT      "round3" $args
  done
done
