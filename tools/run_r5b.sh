mkdir -p gpurun_out/r5c
timeout -k 10 500 python -m pytest tests/test_bm25.py -x -q -m gpu > gpurun_out/r5c/test_bm25.log 2>&1; tail -5 gpurun_out/r5c/test_bm25.log
for cfg in 0 2 3 4; do CCR_BM25_TILE=$cfg timeout -k 10 200 python tools/one_bm25.py --check > gpurun_out/r5c/one_bm25_cfg$cfg.log 2>&1; echo cfg $cfg; tail -2 gpurun_out/r5c/one_bm25_cfg$cfg.log; done
CCR_BM25_TABLE=0 timeout -k 10 200 python tools/one_bm25.py > gpurun_out/r5c/one_bm25_notable.log 2>&1; echo no table; tail -1 gpurun_out/r5c/one_bm25_notable.log
cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5c/prof -- python3 $GRAFT_REPO_ROOT/tools/one_bm25.py > $GRAFT_REPO_ROOT/gpurun_out/r5c/prof.log 2>&1; tail -1 $GRAFT_REPO_ROOT/gpurun_out/r5c/prof.log
