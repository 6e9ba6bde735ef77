mkdir -p gpurun_out/r5d
timeout -k 10 500 python -m pytest tests/test_gpu_inbatch.py tests/test_bm25.py -x -q -m gpu > gpurun_out/r5d/tests.log 2>&1; tail -15 gpurun_out/r5d/tests.log
timeout -k 10 200 python tools/one_inbatch.py > gpurun_out/r5d/inbatch.json 2> gpurun_out/r5d/inbatch.err; cat gpurun_out/r5d/inbatch.json | cut -c1-900
timeout -k 10 200 python tools/one_bm25.py > gpurun_out/r5d/one_bm25.log 2>&1; tail -1 gpurun_out/r5d/one_bm25.log
cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5d/prof -- python3 $GRAFT_REPO_ROOT/tools/one_inbatch.py > $GRAFT_REPO_ROOT/gpurun_out/r5d/prof.log 2>&1; tail -1 $GRAFT_REPO_ROOT/gpurun_out/r5d/prof.log
