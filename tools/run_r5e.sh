mkdir -p gpurun_out/r5e
timeout -k 10 900 python -m pytest tests/test_gpu_encode.py -x -q -m gpu -k "sharded" > gpurun_out/r5e/tests.log 2>&1; tail -15 gpurun_out/r5e/tests.log
