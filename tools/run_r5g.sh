mkdir -p gpurun_out/r5g
timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k streaming > gpurun_out/r5g/tests.log 2>&1; tail -3 gpurun_out/r5g/tests.log
timeout -k 10 300 python tools/exp_small_batches.py > gpurun_out/r5g/small.txt 2>&1; sed -n 3,7p gpurun_out/r5g/small.txt
CCR_NARROW_P12=1 timeout -k 10 300 python tools/exp_small_batches.py > gpurun_out/r5g/small_p12.txt 2>&1; sed -n 4,6p gpurun_out/r5g/small_p12.txt
