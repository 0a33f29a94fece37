timeout -k 10 300 python -m pytest tests/test_gpu_dpie.py -m gpu -x -q > gpurun_out/r4_t6.log 2>&1; echo rc=$?; tail -4 gpurun_out/r4_t6.log
