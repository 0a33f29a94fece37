timeout -k 10 600 python -m pytest tests/test_gpu_lstsq.py -m gpu -x -q > gpurun_out/r4_t9.log 2>&1; echo rc=$?; tail -3 gpurun_out/r4_t9.log
timeout -k 10 200 python tools/prof_kernel.py --workload C3L --mode lstsq --iters 30 2>&1 | grep -v amdgpu
