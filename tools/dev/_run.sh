timeout -k 10 600 python -m pytest tests/test_gpu_user_profile.py tests/test_gpu_lstsq.py -m gpu -x -q > gpurun_out/r4_t11.log 2>&1; echo rc=$?; tail -25 gpurun_out/r4_t11.log
timeout -k 10 300 python tools/dev/user_model_time.py 2>&1 | grep -v amdgpu | head -2
