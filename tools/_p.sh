cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2l
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "many_samples" > gpurun_out/r2l/pytest_many.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r2l/pytest_many.log | cut -c1-300
