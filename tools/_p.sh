cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "nfw_table" > gpurun_out/r2h/pytest_nfw.log 2>&1; echo "rc=$?"; tail -30 gpurun_out/r2h/pytest_nfw.log | cut -c1-300
