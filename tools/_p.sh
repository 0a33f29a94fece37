cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random_shapes" > gpurun_out/pt.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/pt.log | cut -c1-250
