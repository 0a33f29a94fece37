cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lstsq.py tests/test_reference_demo.py -m gpu -q -x > gpurun_out/pt.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/pt.log | cut -c1-200
