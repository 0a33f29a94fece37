cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "psf_supersample" > gpurun_out/pt.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/pt.log | cut -c1-300
