cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/pt.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/pt.log | cut -c1-200
