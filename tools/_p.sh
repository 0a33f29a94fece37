cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
bash tools/dev/sweep_env.sh tools/dev/sweep1.txt
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r2h/pytest_all.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r2h/pytest_all.log | cut -c1-300
