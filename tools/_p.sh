cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2l
echo "X=0" > /tmp/sw.txt
bash tools/dev/sweep_env.sh /tmp/sw.txt
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r2l/pytest_all.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/r2l/pytest_all.log | cut -c1-300
