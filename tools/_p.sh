cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2t
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r2t/pytest_all.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/r2t/pytest_all.log | cut -c1-300
bash tools/collect_profiles.sh r2 2>&1 | tail -1 | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
