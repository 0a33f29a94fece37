cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
python -m pytest tests -m gpu -q > gpurun_out/r2d/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r2d/pytest.log
