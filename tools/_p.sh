cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
python -m pytest tests -m gpu -x -q > gpurun_out/r2c/pytest2.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2c/pytest2.log
python tools/bench_configs.py > gpurun_out/r2c/bench_configs.jsonl 2>gpurun_out/r2c/bench_configs.err; cat gpurun_out/r2c/bench_configs.jsonl | cut -c1-200
