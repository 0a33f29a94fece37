cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" ; tail -5 gpurun_out/r2a/pytest.log
python bench.py > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err; tail -c 3000 gpurun_out/r2a/bench.json
python bench.py --no-cpu-baseline --no-kernel-events > gpurun_out/r2a/bench_noev.json 2>&1; tail -c 600 gpurun_out/r2a/bench_noev.json
python bench.py --no-cpu-baseline --mode svi > gpurun_out/r2a/bench_svi.json 2>&1; tail -c 1500 gpurun_out/r2a/bench_svi.json
python bench.py --no-cpu-baseline --workload C5 --mode svi --steps 100 --warmup 10 > gpurun_out/r2a/bench_c5.json 2>&1; tail -c 1500 gpurun_out/r2a/bench_c5.json
python bench.py --gpus 2 --steps 5 --warmup 1 > gpurun_out/r2a/bench_g2.log 2>&1; echo "gpus2 rc=$?"; tail -3 gpurun_out/r2a/bench_g2.log
