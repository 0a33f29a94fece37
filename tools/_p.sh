cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2m
run() { out=$(env $1 python3 bench.py --no-cpu-baseline --steps 500 --warmup 50 $2 2>/dev/null | tail -1); echo "$1 $2 :: $(echo "$out" | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('value %.4g  ms_per_step %.5f  kernel_ms %.5f' % (r['value'], r['ms_per_step'], r['roofline']['kernel_ms']))")"; }
run "X=0" "--workload C3"
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r2m/pytest_all.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/r2m/pytest_all.log | cut -c1-300
