cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
bash tools/dev/sweep_env.sh tools/dev/sweep1.txt
mkdir -p gpurun_out/r2i
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2i/bench_stats -- python3 $R/bench.py --no-cpu-baseline --steps 500 --warmup 50 > $R/gpurun_out/r2i/bench_under_rocprof.log 2>&1
cd $R
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r2i/bench_stats/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print(r['Name'][:60], r['Calls'], r['AverageNs'], r['MinNs'])
PY
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_inference.py -m gpu -q -x > gpurun_out/r2i/pytest.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/r2i/pytest.log | cut -c1-200
