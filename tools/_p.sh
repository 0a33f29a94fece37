cd $GRAFT_REPO_ROOT
time python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; python3 -c "
import json
r=json.loads(open('gpurun_out/bench_final.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['frac'], r['roofline']['valu_flop_frac'], r['cpu_baseline']['value'], r['cpu_baseline']['sample'][:60])"
time python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline | tail -1 | cut -c1-200
