python -m pytest tests -m gpu -x -q > gpurun_out/r4_t3.log 2>&1; tail -3 gpurun_out/r4_t3.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4_bench_n1b.json 2> gpurun_out/r4_bench_n1b.err; tail -3 gpurun_out/r4_bench_n1b.err
python bench.py --no-cpu-baseline --no-configs > gpurun_out/r4_bench_n1c.json 2> gpurun_out/r4_bench_n1c.err
python bench.py --no-cpu-baseline --no-configs --no-kernel-events > gpurun_out/r4_bench_n1d.json 2> gpurun_out/r4_bench_n1d.err
python - <<'PY'
import json
for f in ("gpurun_out/r4_bench_n1b.json","gpurun_out/r4_bench_n1c.json","gpurun_out/r4_bench_n1d.json"):
    try:
        d=json.loads(open(f).read().strip().split("\n")[-1])
        r=d.get("roofline") or {}
        print(f, d["value"], d["ms_per_step"], r.get("kernel_ms"), r.get("kernel_ms_p10"), r.get("kernel_ms_p90"), r.get("kernel_launches_timed"), r.get("kernel_share_of_step"), r.get("traffic"), r.get("traffic_over_B2"))
        for c in d.get("configs",[]):
            print("   ", c.get("config","")[:50], c.get("ms_per_step"), c.get("kernel_ms"), c.get("valu_flop_frac"), c.get("error"))
    except Exception as e: print(f, "ERR", e)
PY
