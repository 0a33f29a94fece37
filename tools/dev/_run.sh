timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_t8.log 2>&1; echo rc=$?; tail -3 gpurun_out/r4_t8.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4_bench_n1e.json 2> gpurun_out/r4_bench_n1e.err; echo rc=$?
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_bench_n1e.json").read().strip().split("\n")[-1])
r=d["roofline"]; print(d["value"], d["ms_per_step"], r["kernel_ms"], r["kernel_launches_timed"], r["valu_flop_frac"])
for c in d.get("configs",[]):
    print("   ", c.get("config","")[:50], c.get("ms_per_step"), c.get("kernel_ms"), c.get("valu_flop_frac"), c.get("valu_insts_per_pixel"), c.get("shapelet_live_wave_tile_share"), c.get("error"))
PY
