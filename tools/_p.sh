cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r2e/pytest.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/r2e/pytest.log | cut -c1-300
python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['isa'].get('valu_insts_per_pixel'), r['roofline'].get('valu_flop_frac'))"
