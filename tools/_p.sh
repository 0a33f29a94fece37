cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
python -m pytest tests -m gpu -q -x > gpurun_out/r2e/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2e/pytest.log
python bench.py --no-cpu-baseline > gpurun_out/r2e/bench.json 2>&1; python3 -c "
import json
r=json.loads([l for l in open('gpurun_out/r2e/bench.json') if l.startswith('{')][-1]); print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['frac'], r['roofline']['valu_flop_frac'], r['roofline']['isa']['valu_insts_per_pixel'])"
