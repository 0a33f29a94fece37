cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2f
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2f/stats -- python3 $R/bench.py --no-cpu-baseline --steps 400 > $R/gpurun_out/r2f/b.log 2>&1
python3 - <<'PY'
import csv,glob,os
f=sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r2f/stats/*/*_kernel_stats.csv'))[-1]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3)
PY
