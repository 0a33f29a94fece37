cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
python3 tools/dev/prof_demo.py 2 2>&1 | tail -1
for f in 16 32; do
mkdir -p gpurun_out/r2s
cd /tmp && export TMPDIR=/tmp
GIGALENS_HIP_DBGFLAGS=$f rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2s/e$f -- python3 $R/tools/dev/prof_demo.py 2 > $R/gpurun_out/r2s/e$f.log 2>&1
cd $R
python3 - <<PY
import csv,glob
f=sorted(glob.glob('gpurun_out/r2s/e$f/**/*kernel_stats.csv',recursive=True))[-1]
print("flags $f:", [(r['Name'][22:52], round(float(r['AverageNs'])/1e3,1)) for r in csv.DictReader(open(f)) if 'corr_pair' in r['Name']])
PY
done
