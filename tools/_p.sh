cd $GRAFT_REPO_ROOT
python3 tools/bench_configs.py 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    try: r=json.loads(l)
    except: continue
    print(r['config'][:70], r['ms_per_step'])"
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/pt.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/pt.log | cut -c1-200
