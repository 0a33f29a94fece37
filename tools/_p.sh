cd $GRAFT_REPO_ROOT
run() { out=$(python3 bench.py --no-cpu-baseline --steps $2 --warmup 20 --batch $1 2>/dev/null | tail -1); echo "$1 :: $(echo "$out" | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('value %.4g  ms_per_step %.5f  kernel_ms %.5f' % (r['value'], r['ms_per_step'], r['roofline']['kernel_ms']))")"; }
run 64 1000; run 256 1000; run 1024 1000; run 4096 300; run 16384 100
