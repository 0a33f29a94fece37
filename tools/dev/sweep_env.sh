#!/bin/bash
# bench.py under a list of environment settings (one per line of $1: "NAME=VAL NAME=VAL ...") -> value, ms_per_step, kernel_ms
cd ${GRAFT_REPO_ROOT:-.}
while IFS= read -r line; do
  out=$(env $line python3 bench.py --no-cpu-baseline --steps ${STEPS:-1500} --warmup 100 ${BENCH_ARGS:-} 2>/dev/null | tail -1)
  echo "$line :: $(echo "$out" | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('value %.4g  ms_per_step %.5f  kernel_ms %.5f' % (r['value'], r['ms_per_step'], r['roofline']['kernel_ms']))")"
done < "$1"
