#!/bin/bash
# sweep of the tapered dispatch end (GIGALENS_HIP_TAIL_ROWS / _N) on one workload: tools/dev/tail_sweep.sh "rows n_tail" ...  (-1 = automatic)
out=gpurun_out/tail_sweep.txt
mkdir -p gpurun_out
: > $out
for cfg in "$@"; do
  set -- $cfg
  GIGALENS_HIP_TAIL_ROWS=$1 GIGALENS_HIP_TAIL_N=$2 python3 tools/dev/step_time.py ${WORKLOAD:-C2} ${BATCH:-} 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done
cat $out
