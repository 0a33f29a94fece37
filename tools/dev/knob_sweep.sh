#!/bin/bash
# step time of one workload under a list of knob settings: tools/dev/knob_sweep.sh "A=1 B=2" "A=0" ...   (WORKLOAD, BATCH from the environment)
out=gpurun_out/knob_sweep.txt
mkdir -p gpurun_out
: > $out
for cfg in "$@"; do
  ( for kv in $cfg; do export $kv; done; python3 tools/dev/step_time.py ${WORKLOAD:-C2} ${BATCH:-} 2>&1 | grep -v amdgpu.ids >> $out ) || exit 1
done
cat $out
