#!/bin/bash
# A/B of two builds of the library on one box: the shipped one and $1 (a path under the repo), alternating
out=gpurun_out/ab_lib.txt; mkdir -p gpurun_out; : > $out
for rep in 1 2; do
  for lib in "" "$1"; do
    if [ -n "$lib" ]; then export GIGALENS_HIP_LIB=$PWD/$lib; else unset GIGALENS_HIP_LIB; fi
    echo "== lib='${lib}'" >> $out
    python3 tools/dev/step_time.py C2 2>&1 | grep -v amdgpu.ids | cut -c1-120 >> $out || exit 1
    python3 tools/prof_kernel.py --workload C3L --mode lstsq --iters 20 2>&1 | grep lstsq >> $out || exit 1
    python3 tools/dev/prof_demo.py 2 2>&1 | grep "ms per step" >> $out || exit 1
  done
done
cat $out
