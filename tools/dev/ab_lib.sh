#!/bin/bash
# A/B of builds of the library on one box: the shipped one and each path given (under the repo), alternating; WORKLOADS="C2 C3"
out=gpurun_out/ab_lib.txt; mkdir -p gpurun_out; : > $out
for rep in 1 2; do
  for lib in "" "$@"; do
    if [ -n "$lib" ]; then export GIGALENS_HIP_LIB=$PWD/$lib; else unset GIGALENS_HIP_LIB; fi
    for w in ${WORKLOADS:-C2}; do
      echo -n "lib='${lib}' " >> $out
      python3 tools/dev/step_time.py $w 2>&1 | grep -v amdgpu.ids | cut -c1-110 >> $out || exit 1
    done
  done
done
cat $out
