#!/bin/bash
# Phase dissection of the PSF correlation kernels on the reference's demo set-up: tools/dev/psf_phases.sh LIB "ENV=.. ENV=.." ...
# (each further argument is one variant's environment); prints the correlation kernels' average durations per variant.
LIB=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1)); d=$R/gpurun_out/psf/ph$i; rm -rf $d
  env GIGALENS_HIP_LIB=$R/$LIB $v rocprofv3 --kernel-trace --stats -d $d -- python3 $R/tools/dev/prof_demo.py > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
  echo "== $v: $(grep 'ms per step' $d.log)"
  python3 $R/tools/kstats.py $d 12 | grep corr
done
