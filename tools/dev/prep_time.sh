#!/bin/bash
# front-end kernel durations with and without the sort riding in it (rocprofv3 kernel trace of the native call alone)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prep_time; mkdir -p $O
for v in 1 0; do
  export GIGALENS_HIP_ORDER_FUSED=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/f$v -- python3 $R/tools/prof_kernel.py --workload C2 --iters 200 > $O/f$v.log 2>&1 || exit 1
  echo "== ORDER_FUSED=$v"; f=$(find $O/f$v -name '*kernel_stats.csv' | head -1); cut -d, -f1-6 $f | cut -c1-150 | head -6
done
