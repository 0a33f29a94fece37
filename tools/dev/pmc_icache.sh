#!/bin/bash
# instruction-cache and scalar-data-cache counters of the dominant kernel of one workload:  bash tools/dev/pmc_icache.sh <workload>
WL=${1:-C4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_icache_$WL
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python3 $R/tools/prof_kernel.py --workload $WL --iters 6 > $OUT/$n.log 2>&1; }
run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQ_IFETCH GRBM_GUI_ACTIVE
run dc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_BUSY_CYCLES SQC_TC_STALL SQ_INSTS_SMEM GRBM_GUI_ACTIVE
run lvl SQ_IFETCH_LEVEL SQ_IFETCH SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES
python3 $R/tools/pmc_summary.py $OUT 3
