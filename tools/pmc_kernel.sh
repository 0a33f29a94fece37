#!/bin/bash
# PMC passes on the native call of one workload (run on the GPU box):  bash tools/pmc_kernel.sh <tag> <workload> [extra prof_kernel args]
# Each counter group is its own rocprofv3 run with --kernel-trace only, as the pool requires.
set -u
TAG=$1; WL=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python3 $R/tools/prof_kernel.py --workload $WL --iters 6 $EXTRA > $OUT/$n.log 2>&1; }
EXTRA="$*"
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run mix SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD
run clk GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_ANY SQ_INSTS_LDS
if [ "${PMC_MEM:-0}" = "1" ]; then
  # vector-memory path: texture-addresser / L1 busy and stalls, L1 -> L2 requests, matrix-pipe occupancy
  run mem1 SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES
  run mem2 TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
  run mem3 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
  run mem4 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
fi
python3 $R/tools/pmc_summary.py $OUT 4
