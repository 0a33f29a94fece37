#!/bin/bash
# Run on the GPU box (gpurun): collects the rocprofv3 evidence bench.py's numbers are checked against.
#   bash tools/collect_profiles.sh r1
# Counter passes are separate runs with --kernel-trace only (no sys/hip tracing), as the pool requires.
set -u
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -- python3 $R/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
for W in C2 C3 C4 C5 C6; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kernel_stats_$W -- python3 $R/tools/prof_kernel.py --workload $W --iters 20 > $OUT/kernel_stats_$W.log 2>&1
done
# the reference's demo set-up (60x60, supersample 2, 13x13 PSF through subgrid_kernel, 500 samples): PSF / pooling kernels + pair kernels
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kernel_stats_demo -- python3 $R/tools/dev/prof_demo.py 2 > $OUT/kernel_stats_demo.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kernel_stats_C3L -- python3 $R/tools/prof_kernel.py --workload C3L --mode lstsq --iters 8 > $OUT/kernel_stats_C3L.log 2>&1
# the simulate() boundary itself at C2: gl_simulate_fwd writes the image, gl_simulate_bwd reads its cotangent (the pair that moves B1)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kernel_stats_simpair -- python3 $R/tools/prof_kernel.py --workload C2 --mode simpair --iters 40 > $OUT/kernel_stats_simpair.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_simpair_fetch -- python3 $R/tools/prof_kernel.py --workload C2 --mode simpair --iters 8 > $OUT/pmc_simpair_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_simpair_write -- python3 $R/tools/prof_kernel.py --workload C2 --mode simpair --iters 8 > $OUT/pmc_simpair_write.log 2>&1
# the cluster kernel (C4): issue and wait counters
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_C4 -- python3 $R/tools/prof_kernel.py --workload C4 --iters 6 > $OUT/pmc_C4.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_C6 -- python3 $R/tools/prof_kernel.py --workload C6 --iters 6 > $OUT/pmc_C6.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_C3L -- python3 $R/tools/prof_kernel.py --workload C3L --mode lstsq --iters 4 > $OUT/pmc_C3L.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/prof_kernel.py --iters 8 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/tools/prof_kernel.py --iters 8 > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $R/tools/prof_kernel.py --iters 8 > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_mix -- python3 $R/tools/prof_kernel.py --iters 8 > $OUT/pmc_mix.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $OUT/pmc_clk -- python3 $R/tools/prof_kernel.py --iters 8 > $OUT/pmc_clk.log 2>&1
# round 3: the shapelet kernel (table and direct mode): issue / wait / memory-path counters, and the gradient accuracy scan
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kernel_stats_C3direct -- python3 $R/tools/prof_kernel.py --workload C3 --direct --iters 20 > $OUT/kernel_stats_C3direct.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kernel_stats_C3D -- python3 $R/tools/prof_kernel.py --workload C3D --iters 20 > $OUT/kernel_stats_C3D.log 2>&1
for G in "sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" \
         "mix SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
         "clk GRBM_GUI_ACTIVE SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
         "mem TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
         "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $G; N=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_C3_$N -- python3 $R/tools/prof_kernel.py --workload C3 --iters 6 > $OUT/pmc_C3_$N.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_C3direct -- python3 $R/tools/prof_kernel.py --workload C3 --direct --iters 6 > $OUT/pmc_C3direct.log 2>&1
python3 $R/tools/dev/grad_error_distribution.py --n 256 > $OUT/grad_error_distribution.jsonl 2> $OUT/grad_error_distribution.err
python3 $R/tools/dev/grad_accuracy_scan.py --n 8 > $OUT/grad_accuracy.jsonl 2> $OUT/grad_accuracy.err
python3 $R/tools/dev/grad_accuracy_scan.py --configs cases > $OUT/grad_accuracy_cases.jsonl 2>> $OUT/grad_accuracy.err
python3 -m pytest $R/tests/test_gpu_dist.py -m gpu -q > $OUT/two_rank_test.log 2>&1
python3 $R/tools/map_step_time.py > $OUT/map_step_time.log 2>&1
python3 $R/tools/svi_hmc_step_time.py > $OUT/svi_hmc_step_time.log 2>&1
python3 $R/tools/bench_configs.py > $OUT/bench_configs.jsonl 2> $OUT/bench_configs.err
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 $R/bench.py --no-cpu-baseline --mode svi > $OUT/bench_svi.json 2> $OUT/bench_svi.err
python3 $R/bench.py --no-cpu-baseline --workload C5 --steps 200 --warmup 20 > $OUT/bench_C5.json 2> $OUT/bench_C5.err
python3 $R/bench.py --no-cpu-baseline --workload C5 --mode svi --steps 200 --warmup 20 > $OUT/bench_C5_svi.json 2> $OUT/bench_C5_svi.err
# round 4: the N > 1 line (C2 value, svi_step, BASELINE configs[4] in `configs`) rehearsed with two gloo ranks sharing this GPU
GIGALENS_DIST_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 $R/bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err
# the cluster kernel's issue / wait / LDS counters (tools/pmc_kernel.sh: four passes)
bash $R/tools/pmc_kernel.sh ${TAG}_C4 C4 > $OUT/pmc_kernel_C4.log 2>&1
tail -1 $OUT/bench.json
