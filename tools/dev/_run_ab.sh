set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r4_cw_gputests.log 2>&1; tail -4 gpurun_out/r4_cw_gputests.log
python tools/dev/cluster_ab.py C4 1,2 2>&1 | grep -v "^$" | sed "s/ .opt.amdgpu.*//"
python tools/dev/cluster_ab.py C5 1,2 2>&1 | grep -v "^$" | sed "s/ .opt.amdgpu.*//"
bash tools/pmc_kernel.sh r4cw_C4 C4 2>&1 | tail -30
