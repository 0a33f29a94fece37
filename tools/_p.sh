cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2k
GIGALENS_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 100 --warmup 10 > gpurun_out/r2k/bench_2rank_gloo.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/r2k/bench_2rank_gloo.log | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['n_gpus'], r['value'], r['ms_per_step'], r['config']['mode'], r['config']['parallelism'], r.get('sharded_fwdgrad_without_collective'))"
timeout -k 10 300 python bench.py --gpus 2 > gpurun_out/r2k/bench_gpus2_one_device.log 2>&1; echo "rc=$? (expected 2: one device visible)"; tail -2 gpurun_out/r2k/bench_gpus2_one_device.log
