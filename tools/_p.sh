cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
GIGALENS_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 100 --warmup 10 > gpurun_out/r2f/bench_2rank_gloo.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/r2f/bench_2rank_gloo.log | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['n_gpus'], r['value'], r['ms_per_step'], r['config']['mode'], r['config']['parallelism'], r.get('sharded_fwdgrad_without_collective'))" && 
GIGALENS_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 50 --warmup 5 --workload C5 > gpurun_out/r2f/bench_2rank_gloo_C5.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/r2f/bench_2rank_gloo_C5.log | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['n_gpus'], r['value'], r['ms_per_step'], r['config']['mode'], r['config']['parallelism'], r.get('sharded_fwdgrad_without_collective'))"
