cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2u
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 tools/rehearse_2rank.py > gpurun_out/r2u/rehearse_2rank.log 2>&1; echo "rehearse rc=$?"; tail -5 gpurun_out/r2u/rehearse_2rank.log | cut -c1-200
