cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_limits.py tests/test_gpu_parity.py -m gpu -q -k "limits or cluster_kernel or refused or oversized" 2>&1 | grep -E "^E  |^>|passed|failed|^FAILED" | head -30
