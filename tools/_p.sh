cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 tools/dev/hmc_profile.py 2>&1 | grep -v amdgpu | tail -4
