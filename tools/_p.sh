cd $GRAFT_REPO_ROOT
timeout -k 10 800 python3 tools/dev/demo_pipeline.py 2>&1 | grep -v amdgpu.ids | tail -60
