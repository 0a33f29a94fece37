cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/_demo.py
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/demo_stats -- python3 $R/tools/_demo.py > /dev/null 2>&1
