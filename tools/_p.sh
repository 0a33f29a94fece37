cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r2 2>&1 | tail -1 | cut -c1-150
