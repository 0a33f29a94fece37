cd $GRAFT_REPO_ROOT
printf "X=0\nGIGALENS_HIP_CHUNK_PX=5632\nGIGALENS_HIP_CHUNK_PX=4096\n" > /tmp/sw.txt
bash tools/dev/sweep_env.sh /tmp/sw.txt
