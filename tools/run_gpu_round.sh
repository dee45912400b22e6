cd $GRAFT_REPO_ROOT
bash tools/profile_bench.sh final 2>&1 | tail -4 | cut -c1-300
