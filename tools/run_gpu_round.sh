cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_dist_entry.py -x -q 2>&1 | tail -40
