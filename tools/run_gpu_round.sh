cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -1; done
