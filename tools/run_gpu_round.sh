cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -1
SECONDS=200 timeout 500 python tools/fuzz_parity.py 8675309 2>&1 | tail -1
FUZZ_ONLY=signatures SECONDS=60 timeout 300 python tools/fuzz_parity.py 42 2>&1 | tail -1
