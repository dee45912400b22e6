cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
FUZZ_ONLY=fsp4 SECONDS=120 timeout 400 python tools/fuzz_parity.py 77 2>&1 | tail -1
