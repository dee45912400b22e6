cd $GRAFT_REPO_ROOT
FUZZ_ONLY=fsp4 FUZZ_WIDTHS=1025,1100,1500,2000,2048 SECONDS=240 timeout 600 python tools/fuzz_parity.py 11 2>&1 | tail -3
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
