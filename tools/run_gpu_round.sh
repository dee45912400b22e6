cd $GRAFT_REPO_ROOT
SECONDS=240 timeout 400 python3 tools/fuzz_parity.py 7 2>&1 | tail -6
SECONDS=240 timeout 400 python3 tools/fuzz_parity.py 1234 2>&1 | tail -4
