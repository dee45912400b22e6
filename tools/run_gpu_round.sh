cd $GRAFT_REPO_ROOT
FUZZ_ONLY=fsp4 SECONDS=400 timeout 700 python tools/fuzz_parity.py 31337 2>&1 | tail -1
FUZZ_ONLY=fsp4 FUZZ_WIDTHS=1025,1100,1500,2000,2048 SECONDS=300 timeout 600 python tools/fuzz_parity.py 271828 2>&1 | tail -1
FUZZ_ONLY=signatures SECONDS=200 timeout 500 python tools/fuzz_parity.py 1618 2>&1 | tail -1
FUZZ_ONLY=fsp5 SECONDS=200 timeout 500 python tools/fuzz_parity.py 1414 2>&1 | tail -1
