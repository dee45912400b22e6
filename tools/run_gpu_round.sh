cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fsp5.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
FUZZ_ONLY=fsp5 SECONDS=100 timeout 400 python tools/fuzz_parity.py 77 2>&1 | tail -1
timeout 900 python bench.py --workload fsp5 --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['phases_ms'], d['parity_check'])"
