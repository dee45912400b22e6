cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_fsp4.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
for rep in 1 2; do
for lib in "" "$GRAFT_REPO_ROOT/tools/ubench/libem2lsh_nocarry.so"; do
  EM2_LIBRARY=$lib timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lib=[$lib]', d['ms_per_step'], d['phases_ms_rank0'], d['roofline'].get('kernel_ms'), d['parity_check'].get('after_timing_rows'))"
done
done
