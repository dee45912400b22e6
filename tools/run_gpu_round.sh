cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_fsp4.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
for rep in 1 2; do
for lib in "" "$GRAFT_REPO_ROOT/tools/ubench/libem2lsh_old.so"; do
  EM2_LIBRARY=$lib timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lib=[$lib]', d['ms_per_step'], d['phases_ms_rank0'], d['roofline'].get('kernel_ms'), d['roofline'].get('inbox_entries'), d['parity_check'].get('after_timing_rows'))"
done
done
EM2_SCAN_VERBOSE=1 EM2_MATRIX_DIAG=2048 timeout 900 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-check 2>&1 | grep "wave cycles" | tail -1 | cut -c1-300
