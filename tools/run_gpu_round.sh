cd $GRAFT_REPO_ROOT
for L in 1024 2048; do
for seg in 16384 4096; do
  echo "L=$L segment columns $seg"
  EM2_MIN_SEGMENT_COLUMNS=$seg EM2_SCAN_VERBOSE=1 EM2_MATRIX_DIAG=2048 timeout 900 python bench.py --lsh-count $L --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-check 2>&1 | grep "wave cycles\|ms_per_step" | tail -2 | cut -c1-400
done
done
