#!/bin/bash
# One-box sweep of environment settings of the scan (each run parity-gated by bench.py):
#   tools/sweep_env.sh "EM2_MATRIX_RING=0" "EM2_MATRIX_RING=1 EM2_RING_SEGMENT_COLUMNS=4096" ...
cd "$(dirname "$0")/.."
for setting in "$@"; do
  env $setting python bench.py --steps 4 --warmup 2 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json, sys
try:
    d = json.loads(sys.stdin.read()); r = d['roofline']
    print('%-60s step %.1f  kernel %.1f ms  clock %.3f GHz  frac %.3f  scan %.1f' % ('$setting', d['ms_per_step'], r['kernel_ms'], r.get('clock_ghz') or 0, r['frac'], d['phases_ms_rank0']['scan']))
except Exception as error:
    print('%-60s FAILED (%s)' % ('$setting', error))"
done
