#!/bin/bash
# One-box sweep of the departure schedule of the matrix scan (EM2_MATRIX_DEPART_US / _WINDOW_US), baseline interleaved:
#   tools/sweep_depart.sh "period window" ...      (0 0 = no departures; SWEEP_ARGS="--lsh-count 2048" for other bench arguments)
cd "$(dirname "$0")/.."
for setting in "$@"; do
  set -- $setting
  EM2_MATRIX_DEPART_US=$1 EM2_MATRIX_DEPART_WINDOW_US=$2 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline --check-rows 1024 $SWEEP_ARGS 2>/dev/null | tail -1 | python3 -c "
import json, sys
try:
    d = json.loads(sys.stdin.read()); r = d['roofline']
    print('depart %4s us window %4s us: step %.1f  kernel %.2f ms  clock %.3f GHz  frac %.3f  scan %.1f' % ('$1', '$2', d['ms_per_step'], r['kernel_ms'], r.get('clock_ghz') or 0, r['frac'], d['phases_ms_rank0']['scan']))
except Exception as error:
    print('depart $1 window $2 FAILED (%s)' % error)"
done
