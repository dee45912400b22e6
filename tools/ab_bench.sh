#!/bin/bash
# Same-box A/B of two builds of the library (kernel times differ by several per cent between the boxes of the pool, and a
# box's clock drifts: rank builds only by interleaved runs on ONE box):
#   tools/ab_bench.sh path/to/A.so path/to/B.so [rounds] [bench.py arguments]
A=$1; B=$2; ROUNDS=${3:-3}; shift 3 || true
cd "$(dirname "$0")/.."
for round in $(seq 1 "$ROUNDS"); do
  for lib in "$A" "$B"; do
    EM2_LIBRARY=$lib python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
r = d['roofline']
print('%-28s step %.1f ms  kernel %.1f ms  clock %.3f GHz  frac %.3f  scan %.1f  projection %.1f' % ('$(basename $lib)', d['ms_per_step'], r['kernel_ms'], r.get('clock_ghz') or 0, r['frac'], d['phases_ms_rank0']['scan'], d['phases_ms_rank0']['projection']))"
  done
done
