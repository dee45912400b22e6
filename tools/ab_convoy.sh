#!/bin/bash
# Same-box A/B of the convoy of the matrix scan (EM2_MATRIX_CONVOY 0 / 1), 1024 and 2048 bits, alternating:
#   tools/ab_convoy.sh [rounds]
cd "$(dirname "$0")/.."
run() {
  EM2_MATRIX_CONVOY=$1 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline --check-rows 1024 $2 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read()); r = d['roofline']
print('convoy $1 $2: step %.1f  kernel %.2f ms  clock %.3f GHz  frac %.3f  scan %.1f' % (d['ms_per_step'], r['kernel_ms'], r.get('clock_ghz') or 0, r['frac'], d['phases_ms_rank0']['scan']))"
}
for i in $(seq 1 "${1:-3}"); do run 0 ""; run 1 ""; done
for i in $(seq 1 "${1:-3}"); do run 0 "--lsh-count 2048"; run 1 "--lsh-count 2048"; done
