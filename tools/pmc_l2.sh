#!/bin/bash
# L2 (TCC) hit / miss / fabric request counters of one bench.py configuration, per kernel: BENCH_ARGS="..." bash tools/pmc_l2.sh <dir>
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O="$R/gpurun_out/${1:-l2}"; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for c in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCC_EA_WRREQ_sum TCC_WRITE_sum"; do
  d="$O/pmc_$(echo "$c" | tr ' ' '_' | cut -c1-40)"
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$d" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-extra ${BENCH_ARGS:-} > "$d.log" 2>&1 || true
done
python3 "$R/tools/pmc_summary.py" "$O"/pmc_* > "$O/pmc_summary.json"
python3 - <<P
import json
d=json.load(open("$O/pmc_summary.json"))
def walk(x,path=""):
    if isinstance(x,dict):
        for k,v in x.items(): walk(v,path+"/"+str(k))
    else:
        if any(s in path for s in ("fsp4ScanMatrix","projectionScreenQuantized","filterWide","labelProp")) : print(path[:140], x)
walk(d)
P
find "$O" -name "*counter_collection.csv" -delete; find "$O" -name "*kernel_trace.csv" -delete; find "$O" -name "*agent_info.csv" -delete
