cd $GRAFT_REPO_ROOT
O=gpurun_out/r59; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_signatures.py -x -q 2>&1 | tail -3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --check-rows 64 > $R/$O/prof.log 2>&1
f=$(find $R/$O/prof -name "*kernel_stats.csv" | head -1); python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:5]: print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e6,2))
PY
find $R/$O/prof -name "*kernel_trace.csv" -delete
