cd $GRAFT_REPO_ROOT
O=gpurun_out/r68; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 $R/bench.py --workload fsp5 --steps 3 --warmup 1 > $R/$O/fsp5.json 2> $R/$O/fsp5.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_fsp5 -- python3 $R/bench.py --workload fsp5 --steps 3 --warmup 1 --no-check > $R/$O/prof_fsp5.log 2>&1
f=$(find $R/$O/prof_fsp5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$O/fsp5_kernel_stats.csv
find $R/$O/prof_fsp5 -name "*kernel_trace.csv" -delete; find $R/$O/prof_fsp5 -name "*agent_info.csv" -delete
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"]); 
for k,v in d["extra"].items(): print(k, v["ms_per_step"], v.get("phases_ms"))
d=json.loads(open("$O/fsp5.json").read().strip().splitlines()[-1]); print("fsp5", d["ms_per_step"], d["phases_ms"], d["roofline"]["frac"])
PY
