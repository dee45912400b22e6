cd $GRAFT_REPO_ROOT
O=gpurun_out/r47; mkdir -p $O
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json; tail -3 $O/bench.err
bash tools/profile_bench.sh r47/profile 2>&1 | tail -6
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for w in fsp5 chain; do
  timeout 900 python3 $R/bench.py --workload $w --steps 3 --warmup 1 > $R/$O/$w.json 2> $R/$O/$w.err
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$w -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-check > $R/$O/prof_$w.log 2>&1
  f=$(find $R/$O/prof_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $R/$O/${w}_kernel_stats.csv
  find $R/$O/prof_$w -name "*kernel_trace.csv" -delete; find $R/$O/prof_$w -name "*agent_info.csv" -delete
  python3 - <<PY
import json
d=json.loads(open("$R/$O/$w.json").read().strip().splitlines()[-1])
print("$w", round(d["ms_per_step"],1), d["phases_ms"], d.get("roofline") and round(d["roofline"]["frac"],3))
PY
done
