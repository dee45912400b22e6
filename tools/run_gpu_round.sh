cd $GRAFT_REPO_ROOT
O=gpurun_out/r66; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 $R/bench.py --workload chain --steps 3 --warmup 1 > $R/$O/chain.json 2> $R/$O/chain.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_chain -- python3 $R/bench.py --workload chain --steps 3 --warmup 1 --no-check > $R/$O/prof_chain.log 2>&1
f=$(find $R/$O/prof_chain -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$O/chain_kernel_stats.csv
find $R/$O/prof_chain -name "*kernel_trace.csv" -delete; find $R/$O/prof_chain -name "*agent_info.csv" -delete
timeout 600 python3 $R/bench.py --workload chain --cells 200000 --steps 2 --warmup 1 > $R/$O/chain200k.json 2> $R/$O/chain200k.err
cd $R; python - <<PY
import json
for n in ("chain","chain200k"):
    d=json.loads(open("$O/%s.json"%n).read().strip().splitlines()[-1]); print(n, d["ms_per_step"], d["phases_ms"], d["parity_check"])
PY
