cd $GRAFT_REPO_ROOT
O=gpurun_out/r61; mkdir -p $O
T0=$(date +%s); python bench.py > $O/bench.json 2> $O/bench.err; echo wall $(( $(date +%s) - T0 )) s; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["phases_ms_rank0"], d["roofline"]["frac"])
for k,v in d["extra"].items(): print(k, v["ms_per_step"], v.get("phases_ms"), v["parity_check"])
PY
tail -3 $O/bench.err
