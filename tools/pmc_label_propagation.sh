set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lp5; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_WAVE_CYCLES" "SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS"; do
  d="$O/pmc_$(echo "$c" | tr ' ' '_' | cut -c1-40)"
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$d" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-extra --workload chain --no-check > "$d.log" 2>&1 || true
done
python3 "$R/tools/pmc_summary.py" "$O"/pmc_* > "$O/pmc_summary.json"
python3 - <<P
import json
d=json.load(open("$O/pmc_summary.json"))
def walk(x,path=""):
    if isinstance(x,dict):
        for k,v in x.items():
            if "labelProp" in k or "labelProp" in path: walk(v,path+"/"+k)
            elif isinstance(v,dict): walk(v,path+"/"+k)
    else:
        print(path, x)
walk(d)
P
find "$O" -name "*counter_collection.csv" -delete; find "$O" -name "*kernel_trace.csv" -delete; find "$O" -name "*agent_info.csv" -delete
