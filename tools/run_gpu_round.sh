# scratch script of the current GPU round (edited per call)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r36; mkdir -p $O
run() { # name, env...
  name=$1; shift
  env "$@" EM2_SCAN_VERBOSE=1 timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-check > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
    print("$name", round(d["ms_per_step"],1), d["phases_ms_rank0"], round(d["roofline"]["kernel_ms"],1))
except Exception as e:
    print("$name no json", e); print(open("$O/$name.err").read()[-800:])
PY
  grep "matrix kernel" $O/$name.err | tail -1
}
four() { # name ranks cells env...
  name=$1; ranks=$2; cells=$3; shift; shift; shift
  env "$@" EM2_BENCH_SHARE_DEVICE=1 EM2_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 EM2_BLOCKS_PER_CU=1 EM2_SHARDED_MIN_CELLS=1000 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $ranks --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus $ranks --steps 1 --warmup 0 --cells $cells --genes 3000 --no-cpu-baseline --check-rows 96 > $O/$name.out 2> $O/$name.err
  echo "$name rc $? $(grep -h PARITY $O/$name.err | head -2)"
}
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; tail -3 $O/bench.err
bash tools/profile_bench.sh r36/profile 2>&1 | tail -12
exit 0
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L > $R/$O/counters_list.txt 2>/dev/null
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH"; do
  d=$R/$O/pmc_$(echo $c | tr ' ' '_' | cut -c1-30)
  EM2_MATRIX_WALK=1 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check > $d.log 2>&1
done
python3 $R/tools/pmc_summary.py $R/$O/pmc_* > $R/$O/pmc_summary.json 2>/dev/null
find $R/$O -name "*counter_collection.csv" -delete; find $R/$O -name "*kernel_trace.csv" -delete; find $R/$O -name "*agent_info.csv" -delete
python3 - <<PY
import json
d=json.load(open("$R/$O/pmc_summary.json"))
for k,v in d.items():
    if "Matrix" in k or "matrix" in k: print(k, json.dumps(v, indent=0)[:2500])
PY
