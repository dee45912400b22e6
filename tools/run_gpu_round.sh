cd $GRAFT_REPO_ROOT
O=gpurun_out/r63; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fsp4.py -x -q -k "matrix or sharded or layouts" 2>&1 | tail -3
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra --check-rows 64 > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
    print("$name", round(d["ms_per_step"],1), d["phases_ms_rank0"], round(d["roofline"]["kernel_ms"],1), d["parity_check"])
except Exception as e:
    print("$name no json", e); print(open("$O/$name.err").read()[-1500:])
PY
}
run full A=1
run full2 A=1
