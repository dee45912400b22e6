cd $GRAFT_REPO_ROOT
O=gpurun_out/r57; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_signatures.py tests/test_gpu_facade.py -x -q 2>&1 | tail -8
python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["phases_ms_rank0"], d["parity_check"]); print(d["roofline_projection"]); print(d["extra"])
PY
