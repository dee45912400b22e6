cd $GRAFT_REPO_ROOT
O=gpurun_out/r65; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cell_graph.py tests/test_gpu_label_propagation.py -x -q 2>&1 | tail -3
EM2_TIMING=1 timeout 900 python bench.py --workload chain --steps 2 --warmup 1 --no-check > $O/chain.json 2> $O/chain.err; python - <<PY
import json
d=json.loads(open("$O/chain.json").read().strip().splitlines()[-1]); print(d["ms_per_step"], d["phases_ms"], d["config"]["edges"])
PY
grep "timing\|label" $O/chain.err | tail -25
