cd $GRAFT_REPO_ROOT
O=gpurun_out/r67; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fsp5.py tests/test_gpu_fsp7.py -x -q 2>&1 | tail -4
for mode in packed unpacked; do
EM2_FSP5_SELECT=$mode timeout 600 python bench.py --workload fsp5 --steps 3 --warmup 1 > $O/fsp5_$mode.json 2> $O/fsp5_$mode.err
python - <<PY
import json
d=json.loads(open("$O/fsp5_$mode.json").read().strip().splitlines()[-1])
print("$mode", round(d["ms_per_step"],1), d["phases_ms"], d["parity_check"])
PY
done
