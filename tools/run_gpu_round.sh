cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py --lsh-count 2048 --steps 3 --warmup 1 > gpurun_out/bench_2048.json 2> gpurun_out/bench_2048.err; tail -c 600 gpurun_out/bench_2048.err
python - <<'P'
import json
d = json.loads(open("gpurun_out/bench_2048.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["phases_ms_rank0"], {k: d["roofline"][k] for k in ("kernel", "kernel_ms", "frac", "inbox_entries")})
P
for knobs in "EM2_MIN_SEGMENT_COLUMNS=8192" "EM2_MIN_SEGMENT_COLUMNS=32768" "EM2_SCAN_MATRIX_WIDE=0"; do
  env $knobs timeout 900 python bench.py --lsh-count 2048 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$knobs', d['ms_per_step'], d['phases_ms_rank0'], d['roofline'].get('kernel_ms'))"
done
