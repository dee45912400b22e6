cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
timeout 900 python bench.py --workload fsp5 --steps 3 --warmup 1 > gpurun_out/bench_fsp5_$i.json 2>/dev/null
python3 -c "
import json
d=json.loads(open('gpurun_out/bench_fsp5_$i.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['phases_ms'], d['roofline']['frac'])"
done
