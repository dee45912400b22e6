cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
timeout 900 python bench.py --workload fsp5 --steps 3 --warmup 1 > gpurun_out/bench_fsp5.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/fsp5trace -- python3 $GRAFT_REPO_ROOT/bench.py --workload fsp5 --steps 3 --warmup 1 --no-check > /dev/null 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/fsp5trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_fsp5.csv
find $GRAFT_REPO_ROOT/gpurun_out/fsp5trace -name "*kernel_trace.csv" -delete
python3 -c "
import csv,json
d=json.loads(open('$GRAFT_REPO_ROOT/gpurun_out/bench_fsp5.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['phases_ms'], d['roofline']['frac'])
for r in list(csv.DictReader(open('$GRAFT_REPO_ROOT/gpurun_out/kernel_stats_fsp5.csv')))[:7]: print(r['Name'][:90].ljust(90), r['Calls'], float(r['AverageNs'])/1e6, float(r['TotalDurationNs'])/1e6)"
