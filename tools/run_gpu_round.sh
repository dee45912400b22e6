cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_signatures.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
FUZZ_ONLY=signatures SECONDS=60 timeout 300 python tools/fuzz_parity.py 19 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/projtrace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/projtrace_bench.json 2>/dev/null
find $GRAFT_REPO_ROOT/gpurun_out/projtrace -name "*kernel_stats.csv" | head -1 | xargs -I{} python3 -c "
import csv
for r in list(csv.DictReader(open('{}')))[:12]: print(r['Name'][:90].ljust(90), r['Calls'], float(r['AverageNs'])/1e6)"
find $GRAFT_REPO_ROOT/gpurun_out/projtrace -name "*kernel_trace.csv" -delete
cut -c1-400 $GRAFT_REPO_ROOT/gpurun_out/projtrace_bench.json | tail -1
