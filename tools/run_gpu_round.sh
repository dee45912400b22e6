cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py --workload chain --steps 3 --warmup 1 > gpurun_out/bench_chain.json 2>/dev/null
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_default.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-extra > /dev/null 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_default.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/chaintrace -- python3 $GRAFT_REPO_ROOT/bench.py --workload chain --steps 3 --warmup 1 --no-check > /dev/null 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/chaintrace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_chain.csv
find $GRAFT_REPO_ROOT/gpurun_out -name "*kernel_trace.csv" -delete
python3 -c "
import json
d=json.loads(open('$GRAFT_REPO_ROOT/gpurun_out/bench_chain.json').read().strip().splitlines()[-1])
print('chain', d['ms_per_step'], d.get('phases_ms'))
d=json.loads(open('$GRAFT_REPO_ROOT/gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print('default', d['ms_per_step'], d['value'], d['phases_ms_rank0'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline_projection']['kernel_ms'])
print({k:(v.get('ms_per_step') if isinstance(v,dict) else v) for k,v in d.get('extra',{}).items()})"
