cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_signatures.py tests/test_gpu_facade.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/vs -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-extra --no-cpu-baseline --no-check > /dev/null 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/vs -name "*kernel_stats.csv" | head -1 | xargs -I{} grep "vectorStats\|vectorsTo" {} | cut -c1-200
find $GRAFT_REPO_ROOT/gpurun_out/vs -name "*kernel_trace.csv" -delete
