cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 300 gpurun_out/bench_default.err; cut -c1-700 gpurun_out/bench_default.json
