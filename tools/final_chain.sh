set -e
cd $GRAFT_REPO_ROOT
BENCH_ARGS="--workload chain" PROFILE_CONFIG='{"workload":"chain"}' bash tools/profile_bench.sh r03_chain > /dev/null 2>&1 || true
python3 bench.py --workload chain > gpurun_out/r03_bench_chain.json 2> gpurun_out/r03_bench_chain.err || true
tail -1 gpurun_out/r03_bench_chain.json | cut -c1-300
python3 tools/kernel_stats_short.py gpurun_out/r03_chain/kernel_stats.csv | head -4
