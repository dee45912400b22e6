set -e
cd $GRAFT_REPO_ROOT
bash tools/profile_bench.sh r03_default > /dev/null 2>&1 || true
BENCH_ARGS="--workload fsp5" PROFILE_CONFIG='{"workload":"fsp5"}' bash tools/profile_bench.sh r03_fsp5 > /dev/null 2>&1 || true
BENCH_ARGS="--workload chain" PROFILE_CONFIG='{"workload":"chain"}' bash tools/profile_bench.sh r03_chain > /dev/null 2>&1 || true
python3 bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err || true
python3 bench.py --workload fsp5 > gpurun_out/r03_bench_fsp5.json 2> gpurun_out/r03_bench_fsp5.err || true
python3 bench.py --workload chain > gpurun_out/r03_bench_chain.json 2> gpurun_out/r03_bench_chain.err || true
for f in gpurun_out/r03_bench_*.json; do tail -1 $f | cut -c1-400; done
