cd $GRAFT_REPO_ROOT
O=gpurun_out/r62; mkdir -p $O
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; tail -c 200 $O/bench.json; tail -2 $O/bench.err
bash tools/profile_bench.sh r62/profile 2>&1 | tail -3
