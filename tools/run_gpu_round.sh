# scratch script of the current GPU round (edited per call)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r37; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cell_graph.py tests/test_gpu_fsp5.py -x -q 2>&1 | tail -15
timeout 600 python bench.py --workload fsp5 --steps 3 --warmup 1 > $O/fsp5.json 2> $O/fsp5.err; tail -c 2500 $O/fsp5.json; tail -3 $O/fsp5.err
timeout 900 python bench.py --workload chain --steps 2 --warmup 1 > $O/chain.json 2> $O/chain.err; tail -c 2500 $O/chain.json; tail -3 $O/chain.err
timeout 600 python bench.py --workload chain --cells 200000 --steps 2 --warmup 1 > $O/chain200k.json 2> $O/chain200k.err; tail -c 1500 $O/chain200k.json; tail -3 $O/chain200k.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for w in fsp5 chain; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$w -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-check > $R/$O/prof_$w.log 2>&1
  f=$(find $R/$O/prof_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $R/$O/${w}_kernel_stats.csv && head -12 $f
  find $R/$O/prof_$w -name "*kernel_trace.csv" -delete; find $R/$O/prof_$w -name "*agent_info.csv" -delete
done
