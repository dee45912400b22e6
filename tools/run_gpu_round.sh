cd $GRAFT_REPO_ROOT
O=gpurun_out/r58; mkdir -p $O
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; tail -2 $O/bench.err
bash tools/profile_bench.sh r58/profile 2>&1 | tail -4
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for w in chain; do
  timeout 900 python3 $R/bench.py --workload $w --steps 3 --warmup 1 > $R/$O/$w.json 2> $R/$O/$w.err
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$w -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-check > $R/$O/prof_$w.log 2>&1
  f=$(find $R/$O/prof_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $R/$O/${w}_kernel_stats.csv
  find $R/$O/prof_$w -name "*kernel_trace.csv" -delete; find $R/$O/prof_$w -name "*agent_info.csv" -delete
done
timeout 600 python3 $R/bench.py --workload chain --cells 200000 --steps 2 --warmup 1 > $R/$O/chain200k.json 2> $R/$O/chain200k.err
