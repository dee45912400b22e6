cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/mx
rm -rf $O; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -i -E "mfma|SQ_INSTS_VALU\b|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|LDS_BANK|SQ_INST_CYCLES_VMEM|SQ_ACTIVE_INST_LDS|SQ_INSTS_LDS" | head -40 > $O/counters.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 3 --warmup 1 > $O/bench_under_trace.json 2> $O/trace.log
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS"; do
  d=$O/pmc_$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $d.log 2>&1
done
python3 $R/tools/pmc_summary.py $O/pmc_* > $O/pmc_summary.json
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
head -c 1500 $O/kernel_stats.csv | cut -c1-220; cat $O/counters.txt | head -30; tail -2 $O/bench_under_trace.json | cut -c1-400
