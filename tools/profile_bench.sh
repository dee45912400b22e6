#!/bin/bash
# The profile recipe behind profiles/rNN_*: kernel trace + stats of bench.py, then one --pmc pass per counter group
# (the PMC passes never share a run with the trace domains), summed per kernel by tools/pmc_summary.py.
# Run on the GPU box from anywhere:  bash tools/profile_bench.sh [output directory under gpurun_out/]
# BENCH_ARGS="--lsh-count 2048" profiles another configuration of bench.py (e.g. "--workload fsp5"); PROFILE_CONFIG='{"cells": ...}'
# goes into the digest (tools/profile_digest.py -> digest.json: what is committed under profiles/).
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O="$R/gpurun_out/${1:-profile}"
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-extra ${BENCH_ARGS:-} > "$O/bench_under_trace.json" 2> "$O/trace.log" || true
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
  d="$O/pmc_$(echo "$c" | tr ' ' '_' | cut -c1-40)"
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$d" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-extra ${BENCH_ARGS:-} > "$d.log" 2>&1 || true
done
python3 "$R/tools/pmc_summary.py" "$O"/pmc_* > "$O/pmc_summary.json"
find "$O/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$O/kernel_stats.csv"
python3 "$R/tools/profile_digest.py" "$O" "${PROFILE_CONFIG:-}" > "$O/digest.json" || true
find "$O" -name "*counter_collection.csv" -delete; find "$O" -name "*kernel_trace.csv" -delete; find "$O" -name "*agent_info.csv" -delete
head -c 1200 "$O/kernel_stats.csv" | cut -c1-200; tail -1 "$O/bench_under_trace.json" | cut -c1-600
