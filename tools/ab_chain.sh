#!/bin/bash
# Same-box A/B of label propagation builds: tools/ab_chain.sh v0 v1 ...  (libraries expressionmatrix2_amd/libem2lsh_<name>.so, see
# build_variant.sh; "." = the product library).  Two rounds, so that drift of the box shows.
cd "$(dirname "$0")/.."
for round in 1 2; do
  for name in "$@"; do
    lib=$PWD/expressionmatrix2_amd/libem2lsh_$name.so; [ "$name" = "." ] && lib=$PWD/expressionmatrix2_amd/libem2lsh.so
    echo -n "$name: "
    EM2_LIBRARY=$lib EM2_TIMING=1 python bench.py --workload chain --steps 2 --warmup 1 --no-check --no-cpu-baseline 2>&1 | grep "whole call" | tail -2 | sed 's/.*released //' | tr "\n" " "
    echo
  done
done
