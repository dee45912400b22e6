#!/bin/bash
# Builds the working tree's csrc/em2_cluster.hip (or the files named after the variant) into expressionmatrix2_amd/libem2lsh_<name>.so,
# the other objects taken from csrc/build: for same-box A/B runs (EM2_LIBRARY=... ; boxes of the pool differ by a few per cent).
#   tools/build_variant.sh v1 [file.hip ...]
set -e
cd "$(dirname "$0")/../expressionmatrix2_amd/csrc"
name=$1; shift
files=${@:-em2_cluster.hip}
mkdir -p build_variant_$name
objs=$(ls build/*.o)
for f in $files; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -c $f -o build_variant_$name/$(basename $f).o
  objs=$(echo "$objs" | grep -v "build/$(basename $f).o")
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../libem2lsh_$name.so $objs build_variant_$name/*.o -ldl
ls -la ../libem2lsh_$name.so
