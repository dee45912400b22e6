#!/bin/bash
# Builder's loop: build the library, check the compiled matrix walk, and only then spend GPU time on the command given.
#   tools/gpu.sh 'python -m pytest tests -m gpu -x -q'            (GPU_TIMEOUT=seconds, GPU_TAIL=lines)
set -e
cd "$(dirname "$0")/.."
make -C expressionmatrix2_amd/csrc 2>&1 | grep -i "error" -A8 && { echo "BUILD FAILED"; exit 1; }
make -C expressionmatrix2_amd/csrc -q || { echo "BUILD INCOMPLETE"; exit 1; }
mkdir -p /tmp/dis
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-inline-asm --cuda-device-only -S -o /tmp/dis/check.s expressionmatrix2_amd/csrc/em2_scan_symmetric.hip 2>/dev/null
python3 tools/check_matrix_walk_registers.py /tmp/dis/check.s | tail -4
python3 tools/check_matrix_walk_registers.py /tmp/dis/check.s > /dev/null || { echo "REGISTER CHECK FAILED"; exit 1; }
/usr/local/graft/bin/gpurun --timeout ${GPU_TIMEOUT:-1500} -- "$1" 2>&1 | tail -${GPU_TAIL:-30}
