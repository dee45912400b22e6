cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CELLS=1000000 GENES=30000 EM2_TIMING=1 timeout 900 python tools/facade_time.py 2>&1 | grep -v "label propagation: iteration" > gpurun_out/facade_time.txt
wc -l gpurun_out/facade_time.txt
