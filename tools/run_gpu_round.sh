# scratch script of the current GPU round (edited per call)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r40; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_analyze_lsh.py -x -q 2>&1 | tail -25
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
