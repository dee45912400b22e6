cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
EM2_SCAN_VERBOSE=1 SWEEP="EM2_SCAN_MODE=virtual,EM2_VIRTUAL_WORLD=8;EM2_SCAN_MODE=virtual,EM2_VIRTUAL_WORLD=2" REPEATS=2 timeout 900 python tools/scale_check.py sweep > gpurun_out/sweep.txt 2>&1
grep -n "phase\|rank\|sweep" gpurun_out/sweep.txt | cut -c1-330 | tail -30
