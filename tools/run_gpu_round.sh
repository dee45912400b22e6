cd $GRAFT_REPO_ROOT
timeout 600 python - <<'P'
import os, sys, time
sys.path.insert(0, "tests")
import numpy as np, synth
from expressionmatrix2_amd import capi
n, L = 300000, 2048
sig = synth.clustered_signatures(n, L, cluster_count=64, flip=0.15, seed=3)
os.environ["EM2_SCAN_MODE"] = "virtual"; os.environ["EM2_VIRTUAL_WORLD"] = "4"
res = {}
for wide in ("1", "0"):
    os.environ["EM2_SCAN_MATRIX_WIDE"] = wide
    for rep in range(2):
        t = time.time(); pairs, used = capi.find_similar_pairs4(sig, L, 100, 0.2); dt = time.time() - t
        info = capi.dev_find_similar_pairs4_last_launch()
    print("virtual world 4, wide", wide, "wall %.3f s" % dt, {k: info[k] for k in ("form", "matrix_pairs", "inbox_entries")})
    res[wide] = (pairs["cell"].copy(), pairs["similarity"].copy(), used.copy())
print("same bytes:", all(np.array_equal(a, b) for a, b in zip(res["1"], res["0"])))
P
