#!/usr/bin/env python3
"""ExpressionMatrix.findSimilarPairs4 on a data directory, CALLS times in a row: seconds per call and what the library keeps on the
device between calls (free device memory before the first call against after each).  CELLS / GENES / CALLS from the environment."""
import os, sys, time, tempfile, shutil
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from expressionmatrix2_amd import ExpressionMatrix, capi, files, synthetic
C = int(os.environ.get("CELLS", 1000000)); G = int(os.environ.get("GENES", 30000)); calls = int(os.environ.get("CALLS", 24))
toc, data = synthetic.expression_shard(0, C, G, density=0.01, device="cuda")
t_h, g_h, c_h = synthetic.csr_to_host(toc, data)
d = tempfile.mkdtemp(prefix="em2facade", dir="/tmp")
files.create_directory(d, G, t_h, capi.make_counts(g_h, c_h))
del toc, data; torch.cuda.empty_cache(); torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
e = ExpressionMatrix(d)
times, kept = [], []
for rep in range(calls):
    t0 = time.perf_counter(); e.findSimilarPairs4(similarPairsName="P"); times.append(time.perf_counter() - t0)
    kept.append((free0 - torch.cuda.mem_get_info()[0]) / 2**30)
print("findSimilarPairs4 x %d at %d cells: first %.2f s, then min %.3f / median %.3f / max %.3f s; kept on the device between calls: %.1f GB (max %.1f)"
      % (calls, C, times[0], min(times[1:]), float(np.median(times[1:])), max(times[1:]), kept[-1], max(kept)), flush=True)
capi.dev_release_scratch() if hasattr(capi, "dev_release_scratch") else None
print("after em2_dev_release_scratch: %.1f GB" % ((free0 - torch.cuda.mem_get_info()[0]) / 2**30))
shutil.rmtree(d)
