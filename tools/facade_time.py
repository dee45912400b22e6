import os, sys, time, tempfile, shutil
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from expressionmatrix2_amd import ExpressionMatrix, capi, files, synthetic
C = int(os.environ.get("CELLS", 200000)); G = int(os.environ.get("GENES", 20000))
toc, data = synthetic.expression_shard(0, C, G, density=0.01, device="cuda")
t_h, g_h, c_h = synthetic.csr_to_host(toc, data)
d = tempfile.mkdtemp(prefix="em2facade", dir="/tmp")
t0 = time.perf_counter(); files.create_directory(d, G, t_h, capi.make_counts(g_h, c_h)); print("create_directory %.2f s" % (time.perf_counter() - t0))
files.add_gene_set(d, "Half", np.arange(0, G, 2, dtype=np.uint32))
del toc, data; torch.cuda.empty_cache()
e = ExpressionMatrix(d)
for gs in ("AllGenes", "Half"):
    for rep in range(2):
        t0 = time.perf_counter(); e.findSimilarPairs4(geneSetName=gs, similarPairsName="P"); dt = time.perf_counter() - t0
        print("findSimilarPairs4(%s) run %d: %.2f s" % (gs, rep, dt), flush=True)
# BASELINE config E: the consumers of SimilarPairs
for rep in range(2):
    name = "G%d" % rep
    t0 = time.perf_counter(); e.createCellGraph(name, "AllCells", "P", float(os.environ.get("GRAPH_THRESHOLD", 0.2)), 20); dt = time.perf_counter() - t0
    info = e._cell_graph_information(name)
    print("createCellGraph run %d: %.2f s (%d vertices, %d edges)" % (rep, dt, info["vertexCount"], info["edgeCount"]), flush=True)
    t0 = time.perf_counter(); cells, clusters = e.labelPropagationClustering(name); dt = time.perf_counter() - t0
    print("labelPropagationClustering run %d: %.2f s (%d clusters)" % (rep, dt, int(clusters.max()) + 1 if len(clusters) else 0), flush=True)
shutil.rmtree(d)
