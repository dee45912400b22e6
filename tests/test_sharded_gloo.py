"""The N>1 path on CPU: two processes, gloo backend, the sharding / all-gather / collection logic of
expressionmatrix2_amd.sharded with the CPU oracle standing in for the GPU compute of each rank.  The collective
result must equal the single-process oracle result byte for byte."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from expressionmatrix2_amd import sharded

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_cover_cells_contiguously():
    for cells in (1, 2, 3, 7, 64, 65, 1000, 1001):
        for world in (1, 2, 3, 8):
            ranges = [sharded.shard_range(cells, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == cells
            for a, b in zip(ranges, ranges[1:]):
                assert a[1] == b[0]
            assert max(e - b for b, e in ranges) == sharded.shard_size(cells, world)


WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch.distributed as dist
import oracle_binding
from expressionmatrix2_amd import ExpressionMatrix, capi, sharded

class OracleBackend:
    comm_device = "cpu"
    def __init__(self): self.o = oracle_binding.load_oracle()
    def project(self, toc, data, gene_count, vectors, lsh_count):
        return self.o.compute_signatures(toc, data["gene"], data["count"], gene_count, vectors, lsh_count)
    def scan_rows(self, sig, b, e, L, k, thr):
        cell, sim, used = self.o.find_similar_pairs4_rows(sig, L, k, thr, b, e)
        pairs = np.zeros((e - b, k), dtype=capi.PAIR_DTYPE)
        pairs["cell"] = cell; pairs["similarity"] = sim
        return pairs, used
    def bucket_rows(self, sig, b, e, L, k, thr, q, ovf):
        cell, sim, used = self.o.find_similar_pairs5_rows(sig, L, k, thr, q, ovf, b, e)
        pairs = np.zeros((e - b, k), dtype=capi.PAIR_DTYPE)
        pairs["cell"] = cell; pairs["similarity"] = sim
        return pairs, used

dist.init_process_group(backend="gloo")
e = ExpressionMatrix(sys.argv[2])
sharded.find_similar_pairs4_collective(e, "AllGenes", "AllCells", "Sharded", 7, 0.1, 256, 231, dist, OracleBackend())
sharded.find_similar_pairs4_collective(e, "Sub", "Odd", "ShardedSub", 4, 0.0, 128, 9, dist, OracleBackend())
sharded.find_similar_pairs5_collective(e, "AllGenes", "AllCells", "Stored", "Sharded5", 6, 0.1, 8, 1000, dist, OracleBackend())
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 3])
def test_collective_fsp4_equals_single_process(tmp_path, oracle, world):
    import synth
    from expressionmatrix2_amd import ExpressionMatrix, capi, files
    d = str(tmp_path / "data")
    cells, genes = 203, 300
    toc, g, c = synth.expression_matrix(cells, genes, density=0.05, cluster_count=3, seed=11)
    files.create_directory(d, genes, toc, capi.make_counts(g, c))
    files.add_gene_set(d, "Sub", np.arange(0, genes, 3, dtype=np.uint32))
    files.add_cell_set(d, "Odd", np.arange(1, cells, 2, dtype=np.uint32))

    # a stored Lsh object for the findSimilarPairs5 leg (what computeLshSignatures leaves behind)
    full_vectors = oracle.generate_lsh_vectors(genes, 256, 231)
    stored = oracle.compute_signatures(toc, g, c, genes, full_vectors, 256)
    files.write_lsh(d, "Stored", 256, stored)

    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, d], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)

    # single-process expectation with the oracle
    e = ExpressionMatrix(d)
    for name, gs, cs, k, thr, L, seed in [("Sharded", "AllGenes", "AllCells", 7, 0.1, 256, 231),
                                           ("ShardedSub", "Sub", "Odd", 4, 0.0, 128, 9)]:
        n_genes, stoc, sdata = e._subset(gs, cs)
        vectors = oracle.generate_lsh_vectors(n_genes, L, seed)
        sig = oracle.compute_signatures(stoc, sdata["gene"], sdata["count"], n_genes, vectors, L)
        cell, sim, used = oracle.find_similar_pairs4(sig, L, k, thr)
        k2, pairs, u2 = files.read_similar_pairs(d, name)
        assert k2 == k and np.array_equal(u2, used)
        assert np.array_equal(pairs["cell"], cell)
        assert np.array_equal(pairs["similarity"].view(np.uint32), sim.view(np.uint32))
        assert used.sum() > 0
    cell, sim, used = oracle.find_similar_pairs5(stored, 256, 6, 0.1, 8, 1000)
    k2, pairs, u2 = files.read_similar_pairs(d, "Sharded5")
    assert k2 == 6 and np.array_equal(u2, used) and np.array_equal(pairs["cell"], cell)
    assert np.array_equal(pairs["similarity"].view(np.uint32), sim.view(np.uint32))
    assert used.sum() > 0
