"""End to end through the ExpressionMatrix API on a data directory (the drop-in boundary): the files the GPU
path writes must equal what the oracle computes from the same directory."""
import numpy as np
import pytest

import synth
from expressionmatrix2_amd import ExpressionMatrix, capi, files

pytestmark = pytest.mark.gpu


@pytest.fixture()
def data_dir(tmp_path):
    d = str(tmp_path / "data")
    cells, genes = 700, 900
    toc, g, c = synth.expression_matrix(cells, genes, density=0.03, cluster_count=5, seed=21)
    files.create_directory(d, genes, toc, capi.make_counts(g, c))
    files.add_gene_set(d, "HighInformationGenes", np.unique((np.arange(300) * 7) % genes).astype(np.uint32))
    files.add_cell_set(d, "Subset", np.arange(3, cells, 2, dtype=np.uint32))
    return d


def expected(oracle, e, gene_set, cell_set, L, seed, k, thr):
    n_genes, toc, data = e._subset(gene_set, cell_set)
    vectors = oracle.generate_lsh_vectors(n_genes, L, seed)
    sig = oracle.compute_signatures(toc, data["gene"], data["count"], n_genes, vectors, L)
    return sig, oracle.find_similar_pairs4(sig, L, k, thr)


@pytest.mark.parametrize("gene_set,cell_set,k,thr,L,seed", [
    ("AllGenes", "AllCells", 100, 0.2, 1024, 231),           # the reference's defaults
    ("HighInformationGenes", "AllCells", 20, 0.2, 1024, 231),  # tests/CaseStudy1/compute2.py:12 shape
    ("AllGenes", "Subset", 10, 0.0, 128, 7),
    ("HighInformationGenes", "Subset", 5, -0.2, 192, 99),
])
def test_find_similar_pairs4_files(oracle, data_dir, gene_set, cell_set, k, thr, L, seed):
    e = ExpressionMatrix(data_dir)
    e.findSimilarPairs4(geneSetName=gene_set, cellSetName=cell_set, similarPairsName="Lsh", k=k,
                        similarityThreshold=thr, lshCount=L, seed=seed)
    sig, (cell, sim, used) = expected(oracle, e, gene_set, cell_set, L, seed, k, thr)
    k2, pairs, u2 = files.read_similar_pairs(data_dir, "Lsh")
    assert k2 == k and np.array_equal(u2, used)
    assert np.array_equal(pairs["cell"], cell)
    assert np.array_equal(pairs["similarity"].view(np.uint32), sim.view(np.uint32))
    assert used.sum() > 0
    # calling again under the same name silently replaces the object (O_TRUNC, MemoryMappedVector.hpp:351-354)
    e.findSimilarPairs4(geneSetName=gene_set, cellSetName=cell_set, similarPairsName="Lsh", k=2,
                        similarityThreshold=thr, lshCount=L, seed=seed)
    assert files.read_similar_pairs(data_dir, "Lsh")[0] == 2
    e.removeSimilarPairs("Lsh")


def test_compute_lsh_signatures_persists_reference_format(oracle, data_dir):
    e = ExpressionMatrix(data_dir)
    e.computeLshSignatures(lshName="L", lshCount=512, seed=231)
    L, sig = files.read_lsh(data_dir, "L")
    exp_sig, _ = expected(oracle, e, "AllGenes", "AllCells", 512, 231, 1, 0.2)
    assert L == 512 and np.array_equal(sig, exp_sig)
