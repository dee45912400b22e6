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


@pytest.mark.parametrize("gene_set,cell_set", [("AllGenes", "AllCells"), ("HighInformationGenes", "AllCells"),
                                               ("AllGenes", "Subset"), ("HighInformationGenes", "Subset")])
def test_device_subset_equals_host_subset(data_dir, gene_set, cell_set):
    """ExpressionMatrixSubset on the device (em2_dev_subset_count / _fill) against the host restatement of
    src/ExpressionMatrixSubset.cpp:9-42 (em2_matrix_subset): same offsets, same (local gene id, count) entries."""
    import torch
    e = ExpressionMatrix(data_dir)
    n_genes, toc, data = e._subset(gene_set, cell_set)                       # host
    _, g_toc, g_data = e._subset("AllGenes", "AllCells")                    # the global CSR
    cells = e._cell_set(cell_set)
    local_ids = np.full(900, 0xffffffff, dtype=np.uint32)
    if gene_set == "AllGenes":
        local_ids[:] = np.arange(900, dtype=np.uint32)
    else:
        genes = np.unique((np.arange(300) * 7) % 900).astype(np.uint32)
        local_ids[genes] = np.arange(len(genes), dtype=np.uint32)
    lib = capi.load()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8)).cuda()
    d_toc, d_data, d_cells, d_local = d(g_toc), d(g_data), d(cells), d(local_ids)
    ws_bytes = lib.em2_dev_subset_workspace(len(cells))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    out_toc = torch.empty(len(cells) + 1, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    capi.check(lib.em2_dev_subset_count(d_toc.data_ptr(), d_data.data_ptr(), d_cells.data_ptr(), len(cells), d_local.data_ptr(),
                                        900, out_toc.data_ptr(), ws.data_ptr(), ws_bytes, stream))
    torch.cuda.synchronize()
    assert np.array_equal(out_toc.cpu().numpy().view(np.uint64), toc)
    out_data = torch.empty(max(1, int(toc[-1])) * 8, dtype=torch.uint8, device="cuda")
    capi.check(lib.em2_dev_subset_fill(d_toc.data_ptr(), d_data.data_ptr(), d_cells.data_ptr(), len(cells), d_local.data_ptr(),
                                       900, out_toc.data_ptr(), out_data.data_ptr(), stream))
    torch.cuda.synchronize()
    got = out_data.cpu().numpy()[:int(toc[-1]) * 8].view(capi.COUNT_DTYPE)
    assert np.array_equal(got["gene"], data["gene"])
    assert np.array_equal(got["count"].view(np.uint32), data["count"].view(np.uint32))
    assert n_genes == (900 if gene_set == "AllGenes" else len(np.unique((np.arange(300) * 7) % 900)))
