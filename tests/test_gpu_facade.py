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


def _device_subset(toc, data, cells, local_ids, gene_count):
    """em2_dev_subset_count / _fill on device copies of the arrays -> (toc, entries)."""
    import torch
    lib = capi.load()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).copy()).cuda()
    d_toc, d_data, d_cells, d_local = d(toc), d(data), d(cells), d(local_ids)
    ws_bytes = lib.em2_dev_subset_workspace(len(cells))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    out_toc = torch.empty(len(cells) + 1, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    capi.check(lib.em2_dev_subset_count(d_toc.data_ptr(), d_data.data_ptr(), d_cells.data_ptr(), len(cells), d_local.data_ptr(),
                                        gene_count, out_toc.data_ptr(), ws.data_ptr(), ws_bytes, stream))
    torch.cuda.synchronize()
    new_toc = out_toc.cpu().numpy().view(np.uint64)
    out_data = torch.empty(max(1, int(new_toc[-1])) * 8, dtype=torch.uint8, device="cuda")
    capi.check(lib.em2_dev_subset_fill(d_toc.data_ptr(), d_data.data_ptr(), d_cells.data_ptr(), len(cells), d_local.data_ptr(),
                                       gene_count, out_toc.data_ptr(), out_data.data_ptr(), stream))
    torch.cuda.synchronize()
    return new_toc, out_data.cpu().numpy()[:int(new_toc[-1]) * 8].view(capi.COUNT_DTYPE)


@pytest.mark.parametrize("gene_set,cell_set", [("AllGenes", "AllCells"), ("HighInformationGenes", "AllCells"),
                                               ("AllGenes", "Subset"), ("HighInformationGenes", "Subset")])
def test_device_subset_equals_oracle_subset(oracle, data_dir, gene_set, cell_set):
    """ExpressionMatrixSubset on the device (em2_dev_subset_count / _fill) and on the host (em2_matrix_subset) against
    the oracle's restatement of src/ExpressionMatrixSubset.cpp:9-42: same offsets, same (local gene id, count)
    entries."""
    e = ExpressionMatrix(data_dir)
    _, g_toc, g_data = e._subset("AllGenes", "AllCells")                    # the global CSR as stored
    cells = e._cell_set(cell_set)
    genes = np.arange(900, dtype=np.uint32) if gene_set == "AllGenes" else np.unique((np.arange(300) * 7) % 900).astype(np.uint32)
    local_ids = np.full(900, 0xffffffff, dtype=np.uint32)
    local_ids[genes] = np.arange(len(genes), dtype=np.uint32)
    otoc, ogenes, ocounts, sums = oracle.subset(g_toc, g_data["gene"], g_data["count"], genes, local_ids, cells)
    toc, got = _device_subset(g_toc, g_data, cells, local_ids, 900)
    assert np.array_equal(toc, otoc) and np.array_equal(got["gene"], ogenes)
    assert np.array_equal(got["count"].view(np.uint32), ocounts.view(np.uint32))
    n_genes, htoc, hdata = e._subset(gene_set, cell_set)                    # host
    assert n_genes == len(genes) and np.array_equal(htoc, otoc) and np.array_equal(hdata["gene"], ogenes)
    assert np.array_equal(hdata["count"].view(np.uint32), ocounts.view(np.uint32))


def test_device_subset_edge_cases_equal_oracle(oracle):
    """Cells without counts, cells whose genes are all outside the set, a gene set whose local-id vector is shorter
    than the global gene count (GeneSet.hpp:70-77), cells of more than one wave of counts, an empty result."""
    rng = np.random.default_rng(3)
    gene_count = 500
    lengths = np.array([0, 1, 63, 64, 65, 200, 0, 0, 130, 5], dtype=np.int64)
    toc = np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint64)
    genes = np.concatenate([np.sort(rng.choice(gene_count, n, replace=False)) for n in lengths]).astype(np.uint32)
    counts = rng.integers(1, 50, len(genes)).astype(np.float32)
    data = capi.make_counts(genes, counts)
    for gene_ids in (np.arange(gene_count, dtype=np.uint32), np.arange(0, 100, 3, dtype=np.uint32),
                     np.array([499], dtype=np.uint32), np.arange(250, 260, dtype=np.uint32)):
        for cells in (np.arange(len(lengths), dtype=np.uint32), np.array([0, 5, 6, 9], dtype=np.uint32), np.array([6], dtype=np.uint32)):
            local_ids = np.full(gene_count, 0xffffffff, dtype=np.uint32)
            local_ids[gene_ids] = np.arange(len(gene_ids), dtype=np.uint32)
            otoc, ogenes, ocounts, _ = oracle.subset(toc, genes, counts, gene_ids, local_ids[:int(gene_ids[-1]) + 1], cells)
            dtoc, got = _device_subset(toc, data, cells, local_ids, gene_count)
            assert np.array_equal(dtoc, otoc) and np.array_equal(got["gene"], ogenes)
            assert np.array_equal(got["count"].view(np.uint32), ocounts.view(np.uint32))
