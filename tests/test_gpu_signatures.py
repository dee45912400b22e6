"""GPU parity of the signature projection (HIP, FP64, sequential order, no FMA) against the CPU oracle:
signature words must be bit-identical."""
import numpy as np
import pytest

import synth
from expressionmatrix2_amd import capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cells,genes,L,density", [
    (3, 3, 128, 1.0),
    (200, 300, 256, 0.05),
    (500, 2000, 1024, 0.01),
    (257, 1000, 100, 0.02),
    (1000, 5000, 2048, 0.01),
    (64, 50, 64, 0.5),
    (40, 4000, 1000, 0.05),
])
def test_signatures_match_oracle(oracle, cells, genes, L, density):
    toc, g, c = synth.expression_matrix(cells, genes, density=density, cluster_count=4, seed=cells + L)
    vectors = oracle.generate_lsh_vectors(genes, L, 231)
    expect = oracle.compute_signatures(toc, g, c, genes, vectors, L)
    got = capi.compute_signatures(toc, capi.make_counts(g, c), genes, vectors, L)
    assert np.array_equal(got, expect)
    # not degenerate
    ones = np.unpackbits(got.view(np.uint8)).sum()
    assert 0.2 * cells * L < ones < 0.8 * cells * L


def test_toytest1_matrix(oracle):
    """tests/ToyTest1/ExpressionMatrix.csv of the reference: 3 genes x 3 cells (values restated as data)."""
    # Cell0: Gene0=10, Gene1=20 ; Cell1: Gene0=20, Gene1=40(ish) ... dense 3x3 toy; exact values are not
    # important for parity, the shape (3 cells, 3 genes, lshCount 128) is BASELINE config 1.
    dense = np.array([[10., 20., 0.], [0., 30., 10.], [5., 25., 15.]], dtype=np.float32)
    toc = [0]
    genes = []
    counts = []
    for row in dense:
        nz = np.nonzero(row)[0]
        genes += nz.tolist()
        counts += row[nz].tolist()
        toc.append(len(genes))
    vectors = oracle.generate_lsh_vectors(3, 128, 231)
    expect = oracle.compute_signatures(toc, genes, counts, 3, vectors, 128)
    got = capi.compute_signatures(np.array(toc), capi.make_counts(genes, counts), 3, vectors, 128)
    assert np.array_equal(got, expect)
    cell, sim, used = oracle.find_similar_pairs4(got, 128, 100, 0.2)
    pairs, gused = capi.find_similar_pairs4(got, 128, 100, 0.2)
    assert np.array_equal(gused, used) and np.array_equal(pairs["cell"], cell)


def test_empty_cells_and_unsorted_free_rows(oracle):
    """Cells with no expression counts (mean 0, all scalar products 0 -> no bit set)."""
    toc = np.array([0, 0, 2, 2, 5], dtype=np.uint64)
    genes = np.array([1, 3, 0, 2, 4], dtype=np.uint32)
    counts = np.array([1.5, 2.0, 7.0, 1.0, 3.0], dtype=np.float32)
    vectors = oracle.generate_lsh_vectors(5, 192, 7)
    expect = oracle.compute_signatures(toc, genes, counts, 5, vectors, 192)
    got = capi.compute_signatures(toc, capi.make_counts(genes, counts), 5, vectors, 192)
    assert np.array_equal(got, expect)
    assert not got[0].any() and not got[2].any()


def test_screening_pass_defers_to_exact_arithmetic_when_it_cannot_decide(oracle):
    """Hyperplanes 1 + 1e-9*noise: their float copy is exactly 1.0, so the screening pass cannot see what decides
    the signs; every word must go through the exact recomputation to match the oracle."""
    cells, genes, L = 300, 400, 256
    toc, g, c = synth.expression_matrix(cells, genes, density=0.05, cluster_count=3, seed=5)
    idx = np.arange(genes * L, dtype=np.uint64).reshape(genes, L)
    noise = (synth.uniform01(77, idx) - 0.5)
    vectors = 1.0 + 1e-9 * noise
    assert (vectors.astype(np.float32) == 1.0).all()
    expect = oracle.compute_signatures(toc, g, c, genes, vectors, L)
    got = capi.compute_signatures(toc, capi.make_counts(g, c), genes, vectors, L)
    assert np.array_equal(got, expect)
    ones = np.unpackbits(got.view(np.uint8)).sum()
    assert 0.3 * cells * L < ones < 0.7 * cells * L            # the signs are genuinely mixed


@pytest.mark.parametrize("mode", ["exact", "screened"])
def test_exact_only_and_screened_paths_agree_with_oracle(oracle, mode, monkeypatch):
    cells, genes, L = 800, 3000, 1024
    toc, g, c = synth.expression_matrix(cells, genes, density=0.02, cluster_count=6, seed=9)
    vectors = oracle.generate_lsh_vectors(genes, L, 231)
    expect = oracle.compute_signatures(toc, g, c, genes, vectors, L)
    if mode == "exact":
        monkeypatch.setenv("EM2_PROJECTION", "exact")
    got = capi.compute_signatures(toc, capi.make_counts(g, c), genes, vectors, L)
    assert np.array_equal(got, expect)


def test_screening_with_huge_and_tiny_magnitudes(oracle):
    """Counts and hyperplane entries spanning many orders of magnitude (float subnormals included)."""
    cells, genes, L = 200, 300, 128
    toc, g, c = synth.expression_matrix(cells, genes, density=0.1, cluster_count=2, seed=3)
    c = (c * np.float32(1e20)).astype(np.float32)
    c[::7] = np.float32(1e-30)
    idx = np.arange(genes * L, dtype=np.uint64).reshape(genes, L)
    vectors = (synth.uniform01(5, idx) - 0.5) * np.power(10.0, -40.0 * synth.uniform01(6, idx))
    expect = oracle.compute_signatures(toc, g, c, genes, vectors, L)
    got = capi.compute_signatures(toc, capi.make_counts(g, c), genes, vectors, L)
    assert np.array_equal(got, expect)
