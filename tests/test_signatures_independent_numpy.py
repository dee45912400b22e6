"""A second, independent restatement of Lsh::computeCellLshSignatures (src/Lsh.cpp:118-224) and of the per-cell sums it reads
(ExpressionMatrixSubset::computeSums, src/ExpressionMatrixSubset.cpp:46-58) in numpy float64 -- sharing no code with
oracle/em2_oracle.cpp.  The signature of a cell is the sign pattern of sequentially rounded double sums, so the ORDER of the
additions and the separate rounding of every product (no fused multiply-add: the reference is an SSE4.2 build) are what is
restated here: numpy's elementwise `count * v` and `+=` round each operation on its own, one gene (Lsh.cpp:137-144) or one
expression count (:188-198) after the other, vectorised only ACROSS the bits, which are independent.  The hyperplanes are an
input (the normal variates of the reference's Boost are not reproducible here).  The oracle must give the same
64-bit words, including on counts that are no integers, on magnitudes that differ by many orders, and on cells without counts."""
import numpy as np
import pytest

import synth


def compute_signatures(toc, genes, counts, gene_count, vectors, lsh_count):
    cells = len(toc) - 1
    words = (lsh_count - 1) // 64 + 1                                       # :127
    sums = np.zeros(lsh_count, dtype=np.float64)
    for g in range(gene_count):                                             # :137-144, ascending gene id
        sums += vectors[g]
    out = np.zeros((cells, words), dtype=np.uint64)
    for c in range(cells):
        lo, hi = int(toc[c]), int(toc[c + 1])
        sum1 = np.float64(0.0)
        for x in counts[lo:hi]:                                             # computeSums: double += float, stored order
            sum1 = sum1 + np.float64(x)
        mean = sum1 / np.float64(gene_count)                                # :167-168
        sp = -mean * sums                                                   # :180-182
        for g, x in zip(genes[lo:hi], counts[lo:hi]):                       # :188-198, stored (ascending local gene id) order
            sp = sp + np.float64(x) * vectors[int(g)]
        for i in np.nonzero(sp > 0.0)[0]:                                   # :201-206; bit i -> word i >> 6, position 63 - (i & 63)
            out[c, int(i) >> 6] |= np.uint64(1) << np.uint64(63 - (int(i) & 63))
    return out


@pytest.mark.parametrize("cells,genes,L,density,scale", [
    (60, 400, 128, 0.05, None),          # integer counts, as the benchmark's
    (40, 300, 100, 0.08, 0.37),          # fractional counts: every addition rounds
    (30, 500, 1024, 0.03, None),         # the benchmark's width
    (25, 200, 70, 0.2, 1e-3),            # lshCount no multiple of 64; many counts per cell
])
def test_oracle_equals_independent_restatement(oracle, cells, genes, L, density, scale):
    toc, g, c = synth.expression_matrix(cells, genes, density=density, cluster_count=4, seed=cells + genes)
    if scale is not None:
        c = (c * np.float32(scale) + np.float32(0.123)).astype(np.float32)
    vectors = oracle.generate_lsh_vectors(genes, L, 231)
    expect = compute_signatures(toc, g, c, genes, vectors, L)
    got = oracle.compute_signatures(toc, g, c, genes, vectors, L)
    assert np.array_equal(expect, got)


def test_magnitudes_and_empty_cells(oracle):
    rng = np.random.default_rng(5)
    genes, L = 120, 192
    vectors = rng.standard_normal((genes, L)) * np.exp(rng.uniform(-40, 40, size=(genes, 1)))       # rows of very different size
    toc = np.array([0, 0, 5, 5, 40, 41], dtype=np.uint64)                                             # cells 0 and 2 have no counts
    g = np.concatenate([np.sort(rng.choice(genes, 5, replace=False)), np.sort(rng.choice(genes, 35, replace=False)), [7]]).astype(np.uint32)
    c = np.exp(rng.uniform(-20, 20, size=41)).astype(np.float32)
    expect = compute_signatures(toc, g, c, genes, vectors, L)
    got = oracle.compute_signatures(toc, g, c, genes, vectors, L)
    assert np.array_equal(expect, got)
