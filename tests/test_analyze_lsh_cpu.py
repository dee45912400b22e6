"""The oracle's restatement of ExpressionMatrix::analyzeLsh (src/ExpressionMatrixLsh.cpp:1244-1367) against independent
arithmetic.  Parity of this restatement is UNPINNED against the reference binary (ExpressionMatrixSubset.cpp and Lsh.cpp
need Boost and the memory-mapped containers, which do not build here, and the reference ships no test or golden output of
analyzeLsh); what pins it here: the exact similarity equals numpy's Pearson correlation of the dense rows to 1e-12, the
LSH similarity equals cos(pi * mismatches / lshCount), the bins partition the pairs, the csv carries exactly the pairs a
std::mt19937(seed) selects, and the statistics file is the documented function of the bins."""
import math
import os

import numpy as np

import synth


def problem(cells=60, genes=300, L=256, seed=5):
    toc, g, c = synth.expression_matrix(cells, genes, density=0.2, cluster_count=3, seed=seed)
    return toc, g, c.astype(np.float32), genes, L


def signatures(oracle, toc, g, c, genes, L, seed=231):
    vectors = oracle.generate_lsh_vectors(genes, L, seed)
    return oracle.compute_signatures(toc, g, c, genes, vectors, L)


def test_oracle_analyze_lsh_against_numpy(oracle, tmp_path):
    toc, g, c, genes, L = problem()
    n = len(toc) - 1
    sig = signatures(oracle, toc, g, c, genes, L)
    ids = np.arange(100, 100 + n, dtype=np.uint32)
    pairs_csv, stats_csv = str(tmp_path / "Lsh-analysis.csv"), str(tmp_path / "LSH-analysis-statistics.csv")
    out = oracle.analyze_lsh(toc, g, c, genes, sig, L, ids, 231, 0.25, pairs_csv, stats_csv)
    assert out is not None
    dense = np.zeros((n, genes))
    for cell in range(n):
        dense[cell, g[int(toc[cell]):int(toc[cell + 1])]] = c[int(toc[cell]):int(toc[cell + 1])]
    corr = np.corrcoef(dense)
    bits = np.unpackbits(sig.view(np.uint8).reshape(n, -1, 8)[:, :, ::-1].reshape(n, -1), axis=1)[:, :L]
    iu = np.triu_indices(n, 1)                      # row-major upper triangle: the reference's pair order
    assert np.allclose(out["exact"], corr[iu], rtol=0, atol=1e-12)
    mismatches = (bits[iu[0]] != bits[iu[1]]).sum(axis=1)
    assert np.array_equal(out["lsh"], np.cos(mismatches * math.pi / L))
    # bins
    bins = np.floor((out["exact"] + 1.) / (2. / 200)).astype(np.int64)
    assert np.array_equal(np.bincount(bins, minlength=200), out["sum0"].astype(np.int64))
    delta = out["lsh"] - out["exact"]
    for b in np.unique(bins):
        assert math.isclose(out["sum1"][b], delta[bins == b].sum(), rel_tol=1e-9, abs_tol=1e-12)
        assert math.isclose(out["sum2"][b], (delta[bins == b] ** 2).sum(), rel_tol=1e-9, abs_tol=1e-12)
    # the csv: header, then the pairs an mt19937(seed) draw below 0.25 selects, six significant digits, trailing comma
    lines = open(pairs_csv).read().split("\n")
    assert lines[0] == "LocalCellId0,LocalCellId1,GlobalCellId0,GlobalCellId1,ExactSimilarity,LshSimilarity" and lines[-1] == ""
    mt19937 = np.random.MT19937()
    mt19937._legacy_seeding(231)                    # init_genrand(seed) == std::mt19937(seed)
    draws = mt19937.random_raw(len(out["exact"])) / 4294967296.0
    picked = np.nonzero(draws < 0.25)[0]
    assert len(lines) - 2 == len(picked) > 0
    for line, p in zip(lines[1:-1], picked):
        f = line.split(",")
        assert f[6] == "" and int(f[0]) == iu[0][p] and int(f[1]) == iu[1][p]
        assert int(f[2]) == 100 + iu[0][p] and int(f[3]) == 100 + iu[1][p]
        assert f[4] == "%g" % out["exact"][p] and f[5] == "%g" % out["lsh"][p]
    # the statistics: bins with at least two pairs
    stats = open(stats_csv).read().split("\n")
    assert stats[0] == "Similarity,Bias,Rms,RmsTheory" and stats[-1] == ""
    kept = [b for b in range(200) if out["sum0"][b] >= 2]
    assert len(stats) - 2 == len(kept)
    for line, b in zip(stats[1:-1], kept):
        f = [float(x) for x in line.split(",")]
        s = (b + 0.5) * 0.01 - 1.
        theta = math.acos(s)
        p = 1. - theta / math.pi
        assert math.isclose(f[0], s, rel_tol=1e-5, abs_tol=1e-9)
        assert math.isclose(f[1], out["sum1"][b] / out["sum0"][b], rel_tol=1e-5, abs_tol=1e-9)
        assert math.isclose(f[2], math.sqrt(out["sum2"][b] / out["sum0"][b]), rel_tol=1e-5)
        assert math.isclose(f[3], math.pi * math.sqrt(1 - s * s) * math.sqrt(p * (1 - p) / L), rel_tol=1e-5)


def test_oracle_analyze_lsh_assertion_cases(oracle, tmp_path):
    """CZI_ASSERT(bin < binCount) (:1322): two identical cells have exact similarity 1 -> bin 200; a cell without
    variance gives NaN."""
    toc, g, c, genes, L = problem(cells=10, genes=50)
    n = len(toc) - 1
    # make cell 1 a copy of cell 0
    lo, hi = int(toc[0]), int(toc[1])
    g2 = np.concatenate([g[lo:hi], g[lo:hi], g[int(toc[2]):]])
    c2 = np.concatenate([c[lo:hi], c[lo:hi], c[int(toc[2]):]])
    toc2 = np.concatenate([[0, hi - lo, 2 * (hi - lo)], toc[3:] - toc[2] + 2 * (hi - lo)]).astype(np.uint64)
    sig = signatures(oracle, toc2, g2, c2, genes, L)
    ids = np.arange(n, dtype=np.uint32)
    assert oracle.analyze_lsh(toc2, g2, c2, genes, sig, L, ids, 1, 1.0, str(tmp_path / "a.csv"), str(tmp_path / "b.csv")) is None
