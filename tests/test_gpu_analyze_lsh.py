"""ExpressionMatrix.analyzeLsh on the GPU (em2_analyze_lsh: scalar products and mismatch counts of all pairs on the
device, the order-defined accumulation on the host) against the oracle's restatement of
src/ExpressionMatrixLsh.cpp:1244-1367.  Everything is bit-exact: the per-pair doubles, the 200 bins' double sums, both
csv files byte for byte (no floating-point tolerance is needed or used)."""
import os

import numpy as np
import pytest

import synth
from expressionmatrix2_amd import ExpressionMatrix, capi, files

pytestmark = pytest.mark.gpu


def run_both(oracle, tmp_path, toc, g, c, genes, L, ids, seed, downsample):
    vectors = oracle.generate_lsh_vectors(genes, L, seed)
    sig = oracle.compute_signatures(toc, g, c, genes, vectors, L)
    o = oracle.analyze_lsh(toc, g, c, genes, sig, L, ids, seed, downsample, str(tmp_path / "o-pairs.csv"), str(tmp_path / "o-stats.csv"))
    d = capi.analyze_lsh(toc, capi.make_counts(g, c), genes, sig, L, ids, seed, downsample, str(tmp_path / "d-pairs.csv"),
                         str(tmp_path / "d-stats.csv"), per_pair=True)
    return o, d


@pytest.mark.parametrize("cells,genes,L,density,seed,downsample", [
    (300, 900, 1024, 0.05, 231, 0.01),
    (257, 40000, 128, 0.002, 7, 0.5),            # more genes than the LDS vector holds: the global-memory form
    (700, 2000, 192, 0.03, 99, 0.001),
    (2, 50, 64, 0.5, 1, 1.0),                    # one pair
    (130, 36864, 256, 0.001, 3, 0.0),            # the largest gene set of the LDS form; nothing in the csv
])
def test_analyze_lsh_equals_oracle(oracle, tmp_path, cells, genes, L, density, seed, downsample):
    toc, g, c = synth.expression_matrix(cells, genes, density=density, cluster_count=4, seed=seed)
    c = c.astype(np.float32)
    ids = (np.arange(cells, dtype=np.uint32) * 3 + 5).astype(np.uint32)
    o, d = run_both(oracle, tmp_path, toc, g, c, genes, L, ids, seed, downsample)
    assert o is not None
    assert np.array_equal(d["exact"].view(np.uint64), o["exact"].view(np.uint64))
    assert np.array_equal(d["lsh"].view(np.uint64), o["lsh"].view(np.uint64))
    assert np.array_equal(d["sum0"], o["sum0"])
    assert np.array_equal(d["sum1"].view(np.uint64), o["sum1"].view(np.uint64))
    assert np.array_equal(d["sum2"].view(np.uint64), o["sum2"].view(np.uint64))
    for name in ("pairs", "stats"):
        assert open(tmp_path / ("d-%s.csv" % name), "rb").read() == open(tmp_path / ("o-%s.csv" % name), "rb").read()


def test_analyze_lsh_many_chunks(oracle, tmp_path, monkeypatch):
    """More pairs than one chunk of rows holds (2^24): the host walks the chunks in order, the bins and the random draws
    carry over."""
    cells, genes, L = 6000, 300, 64
    toc, g, c = synth.expression_matrix(cells, genes, density=0.03, cluster_count=6, seed=11)
    c = c.astype(np.float32)
    keep = np.diff(toc.astype(np.int64)) > 1          # cells with fewer than two counts have no variance to speak of
    assert keep.all()
    ids = np.arange(cells, dtype=np.uint32)
    o, d = run_both(oracle, tmp_path, toc, g, c, genes, L, ids, 5, 0.0001)
    assert o is not None and len(o["exact"]) > (1 << 24)
    assert np.array_equal(d["exact"].view(np.uint64), o["exact"].view(np.uint64))
    assert np.array_equal(d["sum1"].view(np.uint64), o["sum1"].view(np.uint64))
    assert np.array_equal(d["sum2"].view(np.uint64), o["sum2"].view(np.uint64))
    assert open(tmp_path / "d-pairs.csv", "rb").read() == open(tmp_path / "o-pairs.csv", "rb").read()
    assert open(tmp_path / "d-stats.csv", "rb").read() == open(tmp_path / "o-stats.csv", "rb").read()


def test_analyze_lsh_assertion_and_argument_errors(oracle, tmp_path):
    toc = np.array([0, 3, 6, 8], dtype=np.uint64)
    g = np.array([1, 4, 7, 1, 4, 7, 2, 3], dtype=np.uint32)
    c = np.array([1, 2, 3, 1, 2, 3, 5, 1], dtype=np.float32)          # cells 0 and 1 are equal: similarity 1 -> bin 200
    sig = np.zeros((3, 1), dtype=np.uint64)
    ids = np.arange(3, dtype=np.uint32)
    with pytest.raises(RuntimeError, match="bin < binCount"):
        capi.analyze_lsh(toc, capi.make_counts(g, c), 10, sig, 64, ids, 1, 1.0, str(tmp_path / "p.csv"), str(tmp_path / "s.csv"))
    assert oracle.analyze_lsh(toc, g, c, 10, sig, 64, ids, 1, 1.0, str(tmp_path / "op.csv"), str(tmp_path / "os.csv")) is None
    with pytest.raises(RuntimeError, match="not below geneCount"):
        capi.analyze_lsh(toc, capi.make_counts(g, c), 5, sig, 64, ids, 1, 1.0, str(tmp_path / "p.csv"))
    with pytest.raises(RuntimeError, match="cannot open"):
        capi.analyze_lsh(toc[:3], capi.make_counts(g[:6], c[:6]), 10, sig[:2], 64, ids[:2], 1, 1.0, str(tmp_path / "no" / "p.csv"))


def test_analyze_lsh_through_expression_matrix_api(oracle, tmp_path, monkeypatch):
    """The facade: the files land in the working directory under the reference's names (src/ExpressionMatrixLsh.cpp:1303,
    :1345), equal to the oracle's from the same subset and the same hyperplanes."""
    d = str(tmp_path / "data")
    cells, genes = 400, 700
    toc, g, c = synth.expression_matrix(cells, genes, density=0.3, cluster_count=5, seed=21)      # every cell keeps counts in "Some"
    files.create_directory(d, genes, toc, capi.make_counts(g, c))
    files.add_gene_set(d, "Some", np.unique((np.arange(300) * 7) % genes).astype(np.uint32))
    files.add_cell_set(d, "Odd", np.arange(1, cells, 2, dtype=np.uint32))
    work = tmp_path / "work"
    work.mkdir()
    monkeypatch.chdir(work)
    e = ExpressionMatrix(d)
    e.analyzeLsh("Some", "Odd", 512, 77, 0.05)
    n_genes, s_toc, s_data = e._subset("Some", "Odd")
    vectors = oracle.generate_lsh_vectors(n_genes, 512, 77)
    sig = oracle.compute_signatures(s_toc, s_data["gene"], s_data["count"], n_genes, vectors, 512)
    o = oracle.analyze_lsh(s_toc, s_data["gene"], s_data["count"], n_genes, sig, 512, e._cell_set("Odd"), 77, 0.05,
                           str(tmp_path / "o-pairs.csv"), str(tmp_path / "o-stats.csv"))
    assert o is not None
    assert open(work / "Lsh-analysis.csv", "rb").read() == open(tmp_path / "o-pairs.csv", "rb").read()
    assert open(work / "LSH-analysis-statistics.csv", "rb").read() == open(tmp_path / "o-stats.csv", "rb").read()
    with pytest.raises(RuntimeError, match="Gene set Nope does not exist."):
        e.analyzeLsh("Nope", "Odd", 512, 77, 0.05)
    with pytest.raises(RuntimeError, match="Cell set Nope does not exist."):
        e.analyzeLsh("Some", "Nope", 512, 77, 0.05)
