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


def test_toytest1_matrix(oracle, tmp_path):
    """BASELINE configs[0]: the reference's tests/ToyTest1/ExpressionMatrix.csv (committed as a data fixture,
    tests/golden/ToyTest1_ExpressionMatrix.csv) through ExpressionMatrix.findSimilarPairs4 at lshCount=128, against
    the oracle.  Cell0 = {Gene0:10, Gene1:10}, Cell1 = {Gene1:10, Gene3:10}, Cell2 = {Gene3:20}."""
    import toytest1
    from expressionmatrix2_amd import ExpressionMatrix, files
    gene_count, toc, genes, counts = toytest1.load()
    assert gene_count == 3 and toc.tolist() == [0, 2, 4, 5]
    assert genes.tolist() == [0, 1, 1, 2, 2] and counts.tolist() == [10., 10., 10., 10., 20.]
    d = str(tmp_path / "data")
    files.create_directory(d, gene_count, toc, capi.make_counts(genes, counts))
    e = ExpressionMatrix(d)
    e.findSimilarPairs4(similarPairsName="Lsh", lshCount=128)              # k=100, threshold 0.2, seed 231: the defaults
    vectors = oracle.generate_lsh_vectors(gene_count, 128, 231)
    expect = oracle.compute_signatures(toc, genes, counts, gene_count, vectors, 128)
    got = capi.compute_signatures(toc, capi.make_counts(genes, counts), gene_count, vectors, 128)
    assert np.array_equal(got, expect)
    cell, sim, used = oracle.find_similar_pairs4(expect, 128, 100, 0.2)
    k, pairs, gused = files.read_similar_pairs(d, "Lsh")
    assert k == 100 and np.array_equal(gused, used) and np.array_equal(pairs["cell"], cell)
    assert np.array_equal(pairs["similarity"].view(np.uint32), sim.view(np.uint32))
    # what the data says whatever the hyperplanes: Cell0 is exactly anti-correlated with Cell2 (all 128 bits differ)
    # and finds nobody; Cell1 and Cell2 (exact similarity 0.5) find each other
    assert oracle.mismatch_matrix(expect, 128)[0, 2] == 128
    assert used.tolist() == [0, 1, 1] and cell[1, 0] == 2 and cell[2, 0] == 1


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


def exact_signatures(toc, data, genes, vectors, L):
    """The exact arithmetic alone (sequential FP64, no screening tier): em2_dev_compute_signatures without the auxiliary
    block of the hyperplanes."""
    import torch
    cells = len(toc) - 1
    d_toc = torch.from_numpy(np.ascontiguousarray(toc).view(np.int64)).cuda()
    d_data = torch.from_numpy(np.ascontiguousarray(data).view(np.int64)).cuda()
    d_vectors = torch.from_numpy(np.ascontiguousarray(vectors)).cuda()
    d_sig = torch.zeros((cells, capi.word_count(L)), dtype=torch.int64, device="cuda")
    ws_bytes = capi.dev_compute_signatures_workspace(cells, L)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    capi.dev_compute_signatures(d_toc.data_ptr(), d_data.data_ptr(), cells, genes, d_vectors.data_ptr(), 0, L, d_sig.data_ptr(),
                                ws.data_ptr(), ws_bytes, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return d_sig.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("L", [1024, 1000, 96, 100, 63])
def test_exact_only_and_screened_paths_agree_with_oracle(oracle, L):
    """The first tier by shape: a whole number of 64-bit words (1024) takes the 16-bit fixed-point tier -> float tier on the
    undecided single bits and words -> exact arithmetic; whole 32-bit slices (96) the XCD-sliced float form first; a multiple
    of 4 bits (1000, 100) one block per 1024 bits on the float copy; anything else (63) the exact arithmetic alone -- which
    every width is also held to through the device entry without the hyperplanes' auxiliary block."""
    cells, genes = 800, 3000
    toc, g, c = synth.expression_matrix(cells, genes, density=0.02, cluster_count=6, seed=9)
    vectors = oracle.generate_lsh_vectors(genes, L, 231)
    expect = oracle.compute_signatures(toc, g, c, genes, vectors, L)
    data = capi.make_counts(g, c)
    assert np.array_equal(capi.compute_signatures(toc, data, genes, vectors, L), expect)
    assert np.array_equal(exact_signatures(toc, data, genes, vectors, L), expect)


@pytest.mark.parametrize("cells,genes,L,density,count_scale", [
    (40000, 20000, 1024, 0.01, 1.0),          # BASELINE configs[1]-like rows: ~200 counts per cell
    (20000, 5000, 512, 0.2, 1.0),             # 1000 counts per cell
    (30000, 2000, 256, 0.0015, 1.0),          # three counts per cell, many cells with one or none
    (5000, 3000, 1024, 0.05, 1e30),           # counts whose products overflow single precision in the 16-bit tier
    (5000, 3000, 1024, 0.05, 1e-36),          # ... and underflow it
])
def test_tiers_equal_exact_arithmetic_on_many_cells(cells, genes, L, density, count_scale, monkeypatch):
    """The screening tiers only ever decide a bit when their bound says the exact sequential FP64 sum has that sign: on
    tens of millions of bits the default path (16-bit tier, float tier, exact) must reproduce the exact-arithmetic kernel
    (exact_signatures, itself bit-exact against the oracle in the test above) everywhere."""
    toc, g, c = synth.expression_matrix(cells, genes, density=density, cluster_count=7, seed=cells + L)
    c = (c.astype(np.float64) * count_scale).astype(np.float32)
    idx = np.arange(genes * L, dtype=np.uint64).reshape(genes, L)
    vectors = synth.uniform01(11, idx) - 0.5
    vectors /= np.sqrt((vectors * vectors).sum(axis=0))
    data = capi.make_counts(g, c)
    exact = exact_signatures(toc, data, genes, vectors, L)
    got = capi.compute_signatures(toc, data, genes, vectors, L)
    assert np.array_equal(got, exact)


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


@pytest.mark.parametrize("spread", [0.0, 6.0, 30.0])
def test_fractional_counts_and_the_cell_mean(oracle, spread):
    """The per-cell mean must be the reference's sequentially rounded sum (src/Lsh.cpp:160-170).  cellStatsKernel adds a
    cell's counts lane-parallel when no addition can round (integer counts, or float counts within ~2^20 of each other)
    and in stored order otherwise: non-integer counts with a spread of 0, 6 and 30 decimal orders of magnitude."""
    cells, genes, L = 300, 400, 256
    toc, g, c = synth.expression_matrix(cells, genes, density=0.3, cluster_count=3, seed=11)
    idx = np.arange(len(c), dtype=np.uint64)
    c = (c.astype(np.float64) * 0.37 * np.power(10.0, spread * (synth.uniform01(7, idx) - 0.5))).astype(np.float32)
    vectors = oracle.generate_lsh_vectors(genes, L, 5)
    expect = oracle.compute_signatures(toc, g, c, genes, vectors, L)
    got = capi.compute_signatures(toc, capi.make_counts(g, c), genes, vectors, L)
    assert np.array_equal(got, expect)


@pytest.mark.parametrize("case", ["integers", "all-fractions", "one-fraction", "count-32768", "sum-above-65535", "negative", "at-the-limits"])
def test_integer_first_tier_and_when_it_steps_aside(oracle, monkeypatch, case):
    """The first tier has an exact integer form (v_mad_i32_i16, 32-bit sums) for matrices whose counts are all integers of at
    most 15 bits with sum|count| <= 65535 per cell; the statistics kernel sets a device flag otherwise and the float form runs.
    Same signatures either way, against the oracle: integer counts (integer form), counts that are all fractions (float form),
    and matrices that must step aside -- one fractional count in one cell, a count of 32768, a cell
    whose counts sum above 65535 -- plus negative integers and a cell exactly at both limits (integer form)."""
    cells, genes, L = 700, 1500, 1024
    toc, g, c = synth.expression_matrix(cells, genes, density=0.06, cluster_count=5, seed=21)
    c = c.copy()
    if case == "all-fractions":
        c = (c * np.float32(0.37)).astype(np.float32)
    elif case == "one-fraction":
        c[len(c) // 2] = np.float32(2.5)
    elif case == "count-32768":
        c[7] = np.float32(32768.0)
    elif case == "sum-above-65535":
        lo, hi = int(toc[3]), int(toc[4])
        c[lo:hi] = np.float32(np.ceil(65536.0 / max(1, hi - lo)) + 1)
    elif case == "negative":
        c[::3] = -c[::3]
    elif case == "at-the-limits":
        lo, hi = int(toc[5]), int(toc[6])
        assert hi - lo >= 3
        c[lo:hi] = np.float32(0.0)
        c[lo], c[lo + 1], c[lo + 2] = np.float32(32767.0), np.float32(-32767.0), np.float32(1.0)       # sum|count| = 65535
    vectors = oracle.generate_lsh_vectors(genes, L, 231)
    expect = oracle.compute_signatures(toc, g, c, genes, vectors, L)
    got = capi.compute_signatures(toc, capi.make_counts(g, c), genes, vectors, L)
    assert np.array_equal(got, expect)


@pytest.mark.gpu
@pytest.mark.parametrize("case, L, aux, tier", [("integers", 1024, True, "fixed16-integer"), ("negative", 128, True, "fixed16-integer"),
                                                ("all-fractions", 1024, True, "fixed16-float"), ("one-fraction", 1024, True, "fixed16-float"),
                                                ("count-32768", 1024, True, "fixed16-float"), ("sum-above-65535", 1024, True, "fixed16-float"),
                                                ("integers", 100, True, "float"), ("integers", 96, True, "float"), ("integers", 63, True, "exact"),
                                                ("integers", 1024, False, "exact")])
def test_the_library_says_which_first_tier_ran(oracle, case, L, aux, tier):
    """em2_dev_compute_signatures_tier (include/em2_lsh.h): the first tier the last device call on a workspace ran -- the exact-integer
    form of the 16-bit tier for expression counts, its floating form when the statistics kernel found a count that is no small
    integer, the float copy for widths that are no multiple of 64, the reference's arithmetic alone otherwise.  bench.py prints it
    (roofline_projection.tier).  Signatures against the oracle as everywhere."""
    import torch
    cells, genes = 300, 900
    toc, g, c = synth.expression_matrix(cells, genes, density=0.06, cluster_count=5, seed=23)
    c = c.copy()
    if case == "all-fractions":
        c = (c * np.float32(0.37)).astype(np.float32)
    elif case == "one-fraction":
        c[len(c) // 2] = np.float32(2.5)
    elif case == "count-32768":
        c[7] = np.float32(32768.0)
    elif case == "sum-above-65535":
        lo, hi = int(toc[3]), int(toc[4])
        c[lo:hi] = np.float32(np.ceil(65536.0 / max(1, hi - lo)) + 1)
    elif case == "negative":
        c[::3] = -c[::3]
    vectors = oracle.generate_lsh_vectors(genes, L, 231)
    expect = oracle.compute_signatures(toc, g, c, genes, vectors, L)
    data = capi.make_counts(g, c)
    d_toc = torch.from_numpy(np.ascontiguousarray(toc).view(np.int64)).cuda()
    d_data = torch.from_numpy(np.ascontiguousarray(data).view(np.int64)).cuda()
    d_vectors = torch.from_numpy(np.ascontiguousarray(vectors)).cuda()
    d_sig = torch.zeros((cells, capi.word_count(L)), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    d_aux = torch.empty(capi.dev_vector_aux_bytes(genes, L), dtype=torch.uint8, device="cuda")
    if aux:
        capi.dev_prepare_vectors(d_vectors.data_ptr(), genes, L, d_aux.data_ptr(), stream)
    ws_bytes = capi.dev_compute_signatures_workspace(cells, L)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    capi.dev_compute_signatures(d_toc.data_ptr(), d_data.data_ptr(), cells, genes, d_vectors.data_ptr(), d_aux.data_ptr() if aux else 0, L,
                                d_sig.data_ptr(), ws.data_ptr(), ws_bytes, stream)
    assert capi.dev_compute_signatures_tier(ws.data_ptr(), cells, L, aux) == tier
    assert np.array_equal(d_sig.cpu().numpy().view(np.uint64), expect)
