"""GPU parity of findSimilarPairs5 (slice buckets -> candidate union -> mismatch filter -> keepBest) against the
CPU oracle's restatement of src/ExpressionMatrixLsh.cpp:355-496.  Bit-exact."""
import numpy as np
import pytest

import synth
from expressionmatrix2_amd import ExpressionMatrix, capi, files

pytestmark = pytest.mark.gpu


def check(oracle, sig, L, k, thr, q, ovf):
    cell, sim, used = oracle.find_similar_pairs5(sig, L, k, thr, q, ovf)
    pairs, gused = capi.find_similar_pairs5(sig, L, k, thr, q, ovf)
    assert np.array_equal(gused, used)
    assert np.array_equal(pairs["cell"], cell)
    assert np.array_equal(pairs["similarity"].view(np.uint32), sim.view(np.uint32))
    return used


@pytest.mark.parametrize("n,L,k,thr,q,ovf", [
    (300, 128, 5, 0.2, 8, 1000),
    (300, 1024, 100, 0.2, 14, 1000),      # slices straddle 64-bit words
    (1000, 1024, 10, 0.2, 10, 50),        # overflow rule drops the big buckets
    (1000, 1024, 10, 0.2, 10, 0),         # bucketOverflow = 0: no limit
    (777, 2048, 20, 0.0, 20, 1000),       # BASELINE config D shape (q = 20)
    (500, 192, 7, -0.5, 7, 1000),         # 192 / 7 leaves unused bits
    (400, 100, 3, 0.1, 3, 1000),
    (64, 64, 2, 0.2, 1, 0),               # 1-bit slices: two giant buckets per slice
    (200, 256, 4, 0.2, 19, 1000),
])
def test_fsp5_matches_oracle(oracle, n, L, k, thr, q, ovf):
    # NB the oracle restates the reference's tables literally (2^q vectors per slice): keep q <= 20 here.
    sig = synth.clustered_signatures(n, L, cluster_count=4, flip=0.05, seed=n + q)
    used = check(oracle, sig, L, k, thr, q, ovf)
    if ovf == 0 or ovf >= 1000:
        assert used.sum() > 0


def test_fsp5_slice_longer_than_signature(oracle):
    sig = synth.random_signatures(50, 64)
    pairs, used = capi.find_similar_pairs5(sig, 8, 5, 0.2, 16, 1000)        # lshCount 8 < 16: sliceCount 0
    assert used.sum() == 0 and not pairs["cell"].any()
    cell, sim, oused = oracle.find_similar_pairs5(sig, 8, 5, 0.2, 16, 1000)
    assert oused.sum() == 0


def test_fsp5_wide_slices_equal_narrow_run_on_duplicated_cells():
    """Slices wider than the oracle can tabulate (q = 32): with every cell present twice, each cell's only
    guaranteed bucket mate is its twin at mismatch 0, whatever q is."""
    base = synth.random_signatures(300, 256, seed=17)
    sig = np.concatenate([base, base])
    pairs, used = capi.find_similar_pairs5(sig, 256, 3, 0.9, 32, 1000)
    assert (used >= 1).all()
    twin = (np.arange(600) + 300) % 600
    assert np.array_equal(pairs["cell"][:, 0], twin.astype(np.uint32))
    assert (pairs["similarity"][:, 0] == 1.0).all()


def test_fsp5_identical_cells_long_lists(oracle):
    """Every cell identical: one bucket of 6000 per slice, every candidate passes, lists longer than the 4096-entry LDS
    staging area -> the 12288-entry selection launch; ties everywhere."""
    sig = np.tile(synth.random_signatures(1, 128, seed=5), (6000, 1))
    check(oracle, sig, 128, 9, 0.2, 16, 0)
    # 13000 identical cells: lists beyond every LDS staging area -> cut by one lane in HBM
    many = np.tile(synth.random_signatures(1, 64, seed=6), (13000, 1))
    cell, sim, oused = oracle.find_similar_pairs5_rows(many, 64, 4, 0.2, 16, 0, 0, 40)
    pairs, gused = capi.find_similar_pairs5(many, 64, 4, 0.2, 16, 0)
    assert np.array_equal(gused[:40], oused) and np.array_equal(pairs["cell"][:40], cell)
    assert np.array_equal(pairs["similarity"][:40].view(np.uint32), sim.view(np.uint32))
    assert (gused == 4).all()
    # with the overflow rule every bucket is dropped
    pairs, used = capi.find_similar_pairs5(sig, 128, 9, 0.2, 16, 1000)
    assert used.sum() == 0


def test_fsp5_rejects_zero_slice_length():
    sig = synth.random_signatures(10, 64)
    with pytest.raises(RuntimeError, match="lshSliceLength"):
        capi.find_similar_pairs5(sig, 64, 5, 0.2, 0, 1000)


def test_fsp5_through_expression_matrix_api(oracle, tmp_path):
    d = str(tmp_path / "data")
    cells, genes = 600, 500
    toc, g, c = synth.expression_matrix(cells, genes, density=0.05, cluster_count=4, seed=31)
    files.create_directory(d, genes, toc, capi.make_counts(g, c))
    e = ExpressionMatrix(d)
    e.computeLshSignatures(lshName="L", lshCount=1024, seed=231)
    e.findSimilarPairs5(lshName="L", similarPairsName="P5", k=12, similarityThreshold=0.2, lshSliceLength=12)
    L, sig = files.read_lsh(d, "L")
    cell, sim, used = oracle.find_similar_pairs5(sig, L, 12, 0.2, 12, 1000)
    k2, pairs, u2 = files.read_similar_pairs(d, "P5")
    assert k2 == 12 and np.array_equal(u2, used) and np.array_equal(pairs["cell"], cell)
    assert np.array_equal(pairs["similarity"].view(np.uint32), sim.view(np.uint32))
    assert used.sum() > 0


def test_fsp5_row_shard_through_device_api(oracle):
    import torch
    n, L, k, thr, q, ovf = 2500, 1024, 15, 0.2, 12, 1000
    sig = synth.clustered_signatures(n, L, cluster_count=6, flip=0.08, seed=77)
    d_sig = torch.from_numpy(sig.view(np.int64)).cuda()
    for begin, end in [(0, 900), (900, 901), (901, 2500)]:
        rows = end - begin
        d_pairs = torch.zeros((rows, k, 2), dtype=torch.int32, device="cuda")
        d_used = torch.zeros(rows, dtype=torch.int32, device="cuda")
        capi.dev_find_similar_pairs5(d_sig.data_ptr(), n, begin, end, L, k, thr, q, ovf, d_pairs.data_ptr(),
                                     d_used.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        cell, sim, used = oracle.find_similar_pairs5_rows(sig, L, k, thr, q, ovf, begin, end)
        p = d_pairs.cpu().numpy().view(np.uint32)
        assert np.array_equal(d_used.cpu().numpy().view(np.uint32), used)
        assert np.array_equal(p[:, :, 0], cell) and np.array_equal(p[:, :, 1], sim.view(np.uint32))


def test_fsp5_last_launch_reports_what_the_filter_read(oracle):
    """em2_dev_find_similar_pairs5_last_launch (bench.py's fsp5 roofline input): gathered candidates = the sum over
    cells and slices of the sizes of the non-overflowing buckets the cell falls in, the cell itself included
    (src/ExpressionMatrixLsh.cpp:417-433: the union is taken first, the cell dropped after)."""
    n, L, k, thr, q, ovf = 3000, 256, 10, 0.2, 9, 40
    sig = synth.clustered_signatures(n, L, cluster_count=12, flip=0.1, seed=77)
    check(oracle, sig, L, k, thr, q, ovf)
    info = capi.dev_find_similar_pairs5_last_launch()
    slices = L // q
    assert info["cells"] == n and info["slice_count"] == slices and info["batches"] >= 1
    assert info["filter_ms"] > 0 and info["select_ms"] > 0
    bits = np.unpackbits(sig.view(np.uint8).reshape(n, -1, 8)[:, :, ::-1].reshape(n, -1), axis=1)[:, :L]
    expected = 0
    member = np.zeros((n, n), dtype=bool)            # member[c, o]: o is in one of the buckets c gathers
    for s in range(slices):
        keys = bits[:, s * q:(s + 1) * q].astype(np.uint64) @ (1 << np.arange(q - 1, -1, -1, dtype=np.uint64))
        _, inverse, counts = np.unique(keys, return_inverse=True, return_counts=True)
        size = counts[inverse]
        expected += int(size[size <= ovf].sum())
        member |= (keys[:, None] == keys[None, :]) & (size <= ovf)[:, None]
    assert info["gathered_candidates"] == expected
    # the distinct candidates, i.e. the signatures the filter gathers (unionKernel counts them): the union's size per cell
    assert info["distinct_candidates"] == int(member.sum())


def test_fsp5_union_forms_agree(oracle):
    """The duplicate-free ascending union of a cell's buckets (src/multipleSetUnion.hpp:44-76) through the LDS bitmap
    (unionKernel: up to 256 slices) and through gather + segmented sort (more slices than that: 512 here): both against the
    oracle, on shapes that need several passes of the bitmap's id range would be too large for a unit test -- the pass logic is
    exercised by ids on both sides of a word and of a thread's 17-word share, overflowing buckets, and cells whose union is
    everything."""
    for n, L, k, thr, q, ovf in ((5000, 192, 12, 0.1, 6, 0), (3333, 256, 7, 0.2, 11, 25), (70, 64, 3, -1.0, 1, 0),
                                 (1500, 1024, 9, 0.1, 2, 0), (900, 1024, 5, 0.2, 3, 700)):
        sig = synth.clustered_signatures(n, L, cluster_count=5, flip=0.2, seed=n)
        check(oracle, sig, L, k, thr, q, ovf)


@pytest.mark.parametrize("batch_log2", ["29", "20"])
def test_fsp5_many_batches_agree(oracle, monkeypatch, batch_log2):
    """The filter visits the cells of a batch grouped by a neighbourhood label (smallest id in any of the cell's buckets, two
    rounds of pointer jumping), every XCD working through its own eighth of that order: a schedule, the SimilarPairs are the
    oracle's.  EM2_FSP5_BATCH_LOG2=20 cuts the cells into many batches (a million candidate ids each), whose orders are computed
    one by one."""
    monkeypatch.setenv("EM2_FSP5_BATCH_LOG2", batch_log2)
    for n, L, k, thr, q, ovf in ((6000, 256, 12, 0.1, 8, 0), (9000, 2048, 9, 0.2, 12, 300), (70, 64, 3, -1.0, 1, 0), (3001, 128, 5, 0.0, 7, 0)):
        sig = synth.clustered_signatures(n, L, cluster_count=7, flip=0.2, seed=n + 3)
        check(oracle, sig, L, k, thr, q, ovf)


def test_fsp5_filter_forms_agree(oracle):
    """The candidate filter (src/ExpressionMatrixLsh.cpp:436-457) by shape: 16-byte loads with several candidates in flight for
    an even number of 64-bit words up to 4096 bits (filterWideKernel: 256, 2048, 4096 bits here, and 128 bits = one unit), the
    cooperative 8-byte form for odd word counts (192 bits) and up to 8192 bits (6400), one lane per candidate beyond (8320)."""
    for n, L, k, thr, q, ovf in ((4000, 256, 12, 0.1, 8, 0), (3000, 2048, 9, 0.2, 11, 25), (2500, 192, 5, 0.0, 6, 0), (900, 4096, 4, 0.2, 12, 0),
                                 (1200, 128, 6, 0.1, 7, 0), (500, 6400, 4, 0.2, 12, 0), (400, 8320, 4, 0.2, 12, 0)):
        sig = synth.clustered_signatures(n, L, cluster_count=5, flip=0.2, seed=n + 1)
        check(oracle, sig, L, k, thr, q, ovf)


def test_fsp5_long_lists_all_selection_tiers(oracle):
    """Lists of 4097.., 12289.. candidates with few distinct keys (ties decide who survives keepBest) and with many: the packed
    LDS tiers (up to 16384 entries), the wave-parallel selection in global memory beyond, and -- k = 2500, above the packed
    tiers' 2048 -- the tiers of whole entries must all reproduce libstdc++'s nth_element."""
    rng = np.random.default_rng(17)
    for cells, L, k, flips in ((5000, 64, 7, 3), (5600, 128, 50, 12), (13000, 128, 25, 10), (7000, 256, 100, 40), (17000, 64, 3, 6), (5000, 128, 2500, 30)):
        base = synth.random_signatures(1, L, seed=cells)
        sig = np.tile(base, (cells, 1))
        # flip up to `flips` random bits per cell outside the first slice: one bucket holds everybody, keys vary
        for c in range(cells):
            for bit in rng.integers(16, L, rng.integers(0, flips + 1)):
                sig[c, bit // 64] ^= np.uint64(1) << np.uint64(63 - bit % 64)
        rows = (0, 40)
        cell, sim, oused = oracle.find_similar_pairs5_rows(sig, L, k, 0.2, 16, 0, *rows)
        pairs, gused = capi.find_similar_pairs5(sig, L, k, 0.2, 16, 0)
        assert np.array_equal(gused[:40], oused) and np.array_equal(pairs["cell"][:40], cell)
        assert np.array_equal(pairs["similarity"][:40].view(np.uint32), sim.view(np.uint32))
        cell, sim, oused = oracle.find_similar_pairs5_rows(sig, L, k, 0.2, 16, 0, cells - 30, cells)
        assert np.array_equal(gused[-30:], oused) and np.array_equal(pairs["cell"][-30:], cell)


@pytest.mark.parametrize("cache_mb", [None, "0", "1"])
def test_fsp5_scratch_cache_is_only_a_cache(oracle, monkeypatch, cache_mb):
    """The call's device scratch comes from blocks the process keeps between calls (hipMalloc of gigabytes costs up to 200 ms on
    some hosts): repeated calls of different sizes, with the cache off (EM2_SCRATCH_CACHE_MB=0), too small to keep anything
    useful (1 MB) and at its default, and after em2_dev_release_scratch(), all give the oracle's SimilarPairs."""
    if cache_mb is not None:
        monkeypatch.setenv("EM2_SCRATCH_CACHE_MB", cache_mb)
    for n, L, k, q in ((4000, 256, 9, 8), (2500, 256, 9, 8), (4000, 256, 9, 8), (6000, 128, 5, 7)):
        sig = synth.clustered_signatures(n, L, cluster_count=5, flip=0.2, seed=n)
        check(oracle, sig, L, k, 0.2, q, 0)
        if n == 2500:
            capi.dev_release_scratch()
    capi.dev_release_scratch()
