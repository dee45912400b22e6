"""GPU parity of findSimilarPairs4: the HIP scan kernel, through the C ABI, against the CPU oracle.
Bit-exact on cell ids, float similarity bit patterns and usedCount."""
import numpy as np
import pytest

import synth
from expressionmatrix2_amd import capi
from test_fsp4_cpu import CASES, make

pytestmark = pytest.mark.gpu


def assert_same(gpu_pairs, gpu_used, cell, sim, used):
    assert np.array_equal(gpu_used, used)
    assert np.array_equal(gpu_pairs["cell"], cell)
    assert np.array_equal(gpu_pairs["similarity"].view(np.uint32), sim.view(np.uint32))


@pytest.mark.parametrize("n,L,k,thr,kind", CASES)
def test_fsp4_matches_oracle(oracle, n, L, k, thr, kind):
    sig = make(n, L, kind)
    cell, sim, used = oracle.find_similar_pairs4(sig, L, k, thr)
    pairs, gused = capi.find_similar_pairs4(sig, L, k, thr)
    assert_same(pairs, gused, cell, sim, used)


@pytest.mark.parametrize("L", [1, 63, 64, 65, 128, 192, 320, 512, 1000, 1024, 2048, 3000, 4096])
def test_fsp4_signature_widths(oracle, L):
    sig = synth.clustered_signatures(333, L, cluster_count=3, flip=0.2, seed=L)
    cell, sim, used = oracle.find_similar_pairs4(sig, L, 6, 0.1)
    pairs, gused = capi.find_similar_pairs4(sig, L, 6, 0.1)
    assert_same(pairs, gused, cell, sim, used)


@pytest.mark.parametrize("k", [1, 2, 31, 32, 33, 64, 100, 257])
def test_fsp4_k_values(oracle, k):
    sig = synth.clustered_signatures(900, 256, cluster_count=2, flip=0.1, seed=k)
    cell, sim, used = oracle.find_similar_pairs4(sig, 256, k, 0.2)
    pairs, gused = capi.find_similar_pairs4(sig, 256, k, 0.2)
    assert_same(pairs, gused, cell, sim, used)


def test_fsp4_k_zero_and_nothing_passes(oracle):
    sig = synth.random_signatures(100, 128)
    pairs, used = capi.find_similar_pairs4(sig, 128, 0, 0.2)
    assert used.sum() == 0
    pairs, used = capi.find_similar_pairs4(sig, 128, 5, 1.0)       # nothing is > 1.0
    cell, sim, oused = oracle.find_similar_pairs4(sig, 128, 5, 1.0)
    assert_same(pairs, used, cell, sim, oused)
    assert used.sum() == 0 and not pairs["cell"].any()


def test_fsp4_all_identical_cells(oracle):
    sig = np.tile(synth.random_signatures(1, 256, seed=3), (700, 1))
    cell, sim, used = oracle.find_similar_pairs4(sig, 256, 8, 0.2)
    pairs, gused = capi.find_similar_pairs4(sig, 256, 8, 0.2)
    assert_same(pairs, gused, cell, sim, used)


def test_fsp4_sampled_rows_of_larger_problem(oracle):
    """20k cells: the full O(N^2) oracle is too slow for a unit test, so compare sampled row ranges
    (rows are independent under the per-cell contract)."""
    n, L, k, thr = 20000, 1024, 100, 0.2
    sig = synth.clustered_signatures(n, L, cluster_count=16, flip=0.15, seed=99)
    pairs, gused = capi.find_similar_pairs4(sig, L, k, thr)
    for begin in (0, 6400, 19900):
        end = min(n, begin + 100)
        cell, sim, used = oracle.find_similar_pairs4_rows(sig, L, k, thr, begin, end)
        assert_same(pairs[begin:end], gused[begin:end], cell, sim, used)
    # size-independent properties on the whole result
    assert (gused <= k).all()
    s = pairs["similarity"]
    for row in range(0, n, 997):
        u = int(gused[row])
        assert (np.diff(s[row, :u]) <= 0).all()
        assert row not in pairs["cell"][row, :u]
        assert len(set(pairs["cell"][row, :u].tolist())) == u
        assert not pairs["cell"][row, u:].any() and not s[row, u:].any()


def test_fsp4_row_shard_through_device_api(oracle):
    """The sharded entry point (rows [begin,end) vs all columns) used by the multi-GPU path."""
    import torch
    n, L, k, thr = 3000, 1024, 20, 0.2
    sig = synth.clustered_signatures(n, L, cluster_count=8, flip=0.15, seed=5)
    d_sig = torch.from_numpy(sig.view(np.int64)).cuda()
    for begin, end in [(0, 1000), (1000, 1001), (1001, 3000), (2999, 3000)]:
        rows = end - begin
        ws_bytes = capi.dev_find_similar_pairs4_workspace(n, rows, L, k)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
        d_pairs = torch.empty((rows, k, 2), dtype=torch.int32, device="cuda")
        d_used = torch.empty(rows, dtype=torch.int32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        capi.dev_find_similar_pairs4(d_sig.data_ptr(), n, begin, end, L, k, thr, d_pairs.data_ptr(),
                                     d_used.data_ptr(), ws.data_ptr(), ws_bytes, stream)
        torch.cuda.synchronize()
        pairs = d_pairs.cpu().numpy().view(np.uint32)
        cell, sim, used = oracle.find_similar_pairs4_rows(sig, L, k, thr, begin, end)
        assert np.array_equal(d_used.cpu().numpy().view(np.uint32), used)
        assert np.array_equal(pairs[:, :, 0], cell)
        assert np.array_equal(pairs[:, :, 1], sim.view(np.uint32))


@pytest.fixture()
def scan_knobs():
    """Set / restore the scan kernel's test knobs (read with getenv at every launch)."""
    import os
    saved = {k: os.environ.get(k) for k in ("EM2_MIN_SEGMENT_COLUMNS", "EM2_LOG_CAPACITY", "EM2_SCAN_MODE",
                                             "EM2_BLOCKS_PER_CU", "EM2_FULL_ROW_CELLS", "EM2_INBOX_CAPACITY",
                                             "EM2_PREFIX_PERMILLE", "EM2_TILE_SEGMENTS", "EM2_SCAN_MATRIX", "EM2_MATRIX_CONVOY")}

    def set_knobs(**kw):
        for key, value in kw.items():
            if value is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = str(value)
    yield set_knobs
    for key, value in saved.items():
        if value is None:
            os.environ.pop(key, None)
        else:
            os.environ[key] = value


@pytest.mark.parametrize("min_cols,log_cap", [(64, None), (64, 3), (100, 1), (257, 16), (1000, 2)])
@pytest.mark.parametrize("n,L,k,thr,kind", [(1200, 1024, 25, 0.2, "clustered"), (2000, 256, 7, -0.5, "clustered"),
                                             (900, 128, 3, 0.0, "random")])
def test_fsp4_segment_handoff_speculation_and_log_overflow(oracle, scan_knobs, min_cols, log_cap, n, L, k, thr, kind):
    """Many short column segments (hand-off between waves, speculative look-ahead) and tiny speculative logs
    (overflow -> resume exactly): the result must not depend on any of it."""
    sig = make(n, L, kind)
    cell, sim, used = oracle.find_similar_pairs4(sig, L, k, thr)
    scan_knobs(EM2_MIN_SEGMENT_COLUMNS=min_cols, EM2_LOG_CAPACITY=log_cap)
    pairs, gused = capi.find_similar_pairs4(sig, L, k, thr)
    assert_same(pairs, gused, cell, sim, used)


def test_fsp4_simple_kernel_still_matches(oracle, scan_knobs):
    sig = make(1200, 1024, "clustered")
    cell, sim, used = oracle.find_similar_pairs4(sig, 1024, 25, 0.2)
    scan_knobs(EM2_SCAN_MODE="simple")
    pairs, gused = capi.find_similar_pairs4(sig, 1024, 25, 0.2)
    assert_same(pairs, gused, cell, sim, used)


def test_fsp4_repeated_runs_are_identical(scan_knobs):
    """The speculative path depends on timing (which segments find their predecessor finished); the result must not."""
    sig = synth.clustered_signatures(6000, 512, cluster_count=5, flip=0.15, seed=8)
    scan_knobs(EM2_MIN_SEGMENT_COLUMNS=128, EM2_LOG_CAPACITY=8)
    first = capi.find_similar_pairs4(sig, 512, 50, 0.2)
    for blocks in (1, 2, 4):
        scan_knobs(EM2_BLOCKS_PER_CU=blocks)
        again = capi.find_similar_pairs4(sig, 512, 50, 0.2)
        assert np.array_equal(first[0], again[0]) and np.array_equal(first[1], again[1])


# ---- the symmetric (each unordered pair once) scan: EM2_SCAN_MODE=triangle forces it at test sizes ----

@pytest.mark.parametrize("n,L,k,thr,kind", CASES)
def test_fsp4_symmetric_matches_oracle(oracle, scan_knobs, n, L, k, thr, kind):
    sig = make(n, L, kind)
    cell, sim, used = oracle.find_similar_pairs4(sig, L, k, thr)
    scan_knobs(EM2_SCAN_MODE="triangle", EM2_MIN_SEGMENT_COLUMNS=64, EM2_FULL_ROW_CELLS=64)
    pairs, gused = capi.find_similar_pairs4(sig, L, k, thr)
    assert_same(pairs, gused, cell, sim, used)


@pytest.mark.parametrize("full_rows,min_cols,log_cap", [(0, 64, None), (0, 100, 2), (64, 64, 3), (200, 257, 1),
                                                         (640, 1000, None), (100000, 64, None)])
@pytest.mark.parametrize("n,L,k,thr,kind", [(1200, 1024, 25, 0.2, "clustered"), (2000, 256, 7, -0.5, "clustered"),
                                             (900, 128, 3, 0.0, "random"), (1537, 2048, 10, 0.1, "clustered"),
                                             (777, 64, 300, -1.0, "random")])
def test_fsp4_symmetric_layouts(oracle, scan_knobs, full_rows, min_cols, log_cap, n, L, k, thr, kind):
    """Full-row prefix sizes (none .. everything), segment lengths that do and do not divide the block size,
    tiny speculative logs: the symmetric scan's result must not depend on any of it."""
    sig = make(n, L, kind)
    cell, sim, used = oracle.find_similar_pairs4(sig, L, k, thr)
    scan_knobs(EM2_SCAN_MODE="triangle", EM2_MIN_SEGMENT_COLUMNS=min_cols, EM2_FULL_ROW_CELLS=full_rows,
               EM2_LOG_CAPACITY=log_cap)
    pairs, gused = capi.find_similar_pairs4(sig, L, k, thr)
    assert_same(pairs, gused, cell, sim, used)


@pytest.mark.parametrize("L", [1, 64, 65, 192, 512, 1000, 3000, 4096])
def test_fsp4_symmetric_signature_widths(oracle, scan_knobs, L):
    sig = synth.clustered_signatures(333, L, cluster_count=3, flip=0.2, seed=L)
    cell, sim, used = oracle.find_similar_pairs4(sig, L, 6, 0.1)
    scan_knobs(EM2_SCAN_MODE="triangle", EM2_MIN_SEGMENT_COLUMNS=50, EM2_FULL_ROW_CELLS=0)
    pairs, gused = capi.find_similar_pairs4(sig, L, 6, 0.1)
    assert_same(pairs, gused, cell, sim, used)


def test_fsp4_symmetric_identical_cells_and_increasing_similarity(oracle, scan_knobs):
    """All-equal cells (every candidate ties) and the adversarial order in which every later cell is a better
    match than all earlier ones (every candidate is accepted: the inbox takes the whole upper triangle)."""
    scan_knobs(EM2_SCAN_MODE="triangle", EM2_MIN_SEGMENT_COLUMNS=64, EM2_FULL_ROW_CELLS=0)
    sig = np.tile(synth.random_signatures(1, 256, seed=3), (700, 1))
    cell, sim, used = oracle.find_similar_pairs4(sig, 256, 8, 0.2)
    pairs, gused = capi.find_similar_pairs4(sig, 256, 8, 0.2)
    assert_same(pairs, gused, cell, sim, used)
    # cell i = the last cell with (n-1-i) low bits flipped: similarity to later cells grows with the id
    n, L = 600, 1024
    base = synth.random_signatures(1, L, seed=9)[0]
    sig = np.tile(base, (n, 1))
    for i in range(n):
        flips = n - 1 - i
        for b in range(flips):
            sig[i, b // 64] ^= np.uint64(1) << np.uint64(b % 64)
    cell, sim, used = oracle.find_similar_pairs4(sig, L, 5, 0.0)
    pairs, gused = capi.find_similar_pairs4(sig, L, 5, 0.0)
    assert_same(pairs, gused, cell, sim, used)


def test_fsp4_symmetric_inbox_overflow_falls_back(oracle, scan_knobs):
    sig = make(2000, 256, "clustered")
    cell, sim, used = oracle.find_similar_pairs4(sig, 256, 7, -0.5)
    scan_knobs(EM2_SCAN_MODE="triangle", EM2_MIN_SEGMENT_COLUMNS=64, EM2_FULL_ROW_CELLS=0, EM2_INBOX_CAPACITY=1024)
    pairs, gused = capi.find_similar_pairs4(sig, 256, 7, -0.5)
    assert_same(pairs, gused, cell, sim, used)


def test_fsp4_symmetric_sampled_rows_and_repeatability(oracle, scan_knobs):
    n, L, k, thr = 20000, 1024, 100, 0.2
    sig = synth.clustered_signatures(n, L, cluster_count=16, flip=0.15, seed=99)
    scan_knobs(EM2_SCAN_MODE="persistent")
    ordered = capi.find_similar_pairs4(sig, L, k, thr)
    scan_knobs(EM2_SCAN_MODE="triangle", EM2_FULL_ROW_CELLS=1024)
    first = capi.find_similar_pairs4(sig, L, k, thr)
    assert np.array_equal(first[0], ordered[0]) and np.array_equal(first[1], ordered[1])
    for begin in (0, 1000, 6400, 19900):
        end = min(n, begin + 60)
        cell, sim, used = oracle.find_similar_pairs4_rows(sig, L, k, thr, begin, end)
        assert_same(first[0][begin:end], first[1][begin:end], cell, sim, used)
    for blocks, columns in ((1, 2900), (2, 313), (4, 607)):
        scan_knobs(EM2_BLOCKS_PER_CU=blocks, EM2_MIN_SEGMENT_COLUMNS=columns)
        again = capi.find_similar_pairs4(sig, L, k, thr)
        assert np.array_equal(first[0], again[0]) and np.array_equal(first[1], again[1])


# ---- the multi-GPU symmetric scan with all ranks played on one GPU (EM2_SCAN_MODE=virtual): block-cyclic row
# ownership, prefix phases, deferred tiles, entry exchange, replay ----

@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("n,L,k,thr,kind", [(1200, 1024, 25, 0.2, "clustered"), (2000, 256, 7, -0.5, "clustered"),
                                             (2111, 128, 3, 0.0, "random"), (3000, 2048, 10, 0.1, "clustered"),
                                             (2500, 64, 300, -1.0, "random")])
def test_fsp4_sharded_virtual_world(oracle, scan_knobs, world, n, L, k, thr, kind):
    sig = make(n, L, kind)
    cell, sim, used = oracle.find_similar_pairs4(sig, L, k, thr)
    scan_knobs(EM2_SCAN_MODE="virtual:%d" % world, EM2_MIN_SEGMENT_COLUMNS=64, EM2_TILE_SEGMENTS=5)
    pairs, gused = capi.find_similar_pairs4(sig, L, k, thr)
    assert_same(pairs, gused, cell, sim, used)


@pytest.mark.parametrize("permille,tile_segments,log_cap", [(50, 1, None), (200, 3, 2), (500, 7, None), (900, 256, 1)])
def test_fsp4_sharded_prefix_sizes_and_tilings(oracle, scan_knobs, permille, tile_segments, log_cap):
    sig = make(4000, 512, "clustered")
    cell, sim, used = oracle.find_similar_pairs4(sig, 512, 20, 0.2)
    scan_knobs(EM2_SCAN_MODE="virtual:4", EM2_MIN_SEGMENT_COLUMNS=100, EM2_PREFIX_PERMILLE=permille,
               EM2_TILE_SEGMENTS=tile_segments, EM2_LOG_CAPACITY=log_cap)
    pairs, gused = capi.find_similar_pairs4(sig, 512, 20, 0.2)
    assert_same(pairs, gused, cell, sim, used)


def test_fsp4_sharded_identical_cells_and_overflow_fallback(oracle, scan_knobs):
    sig = np.tile(synth.random_signatures(1, 256, seed=3), (1500, 1))
    cell, sim, used = oracle.find_similar_pairs4(sig, 256, 8, 0.2)
    scan_knobs(EM2_SCAN_MODE="virtual:2", EM2_MIN_SEGMENT_COLUMNS=64)
    pairs, gused = capi.find_similar_pairs4(sig, 256, 8, 0.2)
    assert_same(pairs, gused, cell, sim, used)
    scan_knobs(EM2_INBOX_CAPACITY=1024)          # pools overflow -> the ordered scan runs instead
    pairs, gused = capi.find_similar_pairs4(sig, 256, 8, 0.2)
    assert_same(pairs, gused, cell, sim, used)


def test_fsp4_sharded_sampled_rows_of_larger_problem(oracle, scan_knobs):
    n, L, k, thr = 20000, 1024, 100, 0.2
    sig = synth.clustered_signatures(n, L, cluster_count=16, flip=0.15, seed=99)
    scan_knobs(EM2_SCAN_MODE="persistent")
    ordered = capi.find_similar_pairs4(sig, L, k, thr)
    for world in (2, 8):
        scan_knobs(EM2_SCAN_MODE="virtual:%d" % world)
        got = capi.find_similar_pairs4(sig, L, k, thr)
        assert np.array_equal(got[0], ordered[0]) and np.array_equal(got[1], ordered[1])


@pytest.mark.parametrize("mode", ["persistent", "triangle", "virtual"])
def test_fsp4_golden_digests(scan_knobs, mode):
    """The committed golden digests (tests/golden/oracle_regression.json, made by the oracle at commit time) straight
    against the GPU result, all three scan forms, including the 3000-cell case stored as hashes only."""
    import json
    import os
    from golden.make_golden import digest, make_signatures, regression_cases
    with open(os.path.join(os.path.dirname(__file__), "golden", "oracle_regression.json")) as f:
        golden = json.load(f)
    scan_knobs(EM2_SCAN_MODE=mode, EM2_MIN_SEGMENT_COLUMNS=64, EM2_FULL_ROW_CELLS=128)          # ("virtual" alone: two ranks)
    for case in regression_cases():
        sig = make_signatures(case)
        pairs, used = capi.find_similar_pairs4(sig, case["L"], case["k"], case["thr"])
        cell = np.ascontiguousarray(pairs["cell"])
        sim = np.ascontiguousarray(pairs["similarity"])
        assert digest(cell, sim, used) == golden[case["name"]]["fsp4"], case["name"]


@pytest.mark.parametrize("L", [2048, 4096])
def test_fsp4_symmetric_log_full_at_last_column_below_the_block(oracle, scan_knobs, L):
    """Found by tools/fuzz_parity.py: a speculative log that fills up exactly at the last column below a row block
    must not receive the block's diagonal columns as well (rows 1222 / 1230 / 1254 of this case came out wrong)."""
    sig = synth.clustered_signatures(1500, L, cluster_count=5, flip=0.1, seed=1)
    cell, sim, used = oracle.find_similar_pairs4(sig, L, 10, -0.5)
    scan_knobs(EM2_SCAN_MODE="triangle", EM2_MIN_SEGMENT_COLUMNS=257, EM2_LOG_CAPACITY=16, EM2_FULL_ROW_CELLS=200,
               EM2_BLOCKS_PER_CU=2)
    pairs, gused = capi.find_similar_pairs4(sig, L, 10, -0.5)
    assert_same(pairs, gused, cell, sim, used)


# ---- the matrix-core form of the symmetric scan (1024-bit signatures; EM2_SCAN_MATRIX=0 switches it off) ----

def test_matrix_form_is_the_one_that_runs(oracle, scan_knobs):
    """513..1024 bits, symmetric form, at least one quad of triangle rows: the launch reports form 3 and a non-zero
    number of pairs contracted by v_mfma_scale_f32_32x32x64_f8f6f4; with EM2_SCAN_MATRIX=0 it is form 1 again."""
    sig = make(3000, 1024, "clustered")
    cell, sim, used = oracle.find_similar_pairs4(sig, 1024, 20, 0.2)
    scan_knobs(EM2_SCAN_MODE="triangle", EM2_FULL_ROW_CELLS=256)
    pairs, gused = capi.find_similar_pairs4(sig, 1024, 20, 0.2)
    info = capi.dev_find_similar_pairs4_last_launch()
    assert info["form"] == 3 and info["matrix_pairs"] > 0
    assert_same(pairs, gused, cell, sim, used)
    scan_knobs(EM2_SCAN_MODE="triangle", EM2_FULL_ROW_CELLS=256, EM2_SCAN_MATRIX=0)
    pairs, gused = capi.find_similar_pairs4(sig, 1024, 20, 0.2)
    assert capi.dev_find_similar_pairs4_last_launch()["form"] == 1
    assert_same(pairs, gused, cell, sim, used)


@pytest.mark.parametrize("n,L,k,thr,kind,knobs", [
    (2500, 1024, 10, 0.2, "clustered", dict(EM2_FULL_ROW_CELLS=0, EM2_MIN_SEGMENT_COLUMNS=256)),      # no full rows: the first quad has only its band
    (2500, 1000, 10, 0.2, "clustered", dict(EM2_FULL_ROW_CELLS=0, EM2_MIN_SEGMENT_COLUMNS=256)),      # padded bits count as equal
    (2309, 600, 7, 0.0, "clustered", dict(EM2_FULL_ROW_CELLS=300, EM2_MIN_SEGMENT_COLUMNS=700)),         # cells % 256 != 0: a short last quad, an idle wave
    (1700, 1024, 5, -1.0, "random", dict(EM2_FULL_ROW_CELLS=0, EM2_MIN_SEGMENT_COLUMNS=512)),         # everything passes: logs fill, the walk stops and resumes
    (1700, 1024, 300, -0.5, "clustered", dict(EM2_FULL_ROW_CELLS=256, EM2_MIN_SEGMENT_COLUMNS=256, EM2_LOG_CAPACITY=1)),
    (4000, 1024, 25, 0.5, "clustered", dict(EM2_FULL_ROW_CELLS=512, EM2_MIN_SEGMENT_COLUMNS=1300, EM2_BLOCKS_PER_CU=1)),
    (1300, 1024, 100, 0.2, "equal", dict(EM2_FULL_ROW_CELLS=256)),                                     # all cells identical: dot = 1024 everywhere
])
def test_matrix_form_matches_oracle(oracle, scan_knobs, n, L, k, thr, kind, knobs):
    sig = np.tile(make(1, L, "random"), (n, 1)) if kind == "equal" else make(n, L, kind)
    cell, sim, used = oracle.find_similar_pairs4(sig, L, k, thr)
    scan_knobs(EM2_SCAN_MODE="triangle", **knobs)
    pairs, gused = capi.find_similar_pairs4(sig, L, k, thr)
    assert capi.dev_find_similar_pairs4_last_launch()["form"] == 3
    assert_same(pairs, gused, cell, sim, used)


@pytest.mark.parametrize("convoy", [0, 2, 3, 6])
@pytest.mark.parametrize("n,L,k,thr,kind,knobs", [
    (2500, 1024, 10, 0.2, "clustered", dict(EM2_FULL_ROW_CELLS=0, EM2_MIN_SEGMENT_COLUMNS=512)),
    (4000, 1024, 25, 0.2, "clustered", dict(EM2_FULL_ROW_CELLS=512, EM2_MIN_SEGMENT_COLUMNS=1300)),      # full rows go around as well
    (2309, 600, 7, 0.0, "clustered", dict(EM2_FULL_ROW_CELLS=300, EM2_MIN_SEGMENT_COLUMNS=700)),       # an idle wave
    (1700, 1024, 5, -1.0, "random", dict(EM2_FULL_ROW_CELLS=0, EM2_MIN_SEGMENT_COLUMNS=512)),          # logs fill on either side of the wrap
    (1700, 1024, 300, -0.5, "clustered", dict(EM2_FULL_ROW_CELLS=256, EM2_MIN_SEGMENT_COLUMNS=512, EM2_LOG_CAPACITY=1)),
    (2500, 2048, 10, 0.2, "clustered", dict(EM2_FULL_ROW_CELLS=0, EM2_MIN_SEGMENT_COLUMNS=512)),       # the 2048-bit walk: each pass goes around
    (1700, 2048, 5, -1.0, "random", dict(EM2_FULL_ROW_CELLS=0, EM2_MIN_SEGMENT_COLUMNS=512)),
    (3100, 1500, 40, 0.1, "clustered", dict(EM2_FULL_ROW_CELLS=256, EM2_MIN_SEGMENT_COLUMNS=1500, EM2_LOG_CAPACITY=1)),
])
def test_matrix_form_walks_that_go_around(oracle, scan_knobs, n, L, k, thr, kind, knobs, convoy):
    """The convoy: a walk starts where the other walks of its XCD are and goes around its segment; the replay
    takes the logs in the order of the columns.  EM2_MATRIX_CONVOY=n >= 2 starts EVERY walk 64 (n - 1) columns into its
    segment (the product's setting, 1, follows the other blocks: not reproducible); 0 = no walk goes around.  Logs that fill
    before the wrap send the walk back to the segment's begin, logs that fill behind it are replayed in part."""
    sig = make(n, L, kind)
    cell, sim, used = oracle.find_similar_pairs4(sig, L, k, thr)
    scan_knobs(EM2_SCAN_MODE="triangle", EM2_MATRIX_CONVOY=convoy, **knobs)
    pairs, gused = capi.find_similar_pairs4(sig, L, k, thr)
    info = capi.dev_find_similar_pairs4_last_launch()
    assert info["form"] == 3 and info["matrix_pairs"] > 0
    assert_same(pairs, gused, cell, sim, used)


@pytest.mark.parametrize("convoy", [1, 4, 5, 6])
@pytest.mark.parametrize("L", [1024, 2048])
def test_walk_that_goes_around_with_nearly_full_logs(oracle, scan_knobs, L, convoy):
    """Found by tools/fuzz_parity.py (seed 41): one cluster, k = 5, threshold 0.5 -- nearly every pair is logged.  A walk whose
    logs are nearly full when it has reached its segment's end must not go on into the lower columns (a call of the walk has to
    find room for three tiles' records: the logs of the neighbouring lanes were overwritten, rows got neighbours with the
    similarity of another pair); it starts again at the segment's begin instead."""
    sig = synth.clustered_signatures(1500, L, cluster_count=1, flip=0.1, seed=285543329)
    cell, sim, used = oracle.find_similar_pairs4(sig, L, 5, 0.5)
    scan_knobs(EM2_SCAN_MODE="triangle", EM2_MIN_SEGMENT_COLUMNS=257, EM2_FULL_ROW_CELLS=64, EM2_MATRIX_CONVOY=convoy)
    pairs, gused = capi.find_similar_pairs4(sig, L, 5, 0.5)
    assert capi.dev_find_similar_pairs4_last_launch()["form"] == 3
    assert_same(pairs, gused, cell, sim, used)


@pytest.mark.parametrize("n,L,k,thr,kind", [(2500, 64, 10, 0.2, "clustered"), (2309, 192, 7, 0.0, "clustered"),
                                             (3000, 512, 20, 0.2, "clustered"), (1700, 300, 5, -1.0, "random"),
                                             (2600, 100, 30, 0.1, "clustered"), (2100, 33, 4, 0.3, "random")])
def test_matrix_form_of_narrow_signatures(oracle, scan_knobs, n, L, k, thr, kind):
    """EM2_SCAN_MATRIX=2: every width up to 1024 bits on the matrix cores (the default, 1, takes 129 bits and up) (the signatures zero-extended to 1024 bits: a
    padded bit is +1 on both sides, the dot product is still 1024 - 2 * mismatches).  Same bytes as the oracle and as
    the v_xor/v_bcnt form."""
    sig = make(n, L, kind)
    cell, sim, used = oracle.find_similar_pairs4(sig, L, k, thr)
    scan_knobs(EM2_SCAN_MODE="triangle", EM2_FULL_ROW_CELLS=256, EM2_MIN_SEGMENT_COLUMNS=512, EM2_SCAN_MATRIX=2)
    pairs, gused = capi.find_similar_pairs4(sig, L, k, thr)
    info = capi.dev_find_similar_pairs4_last_launch()
    assert info["form"] == 3 and info["matrix_pairs"] > 0
    assert_same(pairs, gused, cell, sim, used)
    scan_knobs(EM2_SCAN_MODE="triangle", EM2_FULL_ROW_CELLS=256, EM2_MIN_SEGMENT_COLUMNS=512, EM2_SCAN_MATRIX=0)
    pairs, gused = capi.find_similar_pairs4(sig, L, k, thr)
    assert capi.dev_find_similar_pairs4_last_launch()["form"] == 1
    assert_same(pairs, gused, cell, sim, used)


@pytest.mark.parametrize("n,L,k,thr,kind,knobs", [
    (2500, 2048, 10, 0.2, "clustered", dict(EM2_FULL_ROW_CELLS=0, EM2_MIN_SEGMENT_COLUMNS=256)),
    (2500, 1500, 10, 0.2, "clustered", dict(EM2_FULL_ROW_CELLS=0, EM2_MIN_SEGMENT_COLUMNS=256)),      # padded bits count as equal
    (2309, 1025, 7, 0.0, "clustered", dict(EM2_FULL_ROW_CELLS=300, EM2_MIN_SEGMENT_COLUMNS=700)),     # a short last quad, an idle wave
    (1700, 2048, 5, -1.0, "random", dict(EM2_FULL_ROW_CELLS=0, EM2_MIN_SEGMENT_COLUMNS=512)),         # everything passes: the walk stops and resumes
    (1700, 2048, 300, -0.5, "clustered", dict(EM2_FULL_ROW_CELLS=256, EM2_MIN_SEGMENT_COLUMNS=256, EM2_LOG_CAPACITY=1)),
    (4000, 2000, 25, 0.5, "clustered", dict(EM2_FULL_ROW_CELLS=512, EM2_MIN_SEGMENT_COLUMNS=1300, EM2_BLOCKS_PER_CU=1)),
    (1300, 2048, 100, 0.2, "equal", dict(EM2_FULL_ROW_CELLS=256)),                                     # all cells identical: dot = 2048 everywhere
])
def test_matrix_form_of_2048_bit_signatures(oracle, scan_knobs, n, L, k, thr, kind, knobs):
    """1025..2048 bits: the 2048-bit form of the matrix kernel (fsp4ScanMatrixWideKernel: 32 rows per wave and pass, two
    passes over the columns).  Same bytes as the oracle and as the v_xor/v_bcnt form (EM2_SCAN_MATRIX=3: no matrix form above 1024 bits)."""
    sig = np.tile(make(1, L, "random"), (n, 1)) if kind == "equal" else make(n, L, kind)
    cell, sim, used = oracle.find_similar_pairs4(sig, L, k, thr)
    scan_knobs(EM2_SCAN_MODE="triangle", **knobs)
    pairs, gused = capi.find_similar_pairs4(sig, L, k, thr)
    info = capi.dev_find_similar_pairs4_last_launch()
    assert info["form"] == 3 and info["matrix_pairs"] > 0
    assert_same(pairs, gused, cell, sim, used)
    scan_knobs(EM2_SCAN_MODE="triangle", EM2_SCAN_MATRIX=3, **knobs)
    pairs, gused = capi.find_similar_pairs4(sig, L, k, thr)
    assert capi.dev_find_similar_pairs4_last_launch()["form"] == 1
    assert_same(pairs, gused, cell, sim, used)


def test_matrix_form_inbox_overflow_falls_back(oracle, scan_knobs):
    sig = make(2000, 1024, "clustered")
    cell, sim, used = oracle.find_similar_pairs4(sig, 1024, 10, -0.5)
    scan_knobs(EM2_SCAN_MODE="triangle", EM2_FULL_ROW_CELLS=0, EM2_INBOX_CAPACITY=1024)
    pairs, gused = capi.find_similar_pairs4(sig, 1024, 10, -0.5)
    assert capi.dev_find_similar_pairs4_last_launch()["form"] == 4          # every row walked all columns instead, on the matrix cores
    assert_same(pairs, gused, cell, sim, used)
    scan_knobs(EM2_SCAN_MODE="triangle", EM2_FULL_ROW_CELLS=0, EM2_INBOX_CAPACITY=1024, EM2_SCAN_MATRIX=3)
    sig = make(2000, 2048, "clustered")
    cell, sim, used = oracle.find_similar_pairs4(sig, 2048, 10, -0.5)
    pairs, gused = capi.find_similar_pairs4(sig, 2048, 10, -0.5)
    assert capi.dev_find_similar_pairs4_last_launch()["form"] == 0          # no matrix form for this width: the ordered scan
    assert_same(pairs, gused, cell, sim, used)


# ---- the rows form on the matrix cores: rows [rowBegin, rowEnd) x all columns (a rank's shard; the overflow fallback) ----

def run_row_shard(sig, L, k, thr, begin, end, poison=None):
    """em2_dev_find_similar_pairs4 for rows [begin, end): (cell, similarity bits, used) as numpy arrays.  poison: the byte the
    workspace is filled with before the call (what a caller's recycled memory may hold)."""
    import torch
    n = sig.shape[0]
    rows = end - begin
    d_sig = torch.from_numpy(sig.view(np.int64)).cuda()
    ws_bytes = capi.dev_find_similar_pairs4_workspace(n, rows, L, k)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda") if poison is None else \
        torch.full((ws_bytes,), poison, dtype=torch.uint8, device="cuda")
    d_pairs = torch.empty((rows, k, 2), dtype=torch.int32, device="cuda")
    d_used = torch.empty(rows, dtype=torch.int32, device="cuda")
    capi.dev_find_similar_pairs4(d_sig.data_ptr(), n, begin, end, L, k, thr, d_pairs.data_ptr(), d_used.data_ptr(), ws.data_ptr(),
                                 ws_bytes, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    capi.dev_find_similar_pairs4_status(ws.data_ptr(), rows, k, torch.cuda.current_stream().cuda_stream)
    pairs = d_pairs.cpu().numpy().view(np.uint32)
    return pairs[:, :, 0], pairs[:, :, 1], d_used.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("n,L,k,thr,kind,knobs", [
    (3000, 1024, 20, 0.2, "clustered", dict()),
    (3001, 1024, 20, 0.2, "clustered", dict(EM2_MIN_SEGMENT_COLUMNS=256)),                    # a last, partial tile; many hand-offs
    (2309, 600, 7, 0.0, "clustered", dict(EM2_MIN_SEGMENT_COLUMNS=700)),                       # zero-extended fragments
    (1700, 1024, 5, -1.0, "random", dict(EM2_MIN_SEGMENT_COLUMNS=512)),                        # everything passes: the walks stop and resume
    (1700, 1024, 300, -0.5, "clustered", dict(EM2_MIN_SEGMENT_COLUMNS=256, EM2_LOG_CAPACITY=1)),
    (4000, 1024, 25, 0.5, "clustered", dict(EM2_MIN_SEGMENT_COLUMNS=1300, EM2_BLOCKS_PER_CU=1)),
    (2500, 2048, 10, 0.2, "clustered", dict(EM2_MIN_SEGMENT_COLUMNS=512)),                     # the 2048-bit walk
    (2307, 1500, 12, -0.5, "clustered", dict(EM2_MIN_SEGMENT_COLUMNS=256, EM2_LOG_CAPACITY=1)),
    (2500, 1024, 10, 0.2, "clustered", dict(EM2_MIN_SEGMENT_COLUMNS=512, EM2_MATRIX_CONVOY=3)),   # every walk goes around its segment
    (1700, 2048, 5, -1.0, "random", dict(EM2_MIN_SEGMENT_COLUMNS=512, EM2_MATRIX_CONVOY=2)),
])
def test_rows_form_on_the_matrix_cores(oracle, scan_knobs, n, L, k, thr, kind, knobs):
    """A shard of the rows against all columns -- what a rank of the multi-GPU rows form computes -- as FP4 +-1
    dot products: every row block is a full-row block of the matrix kernel.  Shards that begin at a multiple of 32 take their
    row fragments out of the columns' array, others get a copy; short last quads, single rows, the whole problem."""
    sig = make(n, L, kind)
    scan_knobs(EM2_SCAN_MODE="rows", **knobs)
    for begin, end in [(0, n), (0, 1000), (1000, 1001), (1001, n), (n - 1, n), (37, n - 100), (1024, 1024 + 256), (960, 960 + 300)]:
        cell, sim, used = oracle.find_similar_pairs4_rows(sig, L, k, thr, begin, end)
        gcell, gsim, gused = run_row_shard(sig, L, k, thr, begin, end)
        info = capi.dev_find_similar_pairs4_last_launch()
        assert info["form"] == 4 and info["matrix_pairs"] > 0, (begin, end)
        assert np.array_equal(gused, used), (begin, end)
        assert np.array_equal(gcell, cell), (begin, end)
        assert np.array_equal(gsim, sim.view(np.uint32)), (begin, end)


@pytest.mark.parametrize("k", [79, 80, 81, 85, 99, 101, 171])
def test_matrix_form_with_k_around_the_walk_blocks_place(oracle, scan_knobs, k):
    """The walk's per-wave LDS block lives inside the wave's selection area (2k entries) when that is large enough -- from k = 80
    on, 8 bytes in where an odd k puts the area at 8 (mod 16) -- and behind the tiles otherwise: every side of those borders."""
    sig = synth.clustered_signatures(3000, 1024, cluster_count=4, flip=0.2, seed=k)
    cell, sim, used = oracle.find_similar_pairs4(sig, 1024, k, 0.0)
    scan_knobs(EM2_SCAN_MODE="triangle", EM2_MIN_SEGMENT_COLUMNS=512, EM2_FULL_ROW_CELLS=256)
    pairs, gused = capi.find_similar_pairs4(sig, 1024, k, 0.0)
    assert capi.dev_find_similar_pairs4_last_launch()["form"] == 3
    assert_same(pairs, gused, cell, sim, used)


@pytest.mark.parametrize("n,L", [(3000, 1024), (2977, 1024), (2500, 2048), (3040, 600)])
def test_rows_form_shard_at_32_mod_64_that_ends_with_the_cells(oracle, scan_knobs, n, L):
    """a shard that begins at 32 (mod 64) and ends with the cells has a last 64-row block that reaches past the
    columns' fragment array; its fragments must not be addressed in place there.  The workspace is poisoned (0x77: FP4
    magnitudes above 1 in every nibble), so that a read past the array shows."""
    k, thr = 20, 0.2
    sig = make(n, L, "clustered")
    scan_knobs(EM2_SCAN_MODE="rows", EM2_MIN_SEGMENT_COLUMNS=512)
    for begin in (32, 1056, (n // 64) * 64 - 32, (n // 64) * 64 - 96):
        for end in (n, n - 1):
            cell, sim, used = oracle.find_similar_pairs4_rows(sig, L, k, thr, begin, end)
            for poison in (0x77, 0xff, 0x00):
                gcell, gsim, gused = run_row_shard(sig, L, k, thr, begin, end, poison=poison)
                assert capi.dev_find_similar_pairs4_last_launch()["form"] == 4
                assert np.array_equal(gused, used), (begin, end, poison)
                assert np.array_equal(gcell, cell), (begin, end, poison)
                assert np.array_equal(gsim, sim.view(np.uint32)), (begin, end, poison)


@pytest.mark.parametrize("n,L,k,thr,kind", CASES)
def test_rows_form_whole_problem_matches_oracle(oracle, scan_knobs, n, L, k, thr, kind):
    """EM2_SCAN_MODE=rows: every launch that has a matrix form takes the rows form (the others run as ever)."""
    sig = make(n, L, kind)
    cell, sim, used = oracle.find_similar_pairs4(sig, L, k, thr)
    scan_knobs(EM2_SCAN_MODE="rows")
    pairs, gused = capi.find_similar_pairs4(sig, L, k, thr)
    expected = 4 if (128 < L <= 2048 and n >= 64 and k <= 682) else 0
    assert capi.dev_find_similar_pairs4_last_launch()["form"] == expected
    assert_same(pairs, gused, cell, sim, used)


def test_rows_form_is_the_default_for_a_large_shard(oracle, scan_knobs):
    """Without any knob: 2^31 (row, column) pairs and more go to the matrix cores, smaller launches keep the ordered scan."""
    n, L, k, thr = 70000, 1024, 20, 0.2
    sig = synth.clustered_signatures(n, L, cluster_count=32, flip=0.15, seed=12)
    assert capi.dev_find_similar_pairs4_form_for(n, 35000, L) == 4
    assert capi.dev_find_similar_pairs4_form_for(n, 20000, L) == 0
    begin, end = 30001, 65001
    gcell, gsim, gused = run_row_shard(sig, L, k, thr, begin, end)
    assert capi.dev_find_similar_pairs4_last_launch()["form"] == 4
    for b in (begin, begin + 17003, end - 200):
        cell, sim, used = oracle.find_similar_pairs4_rows(sig, L, k, thr, b, b + 200)
        assert np.array_equal(gused[b - begin:b - begin + 200], used)
        assert np.array_equal(gcell[b - begin:b - begin + 200], cell)
        assert np.array_equal(gsim[b - begin:b - begin + 200], sim.view(np.uint32))


@pytest.mark.parametrize("world,n,permille", [(2, 6000, 200), (4, 9000, 300), (3, 7000, 250), (8, 9000, 100)])
def test_sharded_virtual_world_tiles_on_the_matrix_cores(oracle, scan_knobs, world, n, permille):
    """The tile phase of the sharded scan with 1024-bit signatures: prefixes of whole quads, fsp4TileMatrixKernel; the
    same run with EM2_SCAN_MATRIX=0 gives the same bytes."""
    sig = make(n, 1024, "clustered")
    cell, sim, used = oracle.find_similar_pairs4(sig, 1024, 20, 0.2)
    for matrix in (1, 0):
        scan_knobs(EM2_SCAN_MODE="virtual:%d" % world, EM2_PREFIX_PERMILLE=permille, EM2_TILE_SEGMENTS=3,
                   EM2_SCAN_MATRIX=matrix)
        pairs, gused = capi.find_similar_pairs4(sig, 1024, 20, 0.2)
        info = capi.dev_find_similar_pairs4_last_launch()
        assert info["form"] == 2 and (info["matrix_pairs"] > 0) == bool(matrix)
        assert_same(pairs, gused, cell, sim, used)


@pytest.mark.parametrize("world,n,L,permille", [(2, 6000, 2048, 200), (4, 9000, 1500, 300), (3, 5000, 2048, 250), (8, 9000, 1025, 100)])
def test_sharded_virtual_world_2048_bit_tiles_on_the_matrix_cores(oracle, scan_knobs, world, n, L, permille):
    """The same with 1025..2048-bit signatures: phases 0 / 1 on fsp4ScanMatrixWideKernel, the tiles on
    fsp4TileMatrixWideKernel; EM2_SCAN_MATRIX=3 (v_xor/v_bcnt everywhere at this width) gives the same bytes."""
    sig = make(n, L, "clustered")
    cell, sim, used = oracle.find_similar_pairs4(sig, L, 20, 0.2)
    for wide in (1, 0):
        scan_knobs(EM2_SCAN_MODE="virtual:%d" % world, EM2_PREFIX_PERMILLE=permille, EM2_TILE_SEGMENTS=3,
                   EM2_SCAN_MATRIX=1 if wide else 3)
        pairs, gused = capi.find_similar_pairs4(sig, L, 20, 0.2)
        info = capi.dev_find_similar_pairs4_last_launch()
        assert info["form"] == 2 and (info["matrix_pairs"] > 0) == bool(wide)
        assert_same(pairs, gused, cell, sim, used)


def test_sharded_tile_walk_repeats_bit_identically(scan_knobs):
    """The deferred square of the sharded scan on the matrix cores (hand-scheduled walk, both sides deferred) against
    the v_xor/v_bcnt form on a problem of a few thousand tiles, many times over: a tile piece that reaches LDS
    late or in the wrong place shows as a handful of mismatch counts that are off by one or two (found this way:
    global_load_lds with a scalar base and a 32-bit lane offset left a 1 KB piece of a tile stale in about every
    second run; the 64-bit lane address form does not)."""
    sig = synth.clustered_signatures(24000, 1024, cluster_count=12, flip=0.2, seed=77)
    scan_knobs(EM2_SCAN_MODE="virtual:4", EM2_SCAN_MATRIX=0)
    reference = capi.find_similar_pairs4(sig, 1024, 40, 0.2)
    scan_knobs(EM2_SCAN_MODE="virtual:4", EM2_SCAN_MATRIX=1)
    for attempt in range(25):
        again = capi.find_similar_pairs4(sig, 1024, 40, 0.2)
        assert np.array_equal(reference[1], again[1]), attempt
        assert np.array_equal(reference[0]["cell"], again[0]["cell"]), attempt
        assert np.array_equal(reference[0]["similarity"].view(np.uint32), again[0]["similarity"].view(np.uint32)), attempt


def test_threshold_sweep_evicts_cached_tables(oracle):
    """The per-device cache of lookup tables keeps 16 (lshCount, threshold) sets; a sweep over more thresholds frees the least
    recently used ones, and coming back to an evicted threshold rebuilds it."""
    sig = make(400, 128, "clustered")
    thresholds = [round(-0.9 + 0.09 * i, 3) for i in range(20)] + [-0.9, 0.0]
    for thr in thresholds:
        cell, sim, used = oracle.find_similar_pairs4(sig, 128, 5, thr)
        pairs, gused = capi.find_similar_pairs4(sig, 128, 5, thr)
        assert_same(pairs, gused, cell, sim, used)


@pytest.mark.parametrize("name,value", [("EM2_MATRIX_DIAG", v) for v in (1, 16, 32, 64, 128, 256, 512, 2048, 4096)] +
                         [("EM2_PROJECTION_DIAG", v) for v in (1, 2)])
def test_measurement_knobs_do_nothing_in_the_product_library(oracle, monkeypatch, name, value):
    """The knobs that switch parts of the kernels off exist in libem2lsh_diag.so only (csrc/Makefile, -DEM2_DIAG): with the
    product library every one of them leaves signatures and SimilarPairs bit-identical to the oracle -- in the matrix-core
    form of the scan (forced at this size), where EM2_MATRIX_DIAG used to act, and in the 16-bit projection tier."""
    cells, genes, L, k, thr = 3000, 800, 1024, 10, 0.2
    toc, g, c = synth.expression_matrix(cells, genes, density=0.03, cluster_count=6, seed=17)
    vectors = capi.lsh_generate_vectors(genes, L, 231)
    expected_sig = oracle.compute_signatures(toc, g, c, genes, vectors, L)
    cell, sim, used = oracle.find_similar_pairs4(expected_sig, L, k, thr)
    monkeypatch.setenv(name, str(value))
    monkeypatch.setenv("EM2_SCAN_MODE", "triangle")
    monkeypatch.setenv("EM2_FULL_ROW_CELLS", "256")
    sig = capi.compute_signatures(toc, capi.make_counts(g, c), genes, vectors, L)
    assert np.array_equal(sig, expected_sig)
    pairs, gused = capi.find_similar_pairs4(sig, L, k, thr)
    assert capi.dev_find_similar_pairs4_last_launch()["form"] == 3
    assert_same(pairs, gused, cell, sim, used)
