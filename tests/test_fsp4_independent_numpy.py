"""A second, independent restatement of findSimilarPairs4's per-cell contract (src/ExpressionMatrixLsh.cpp:200-285) --
plain Python / numpy, sharing no code with oracle/em2_oracle.cpp -- whose selection step is the REFERENCE'S OWN keepBest
(src/heap.hpp:116-126 compiled in place, oracle/_ref/libem2ref.so: em2ref_keep_best), and whose final order is the
reference's own comparator (src/orderPairs.hpp:44-52: em2ref_sort_pairs).

The oracle as a whole cannot be checked against a build of the reference (Boost is absent from this image); this test
narrows that: the oracle's pair loop, its acceptance tests (double similarity against the double threshold and against the
float cut-off, :244-252), its flush rule (:247-250) and its final cut (:265-269) must agree, on every golden case and on
the parametrised cases of test_fsp4_cpu.py, with a loop written from the reference text whose every nth_element is the
reference's.  Runs where /root/reference exists (the fixtures it guards are committed)."""
import json
import math
import os

import numpy as np
import pytest

from test_fsp4_cpu import CASES, make

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def similarity_table(lsh_count):
    # src/Lsh.cpp:229-249: cos(double(m) * pi / double(lshCount)) with the C library's cos (math.cos calls it)
    return [math.cos(float(m) * math.pi / float(lsh_count)) for m in range(lsh_count + 1)]


def find_similar_pairs4(sig, lsh_count, k, threshold, ref):
    n = sig.shape[0]
    table = similarity_table(lsh_count)
    out_cell = np.zeros((n, k), dtype=np.uint32)
    out_sim = np.zeros((n, k), dtype=np.float32)
    out_used = np.zeros(n, dtype=np.uint32)
    for c in range(n):
        mismatches = np.bitwise_count(sig ^ sig[c]).sum(axis=1)             # countMismatches, src/BitSet.hpp:277-288
        cells, sims = [], []
        cell_threshold = np.float32(threshold)                              # :207
        for o in range(n):
            if o == c:
                continue
            similarity = table[int(mismatches[o])]                          # double
            if similarity > threshold and similarity > float(cell_threshold):  # :244-245, :252
                cells.append(o)
                sims.append(np.float32(similarity))
                if len(cells) == 2 * k:                                     # :247-250, :253-257
                    kept_cells, kept_sims = ref.keep_best(cells, sims, k)
                    cells, sims = kept_cells.tolist(), [np.float32(x) for x in kept_sims]
                    cell_threshold = sims[-1]
        if len(cells) > k:                                                  # :265-269
            kept_cells, kept_sims = ref.keep_best(cells, sims, k)
            cells, sims = kept_cells.tolist(), kept_sims.tolist()
        if cells:                                                           # SimilarPairs::copy + sort, src/SimilarPairs.cpp:369-405
            sorted_cells, sorted_sims = ref.sort_pairs(cells, sims)
            out_cell[c, :len(cells)] = sorted_cells
            out_sim[c, :len(cells)] = sorted_sims
        out_used[c] = len(cells)
    return out_cell, out_sim, out_used


@pytest.mark.parametrize("n,L,k,thr,kind", CASES)
def test_oracle_equals_independent_restatement(oracle, reflib, n, L, k, thr, kind):
    sig = make(n, L, kind)
    expect = find_similar_pairs4(sig, L, k, thr, reflib)
    got = oracle.find_similar_pairs4(sig, L, k, thr)
    for x, y in zip(expect, got):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))


def test_golden_digests_equal_independent_restatement(reflib):
    """The committed fsp4 digests (tests/golden/oracle_regression.json: what bench.py and the GPU tests compare the
    kernels with) reproduced without the oracle."""
    from golden.make_golden import digest, make_signatures, regression_cases
    with open(os.path.join(GOLDEN, "oracle_regression.json")) as f:
        golden = json.load(f)
    for case in regression_cases():
        cell, sim, used = find_similar_pairs4(make_signatures(case), case["L"], case["k"], case["thr"], reflib)
        assert digest(cell, sim, used) == golden[case["name"]]["fsp4"], case["name"]
