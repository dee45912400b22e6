"""Oracle restatement of findSimilarPairs7 (src/ExpressionMatrixLsh.cpp:507-827): properties that follow from the
reference's text.  The reference holds no test or golden output for this function (it calls it prototype code), so
the GPU path is pinned to this restatement only (tests/test_gpu_fsp7.py)."""
import numpy as np
import pytest

import synth


def test_unlimited_check_with_one_bit_slices_sees_every_cell(oracle):
    """Slices of 1 bit: any two cells share a bucket in some slice unless they are exact complements, so with no
    check limit every cell is a candidate and the result is the k best by (mismatch, id) below the threshold --
    i.e. findSimilarPairs4's set when nothing ties at the cut (fsp4 breaks ties differently)."""
    L, k = 128, 6
    sig = synth.clustered_signatures(300, L, cluster_count=3, flip=0.1, seed=11)
    cell, sim, used = oracle.find_similar_pairs7(sig, L, k, 0.2, [1], 10**6, 10)
    m = oracle.mismatch_matrix(sig, L).astype(np.int64)
    table = oracle.similarity_table(L)
    threshold = next(i for i in range(L + 1) if table[i] < 0.2) - 1
    for c in range(0, 300, 17):
        keys = sorted((int(m[c, o]), o) for o in range(300) if o != c and m[c, o] < threshold)[:k]
        assert used[c] == len(keys)
        assert [o for _, o in keys] == cell[c, :used[c]].tolist()
        assert np.array_equal(sim[c, :used[c]], np.array([table[mm] for mm, _ in keys], dtype=np.float32))


def test_max_check_cuts_the_candidate_sequence(oracle):
    L = 256
    sig = synth.clustered_signatures(400, L, cluster_count=2, flip=0.05, seed=5)
    full = oracle.find_similar_pairs7(sig, L, 10, 0.2, [16, 8], 10**6, 12)
    few = oracle.find_similar_pairs7(sig, L, 10, 0.2, [16, 8], 3, 12)
    assert (few[2] <= 3).all() and (few[2] <= full[2]).all()
    assert (full[2] == 10).any()
    # maxCheck == 0 is not "no limit" in the reference: the size test after a bucket (:663) holds while the candidate
    # list is empty, so a cell whose FIRST bucket holds nobody else ends with no neighbours at all
    zero = oracle.find_similar_pairs7(sig, L, 10, 0.2, [16, 8], 0, 12)
    assert (zero[2] == 0).any() and (zero[2] <= full[2]).all()


def test_argument_errors(oracle):
    sig = synth.random_signatures(10, 64)
    with pytest.raises(ValueError):
        oracle.find_similar_pairs7(sig, 64, 3, 0.2, [8, 8], 5, 10)       # not decreasing
    with pytest.raises(ValueError):
        oracle.find_similar_pairs7(sig, 64, 3, 0.2, [65], 5, 10)         # above 64
    with pytest.raises(ValueError):
        oracle.find_similar_pairs7(sig, 64, 3, -1.5, [8], 5, 10)         # no mismatch count is below the threshold
