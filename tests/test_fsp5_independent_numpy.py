"""A second, independent restatement of findSimilarPairs5 (src/ExpressionMatrixLsh.cpp:312-501) -- plain Python / numpy,
sharing no code with oracle/em2_oracle.cpp -- whose selection step is the REFERENCE'S OWN keepBest (src/heap.hpp:116-126
compiled in place, oracle/_ref/libem2ref.so: em2ref_keep_best) and whose final order is the reference's own comparator
(src/orderPairs.hpp:44-52: em2ref_sort_pairs).

The oracle as a whole cannot be checked against a build of the reference (Boost is absent from this image).  This test narrows
that for the bucketed path the way test_fsp4_independent_numpy.py does for the all-pairs one: the oracle's slice values
(BitSetPointer::getBits, src/BitSet.hpp:111-119: first bit of a slice most significant; bit i of a signature in word i >> 6 at
position 63 - (i & 63), :48-62), its tables in ascending cell id (:377-389), the overflow rule (:419), the ascending,
duplicate-free union (src/multipleSetUnion.hpp:44-76), the acceptance test in ascending candidate order (:436-445: double
similarity against the double threshold, stored as float), the single keepBest (:457) and the final copy + sort (:489-496) must
agree with a loop written from the reference text whose nth_element is the reference's.  Runs where /root/reference exists."""
import math

import numpy as np
import pytest

import synth


def similarity_table(lsh_count):
    # src/Lsh.cpp:229-249: cos(double(m) * pi / double(lshCount)) with the C library's cos (math.cos calls it)
    return [math.cos(float(m) * math.pi / float(lsh_count)) for m in range(lsh_count + 1)]


def signature_bits(sig, lsh_count):
    # bit i of a cell: word i >> 6, position 63 - (i & 63) (src/BitSet.hpp:48-62): the big-endian bytes of the words, bit by bit
    n = sig.shape[0]
    return np.unpackbits(sig.astype(">u8").view(np.uint8).reshape(n, -1), axis=1)[:, :lsh_count]


def find_similar_pairs5(sig, lsh_count, k, threshold, slice_length, bucket_overflow, ref):
    n = sig.shape[0]
    table = similarity_table(lsh_count)
    bits = signature_bits(sig, lsh_count)
    slice_count = lsh_count // slice_length                                   # :355
    weights = [1 << (slice_length - 1 - j) for j in range(slice_length)]      # getBits: the last bit least significant
    # tables[sliceId][sliceValue] = cell ids in ascending order (:377-389)
    values = [[int(sum(int(b) * w for b, w in zip(bits[c, s * slice_length:(s + 1) * slice_length], weights))) for s in range(slice_count)]
              for c in range(n)]
    tables = [dict() for _ in range(slice_count)]
    for c in range(n):
        for s in range(slice_count):
            tables[s].setdefault(values[c][s], []).append(c)
    out_cell = np.zeros((n, k), dtype=np.uint32)
    out_sim = np.zeros((n, k), dtype=np.float32)
    out_used = np.zeros(n, dtype=np.uint32)
    for c in range(n):
        union = set()
        for s in range(slice_count):
            bucket = tables[s][values[c][s]]
            if bucket_overflow == 0 or len(bucket) <= bucket_overflow:        # :419
                union.update(bucket)
        cells, sims = [], []
        for o in sorted(union):                                               # multipleSetUnion: ascending, duplicate-free
            if o == c:                                                        # :437-439
                continue
            mismatches = int(np.bitwise_count(sig[c] ^ sig[o]).sum())         # countMismatches, src/BitSet.hpp:277-288
            similarity = table[mismatches]                                    # double
            if similarity > threshold:                                        # :441
                cells.append(o)
                sims.append(np.float32(similarity))
        if len(cells) > k:                                                    # keepBest, :457
            kept_cells, kept_sims = ref.keep_best(cells, sims, k)
            cells, sims = kept_cells.tolist(), kept_sims.tolist()
        if cells:                                                             # SimilarPairs::copy + sort (:489-496)
            sorted_cells, sorted_sims = ref.sort_pairs(cells, sims)
            out_cell[c, :len(cells)] = sorted_cells
            out_sim[c, :len(cells)] = sorted_sims
        out_used[c] = len(cells)
    return out_cell, out_sim, out_used


@pytest.mark.parametrize("n,L,k,thr,q,overflow,clusters,flip", [
    (300, 128, 5, 0.2, 8, 0, 4, 0.1),          # plain
    (300, 128, 100, -0.5, 4, 0, 4, 0.1),        # short slices: large buckets, long candidate lists, many ties at the cut
    (257, 192, 7, 0.0, 13, 25, 3, 0.15),        # slices that straddle words (13 does not divide 64), the overflow rule
    (120, 100, 3, 0.3, 7, 0, 2, 0.05),          # lshCount no multiple of 64 nor of the slice length: the remainder bits are unused
    (64, 64, 10, -1.0, 1, 0, 1, 0.5),           # one-bit slices: every cell shares a bucket with everybody
    (200, 256, 4, 0.2, 20, 5, 5, 0.02),         # 20-bit slices as in BASELINE configs[3], tiny overflow limit
])
def test_oracle_equals_independent_restatement(oracle, reflib, n, L, k, thr, q, overflow, clusters, flip):
    sig = synth.clustered_signatures(n, L, cluster_count=clusters, flip=flip, seed=n + L)
    expect = find_similar_pairs5(sig, L, k, thr, q, overflow, reflib)
    got = oracle.find_similar_pairs5(sig, L, k, thr, q, overflow)
    for x, y in zip(expect, got):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
