"""A second, independent restatement of findSimilarPairs7 (src/ExpressionMatrixLsh.cpp:507-827) in plain Python -- sharing no
code with oracle/em2_oracle.cpp -- whose bucket ids are hashed by the REFERENCE'S OWN MurmurHash64A (src/MurmurHash2.cpp:96-137
compiled in place, oracle/_ref/libem2ref.so: em2ref_murmur_hash_64a).  Its selection needs no reference code: keepBest runs with
std::less on (mismatch, cell) pairs, a total order, so the k kept and their order after the sort are the k smallest pairs.

What this cross-checks in the oracle: slice values (getBits: first bit most significant), bucket id = the slice value when the
slice is shorter than log2BucketCount, else MurmurHash64A(value as 8 little-endian bytes, seed 231) & mask (:645-649), tables in
ascending cell id for every slice length and slice (findSimilarPairs7AssignCellsToBuckets, :721-827), the walk over lengths ->
slices -> bucket members with the visited bitmap and the maxCheck stop that is tested only after a candidate was added and after
each bucket / slice / length (:650-675), `mismatch < threshold` with threshold = (first count whose similarity is below the
threshold) - 1 (src/Lsh.hpp:86-95), and float(similarityTable[mismatch]) stored in (mismatch, cell) order.  Runs where
/root/reference exists."""
import math

import numpy as np
import pytest

import synth


def similarity_table(lsh_count):
    return [math.cos(float(m) * math.pi / float(lsh_count)) for m in range(lsh_count + 1)]      # src/Lsh.cpp:229-249


def find_similar_pairs7(sig, lsh_count, k, threshold, slice_lengths, max_check, log2_bucket_count, ref):
    n = sig.shape[0]
    table = similarity_table(lsh_count)
    mismatch_threshold = next(m for m in range(lsh_count + 1) if table[m] < threshold) - 1       # Lsh.hpp:86-95
    bits = np.unpackbits(sig.astype(">u8").view(np.uint8).reshape(n, -1), axis=1)[:, :lsh_count]
    bucket_mask = (1 << log2_bucket_count) - 1

    def bucket_id(cell, length, slice_id):
        value = 0
        for b in bits[cell, slice_id * length:(slice_id + 1) * length]:                            # getBits: last bit least significant
            value = (value << 1) + int(b)
        if length < log2_bucket_count:                                                            # :645-649
            return value
        return ref.murmur(np.array([value], dtype="<u8")) & bucket_mask

    # findSimilarPairs7AssignCellsToBuckets: tables[lengthId][sliceId][bucketId] = cells in ascending id
    ids = [[[bucket_id(c, length, s) for s in range(lsh_count // length)] for length in slice_lengths] for c in range(n)]
    tables = [[dict() for _ in range(lsh_count // length)] for length in slice_lengths]
    for c in range(n):
        for li, length in enumerate(slice_lengths):
            for s in range(lsh_count // length):
                tables[li][s].setdefault(ids[c][li][s], []).append(c)
    out_cell = np.zeros((n, k), dtype=np.uint32)
    out_sim = np.zeros((n, k), dtype=np.float32)
    out_used = np.zeros(n, dtype=np.uint32)
    for c in range(n):
        seen, candidates, neighbors = set(), 0, []
        stop = False
        for li, length in enumerate(slice_lengths):
            for s in range(lsh_count // length):
                for o in tables[li][s][ids[c][li][s]]:
                    if o == c or o in seen:
                        continue
                    seen.add(o)
                    candidates += 1
                    mismatch = int(np.bitwise_count(sig[c] ^ sig[o]).sum())
                    if mismatch < mismatch_threshold:
                        neighbors.append((mismatch, o))
                    if candidates == max_check:                                                    # :659-661
                        break
                if candidates == max_check:                                                        # :663-665 (true for 0 == 0 as well)
                    stop = True
                    break
            if stop:
                break
        neighbors = sorted(neighbors)[:k]                                                         # keepBest(std::less) + sort: a total order
        for j, (mismatch, o) in enumerate(neighbors):                                             # addUnsymmetricNoCheck, :680-686
            out_cell[c, j] = o
            out_sim[c, j] = np.float32(table[mismatch])
        out_used[c] = len(neighbors)
    return out_cell, out_sim, out_used


@pytest.mark.parametrize("n,L,k,thr,lengths,max_check,log2b,clusters,flip", [
    (300, 128, 6, 0.2, [16, 8], 10 ** 6, 10, 3, 0.1),        # 16-bit slices hashed (16 >= 10), 8-bit slices direct
    (300, 128, 6, 0.2, [16, 8], 7, 10, 3, 0.1),              # the candidate limit cuts the walk
    (200, 192, 10, 0.0, [33, 13, 5], 50, 12, 4, 0.15),       # slices that straddle words; three lengths; 192 % 33 != 0
    (150, 64, 4, 0.5, [64], 10 ** 6, 20, 2, 0.02),           # one slice = the whole signature, hashed
    (120, 100, 3, 0.3, [7, 3], 0, 4, 2, 0.05),               # maxCheck 0: stops after the first bucket that adds nobody
    (257, 256, 100, -0.5, [8], 10 ** 6, 8, 1, 0.3),          # 8 >= 8: hashed; k above what most cells find
])
def test_oracle_equals_independent_restatement(oracle, reflib, n, L, k, thr, lengths, max_check, log2b, clusters, flip):
    sig = synth.clustered_signatures(n, L, cluster_count=clusters, flip=flip, seed=n + L)
    expect = find_similar_pairs7(sig, L, k, thr, lengths, max_check, log2b, reflib)
    got = oracle.find_similar_pairs7(sig, L, k, thr, lengths, max_check, log2b)
    for x, y in zip(expect, got):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
