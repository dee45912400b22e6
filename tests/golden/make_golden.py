"""Generates the committed fixtures under tests/golden/ .  Run in the build container (needs /root/reference
for oracle/_ref):   python tests/golden/make_golden.py

What is recorded, and from where:
  reference_known_answers.json
      - testKeepBest: the inputs of the reference's print-only self test (src/heap.cpp:32-38) and what the
        reference's own keepBest (src/heap.hpp:116-126, compiled in place into oracle/_ref) returns for them;
      - multipleSetUnionTest: inputs and the documented output of src/multipleSetUnion.cpp:9-23;
      - MurmurHash64A(seed 231) of a few byte strings, computed by src/MurmurHash2.cpp compiled in place.
  keepbest_reference_header.npz
      - tie-heavy (cell, similarity) lists and the result of the reference's keepBest with
        OrderPairsBySecondGreater on them (the exact call of ExpressionMatrixLsh.cpp:247,254,267,457).
  oracle_regression.json
      - SHA-256 of the oracle's outputs on seeded inputs.  These are NOT reference outputs (the reference's
        driver cannot be built here); they pin the oracle against accidental change.
Fixtures are data only; no reference source text is stored.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding  # noqa: E402
import synth  # noqa: E402
from test_oracle_components import tie_heavy_pairs  # noqa: E402


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def regression_cases():
    return [
        dict(name="clustered_300_L128_k5_thr0.2", n=300, L=128, k=5, thr=0.2, kind="clustered"),
        dict(name="clustered_300_L1024_k100_thr0.2", n=300, L=1024, k=100, thr=0.2, kind="clustered"),
        dict(name="clustered_700_L1024_k10_thr-0.5", n=700, L=1024, k=10, thr=-0.5, kind="clustered"),
        dict(name="random_500_L256_k7_thr0.0", n=500, L=256, k=7, thr=0.0, kind="random"),
        dict(name="clustered_600_L2048_k20_thr0.2", n=600, L=2048, k=20, thr=0.2, kind="clustered"),
        dict(name="clustered_3000_L1024_k100_thr0.2", n=3000, L=1024, k=100, thr=0.2, kind="clustered"),
    ]


def bucketed_cases():
    """(name, signatures, arguments) of the fsp5 / fsp7 / cell-graph regression digests."""
    sig = synth.clustered_signatures(900, 512, cluster_count=6, flip=0.12, seed=99)
    return sig, {
        "fsp5_900_L512_k10_q14_overflow1000": dict(k=10, thr=0.2, q=14, overflow=1000),
        "fsp5_900_L512_k25_q9_overflow0": dict(k=25, thr=0.0, q=9, overflow=0),
        "fsp7_900_L512_k10_lengths24_12_check60_log2b16": dict(k=10, thr=0.2, lengths=[24, 12], max_check=60, log2b=16),
        "fsp7_900_L512_k30_lengths64_7_check0_log2b5": dict(k=30, thr=0.1, lengths=[64, 7], max_check=0, log2b=5),
    }


def bucketed_digests(oracle):
    sig, cases = bucketed_cases()
    out = {}
    for name, a in cases.items():
        if name.startswith("fsp5"):
            out[name] = digest(*oracle.find_similar_pairs5(sig, 512, a["k"], a["thr"], a["q"], a["overflow"]))
        else:
            out[name] = digest(*oracle.find_similar_pairs7(sig, 512, a["k"], a["thr"], a["lengths"], a["max_check"], a["log2b"]))
    cell, sim, used = oracle.find_similar_pairs4(sig, 512, 20, 0.2)
    ids = np.arange(900, dtype=np.uint32)
    out["cellgraph_900_thr0.5_k5"] = digest(*oracle.cell_graph_edges(cell, sim, used, ids, ids, 0.5, 5))
    return out


def info_cases():
    """(k, geneSetName, geneSetHash, cellSetName, cellSetHash) of the recorded SimilarPairs::Info objects."""
    return [(100, "AllGenes", 0x0123456789abcdef, "AllCells", 0xfedcba9876543210),
            (1, "", 0, "", 0),
            (3, "HighInformationGenes", 2**64 - 1, "c", 1),
            (2**40 + 7, "g" * 230, 5, "x" * 240, 6),          # (file names end at 255 bytes: GeneSet-<name>-GlobalIds)
            (9, "g" * 255, 7, "x" * 254, 8)]


def make_signatures(case):
    if case["kind"] == "clustered":
        return synth.clustered_signatures(case["n"], case["L"], cluster_count=8, flip=0.15, seed=4242)
    return synth.random_signatures(case["n"], case["L"], seed=4242)


def main():
    oracle = oracle_binding.load_oracle()
    ref = oracle_binding.load_ref()
    assert ref is not None, "needs /root/reference to build oracle/_ref"

    values = [35, 9, 14, 39, 17, 10, 18, 28, 19, 36, 7, 43, 16]
    known = {
        "testKeepBest": {"input": values, "k": 6,
                         "kept_in_order": ref.keep_best_int_greater(values, 6).tolist()},
        "multipleSetUnionTest": {"input": [[3, 7, 10], [2, 7, 25], [7, 10], [3, 8, 25, 40]],
                                 "output": [2, 3, 7, 8, 10, 25, 40]},
        "murmur64a_seed231": [],
    }
    for n in [0, 1, 5, 8, 13, 64, 257]:
        data = (synth.hash_u64(5, np.arange(n, dtype=np.uint64)) & np.uint64(0xFF)).astype(np.uint8)
        known["murmur64a_seed231"].append({"bytes": data.tolist(), "hash": str(ref.murmur(data))})
    # SimilarPairs::Info (src/SimilarPairs.hpp:190-203) as the reference's own StaticString255 lays it out
    # (src/ShortStaticString.hpp compiled in place, oracle/ref_layout.cpp): size, member offsets, and the bytes of a few
    # filled-in objects, hex
    layout = oracle_binding.load_ref_layout()
    assert layout is not None, "needs /root/reference to build oracle/_ref"
    known["similarPairsInfo"] = {"size": layout.info_size(), "offsets": layout.info_offsets(), "objects": []}
    for k, gene_set, gene_hash, cell_set, cell_hash in info_cases():
        known["similarPairsInfo"]["objects"].append({
            "k": k, "geneSetName": gene_set, "geneSetHash": str(gene_hash), "cellSetName": cell_set, "cellSetHash": str(cell_hash),
            "bytes": layout.make_info(k, gene_set, gene_hash, cell_set, cell_hash).hex()})
    with open(os.path.join(HERE, "reference_known_answers.json"), "w") as f:
        json.dump(known, f, indent=1)

    arrays = {}
    i = 0
    for (n, k, distinct) in [(200, 100, 5), (200, 100, 30), (200, 100, 300), (40, 20, 4), (199, 100, 12),
                             (101, 100, 3), (1000, 100, 60), (16, 3, 2)]:
        for seed in range(4):
            cell, sim = tie_heavy_pairs(n, distinct, 1000 + seed)
            oc, osim = ref.keep_best(cell, sim, k)
            arrays["in_cell_%d" % i] = cell
            arrays["in_sim_%d" % i] = sim
            arrays["k_%d" % i] = np.int64(k)
            arrays["out_cell_%d" % i] = oc
            arrays["out_sim_%d" % i] = osim
            i += 1
    arrays["case_count"] = np.int64(i)
    np.savez_compressed(os.path.join(HERE, "keepbest_reference_header.npz"), **arrays)

    regression = {}
    for case in regression_cases():
        sig = make_signatures(case)
        cell, sim, used = oracle.find_similar_pairs4(sig, case["L"], case["k"], case["thr"])
        regression[case["name"]] = {"signatures": digest(sig), "fsp4": digest(cell, sim, used),
                                    "used_sum": int(used.sum())}
    toc, genes, counts = synth.expression_matrix(200, 300, density=0.05, cluster_count=4, seed=77)
    vectors = oracle.generate_lsh_vectors(300, 256, 231)
    sig = oracle.compute_signatures(toc, genes, counts, 300, vectors, 256)
    regression["projection_200x300_L256_seed231"] = {"vectors": digest(vectors), "signatures": digest(sig)}
    for L in (128, 1024, 2048):
        t = oracle.similarity_table(L)
        regression["similarity_table_%d" % L] = {"double_bits": digest(t), "float_bits": digest(t.astype(np.float32))}
    regression.update(bucketed_digests(oracle))
    with open(os.path.join(HERE, "oracle_regression.json"), "w") as f:
        json.dump(regression, f, indent=1)
    print("wrote fixtures:", i, "keepBest cases,", len(regression), "regression digests")


if __name__ == "__main__":
    main()
