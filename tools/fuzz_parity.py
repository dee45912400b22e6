#!/usr/bin/env python3
"""Randomised parity sweep on a GPU box (not part of pytest): random shapes, thresholds, k and scan-form knobs, every
result compared bit for bit with the CPU oracle.  SECONDS=300 python3 tools/fuzz_parity.py [seed]
FUZZ_ONLY=fsp4 restricts the sweep to one path, FUZZ_WIDTHS=1100,1500,2048 to those signature widths, FUZZ_MODE=triangle
to one scan form with the matrix cores on."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding  # noqa: E402
import synth  # noqa: E402
from label_graphs import fast_graph  # noqa: E402
from expressionmatrix2_amd import capi  # noqa: E402

KNOBS = ("EM2_SCAN_MODE", "EM2_MIN_SEGMENT_COLUMNS", "EM2_LOG_CAPACITY", "EM2_FULL_ROW_CELLS",
         "EM2_PREFIX_PERMILLE", "EM2_TILE_SEGMENTS", "EM2_BLOCKS_PER_CU", "EM2_SCAN_MATRIX", "EM2_MATRIX_CONVOY")


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    rng = np.random.default_rng(seed)
    oracle = oracle_binding.load_oracle()
    deadline = time.time() + float(os.environ.get("SECONDS", "120"))
    runs = {"fsp4": 0, "fsp5": 0, "fsp7": 0, "signatures": 0, "graph": 0, "labels": 0}
    while time.time() < deadline:
        for key in KNOBS:
            os.environ.pop(key, None)
        if os.environ.get("FUZZ_ONLY") == "labels" or (rng.random() < 0.15 and not os.environ.get("FUZZ_ONLY")):
            # label propagation over a random k-NN-like graph: every schedule against the serial oracle
            vertices = int(rng.choice([2, 3, 50, 64, 65, 1000, 5000, 30000, 70000]))
            degree = int(rng.choice([1, 2, 5, 12, 30, 50]))
            hubs = int(rng.choice([0, 0, 1, 4, 30]))
            case = dict(vertices=vertices, degree=degree, clusters=int(rng.choice([1, 3, 10, 40])), hubs=min(hubs, vertices),
                        hub_degree=int(rng.choice([70, 150, 300, 500, 700, 3000])), parallel=int(rng.choice([0, 0, 5, 200])),
                        ties=int(rng.choice([0, 0, 2, 16])), graph_seed=int(rng.integers(1 << 30)),
                        seed=int(rng.choice([231, 1, 2 ** 33 + 7])), stable=int(rng.choice([0, 1, 3])),
                        max_iterations=int(rng.choice([0, 1, 3, 100])), ticket=str(rng.choice(["", "", "1", "3", "8"])),
                        pool=str(rng.choice(["", "", "0", "1", "3"])))
            cells, v0, v1, s = fast_graph(np.random.default_rng(case["graph_seed"]), vertices, degree, case["clusters"],
                                          case["hubs"], case["hub_degree"], case["parallel"] if vertices > 3 else 0, case["ties"])
            for name, key in (("EM2_LABEL_TICKET_BATCH", "ticket"), ("EM2_LABEL_POOL_AREAS", "pool")):
                if case[key]:
                    os.environ[name] = case[key]
            got = capi.cell_graph_label_propagation(cells, v0, v1, s, case["seed"], case["stable"], case["max_iterations"])
            for name in ("EM2_LABEL_TICKET_BATCH", "EM2_LABEL_POOL_AREAS"):
                os.environ.pop(name, None)
            expect = oracle.label_propagation(cells, v0, v1, s, case["seed"], case["stable"], case["max_iterations"])
            if got[1] != expect[1] or not np.array_equal(got[0], expect[0]):
                raise SystemExit("PARITY FAILURE labels %r" % case)
            runs["labels"] += 1
            continue
        n = int(rng.choice([1, 2, 63, 64, 65, 200, 500, 1000, 1500, 2500, 4000]))
        L = int(rng.choice([1, 32, 64, 100, 128, 192, 256, 512, 600, 1000, 1024, 1024, 1024, 1100, 2000, 2048, 2048, 4096]))
        if os.environ.get("FUZZ_WIDTHS"):
            L = int(rng.choice([int(x) for x in os.environ["FUZZ_WIDTHS"].split(",")]))
        k = int(rng.choice([1, 2, 5, 10, 33, 100, 300]))
        thr = float(rng.choice([-1.0, -0.5, 0.0, 0.1, 0.2, 0.5, 0.9]))
        clusters = int(rng.choice([1, 2, 5, 20]))
        flip = float(rng.choice([0.0, 0.02, 0.1, 0.3, 0.5]))
        sig_seed = int(rng.integers(1 << 30))
        sig = synth.clustered_signatures(n, L, cluster_count=clusters, flip=flip, seed=sig_seed)
        what = rng.choice(["fsp4", "fsp4", "fsp4", "fsp5", "fsp7", "signatures", "graph"])
        what = os.environ.get("FUZZ_ONLY", what)
        label = dict(n=n, L=L, k=k, thr=thr, clusters=clusters, flip=flip, sig_seed=sig_seed)
        if what == "fsp4":
            knobs = {"EM2_SCAN_MODE": str(rng.choice(["persistent", "triangle", "virtual:%d" % int(rng.choice([1, 2, 3, 4, 8])), "simple", "rows"])),
                     "EM2_MIN_SEGMENT_COLUMNS": str(int(rng.choice([64, 100, 257, 1000, 4096]))),
                     "EM2_LOG_CAPACITY": str(int(rng.choice([1, 3, 16, 256]))),
                     "EM2_FULL_ROW_CELLS": str(int(rng.choice([0, 64, 200, 1000, 100000]))),
                     "EM2_PREFIX_PERMILLE": str(int(rng.choice([50, 200, 500, 900]))),
                     "EM2_TILE_SEGMENTS": str(int(rng.choice([1, 3, 17, 256]))),
                     "EM2_BLOCKS_PER_CU": str(int(rng.choice([1, 2, 4]))),
                     "EM2_SCAN_MATRIX": str(int(rng.choice([0, 1, 1, 2, 3]))),
                     # (the convoy of the matrix walks: off, following the other blocks, or every walk n - 1 pairs of tiles in)
                     "EM2_MATRIX_CONVOY": str(int(rng.choice([0, 1, 1, 2, 3, 5, 9, 30])))}
            if os.environ.get("FUZZ_MODE"):
                knobs["EM2_SCAN_MODE"] = os.environ["FUZZ_MODE"] if os.environ["FUZZ_MODE"] != "virtual" else knobs["EM2_SCAN_MODE"] if knobs["EM2_SCAN_MODE"].startswith("virtual") else "virtual:2"
                knobs["EM2_SCAN_MATRIX"] = "1"
            os.environ.update(knobs)
            label.update(knobs)
            cell, sim, used = oracle.find_similar_pairs4(sig, L, k, thr)
            pairs, gused = capi.find_similar_pairs4(sig, L, k, thr)
        elif what == "fsp5":
            q = int(rng.choice([1, 3, 8, 13, 20]))
            overflow = int(rng.choice([0, 5, 1000]))
            knobs = {"EM2_FSP5_BATCH_LOG2": str(int(rng.choice([20, 22, 29]))), "EM2_SCRATCH_CACHE_MB": str(int(rng.choice([0, 1, 4096])))}
            os.environ.update(knobs)
            label.update(q=q, overflow=overflow, **knobs)
            cell, sim, used = oracle.find_similar_pairs5(sig, L, k, thr, q, overflow)
            pairs, gused = capi.find_similar_pairs5(sig, L, k, thr, q, overflow)
            for key in knobs:
                os.environ.pop(key, None)
        elif what == "fsp7":
            lengths = sorted(set(int(x) for x in rng.choice([1, 2, 5, 8, 13, 16, 24, 33, 64], size=int(rng.integers(1, 4)))), reverse=True)
            max_check = int(rng.choice([0, 1, 7, 100, 100000]))
            log2b = int(rng.choice([4, 10, 16, 24]))
            if thr <= -1.0:
                thr = -0.9            # the reference asserts when no mismatch count is below the threshold
            label.update(lengths=lengths, max_check=max_check, log2b=log2b, thr=thr)
            cell, sim, used = oracle.find_similar_pairs7(sig, L, k, thr, lengths, max_check, log2b)
            pairs, gused = capi.find_similar_pairs7(sig, L, k, thr, lengths, max_check, log2b)
        elif what == "graph":
            cell, sim, used = oracle.find_similar_pairs4(sig, L, k, min(thr, 0.2))
            total = n + int(rng.integers(0, n + 1))
            sp_cells = np.sort(rng.choice(total, n, replace=False)).astype(np.uint32)
            graph_cells = rng.permutation(total)[:int(rng.integers(1, total + 1))].astype(np.uint32)
            g_thr = float(rng.choice([-1.0, 0.0, 0.3, 0.5, 0.9]))
            max_conn = int(rng.choice([0, 1, 3, 20, 1000]))
            expect = oracle.cell_graph_edges(cell, sim, used, sp_cells, graph_cells, g_thr, max_conn)
            pairs = np.zeros(cell.shape, dtype=capi.PAIR_DTYPE)
            pairs["cell"] = cell
            pairs["similarity"] = sim
            got = capi.cell_graph_edges(pairs, used, sp_cells, graph_cells, g_thr, max_conn)
            if not all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(expect, got)):
                raise SystemExit("PARITY FAILURE graph %r" % dict(label, g_thr=g_thr, max_conn=max_conn))
            runs[what] += 1
            continue
        else:
            cells, genes = int(rng.choice([1, 50, 300, 1000])), int(rng.choice([1, 10, 200, 1500]))
            toc, g, c = synth.expression_matrix(cells, genes, density=float(rng.choice([0.0, 0.01, 0.2])), cluster_count=3,
                                                seed=int(rng.integers(1 << 30)))
            vectors = oracle.generate_lsh_vectors(genes, L, int(rng.integers(1 << 20)))
            expect = oracle.compute_signatures(toc, g, c, genes, vectors, L)
            got = capi.compute_signatures(toc, capi.make_counts(g, c), genes, vectors, L)
            if not np.array_equal(expect, got):
                raise SystemExit("PARITY FAILURE signatures %r" % dict(cells=cells, genes=genes, L=L))
            runs[what] += 1
            continue
        ok = (np.array_equal(gused, used) and np.array_equal(pairs["cell"], cell) and
              np.array_equal(pairs["similarity"].view(np.uint32), sim.view(np.uint32)))
        if not ok:
            # which rows, and where in them (the first few)
            rows = np.nonzero((gused != used) | (pairs["cell"] != cell).any(axis=1) |
                              (pairs["similarity"].view(np.uint32) != sim.view(np.uint32)).any(axis=1))[0]
            print("differing rows: %d of %d, first %s" % (len(rows), len(used), rows[:12].tolist()))
            for r in rows[:4]:
                print(" row %d: used %d / expected %d; cells %s / expected %s" % (r, gused[r], used[r], pairs["cell"][r][:12].tolist(),
                                                                                 cell[r][:12].tolist()))
            raise SystemExit("PARITY FAILURE %s %r" % (what, label))
        runs[what] += 1
    print("fuzz ok", runs)


if __name__ == "__main__":
    main()
