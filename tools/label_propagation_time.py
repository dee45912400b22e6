"""Times em2_cell_graph_label_propagation on a synthetic k-NN-like graph and (optionally) checks it against the
oracle's serial restatement, which doubles as the CPU time of the reference's algorithm.
    python tools/label_propagation_time.py VERTICES DEGREE [--oracle]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from expressionmatrix2_amd import capi          # noqa: E402
from label_graphs import fast_graph             # noqa: E402


def main():
    n, degree = int(sys.argv[1]), int(sys.argv[2])
    rng = np.random.default_rng(1)
    cells, v0, v1, sim = fast_graph(rng, n, degree, 64)
    print("vertices %d edges %d" % (n, len(v0)), flush=True)
    capi.cell_graph_label_propagation(cells[:1000], v0[:0], v1[:0], sim[:0])        # library + device warm-up
    for _ in range(2):
        t = time.time()
        got, iterations = capi.cell_graph_label_propagation(cells, v0, v1, sim)
        elapsed = time.time() - t
        print("gpu: %.3f s, %d iterations, %d clusters" % (elapsed, iterations, int(got.max()) + 1), flush=True)
        for line in capi.last_timing().splitlines() if hasattr(capi, "last_timing") else []:
            print("   ", line)
    if "--oracle" in sys.argv:
        import oracle_binding
        oracle = oracle_binding.load_oracle()
        t = time.time()
        expected, expected_iterations = oracle.label_propagation(cells, v0, v1, sim)
        print("oracle (serial CPU): %.3f s, %d iterations" % (time.time() - t, expected_iterations))
        print("identical:", bool(np.array_equal(got, expected)) and iterations == expected_iterations)


if __name__ == "__main__":
    main()
