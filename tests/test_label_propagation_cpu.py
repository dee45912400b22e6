"""Label propagation over a cell graph: the host implementation behind
em2_cell_graph_label_propagation against the oracle's literal restatement of
CellGraph::labelPropagationClustering (src/CellGraph.cpp:443-612).  Host code on both sides, so no GPU is needed."""
import os

import numpy as np
import pytest

from expressionmatrix2_amd import capi


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(capi.LIBRARY_PATH):
        capi.build_library()
    return capi.load()


def random_graph(rng, vertex_count, degree, clusters, tie_levels=0, sorted_ids=True):
    """A k-NN-like graph: every vertex proposes `degree` neighbours, mostly inside its own block; duplicate and
    self edges are dropped, the first proposal of a pair fixes its position (the add_edge order)."""
    block = rng.integers(0, clusters, vertex_count)
    seen = set()
    v0, v1, sim = [], [], []
    for v in range(vertex_count):
        same = np.flatnonzero(block == block[v])
        for _ in range(degree):
            w = int(rng.choice(same)) if rng.random() < 0.8 else int(rng.integers(0, vertex_count))
            key = (min(v, w), max(v, w))
            if w == v or key in seen:
                continue
            seen.add(key)
            v0.append(v)
            v1.append(w)
            s = rng.random() * 0.8 + 0.2
            if tie_levels:
                s = np.floor(s * tie_levels) / tie_levels
            sim.append(s if rng.random() < 0.9 else -s * 0.1)
    cells = np.sort(rng.choice(10 * vertex_count, vertex_count, replace=False)).astype(np.uint32)
    if not sorted_ids:
        cells = cells[rng.permutation(vertex_count)]
    return cells, np.array(v0, np.uint32), np.array(v1, np.uint32), np.array(sim, np.float32)


def test_two_vertices_by_hand(lib):
    # Whichever vertex the shuffle puts first adopts the other's label, and the second then finds itself
    # consistent: one cluster, numbered 0; one changing iteration and three stable ones.
    clusters, iterations = capi.cell_graph_label_propagation([5, 9], [0], [1], [0.5])
    assert clusters.tolist() == [0, 0]
    assert iterations == 4


def test_isolated_vertices_keep_their_own_cluster(lib, oracle):
    # keepIsolatedVertices=True graphs: singleton clusters are numbered by decreasing original label
    # (std::greater on (size, id), CellGraph.cpp:579).
    cells = np.array([3, 8, 20, 21, 40], np.uint32)
    clusters, iterations = capi.cell_graph_label_propagation(cells, [1], [2], [0.7])
    assert clusters[1] == clusters[2] == 0
    assert clusters[[4, 3, 0]].tolist() == [1, 2, 3]
    expected, expected_iterations = oracle.label_propagation(cells, [1], [2], [0.7])
    assert clusters.tolist() == expected.tolist() and iterations == expected_iterations


def test_no_edges_and_no_vertices(lib):
    clusters, iterations = capi.cell_graph_label_propagation(np.arange(4), [], [], [])
    assert clusters.tolist() == [3, 2, 1, 0] and iterations == 3
    clusters, iterations = capi.cell_graph_label_propagation([], [], [], [])
    assert len(clusters) == 0 and iterations == 0


def test_edge_naming_a_missing_vertex_is_refused(lib):
    with pytest.raises((RuntimeError, ValueError), match="vertex that does not exist"):
        capi.cell_graph_label_propagation([1, 2], [0], [2], [0.5])


@pytest.mark.parametrize("vertex_count,degree,clusters,tie_levels,sorted_ids,seed,stable,max_iterations", [
    (50, 3, 3, 0, True, 231, 3, 100),
    (400, 6, 5, 0, True, 231, 3, 100),
    (400, 6, 5, 4, True, 7, 3, 100),          # heavy weight ties: first-entry and first-maximum rules decide
    (1500, 10, 12, 16, True, 231, 3, 100),
    (1500, 10, 12, 0, False, 99, 2, 100),     # vertex order differs from cell id order (shuffle input is by cell id)
    (3000, 20, 40, 0, True, 231, 3, 2),       # stopped by maxIterationCount
    (3000, 20, 40, 8, True, 2 ** 40 + 5, 1, 100),   # seed beyond 32 bits (std::mt19937 takes it modulo 2^32)
    (800, 4, 2, 0, True, 231, 0, 100),        # threshold 0: leaves after the first iteration that changes something
])
def test_matches_oracle(lib, oracle, vertex_count, degree, clusters, tie_levels, sorted_ids, seed, stable, max_iterations):
    rng = np.random.default_rng(vertex_count * 31 + degree)
    cells, v0, v1, sim = random_graph(rng, vertex_count, degree, clusters, tie_levels, sorted_ids)
    got, iterations = capi.cell_graph_label_propagation(cells, v0, v1, sim, seed, stable, max_iterations)
    expected, expected_iterations = oracle.label_propagation(cells, v0, v1, sim, seed, stable, max_iterations)
    assert iterations == expected_iterations
    assert np.array_equal(got, expected)
    # cluster numbers are contiguous from 0 and ordered by decreasing size
    sizes = np.bincount(got)
    assert sizes.min() > 0 and np.all(np.diff(sizes) <= 0)
