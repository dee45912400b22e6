"""CPU side of SURVEY.md 8(f) row 2: the oracle's restatement of CellGraph::labelPropagationClustering
(src/CellGraph.cpp:443-612) on cases small enough to work out by hand, and the refusal of the product entry point to
run without a device.  The parity tests proper are tests/test_gpu_label_propagation.py."""
import numpy as np
import pytest

from expressionmatrix2_amd import capi


def test_oracle_two_vertices_by_hand(oracle):
    # Whichever vertex the shuffle puts first adopts the other's label; the push (+0.5 on the new label, -0.5 on the
    # old, ClusterTable::addWeight) leaves the second vertex consistent: one cluster, one changing iteration and
    # three stable ones.
    clusters, iterations = oracle.label_propagation([5, 9], [0], [1], [0.5])
    assert clusters.tolist() == [0, 0] and iterations == 4


def test_oracle_renumbering_by_size_then_by_decreasing_label(oracle):
    # std::sort with std::greater on (size, label), CellGraph.cpp:575-579: the pair {8,20} first, then the
    # singletons 40, 21, 3.
    clusters, iterations = oracle.label_propagation([3, 8, 20, 21, 40], [1], [2], [0.7])
    assert clusters.tolist() == [3, 0, 0, 2, 1]
    clusters, iterations = oracle.label_propagation(np.arange(4), [], [], [])
    assert clusters.tolist() == [3, 2, 1, 0] and iterations == 3


def test_oracle_triangle_with_a_tail(oracle):
    # 0-1-2 triangle of weight 1 and a tail 2-3 of weight 0.25: every shuffle ends in one cluster, because a vertex
    # with a single neighbour always follows it and the triangle's weights dominate the tail's.
    for seed in range(8):
        clusters, _ = oracle.label_propagation([10, 11, 12, 13], [0, 1, 0, 2], [1, 2, 2, 3], [1, 1, 1, 0.25], seed)
        assert clusters.tolist() == [0, 0, 0, 0]


def test_product_refuses_to_run_without_a_device():
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        capi.cell_graph_label_propagation([5, 9], [0], [1], [0.5])
