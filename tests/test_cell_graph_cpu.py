"""Oracle restatement of CellGraph::CellGraph (src/CellGraph.cpp:33-117) against hand-worked cases.  The reference
holds no test for this constructor (its tests/ are case-study scripts whose graphs are only drawn), so these known
answers are worked from the constructor's text; the GPU path is compared with the oracle in test_gpu_cell_graph.py."""
import numpy as np

import synth


def pairs_from_lists(lists, k):
    n = len(lists)
    cell = np.zeros((n, k), dtype=np.uint32)
    sim = np.zeros((n, k), dtype=np.float32)
    used = np.zeros(n, dtype=np.uint32)
    for i, entries in enumerate(lists):
        used[i] = len(entries)
        for j, (c, s) in enumerate(entries):
            cell[i, j] = c
            sim[i, j] = s
    return cell, sim, used


def edges(oracle, lists, k, sp_cells, graph_cells, thr, max_conn):
    cell, sim, used = pairs_from_lists(lists, k)
    v0, v1, s = oracle.cell_graph_edges(cell, sim, used, sp_cells, graph_cells, thr, max_conn)
    # the hash-table form used at a million cells is the same function
    h0, h1, hs = oracle.cell_graph_edges(cell, sim, used, sp_cells, graph_cells, thr, max_conn, hashed=True)
    assert np.array_equal(v0, h0) and np.array_equal(v1, h1) and np.array_equal(s.view(np.uint32), hs.view(np.uint32))
    return list(zip(v0.tolist(), v1.tolist())), s


def test_mutual_pairs_give_one_edge_in_first_seen_direction(oracle):
    lists = [[(1, 0.9), (2, 0.8)], [(0, 0.9), (2, 0.7)], [(0, 0.8), (1, 0.7)]]
    e, s = edges(oracle, lists, 2, [0, 1, 2], [0, 1, 2], 0.5, 20)
    assert e == [(0, 1), (0, 2), (1, 2)]
    assert np.array_equal(s, np.array([0.9, 0.8, 0.7], dtype=np.float32))


def test_one_sided_pair_from_later_vertex_points_backwards(oracle):
    # vertex 2 lists 0 but 0 does not list 2: the edge appears when 2 is processed, as (2, 0)
    lists = [[(1, 0.9)], [(0, 0.9)], [(0, 0.6)]]
    e, _ = edges(oracle, lists, 1, [0, 1, 2], [0, 1, 2], 0.5, 20)
    assert e == [(0, 1), (2, 0)]


def test_threshold_breaks_and_connectivity_caps(oracle):
    lists = [[(1, 0.9), (2, 0.6), (3, 0.4)], [(0, 0.9)], [(0, 0.6)], [(0, 0.4)]]
    e, _ = edges(oracle, lists, 3, [0, 1, 2, 3], [0, 1, 2, 3], 0.5, 20)
    assert e == [(0, 1), (0, 2)]                        # 0.4 < 0.5 stops the scan of every list
    e, _ = edges(oracle, lists, 3, [0, 1, 2, 3], [0, 1, 2, 3], 0.0, 1)
    assert e == [(0, 1), (2, 0), (3, 0)]                # cell 0 keeps its best only; 2 and 3 still point at it
    e, _ = edges(oracle, lists, 3, [0, 1, 2, 3], [0, 1, 2, 3], 0.0, 0)
    assert e == [(0, 1), (0, 2), (0, 3)]                # 0 never equals a size after push_back: no cap (:101)


def test_similarity_equal_to_threshold_is_kept(oracle):
    half = np.float32(0.5)
    lists = [[(1, half)], [(0, half)]]
    assert edges(oracle, lists, 1, [0, 1], [0, 1], 0.5, 20)[0] == [(0, 1)]
    # float 0.2 promoted to double is above the double 0.2 only if float(0.2) >= 0.2: it is (0.200000003)
    lists = [[(1, np.float32(0.2))], []]
    assert edges(oracle, lists, 1, [0, 1], [0, 1], 0.2, 20)[0] == [(0, 1)]
    lists = [[(1, np.float32(0.1))], []]                # float(0.1) = 0.100000001 >= 0.1
    assert edges(oracle, lists, 1, [0, 1], [0, 1], 0.1, 20)[0] == [(0, 1)]
    lists = [[(1, np.float32(0.7))], []]                # float(0.7) = 0.699999988 < 0.7
    assert edges(oracle, lists, 1, [0, 1], [0, 1], 0.7, 20)[0] == []


def test_two_cell_sets(oracle):
    # SimilarPairs built on cells {10,20,30,40}; the graph on {40,20,99,10} in that (vertex) order.
    sp_cells = [10, 20, 30, 40]
    lists = [[(1, 0.9), (2, 0.85), (3, 0.8)],           # 10: 20, 30, 40
             [(0, 0.9), (3, 0.6)],                      # 20: 10, 40
             [(0, 0.85)],                               # 30: 10
             [(2, 0.95), (0, 0.8)]]                     # 40: 30, 10
    graph_cells = [40, 20, 99, 10]
    e, s = edges(oracle, lists, 3, sp_cells, graph_cells, 0.5, 2)
    # vertex 0 (cell 40): 30 is not in the graph (skipped without using a slot), then 10 -> (0,3)
    # vertex 1 (cell 20): 10 -> (1,3); 40 -> (1,0)
    # vertex 2 (cell 99): not in the SimilarPairs cell set, skipped
    # vertex 3 (cell 10): 20 exists; 30 absent; 40 exists
    assert e == [(0, 3), (1, 3), (1, 0)]
    assert np.array_equal(s, np.array([0.8, 0.9, 0.6], dtype=np.float32))


def test_knn_property_on_fsp4_output(oracle):
    L, k = 256, 12
    sig = synth.clustered_signatures(400, L, cluster_count=6, flip=0.1, seed=3)
    cell, sim, used = oracle.find_similar_pairs4(sig, L, k, 0.2)
    ids = np.arange(400, dtype=np.uint32)
    v0, v1, s = oracle.cell_graph_edges(cell, sim, used, ids, ids, 0.5, 5)
    assert len(v0) > 0
    undirected = set(map(tuple, np.sort(np.stack([v0, v1], 1), 1).tolist()))
    assert len(undirected) == len(v0)                   # no parallel edges
    assert np.all(s >= np.float32(0.5))
    # every edge is within the best 5 above-threshold pairs of its first vertex
    for a, b, w in zip(v0.tolist(), v1.tolist(), s.tolist()):
        best = [(c, x) for c, x in zip(cell[a, :used[a]].tolist(), sim[a, :used[a]].tolist()) if x >= 0.5][:5]
        assert (b, np.float32(w)) in [(c, np.float32(x)) for c, x in best]


def test_hashed_form_equals_the_literal_form_on_random_inputs(oracle):
    """em2o_cell_graph_edges_hashed (what the 1M-cell GPU test checks against) against the literal std::map / std::set
    restatement: random SimilarPairs incl. duplicate and self pairs, graph cell sets that are unsorted, overlap the
    SimilarPairs cell set partially and name a cell twice (the vertex table keeps the first), several caps."""
    rng = np.random.default_rng(12)
    for trial in range(30):
        n, k = int(rng.integers(1, 300)), int(rng.integers(1, 9))
        sp_cells = np.sort(rng.choice(1000, size=n, replace=False)).astype(np.uint32)
        cell = rng.integers(0, n, size=(n, k)).astype(np.uint32)
        sim = -np.sort(-rng.random((n, k)).astype(np.float32), axis=1)
        used = rng.integers(0, k + 1, size=n).astype(np.uint32)
        graph_cells = rng.choice(1000, size=int(rng.integers(1, 400)), replace=True).astype(np.uint32)
        for max_conn in (0, 1, 3, 20):
            a = oracle.cell_graph_edges(cell, sim, used, sp_cells, graph_cells, 0.3, max_conn)
            b = oracle.cell_graph_edges(cell, sim, used, sp_cells, graph_cells, 0.3, max_conn, hashed=True)
            assert all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in zip(a, b))


def test_fsp5_listed_cells_equal_the_full_run(oracle):
    """em2o_find_similar_pairs5_cells (flat bucket tables, listed cells) against em2o_find_similar_pairs5 (the reference's
    vector-of-vectors tables, every cell)."""
    for L, q, overflow in ((256, 6, 0), (256, 7, 30), (200, 9, 1000)):
        sig = synth.clustered_signatures(700, L, cluster_count=5, flip=0.12, seed=L + q)
        cell, sim, used = oracle.find_similar_pairs5(sig, L, 7, 0.1, q, overflow)
        listed = np.array([699, 0, 5, 5, 340], dtype=np.uint32)
        c, s, u = oracle.find_similar_pairs5_cells(sig, L, 7, 0.1, q, overflow, listed)
        assert np.array_equal(c, cell[listed]) and np.array_equal(s.view(np.uint32), sim[listed].view(np.uint32))
        assert np.array_equal(u, used[listed])
