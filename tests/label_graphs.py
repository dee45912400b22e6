"""Graph generators shared by the label propagation tests."""
import numpy as np


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


def fast_graph(rng, vertex_count, degree, clusters, hubs=0, hub_degree=0, parallel_edges=0, tie_levels=0):
    """The same shape built with array operations, for sizes the loop above is too slow for.  hubs vertices get
    hub_degree extra edges each (degree > 64 takes the staged-candidate path of the kernel); parallel_edges existing
    edges are repeated (boost's listS graph would hold them twice; cell_graph_edges never produces them)."""
    block = rng.integers(0, clusters, vertex_count)
    order = np.argsort(block, kind="stable")
    starts = np.searchsorted(block[order], np.arange(clusters + 1))
    src = np.repeat(np.arange(vertex_count), degree)
    b = block[src]
    pick = (rng.random(len(src)) * (starts[b + 1] - starts[b])).astype(np.int64) + starts[b]
    dst = order[pick]
    far = rng.random(len(src)) < 0.1
    dst[far] = rng.integers(0, vertex_count, int(far.sum()))
    if hubs:
        hub = rng.choice(vertex_count, hubs, replace=False)
        src = np.concatenate([src, rng.integers(0, vertex_count, hubs * hub_degree)])
        dst = np.concatenate([dst, np.repeat(hub, hub_degree)])
        shuffle = rng.permutation(len(src))
        src, dst = src[shuffle], dst[shuffle]
    keep = src != dst
    src, dst = src[keep], dst[keep]
    key = np.minimum(src, dst).astype(np.int64) * vertex_count + np.maximum(src, dst)
    _, first = np.unique(key, return_index=True)
    first.sort()
    v0, v1 = src[first].astype(np.uint32), dst[first].astype(np.uint32)
    sim = (0.2 + 0.8 * rng.random(len(v0)))
    if tie_levels:
        sim = np.floor(sim * tie_levels) / tie_levels
    sim = sim.astype(np.float32)
    if parallel_edges:
        again = rng.integers(0, len(v0), parallel_edges)
        at = np.sort(rng.integers(0, len(v0), parallel_edges))
        v0 = np.insert(v0, at, v0[again])
        v1 = np.insert(v1, at, v1[again])
        sim = np.insert(sim, at, (0.2 + 0.8 * rng.random(parallel_edges)).astype(np.float32))
    cells = np.sort(rng.choice(10 * vertex_count, vertex_count, replace=False)).astype(np.uint32)
    return cells, v0, v1, sim
