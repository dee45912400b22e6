// em2_cluster.cpp -- label propagation over a cell graph, the consumer of em2_cell_graph_edges.
//
// Reference: CellGraph::labelPropagationClustering (src/CellGraph.cpp:443-612) with ClusterTable
// (src/CellGraph.hpp:50-121).  The algorithm is sequential by definition: every vertex update reads the labels its
// neighbours were given earlier in the same std::shuffle order, and the float weights of a vertex's cluster table
// are accumulated in the order those neighbour updates happen, so the result is tied to one serial schedule.  It
// therefore runs on the host, over the edge list the GPU produced; there is no device variant to fall back from.
//
// Layout (instead of the reference's per-vertex std::vector inside a BGL vertex):
//   adjacency   CSR (offsets, neighbour, weight); the edges of a vertex in add_edge order, which is the order
//               out_edges() walks adjacency_list<listS,listS,undirectedS>
//   tables      one arena of (cluster, weight) entries; a vertex owns a contiguous run that starts with room for its
//               degree and moves to the end of the arena, doubled, when it fills (entries are never removed, as in
//               the reference, so their order is the order of first appearance)
#include "em2_host.h"

#include <algorithm>
#include <cstdint>
#include <limits>
#include <numeric>
#include <random>
#include <vector>

namespace em2 {
namespace host {

namespace {

struct Entry {
    uint32_t cluster;
    float weight;
};

struct Table {
    uint64_t begin = 0;
    uint32_t size = 0;
    uint32_t capacity = 0;
    uint32_t bestCluster = std::numeric_limits<uint32_t>::max();
    float bestWeight = -1.f;
};

class Tables {
public:
    Tables(const std::vector<uint64_t>& offsets, uint32_t vertexCount) : tables(vertexCount)
    {
        uint64_t total = 0;
        for (uint32_t v = 0; v < vertexCount; v++) {
            const uint64_t degree = offsets[v + 1] - offsets[v];
            tables[v].begin = total;
            tables[v].capacity = uint32_t(degree);
            total += degree;
        }
        arena.resize(total);
    }

    bool empty(uint32_t v) const { return tables[v].size == 0; }
    uint32_t best(uint32_t v) const { return tables[v].bestCluster; }

    // ClusterTable::addWeightQuick (CellGraph.hpp:66-69).
    void append(uint32_t v, uint32_t cluster, float weight)
    {
        Table& t = tables[v];
        if (t.size == t.capacity) grow(t);
        arena[t.begin + t.size++] = Entry{cluster, weight};
    }

    // ClusterTable::findBestCluster (CellGraph.hpp:104-114): the first entry with the strictly largest weight
    // above -1.
    void findBest(uint32_t v)
    {
        Table& t = tables[v];
        t.bestCluster = std::numeric_limits<uint32_t>::max();
        t.bestWeight = -1.f;
        const Entry* e = arena.data() + t.begin;
        for (uint32_t i = 0; i < t.size; i++) {
            if (e[i].weight > t.bestWeight) {
                t.bestWeight = e[i].weight;
                t.bestCluster = e[i].cluster;
            }
        }
    }

    // ClusterTable::addWeight (CellGraph.hpp:70-99).
    void add(uint32_t v, uint32_t cluster, float weight)
    {
        Table& t = tables[v];
        Entry* e = arena.data() + t.begin;
        for (uint32_t i = 0; i < t.size; i++) {
            if (e[i].cluster != cluster) continue;
            e[i].weight += weight;
            if (cluster == t.bestCluster) {
                if (weight < 0.) findBest(v);
                else t.bestWeight = e[i].weight;
            } else if (e[i].weight > t.bestWeight) {
                t.bestCluster = cluster;
                t.bestWeight = e[i].weight;
            }
            return;
        }
        append(v, cluster, weight);
        if (weight > t.bestWeight) {
            t.bestCluster = cluster;
            t.bestWeight = weight;
        }
    }

private:
    void grow(Table& t)
    {
        const uint32_t capacity = std::max<uint32_t>(4, t.capacity * 2);
        const uint64_t begin = arena.size();
        arena.resize(begin + capacity);
        std::copy(arena.begin() + t.begin, arena.begin() + t.begin + t.size, arena.begin() + begin);
        t.begin = begin;
        t.capacity = capacity;
    }

    std::vector<Table> tables;
    std::vector<Entry> arena;
};

}  // namespace

uint64_t labelPropagation(const uint32_t* vertexCellIds, uint32_t vertexCount, const uint32_t* edgeVertex0,
                          const uint32_t* edgeVertex1, const float* edgeSimilarity, uint64_t edgeCount, uint64_t seed,
                          uint64_t stableIterationCountThreshold, uint64_t maxIterationCount, uint32_t* clusterIds)
{
    for (uint64_t e = 0; e < edgeCount; e++) {
        if (edgeVertex0[e] >= vertexCount || edgeVertex1[e] >= vertexCount) {
            throw Error{EM2_ERROR_INVALID_ARGUMENT, "em2_cell_graph_label_propagation: an edge names a vertex that does not exist"};
        }
    }

    // out_edges() of every vertex, in add_edge order.
    std::vector<uint64_t> offsets(size_t(vertexCount) + 1, 0);
    for (uint64_t e = 0; e < edgeCount; e++) {
        ++offsets[edgeVertex0[e] + 1];
        ++offsets[edgeVertex1[e] + 1];
    }
    std::partial_sum(offsets.begin(), offsets.end(), offsets.begin());
    std::vector<uint32_t> neighbour(2 * edgeCount);
    std::vector<float> weight(2 * edgeCount);
    {
        std::vector<uint64_t> cursor(offsets.begin(), offsets.end() - 1);
        for (uint64_t e = 0; e < edgeCount; e++) {
            const uint32_t a = edgeVertex0[e], b = edgeVertex1[e];
            neighbour[cursor[a]] = b;
            weight[cursor[a]++] = edgeSimilarity[e];
            neighbour[cursor[b]] = a;
            weight[cursor[b]++] = edgeSimilarity[e];
        }
    }

    // :459-476 every vertex starts in the cluster named by its own cell id; tables hold the neighbours' clusters.
    for (uint32_t v = 0; v < vertexCount; v++) clusterIds[v] = vertexCellIds[v];
    Tables tables(offsets, vertexCount);
    for (uint32_t v = 0; v < vertexCount; v++) {
        for (uint64_t i = offsets[v]; i < offsets[v + 1]; i++) tables.append(v, clusterIds[neighbour[i]], weight[i]);
        tables.findBest(v);
    }

    // :484-489 the shuffle starts from the vertices in ascending cell id (std::map order; on equal ids the map kept
    // the first vertex only).
    std::vector<uint32_t> allVertices(vertexCount);
    std::iota(allVertices.begin(), allVertices.end(), 0u);
    if (!std::is_sorted(vertexCellIds, vertexCellIds + vertexCount, [](uint32_t a, uint32_t b) { return a <= b; })) {
        std::stable_sort(allVertices.begin(), allVertices.end(),
                         [&](uint32_t a, uint32_t b) { return vertexCellIds[a] < vertexCellIds[b]; });
        allVertices.erase(std::unique(allVertices.begin(), allVertices.end(),
                                      [&](uint32_t a, uint32_t b) { return vertexCellIds[a] == vertexCellIds[b]; }),
                          allVertices.end());
    }

    std::mt19937 randomGenerator(seed);
    std::vector<uint32_t> shuffled;
    uint64_t stable = 0;
    uint64_t iterations = 0;
    while (iterations < maxIterationCount) {
        ++iterations;
        uint64_t changes = 0;
        shuffled = allVertices;
        std::shuffle(shuffled.begin(), shuffled.end(), randomGenerator);
        for (const uint32_t v : shuffled) {
            if (tables.empty(v)) continue;
            const uint32_t to = tables.best(v);
            const uint32_t from = clusterIds[v];
            if (from == to) continue;
            clusterIds[v] = to;
            ++changes;
            for (uint64_t i = offsets[v]; i < offsets[v + 1]; i++) {
                tables.add(neighbour[i], to, weight[i]);
                tables.add(neighbour[i], from, -weight[i]);
            }
        }
        stable = changes ? 0 : stable + 1;
        if (stable == stableIterationCountThreshold) break;
    }

    // :561-596 renumber by decreasing size; equal sizes by decreasing old id (std::greater on the pair).
    std::vector<uint32_t> sortedIds(clusterIds, clusterIds + vertexCount);
    std::sort(sortedIds.begin(), sortedIds.end());
    struct Cluster {
        uint64_t size;
        uint32_t id;
    };
    std::vector<Cluster> clusters;
    for (size_t i = 0; i < sortedIds.size();) {
        size_t j = i;
        while (j < sortedIds.size() && sortedIds[j] == sortedIds[i]) ++j;
        clusters.push_back(Cluster{uint64_t(j - i), sortedIds[i]});
        i = j;
    }
    std::sort(clusters.begin(), clusters.end(), [](const Cluster& a, const Cluster& b) {
        return a.size != b.size ? a.size > b.size : a.id > b.id;
    });
    std::vector<std::pair<uint32_t, uint32_t>> renumber(clusters.size());
    for (uint32_t i = 0; i < clusters.size(); i++) renumber[i] = std::make_pair(clusters[i].id, i);
    std::sort(renumber.begin(), renumber.end());
    for (uint32_t v = 0; v < vertexCount; v++) {
        const auto it = std::lower_bound(renumber.begin(), renumber.end(), std::make_pair(clusterIds[v], 0u));
        clusterIds[v] = it->second;
    }
    return iterations;
}

}  // namespace host
}  // namespace em2
