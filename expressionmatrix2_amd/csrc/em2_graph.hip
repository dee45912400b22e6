// em2_graph.hip -- the edges of the k-NN cell similarity graph, SURVEY.md 8(f) row 1: CellGraph::CellGraph
// (src/CellGraph.cpp:33-117), the immediate consumer of SimilarPairs (createCellGraph,
// src/ExpressionMatrix.cpp:1795-1845).
//
// Reference: vertices are added in cell-set order; then for every cell in that order the first <= maxConnectivity
// stored pairs with similarity >= similarityThreshold (float promoted to double, :92) whose other cell is also in
// the graph's cell set are selected (:86-103) and an undirected edge is added for each unless it already exists
// (:108-117).  An edge (v0,v1) can only exist already if the earlier of the two vertices selected the later one.
// So with sel(v) = the selected list of vertex v:
//     edges, in insertion order = for v0 ascending, for v1 in sel(v0) in order: keep unless (v1 < v0 and v0 in sel(v1)).
// Integer / index work only: one thread per vertex builds sel(v), one thread per vertex filters and counts, an
// exclusive scan places the survivors.

#include "em2_device.h"

#include <cstring>

#include <rocprim/rocprim.hpp>

namespace em2 {
namespace {

constexpr uint32_t kInvalid = 0xffffffffu;

// Position of id in the ascending ids[0, n), or kInvalid.  A set of consecutive ids (AllCells; any stored set without gaps:
// ids[n - 1] - ids[0] == n - 1, which the caller passes as `consecutive`) needs no search: twenty dependent loads per look-up,
// a hundred look-ups per vertex, were most of selectNeighboursKernel.
__device__ __forceinline__ uint32_t findSorted(const uint32_t* __restrict__ ids, uint32_t n, uint32_t id, bool consecutive, uint32_t first)
{
    if (consecutive) return id - first < n ? id - first : kInvalid;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        if (ids[mid] < id) lo = mid + 1u;
        else hi = mid;
    }
    return (lo < n && ids[lo] == id) ? lo : kInvalid;
}

// sel(v0): up to maxConnectivity (vertex, similarity) entries per graph vertex.
__global__ void __launch_bounds__(256)
selectNeighboursKernel(const PairOut* __restrict__ pairs, const uint32_t* __restrict__ usedCount, uint32_t spCellCount,
                       uint32_t k, const uint32_t* __restrict__ spCellSet, const uint32_t* __restrict__ graphSortedIds,
                       const uint32_t* __restrict__ graphVertexOfSorted, const uint32_t* __restrict__ graphCellSet,
                       uint32_t graphCellCount, double similarityThreshold, uint32_t maxConnectivity,
                       uint32_t* __restrict__ selVertex, float* __restrict__ selSimilarity, uint32_t* __restrict__ selCount,
                       bool spConsecutive, uint32_t spFirst, bool graphConsecutive, uint32_t graphFirst)
{
    const uint32_t v0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (v0 >= graphCellCount) return;
    uint32_t n = 0;
    // (a set of consecutive ids in ascending order is arithmetic: the caller passes no array for it)
    const uint32_t id0 = graphCellSet ? graphCellSet[v0] : graphFirst + v0;
    const uint32_t local0 = findSorted(spCellSet, spCellCount, id0, spConsecutive, spFirst);      // getLocalCellId (:70)
    if (local0 != kInvalid) {
        const PairOut* p = pairs + size_t(local0) * k;
        const uint32_t used = usedCount[local0];
        for (uint32_t j = 0; j < used; ++j) {
            const float similarity = p[j].similarity;
            if (double(similarity) < similarityThreshold) break;                        // :92
            const uint32_t id1 = spConsecutive ? spFirst + p[j].cell : spCellSet[p[j].cell];
            const uint32_t sorted1 = findSorted(graphSortedIds, graphCellCount, id1, graphConsecutive, graphFirst);
            if (sorted1 == kInvalid) continue;                                          // :96-99
            selVertex[size_t(v0) * maxConnectivity + n] = graphVertexOfSorted ? graphVertexOfSorted[sorted1] : sorted1;
            selSimilarity[size_t(v0) * maxConnectivity + n] = similarity;
            if (++n == maxConnectivity) break;                                          // :101-103
        }
    }
    selCount[v0] = n;
}

template <bool WRITE>
__global__ void __launch_bounds__(256)
filterEdgesKernel(const uint32_t* __restrict__ selVertex, const float* __restrict__ selSimilarity,
                  const uint32_t* __restrict__ selCount, uint32_t graphCellCount, uint32_t maxConnectivity,
                  uint32_t* __restrict__ keptCount, const uint64_t* __restrict__ offsets, uint32_t* __restrict__ edge0,
                  uint32_t* __restrict__ edge1, float* __restrict__ edgeSimilarity)
{
    const uint32_t v0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (v0 >= graphCellCount) return;
    const uint32_t n = selCount[v0];
    uint32_t kept = 0;
    uint64_t out = WRITE ? offsets[v0] : 0;
    for (uint32_t j = 0; j < n; ++j) {
        const uint32_t v1 = selVertex[size_t(v0) * maxConnectivity + j];
        bool exists = false;
        // The same neighbour earlier in this list (never the case for fsp4/fsp5 output, but boost::edge would find it).
        for (uint32_t t = 0; t < j; ++t) exists |= selVertex[size_t(v0) * maxConnectivity + t] == v1;
        if (v1 < v0) {                                   // v1 was processed first: did it select v0?
            const uint32_t n1 = selCount[v1];
            for (uint32_t t = 0; t < n1; ++t) exists |= selVertex[size_t(v1) * maxConnectivity + t] == v0;
        }
        if (!exists) {
            if (WRITE) {
                edge0[out] = v0;
                edge1[out] = v1;
                edgeSimilarity[out] = selSimilarity[size_t(v0) * maxConnectivity + j];
                ++out;
            }
            ++kept;
        }
    }
    if (!WRITE) keptCount[v0] = kept;
}

struct Buffer {
    void* p = nullptr;
    ~Buffer() { if (p) (void)hipFree(p); }
    hipError_t allocate(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

#define EM2_TRY(call)                        \
    do {                                     \
        hipError_t em2Err_ = (call);         \
        if (em2Err_ != hipSuccess) return em2Err_; \
    } while (0)

}  // namespace

// All pointers are device pointers; graphSortedIds / graphVertexOfSorted = the graph's cell set sorted by id and the
// vertex (position in the set) of each sorted entry.  maxConnectivity is the effective one (1..k, see em2_capi.hip);
// edge arrays have capacity graphCellCount*maxConnectivity.
hipError_t runCellGraphEdges(const PairOut* pairs, const uint32_t* usedCount, uint32_t spCellCount, uint32_t k,
                             const uint32_t* spCellSet, const uint32_t* graphCellSet, const uint32_t* graphSortedIds,
                             const uint32_t* graphVertexOfSorted, uint32_t graphCellCount, double similarityThreshold,
                             uint32_t maxConnectivity, uint32_t* edge0, uint32_t* edge1, float* edgeSimilarity,
                             uint64_t* edgeCountHost, hipStream_t stream, bool spConsecutive, uint32_t spFirst, bool graphConsecutive,
                             uint32_t graphFirst)
{
    *edgeCountHost = 0;
    if (graphCellCount == 0 || maxConnectivity == 0) return hipSuccess;
    Buffer selVertex, selSim, selCount, kept, offsets, temp;
    const size_t slots = size_t(graphCellCount) * maxConnectivity;
    EM2_TRY(selVertex.allocate(slots * sizeof(uint32_t)));
    EM2_TRY(selSim.allocate(slots * sizeof(float)));
    EM2_TRY(selCount.allocate(size_t(graphCellCount) * sizeof(uint32_t)));
    EM2_TRY(kept.allocate(size_t(graphCellCount) * sizeof(uint32_t)));
    EM2_TRY(offsets.allocate((size_t(graphCellCount) + 1) * sizeof(uint64_t)));
    const dim3 grid((graphCellCount + 255u) / 256u), block(256);
    selectNeighboursKernel<<<grid, block, 0, stream>>>(pairs, usedCount, spCellCount, k, spCellSet, graphSortedIds,
                                                       graphVertexOfSorted, graphCellSet, graphCellCount, similarityThreshold,
                                                       maxConnectivity, selVertex.as<uint32_t>(), selSim.as<float>(),
                                                       selCount.as<uint32_t>(), spConsecutive, spFirst, graphConsecutive, graphFirst);
    EM2_TRY(hipGetLastError());
    filterEdgesKernel<false><<<grid, block, 0, stream>>>(selVertex.as<uint32_t>(), selSim.as<float>(), selCount.as<uint32_t>(),
                                                         graphCellCount, maxConnectivity, kept.as<uint32_t>(), nullptr, nullptr,
                                                         nullptr, nullptr);
    EM2_TRY(hipGetLastError());
    size_t tempBytes = 0;
    EM2_TRY(rocprim::exclusive_scan(nullptr, tempBytes, kept.as<uint32_t>(), offsets.as<uint64_t>(), uint64_t(0),
                                    size_t(graphCellCount), rocprim::plus<uint64_t>(), stream));
    EM2_TRY(temp.allocate(tempBytes));
    EM2_TRY(rocprim::exclusive_scan(temp.p, tempBytes, kept.as<uint32_t>(), offsets.as<uint64_t>(), uint64_t(0),
                                    size_t(graphCellCount), rocprim::plus<uint64_t>(), stream));
    filterEdgesKernel<true><<<grid, block, 0, stream>>>(selVertex.as<uint32_t>(), selSim.as<float>(), selCount.as<uint32_t>(),
                                                        graphCellCount, maxConnectivity, nullptr, offsets.as<uint64_t>(), edge0,
                                                        edge1, edgeSimilarity);
    EM2_TRY(hipGetLastError());
    uint64_t lastOffset = 0;
    uint32_t lastKept = 0;
    EM2_TRY(hipMemcpyAsync(&lastOffset, offsets.as<uint64_t>() + (graphCellCount - 1), sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    EM2_TRY(hipMemcpyAsync(&lastKept, kept.as<uint32_t>() + (graphCellCount - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    EM2_TRY(hipStreamSynchronize(stream));
    *edgeCountHost = lastOffset + lastKept;
    return hipSuccess;
}

}  // namespace em2
