// em2_subset.hip -- ExpressionMatrixSubset::ExpressionMatrixSubset (src/ExpressionMatrixSubset.cpp:9-42) on the
// device: the global CSR restricted to a cell set and a gene set, gene ids remapped to the gene set's local ids.
// HighInformationGenes runs no longer rewrite the CSR on the host.
//
// One wave per cell.  Entries of a cell keep their order (the reference pushes them in storage order, :29-38), so the
// fill is an ordered compaction: ballot of the kept lanes + prefix popcount gives each kept entry its slot.
// Index / byte work only; the counts are copied bit for bit.

#include "em2_device.h"

#include <cstring>

#include <rocprim/rocprim.hpp>

namespace em2 {
namespace {

constexpr uint32_t kInvalid = 0xffffffffu;

__device__ __forceinline__ uint32_t lanesBelowMask(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
}

template <bool FILL>
__global__ void __launch_bounds__(256)
subsetKernel(const uint64_t* __restrict__ globalToc, const CountIn* __restrict__ globalData, const uint32_t* __restrict__ cellIds,
             uint32_t cellCount, const uint32_t* __restrict__ geneLocalIds, uint32_t globalGeneCount,
             uint64_t* __restrict__ counts, const uint64_t* __restrict__ toc, CountIn* __restrict__ outData)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t cell = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (cell >= cellCount) return;
    const uint32_t global = cellIds ? cellIds[cell] : cell;
    const uint64_t begin = globalToc[global];
    const uint64_t end = globalToc[global + 1u];
    uint64_t kept = 0;
    uint64_t out = FILL ? toc[cell] : 0;
    for (uint64_t base = begin; base < end; base += 64u) {
        const uint64_t j = base + lane;
        CountIn e;
        e.gene = 0;
        e.count = 0.f;
        uint32_t localGene = kInvalid;
        if (j < end) {
            e = globalData[j];
            if (e.gene < globalGeneCount) localGene = geneLocalIds[e.gene];      // GeneSet::getLocalGeneId
        }
        const bool keep = localGene != kInvalid;
        const uint64_t mask = __builtin_amdgcn_ballot_w64(keep);
        if (FILL && keep) {
            CountIn o;
            o.gene = localGene;
            o.count = e.count;
            outData[out + lanesBelowMask(mask)] = o;
        }
        const uint32_t n = uint32_t(__builtin_popcountll(mask));
        kept += n;
        out += n;
    }
    if (!FILL && lane == 0u) counts[cell] = kept;
}

}  // namespace

size_t subsetWorkspaceBytes(uint32_t cellCount)
{
    size_t temp = 0;
    uint64_t* none = nullptr;
    (void)rocprim::exclusive_scan(nullptr, temp, none, none, uint64_t(0), size_t(cellCount) + 1u, rocprim::plus<uint64_t>(),
                                  hipStream_t(nullptr));
    return ((size_t(cellCount) + 1u) * sizeof(uint64_t) + 255u) / 256u * 256u + temp + 256u;
}

// toc[0..cellCount] = exclusive scan of the kept entries per cell (toc[cellCount] = total).
hipError_t launchSubsetCount(const uint64_t* globalToc, const CountIn* globalData, const uint32_t* cellIds, uint32_t cellCount,
                             const uint32_t* geneLocalIds, uint32_t globalGeneCount, uint64_t* toc, void* workspace,
                             size_t workspaceBytes, hipStream_t stream)
{
    if (workspaceBytes < subsetWorkspaceBytes(cellCount)) return hipErrorInvalidValue;
    uint64_t* counts = static_cast<uint64_t*>(workspace);
    const size_t countBytes = ((size_t(cellCount) + 1u) * sizeof(uint64_t) + 255u) / 256u * 256u;
    void* temp = static_cast<char*>(workspace) + countBytes;
    size_t tempBytes = workspaceBytes - countBytes;
    hipError_t e = hipMemsetAsync(counts + cellCount, 0, sizeof(uint64_t), stream);
    if (e != hipSuccess) return e;
    if (cellCount) {
        subsetKernel<false><<<dim3((cellCount + 3u) / 4u), dim3(256), 0, stream>>>(globalToc, globalData, cellIds, cellCount,
                                                                                     geneLocalIds, globalGeneCount, counts, nullptr, nullptr);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return rocprim::exclusive_scan(temp, tempBytes, counts, toc, uint64_t(0), size_t(cellCount) + 1u, rocprim::plus<uint64_t>(), stream);
}

hipError_t launchSubsetFill(const uint64_t* globalToc, const CountIn* globalData, const uint32_t* cellIds, uint32_t cellCount,
                            const uint32_t* geneLocalIds, uint32_t globalGeneCount, const uint64_t* toc, CountIn* outData,
                            hipStream_t stream)
{
    if (cellCount == 0) return hipSuccess;
    subsetKernel<true><<<dim3((cellCount + 3u) / 4u), dim3(256), 0, stream>>>(globalToc, globalData, cellIds, cellCount, geneLocalIds,
                                                                                globalGeneCount, nullptr, toc, outData);
    return hipGetLastError();
}

}  // namespace em2
