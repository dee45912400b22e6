// em2_project.hip -- LSH signature projection on gfx950, bit-identical to
// ExpressionMatrixSubset::computeSums (src/ExpressionMatrixSubset.cpp:47-58, sum1 only) and
// Lsh::computeCellLshSignatures (src/Lsh.cpp:118-224).
//
// Exactness: the sign of a sequential FP64 sum decides each bit, so the order of operations is part of
// the contract:  sp = (-mean) * S[i];  then for every stored count of the cell in stored order
// sp = sp + (double(count) * U[g][i])  with the product rounded before the add (the reference is built for
// SSE4.2: no FMA).  This file is compiled with -ffp-contract=off and uses __dmul_rn/__dadd_rn.
// No MFMA: an MFMA contraction would change the summation order.
//
// Mapping: one lane = one signature bit of one cell; the 64 lanes of a wave are 64 consecutive bits, so
// every expression count costs the wave one coalesced 512-byte read of the gene's hyperplane row segment and
// the finished 64 bits leave as one ballot -> one uint64 store (MSB-first, src/BitSet.hpp:48-62).
// The cell's (gene, count) stream is wave-uniform and comes through the scalar unit.  grid.y walks 256-bit
// column chunks of the hyperplane matrix so the chunk in use (geneCount x 256 x 8 B) stays cache resident.

#include "em2_device.h"

namespace em2 {
namespace {

typedef const __attribute__((address_space(4))) uint64_t* ScalarPtr64;

constexpr int kCellsPerBlock = 16;

__global__ void __launch_bounds__(256)
cellMeansKernel(const uint64_t* __restrict__ toc, const CountIn* __restrict__ data, uint32_t cellCount,
                uint32_t geneCount, double* __restrict__ means)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cellCount) return;
    double sum1 = 0.;
    const uint64_t end = toc[c + 1];
    for (uint64_t j = toc[c]; j < end; ++j) {
        sum1 = __dadd_rn(sum1, double(data[j].count));      // ExpressionMatrixSubset.cpp:52-55
    }
    means[c] = sum1 / double(geneCount);                     // Lsh.cpp:168
}

__global__ void __launch_bounds__(256)
vectorSumsKernel(const double* __restrict__ vectors, uint32_t geneCount, uint32_t lshCount,
                 double* __restrict__ sums)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lshCount) return;
    double s = 0.;
    for (uint32_t g = 0; g < geneCount; ++g) {
        s = __dadd_rn(s, vectors[size_t(g) * lshCount + i]);  // Lsh.cpp:137-144
    }
    sums[i] = s;
}

__global__ void __launch_bounds__(256)
projectionKernel(const uint64_t* __restrict__ toc, const CountIn* __restrict__ data, uint32_t cellCount,
                 const double* __restrict__ vectors, const double* __restrict__ vectorSums,
                 const double* __restrict__ means, uint32_t lshCount, uint32_t wordCount,
                 uint64_t* __restrict__ signatures)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t word = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (word >= wordCount) return;
    const uint32_t bit = word * 64u + lane;
    const bool bitValid = bit < lshCount;
    const double* column = vectors + (bitValid ? bit : 0u);
    const double s = bitValid ? vectorSums[bit] : 0.;
    ScalarPtr64 entries = (ScalarPtr64)(uintptr_t)data;

    const uint32_t cellBegin = blockIdx.x * kCellsPerBlock;
    const uint32_t cellEnd = min(cellBegin + kCellsPerBlock, cellCount);
    for (uint32_t c = cellBegin; c < cellEnd; ++c) {
        const uint64_t jBegin = toc[c];
        const uint64_t jEnd = toc[c + 1];
        double sp = __dmul_rn(-means[c], s);                 // Lsh.cpp:180-182
        uint64_t j = jBegin;
        for (; j + 4 <= jEnd; j += 4) {
            double u[4], x[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint64_t e = entries[j + t];          // {gene (low 32), count bits (high 32)}
                u[t] = column[size_t(uint32_t(e)) * lshCount];
                x[t] = double(__uint_as_float(uint32_t(e >> 32)));
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) sp = __dadd_rn(sp, __dmul_rn(x[t], u[t]));   // Lsh.cpp:188-198
        }
        for (; j < jEnd; ++j) {
            const uint64_t e = entries[j];
            const double u = column[size_t(uint32_t(e)) * lshCount];
            const double x = double(__uint_as_float(uint32_t(e >> 32)));
            sp = __dadd_rn(sp, __dmul_rn(x, u));
        }
        const uint64_t mask = __builtin_amdgcn_ballot_w64(bitValid && sp > 0.);      // Lsh.cpp:201-206
        if (lane == 0u) signatures[size_t(c) * wordCount + word] = __brevll(mask);  // bit i -> 63-(i&63)
    }
}

}  // namespace


hipError_t launchCellMeans(const uint64_t* toc, const CountIn* data, uint32_t cellCount, uint32_t geneCount,
                           double* means, hipStream_t stream)
{
    if (cellCount == 0) return hipSuccess;
    cellMeansKernel<<<dim3((cellCount + 255u) / 256u), dim3(256), 0, stream>>>(toc, data, cellCount, geneCount,
                                                                                means);
    return hipGetLastError();
}

hipError_t launchVectorSums(const double* vectors, uint32_t geneCount, uint32_t lshCount, double* sums,
                            hipStream_t stream)
{
    if (lshCount == 0) return hipSuccess;
    vectorSumsKernel<<<dim3((lshCount + 63u) / 64u), dim3(64), 0, stream>>>(vectors, geneCount, lshCount, sums);
    return hipGetLastError();
}

hipError_t launchProjection(const uint64_t* toc, const CountIn* data, uint32_t cellCount, const double* vectors,
                            const double* vectorSums, const double* means, uint32_t lshCount,
                            uint64_t* signatures, hipStream_t stream)
{
    if (cellCount == 0 || lshCount == 0) return hipSuccess;
    const uint32_t wordCount = (lshCount - 1u) / 64u + 1u;
    const dim3 grid((cellCount + kCellsPerBlock - 1u) / kCellsPerBlock, (wordCount + 3u) / 4u);
    projectionKernel<<<grid, dim3(256), 0, stream>>>(toc, data, cellCount, vectors, vectorSums, means, lshCount,
                                                     wordCount, signatures);
    return hipGetLastError();
}

}  // namespace em2
