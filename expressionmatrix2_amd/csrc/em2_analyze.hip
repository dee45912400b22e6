// em2_analyze.hip -- ExpressionMatrix::analyzeLsh (src/ExpressionMatrixLsh.cpp:1244-1367): every unordered pair of
// cells of an expression matrix subset, its exact similarity (ExpressionMatrixSubset::computeCellSimilarity,
// src/ExpressionMatrixSubset.cpp:83-133) against its LSH similarity (Lsh::computeCellSimilarity, src/Lsh.cpp:254-265).
//
// The reference's loop is serial in two ways that define its output: the bins accumulate doubles in pair order
// (:1323-1325), and the csv is downsampled by one draw of a seeded mt19937 per pair, in pair order (:1329).  What is
// O(pairs x counts per cell) -- the sparse scalar product of each pair (:86-108) and the mismatch count of its two
// signatures -- is done here on the device, bit for bit as the reference does it: the products are float products, the
// sum a double sum in ascending gene order.  What is O(pairs) and order-defined -- correlation coefficient from the
// scalar product, bins, the random draw, the csv lines -- stays with the host (analyzeLshHost below), which walks the
// pairs in the reference's order over the device's output, chunk of rows by chunk of rows.
//
// Device layout: one block per row cell i.  The counts of cell i are scattered into a dense float vector over the
// local gene ids, with a presence bitmap, in LDS (36864 genes fit) or, for larger gene sets, in a global scratch vector
// per block; thread t then takes the cells j = i + 1 + t, i + 1 + t + 256, ... and walks cell j's counts once:
// for the genes both cells have, in ascending gene order, scalarProduct += count_i * count_j -- the pairs the
// reference's two-pointer merge visits, in the same order.

#include "em2_device.h"

#include <cmath>
#include <cstdio>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "em2_tables.h"

namespace em2 {
namespace {

constexpr uint32_t kDenseLdsGenes = 36864;          // 144 KB of floats + 4.5 KB of bitmap

template <bool IN_LDS>
__global__ void __launch_bounds__(256)
analyzePairsKernel(const uint64_t* __restrict__ toc, const CountIn* __restrict__ data, uint32_t cellCount, uint32_t geneCount,
                   const uint64_t* __restrict__ sig, uint32_t words, uint32_t rowBegin, float* __restrict__ denseScratch,
                   uint32_t* __restrict__ presentScratch, double* __restrict__ scalarProducts, uint32_t* __restrict__ mismatches)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
    const uint32_t bitmapWords = (geneCount + 31u) / 32u;
    float* dense = IN_LDS ? reinterpret_cast<float*>(ldsRaw) : denseScratch + size_t(blockIdx.x) * geneCount;
    uint32_t* present = IN_LDS ? reinterpret_cast<uint32_t*>(ldsRaw + size_t(geneCount) * 4u)
                               : presentScratch + size_t(blockIdx.x) * bitmapWords;
    const uint32_t i = rowBegin + blockIdx.x;
    for (uint32_t w = threadIdx.x; w < bitmapWords; w += blockDim.x) present[w] = 0u;
    __syncthreads();
    const uint64_t begin0 = toc[i], end0 = toc[i + 1u];
    for (uint64_t p = begin0 + threadIdx.x; p < end0; p += blockDim.x) {
        const CountIn c = data[p];
        dense[c.gene] = c.count;
        atomicOr(present + (c.gene >> 5), 1u << (c.gene & 31u));
    }
    __syncthreads();
    // pairs of row i start at this offset of the chunk's output: rows rowBegin .. i-1 have cellCount - 1 - r pairs each
    const uint64_t below = uint64_t(i) * (uint64_t(i) + 1u) / 2u - uint64_t(i);              // 0 + 1 + ... + (i - 1)
    const uint64_t belowBegin = uint64_t(rowBegin) * (uint64_t(rowBegin) + 1u) / 2u - uint64_t(rowBegin);
    const uint64_t offset = uint64_t(i - rowBegin) * uint64_t(cellCount - 1u) - (below - belowBegin);
    const uint64_t* sig0 = sig + size_t(i) * words;
    for (uint32_t j = i + 1u + threadIdx.x; j < cellCount; j += blockDim.x) {
        double scalarProduct = 0.;
        const uint64_t end1 = toc[j + 1u];
        for (uint64_t p = toc[j]; p < end1; ++p) {
            const CountIn c = data[p];
            if ((present[c.gene >> 5] >> (c.gene & 31u)) & 1u) {
                const float product = dense[c.gene] * c.count;          // it0->second * it1->second: a float product (:103)
                scalarProduct += double(product);
            }
        }
        const uint64_t* sig1 = sig + size_t(j) * words;
        uint32_t m = 0;
        for (uint32_t w = 0; w < words; ++w) m += uint32_t(__builtin_popcountll(sig0[w] ^ sig1[w]));
        const uint64_t at = offset + (j - i - 1u);
        scalarProducts[at] = scalarProduct;
        mismatches[at] = m;
    }
}

}  // namespace


uint64_t analyzePairCount(uint32_t cellCount, uint32_t rowBegin, uint32_t rowEnd)
{
    uint64_t n = 0;
    for (uint32_t r = rowBegin; r < rowEnd; ++r) n += uint64_t(cellCount - 1u - r);
    return n;
}

size_t analyzeScratchBytes(uint32_t geneCount, uint32_t rowCount)
{
    if (geneCount <= kDenseLdsGenes) return 0;
    return (size_t(geneCount) * 4u + size_t((geneCount + 31u) / 32u) * 4u) * rowCount;
}

// Rows [rowBegin, rowEnd) against the cells above them; scalarProducts / mismatches hold analyzePairCount entries, row
// by row, within a row by ascending second cell.  scratch: analyzeScratchBytes(geneCount, rowEnd - rowBegin).
hipError_t launchAnalyzePairs(const uint64_t* toc, const CountIn* data, uint32_t cellCount, uint32_t geneCount,
                              const uint64_t* signatures, uint32_t words, uint32_t rowBegin, uint32_t rowEnd, void* scratch,
                              double* scalarProducts, uint32_t* mismatches, hipStream_t stream)
{
    if (rowEnd <= rowBegin) return hipSuccess;
    const uint32_t rows = rowEnd - rowBegin;
    if (geneCount <= kDenseLdsGenes) {
        const size_t lds = size_t(geneCount) * 4u + size_t((geneCount + 31u) / 32u) * 4u;
        const void* kernel = reinterpret_cast<const void*>(&analyzePairsKernel<true>);
        hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
        analyzePairsKernel<true><<<dim3(rows), dim3(256), lds, stream>>>(toc, data, cellCount, geneCount, signatures, words, rowBegin,
                                                                        nullptr, nullptr, scalarProducts, mismatches);
    } else {
        float* dense = static_cast<float*>(scratch);
        uint32_t* present = reinterpret_cast<uint32_t*>(static_cast<char*>(scratch) + size_t(geneCount) * 4u * rows);
        analyzePairsKernel<false><<<dim3(rows), dim3(256), 0, stream>>>(toc, data, cellCount, geneCount, signatures, words, rowBegin,
                                                                       dense, present, scalarProducts, mismatches);
    }
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------------
// The host half: everything of ExpressionMatrixLsh.cpp:1286-1364 that is defined by the order of the pairs.
// ---------------------------------------------------------------------------------------------------------

struct AnalyzeLshState {
    static constexpr size_t binCount = 200;                          // :1297
    std::vector<uint64_t> sum0 = std::vector<uint64_t>(binCount, 0);
    std::vector<double> sum1 = std::vector<double>(binCount, 0.);
    std::vector<double> sum2 = std::vector<double>(binCount, 0.);
    std::mt19937 randomSource;                                        // boost::mt19937 has std::mt19937's parameters
    std::ofstream csvOut;
    std::vector<double> similarityTable;
};

AnalyzeLshState* analyzeLshBegin(uint32_t lshCount, uint32_t seed, const char* pairsCsvPath)
{
    AnalyzeLshState* s = new AnalyzeLshState;
    s->randomSource.seed(seed);
    s->similarityTable.resize(size_t(lshCount) + 1);
    computeSimilarityTable(lshCount, s->similarityTable.data());
    s->csvOut.open(pairsCsvPath);
    if (!s->csvOut) {
        delete s;
        return nullptr;
    }
    s->csvOut << "LocalCellId0,LocalCellId1,GlobalCellId0,GlobalCellId1,ExactSimilarity,LshSimilarity\n";
    return s;
}

// The pairs of rows [rowBegin, rowEnd), in order.  Returns false where the reference's CZI_ASSERT(bin < binCount)
// throws (:1322).
bool analyzeLshRows(AnalyzeLshState* s, const double* sums, uint32_t cellCount, uint32_t geneCount, const uint32_t* globalCellIds,
                    uint32_t rowBegin, uint32_t rowEnd, const double* scalarProducts, const uint32_t* mismatches, double csvDownsample,
                    double* exactOut, double* lshOut)
{
    const double binWidth = 2. / double(AnalyzeLshState::binCount);
    const double factor = 1.0 / (double(0xffffffffu) + 1.0);          // boost::uniform_01 over a 32-bit engine: eng() * 2^-32
    const double n = double(geneCount);
    size_t at = 0;
    for (uint32_t localCellId0 = rowBegin; localCellId0 < rowEnd; localCellId0++) {
        const double s10 = sums[2 * size_t(localCellId0)], s20 = sums[2 * size_t(localCellId0) + 1];
        for (uint32_t localCellId1 = localCellId0 + 1; localCellId1 < cellCount; localCellId1++, at++) {
            const double s11 = sums[2 * size_t(localCellId1)], s21 = sums[2 * size_t(localCellId1) + 1];
            const double numerator = n * scalarProducts[at] - s10 * s11;                                  // ExpressionMatrixSubset.cpp:118
            const double denominator = std::sqrt((n * s20 - s10 * s10) * (n * s21 - s11 * s11));
            const double exactSimilarity = numerator / denominator;
            const double lshSimilarity = s->similarityTable[mismatches[at]];
            const double delta = lshSimilarity - exactSimilarity;
            const size_t bin = size_t(std::floor((exactSimilarity + 1.) / binWidth));
            if (!(bin < AnalyzeLshState::binCount)) return false;
            ++(s->sum0[bin]);
            s->sum1[bin] += delta;
            s->sum2[bin] += delta * delta;
            if (exactOut) exactOut[at] = exactSimilarity;
            if (lshOut) lshOut[at] = lshSimilarity;
            if (double(s->randomSource()) * factor < csvDownsample) {
                s->csvOut << localCellId0 << ",";
                s->csvOut << localCellId1 << ",";
                s->csvOut << globalCellIds[localCellId0] << ",";
                s->csvOut << globalCellIds[localCellId1] << ",";
                s->csvOut << exactSimilarity << ",";
                s->csvOut << lshSimilarity << ",\n";
            }
        }
    }
    return true;
}

// :1345-1364.  Deletes the state.
bool analyzeLshEnd(AnalyzeLshState* s, uint32_t lshCount, const char* statisticsCsvPath, uint64_t* sum0, double* sum1, double* sum2)
{
    bool ok = true;
    if (statisticsCsvPath) {
        std::ofstream statsOut(statisticsCsvPath);
        ok = bool(statsOut);
        statsOut << "Similarity,Bias,Rms,RmsTheory\n";
        const double binWidth = 2. / double(AnalyzeLshState::binCount);
        for (size_t bin = 0; bin < AnalyzeLshState::binCount; bin++) {
            if (s->sum0[bin] < 2) continue;
            const double pi = 3.141592653589793238462643383279502884;      // boost::math::double_constants::pi
            const double similarity = (double(bin) + 0.5) * binWidth - 1.;
            const double sinTheta = std::sqrt(1. - similarity * similarity);
            const double theta = std::acos(similarity);
            const double p = 1. - theta / pi;
            const double theoreticalSigma = pi * sinTheta * std::sqrt(p * (1. - p) / double(lshCount));
            const double s0 = double(s->sum0[bin]);
            const double average = s->sum1[bin] / s0;
            const double sigma = std::sqrt(s->sum2[bin] / s0);
            statsOut << similarity << ",";
            statsOut << average << ",";
            statsOut << sigma << ",";
            statsOut << theoreticalSigma << "\n";
        }
    }
    for (size_t bin = 0; bin < AnalyzeLshState::binCount; bin++) {
        if (sum0) sum0[bin] = s->sum0[bin];
        if (sum1) sum1[bin] = s->sum1[bin];
        if (sum2) sum2[bin] = s->sum2[bin];
    }
    s->csvOut.close();
    delete s;
    return ok;
}

}  // namespace em2
