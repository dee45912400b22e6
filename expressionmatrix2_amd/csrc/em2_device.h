// em2_device.h -- internal declarations shared by the HIP translation units and the C ABI glue.
#ifndef EM2_DEVICE_H
#define EM2_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "em2_select.h"

namespace em2 {

// Layout-compatible with std::pair<CellId, float> (src/SimilarPairs.hpp:53-56) and em2_pair (include/em2_lsh.h).
struct PairOut {
    uint32_t cell;
    float similarity;
};

// Layout-compatible with std::pair<GeneId, float> (src/ExpressionMatrixSubset.hpp:36) and em2_count.
struct CountIn {
    uint32_t gene;
    float count;
};

// Device-resident copies of the tables of em2_tables.h.
struct DeviceTables {
    const uint32_t* keyOfMismatch;   // [lshCount+1]
    const int32_t* acceptMaxByKey;   // [keyCount]
    const float* keySimilarity;      // [keyCount]
    int32_t mGlobal;
    int32_t mMaxInitial;
    bool identityKeys;               // keyOfMismatch[m] == m for every m
};

// Signature words per cell padded to a supported kernel width (in 32-bit words); 0 if lshCount is too large.
uint32_t paddedDwords(uint32_t lshCount);

// Copies [cellCount][wordCount] uint64 signatures into the zero-padded [cellCount][paddedDwords/2] layout.
hipError_t launchRepackSignatures(const uint64_t* src, uint32_t cellCount, uint32_t wordCount,
                                  uint32_t* dst, uint32_t paddedDw, hipStream_t stream);

// Largest k the scan kernel supports (limited by the per-wave LDS staging area).
uint32_t fsp4MaxK();

// findSimilarPairs4 for rows [rowBegin,rowEnd) against all cellCount columns.
//   sig32    [cellCount][paddedDw] device
//   buffers  [(rowEnd-rowBegin)][2k] Entry, device scratch
//   outPairs [(rowEnd-rowBegin)][k], outUsed [(rowEnd-rowBegin)]
//   control  fsp4ControlBytes(rows) bytes of device scratch for the persistent kernel's hand-off state (NULL
//            selects the simple one-wave-per-row-block kernel)
hipError_t launchFsp4Scan(const uint32_t* sig32, uint32_t paddedDw, uint32_t cellCount,
                          uint32_t rowBegin, uint32_t rowEnd, uint32_t k, const DeviceTables& tables,
                          Entry* buffers, PairOut* outPairs, uint32_t* outUsed, void* control,
                          hipStream_t stream, void* symmetricWorkspace = nullptr);
size_t fsp4ControlBytes(uint32_t rowCount);
// Extra scratch for the symmetric (each unordered pair once) form of the scan, which launchFsp4Scan uses when all
// rows of the problem are in one launch and this workspace is given; 0 when that form would not be used.
size_t fsp4SymmetricBytes(uint32_t cellCount, uint32_t rowCount, uint32_t paddedDw);
bool fsp4UsesSymmetricScan(uint32_t cellCount, uint32_t rowCount, uint32_t paddedDw);
// Whether a launch of rowCount rows against cellCount columns that is not symmetric takes the rows form on the matrix
// cores (form 4: every row walks all columns as FP4 dot products); fsp4SymmetricBytes then sizes its workspace.
bool fsp4UsesRowsMatrixScan(uint32_t cellCount, uint32_t rowCount, uint32_t paddedDw);
// Whether signatures of this padded width take the matrix-core form of the symmetric scan (EM2_SCAN_MATRIX included).
bool fsp4MatrixFormWanted(uint32_t paddedDw);
struct Fsp4LaunchInfo {
    int form;                 // 0 ordered rows x columns, 1 symmetric, 2 sharded symmetric, 3 symmetric on the matrix cores, 4 rows x columns on the matrix cores
    double scanKernelMs;      // duration of the scan kernel proper when the launcher measured it (symmetric form), else -1
    double waveColumnSteps;   // (64-row wave, column) steps executed: x 64 lanes x 2*W32 = v_xor/v_bcnt lane-ops
    double inboxEntries;      // symmetric form: entries (incl. chunk padding) sorted and replayed
    double segments;
    double fullRowCells;
    double matrixPairs;       // form 3: (row, column) pairs contracted on the matrix cores
    double matrixKernelMs;    // form 3: duration of fsp4ScanMatrixKernel alone
    double matrixClockGHz;    // form 3: the shader clock that kernel ran at (block-lifetime s_memtime / s_memrealtime sums), 0 if unknown
};
Fsp4LaunchInfo fsp4LastLaunchInfo();      // of the calling thread's last launchFsp4Scan
hipError_t readFsp4Error(const void* control, uint32_t rowCount, hipStream_t stream, uint32_t* error);

// Sharded symmetric scan (em2_scan.hip): plan of one rank, one launch per phase; the collectives between the phases
// belong to the caller.
struct Fsp4ShardPlan {
    bool eligible;
    uint32_t cellCount, world, rank, k;
    uint32_t blocks, prefixBlocks, prefixCells, ownBlocks, maxOwnBlocks, ownPrefixBlocks;
    uint64_t capLocal, capGathered;
    size_t sortTempBytes;
    size_t offLists, offControl, offSnap, offTable, offInboxControl, offPool, offFragments, offTerms, rankBytes, offGathered, offSorted, offTemp, totalBytes;
};
Fsp4ShardPlan fsp4ShardPlan(uint32_t cellCount, uint32_t k, uint32_t rank, uint32_t world);
hipError_t launchFsp4ShardPhase(const Fsp4ShardPlan& plan, int phase, const uint32_t* sig32, uint32_t paddedDw,
                                const DeviceTables& tables, void* rankWs, void* exchangeWs, PairOut* outPairs,
                                uint32_t* outUsed, uint64_t gatheredCount, hipStream_t stream);
hipError_t readFsp4ShardStatus(const Fsp4ShardPlan& plan, const void* rankWs, hipStream_t stream, uint64_t* used,
                               uint32_t* overflow, uint32_t* error);

// ExpressionMatrixSubset::computeSums (sum1 only) -> mean = sum1 / geneCount, per cell.
hipError_t launchCellMeans(const uint64_t* toc, const CountIn* data, uint32_t cellCount, uint32_t geneCount,
                           double* means, hipStream_t stream);

// lshVectorsSums[i] = sum over genes (ascending) of vectors[g][i]   (Lsh.cpp:137-144)
hipError_t launchVectorSums(const double* vectors, uint32_t geneCount, uint32_t lshCount, double* sums,
                            hipStream_t stream);

// Lsh::computeCellLshSignatures for the cellCount cells of the CSR (toc[0..cellCount], data); writes
// signatures[cellCount][wordCount].  means[cellCount] from launchCellMeans.
hipError_t launchProjection(const uint64_t* toc, const CountIn* data, uint32_t cellCount, const double* vectors,
                            const double* vectorSums, const double* means, uint32_t lshCount,
                            uint64_t* signatures, hipStream_t stream);

// Screened projection (em2_project.hip): aux = per-bit sums and max |U|, and a float copy of the hyperplanes,
// built once per hyperplane matrix; requires lshCount % 4 == 0.  Result identical to launchProjection.
size_t vectorAuxBytes(uint32_t geneCount, uint32_t lshCount);
hipError_t launchPrepareVectors(const double* vectors, uint32_t geneCount, uint32_t lshCount, void* aux,
                                hipStream_t stream);
size_t projectionScreenedWorkspaceBytes(uint32_t cellCount, uint32_t lshCount);
hipError_t launchProjectionScreened(const uint64_t* toc, const CountIn* data, uint32_t cellCount, uint32_t geneCount,
                                    const double* vectors, const void* aux, uint32_t lshCount, uint64_t* signatures,
                                    void* workspace, hipStream_t stream);

// findSimilarPairs5 (src/ExpressionMatrixLsh.cpp:355-496): bucket tables over all cells, results for the cells
// [rowBegin,rowEnd).  Allocates its own scratch and synchronises the stream.  q = lshSliceLength in [1,32].
// What the calling thread's last runFsp5 did (benchmarks): candidate ids gathered from the buckets (duplicates and the
// cell itself included), cells queried, slices, batches, and the HIP-event durations of the candidate filter (the
// gather of candidate signatures + popcounts) and of the selection over all batches.
struct Fsp5LaunchInfo {
    double gatheredCandidates, cells, sliceCount, batches, filterMs, selectMs;
    double distinctCandidates;      // what the filter read: the sizes of the cells' duplicate-free unions (the cell itself included); -1 in the sort form
};
Fsp5LaunchInfo fsp5LastLaunchInfo();
// Frees the device scratch runFsp5 keeps between calls (em2_fsp5.hip: ScratchCache).
void fsp5ReleaseScratch();
void fsp4SetInboxEntriesPerCell(uint32_t entries);       // (this thread's symmetric scans: 0 = the default of 1024; em2_scan_symmetric.hip)
// (em2_fsp5.hip) blocks of that cache for the library's other host-buffer entry points
void* scratchTake(size_t bytes, size_t* got);
void scratchGive(void* p, size_t bytes);

hipError_t runFsp5(const uint64_t* d_sig, uint32_t cellCount, uint32_t rowBegin, uint32_t rowEnd, uint32_t lshCount,
                   uint32_t k, uint32_t q, uint64_t bucketOverflow, const DeviceTables& tables, PairOut* d_pairs,
                   uint32_t* d_used, hipStream_t stream);

// ExpressionMatrixSubset::ExpressionMatrixSubset (src/ExpressionMatrixSubset.cpp:9-42) on the device (em2_subset.hip).
// cellIds == NULL means all cells in order; geneLocalIds[globalGeneId] = local id or 0xffffffff.
size_t subsetWorkspaceBytes(uint32_t cellCount);
hipError_t launchSubsetCount(const uint64_t* globalToc, const CountIn* globalData, const uint32_t* cellIds, uint32_t cellCount,
                             const uint32_t* geneLocalIds, uint32_t globalGeneCount, uint64_t* toc, void* workspace,
                             size_t workspaceBytes, hipStream_t stream);
hipError_t launchSubsetFill(const uint64_t* globalToc, const CountIn* globalData, const uint32_t* cellIds, uint32_t cellCount,
                            const uint32_t* geneLocalIds, uint32_t globalGeneCount, const uint64_t* toc, CountIn* outData,
                            hipStream_t stream);

// findSimilarPairs7 (src/ExpressionMatrixLsh.cpp:507-827), results for the cells [rowBegin,rowEnd).  Allocates its own
// scratch and synchronises the stream.
uint32_t fsp7MaxK();
hipError_t runFsp7(const uint64_t* d_sig, uint32_t cellCount, uint32_t rowBegin, uint32_t rowEnd, uint32_t lshCount, uint32_t k,
                   const int32_t* sliceLengths, uint32_t sliceLengthCount, uint32_t maxCheck, uint32_t log2BucketCount,
                   uint64_t mismatchThreshold, const DeviceTables& tables, PairOut* d_pairs, uint32_t* d_used, hipStream_t stream);

// CellGraph::CellGraph (src/CellGraph.cpp:33-117): edges of the k-NN graph in the reference's insertion order.
hipError_t runCellGraphEdges(const PairOut* pairs, const uint32_t* usedCount, uint32_t spCellCount, uint32_t k,
                             const uint32_t* spCellSet, const uint32_t* graphCellSet, const uint32_t* graphSortedIds,
                             const uint32_t* graphVertexOfSorted, uint32_t graphCellCount, double similarityThreshold,
                             uint32_t maxConnectivity, uint32_t* edge0, uint32_t* edge1, float* edgeSimilarity,
                             uint64_t* edgeCountHost, hipStream_t stream, bool spConsecutive, uint32_t spFirst, bool graphConsecutive,
                             uint32_t graphFirst);

// CellGraph::labelPropagationClustering (src/CellGraph.cpp:443-612) on the device (em2_cluster.hip).  Host buffers;
// labels[v] = the raw label (a cell id) of vertex v after the last iteration.
hipError_t runLabelPropagation(const uint32_t* vertexCellIds, uint32_t vertexCount, const uint32_t* edgeVertex0,
                               const uint32_t* edgeVertex1, const float* edgeSimilarity, uint64_t edgeCount,
                               const uint32_t* shuffleInput, uint64_t seed, uint64_t stableIterationCountThreshold,
                               uint64_t maxIterationCount, uint32_t* labels, uint64_t* iterationCount, uint32_t* error,
                               hipStream_t stream);

// em2_analyze.hip: ExpressionMatrix::analyzeLsh (src/ExpressionMatrixLsh.cpp:1244-1367).  Device half: per unordered pair
// of rows [rowBegin, rowEnd) x the cells above each, the sparse scalar product of the two cells' counts (float products,
// double sum, ascending gene order) and the mismatch count of their signatures.  Host half: the order-defined rest.
uint64_t analyzePairCount(uint32_t cellCount, uint32_t rowBegin, uint32_t rowEnd);
size_t analyzeScratchBytes(uint32_t geneCount, uint32_t rowCount);
hipError_t launchAnalyzePairs(const uint64_t* toc, const CountIn* data, uint32_t cellCount, uint32_t geneCount,
                              const uint64_t* signatures, uint32_t words, uint32_t rowBegin, uint32_t rowEnd, void* scratch,
                              double* scalarProducts, uint32_t* mismatches, hipStream_t stream);
struct AnalyzeLshState;
AnalyzeLshState* analyzeLshBegin(uint32_t lshCount, uint32_t seed, const char* pairsCsvPath);
bool analyzeLshRows(AnalyzeLshState* s, const double* sums, uint32_t cellCount, uint32_t geneCount, const uint32_t* globalCellIds,
                    uint32_t rowBegin, uint32_t rowEnd, const double* scalarProducts, const uint32_t* mismatches, double csvDownsample,
                    double* exactOut, double* lshOut);
bool analyzeLshEnd(AnalyzeLshState* s, uint32_t lshCount, const char* statisticsCsvPath, uint64_t* sum0, double* sum1, double* sum2);

}  // namespace em2

#endif
