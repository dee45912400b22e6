/* em2_lsh.h -- C ABI of the MI355X implementation of ExpressionMatrix2's LSH similar-pairs path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.  Each entry point names the
 * reference interface it replaces (file:line under the ExpressionMatrix2 source tree).  The reference is a C++
 * library with a pybind11 module; a maintainer binds these functions from ExpressionMatrixLsh.cpp /
 * PythonModule.cpp as shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns EM2_OK (0) or an EM2_ERROR_* code; em2_last_error() returns the message of the
 *     last failure on the calling thread.  Where the reference throws std::runtime_error with a fixed text
 *     ("Gene set X does not exist." ...) the message is that text and the code is EM2_ERROR_RUNTIME.
 *   - "dev" functions take DEVICE pointers valid on the current HIP device and a hipStream_t passed as void*;
 *     they enqueue work and return without synchronising.  They never allocate.
 *   - the other compute functions take HOST pointers, run on the current HIP device and return when the
 *     result is in the output buffers.  They fail with EM2_ERROR_NO_DEVICE when no GPU is present: there is
 *     no CPU fallback in this library.
 *   - cell ids are local to the cell set, gene ids local to the gene set (src/SimilarPairs.hpp:28-33).
 */
#ifndef EM2_LSH_H
#define EM2_LSH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EM2_OK 0
#define EM2_ERROR_INVALID_ARGUMENT 1
#define EM2_ERROR_NO_DEVICE 2
#define EM2_ERROR_HIP 3
#define EM2_ERROR_IO 4
#define EM2_ERROR_RUNTIME 5
#define EM2_ERROR_UNSUPPORTED 6

/* Layout of std::pair<CellId, float> (src/SimilarPairs.hpp:53-56): one stored neighbour of a cell. */
typedef struct em2_pair {
    uint32_t cell;
    float similarity;
} em2_pair;

/* Layout of std::pair<GeneId, float> (src/ExpressionMatrixSubset.hpp:36): one stored expression count. */
typedef struct em2_count {
    uint32_t gene;
    float count;
} em2_count;

int em2_abi_version(void);
const char* em2_last_error(void);

/* ------------------------------------------------------------------------------------------------------
 * Host-side pieces of the path (no GPU needed).
 * ------------------------------------------------------------------------------------------------------ */

/* Lsh::generateLshVectors (src/Lsh.cpp:68-113).  vectors is gene-major [geneCount][lshCount] like
 * Lsh::lshVectors (src/Lsh.hpp:104-113).  Generator: mt19937(seed) + the Box-Muller normal_distribution of
 * Boost <= 1.55, normalised per hyperplane.  Boost is not vendored by the reference and its
 * normal_distribution changed algorithm in 1.56, so a bit-for-bit match with a particular reference build is
 * not claimed: pass that build's own hyperplanes to em2_compute_signatures instead ("Oracle"). */
int em2_lsh_generate_vectors(uint32_t geneCount, uint32_t lshCount, uint32_t seed, double* vectors);

/* Lsh::computeSimilarityTable (src/Lsh.cpp:229-249): table[m] = cos(m*pi/lshCount), m = 0..lshCount. */
int em2_lsh_similarity_table(uint32_t lshCount, double* table);

/* MurmurHash64A as used by MemoryMapped::Vector::hash (src/MemoryMappedVector.hpp:715-723, seed 231). */
uint64_t em2_murmur_hash_64a(const void* key, int len, uint64_t seed);

/* ------------------------------------------------------------------------------------------------------
 * Device management.
 * ------------------------------------------------------------------------------------------------------ */
int em2_device_count(int* count);
int em2_set_device(int device);

/* ------------------------------------------------------------------------------------------------------
 * Host-buffer entry points: what a reference-side binding calls.
 * ------------------------------------------------------------------------------------------------------ */

/* ExpressionMatrixSubset::computeSums (src/ExpressionMatrixSubset.cpp:47-58) + Lsh::computeCellLshSignatures
 * (src/Lsh.cpp:118-224).  CSR: toc[cellCount+1] offsets into data, gene ids ascending within a cell and
 * < geneCount.  vectors as above.  signatures: cellCount * ((lshCount-1)/64+1) words, cell-major, bit i of a
 * cell in word i>>6 at bit position 63-(i&63) (src/BitSet.hpp:48-62). */
int em2_compute_signatures(const uint64_t* toc, const em2_count* data, uint32_t cellCount, uint32_t geneCount,
                           const double* vectors, uint32_t lshCount, uint64_t* signatures);

/* The pair loop, selection and storage order of ExpressionMatrix::findSimilarPairs4
 * (src/ExpressionMatrixLsh.cpp:200-285) + SimilarPairs::copy/sort (src/SimilarPairs.cpp:369-405).
 * pairs: cellCount*k slots, cell c at [c*k, c*k+usedCount[c]), sorted by similarity descending then cell id
 * ascending; unused slots are zero (as in a freshly created SimilarPairs-*-Pairs file). */
int em2_find_similar_pairs4(const uint64_t* signatures, uint32_t cellCount, uint32_t lshCount, uint32_t k,
                            double similarityThreshold, em2_pair* pairs, uint32_t* usedCount);

/* ExpressionMatrix::findSimilarPairs7 after its lookups (src/ExpressionMatrixLsh.cpp:563-690 with
 * findSimilarPairs7AssignCellsToBuckets :727-827): LSH buckets of several slice lengths (decreasing, each 1..64 bits;
 * buckets of slices with at least log2BucketCount bits are MurmurHash64A(value, seed 231) & (2^log2BucketCount - 1)),
 * per cell the first maxCheck distinct bucket-mates in (length, slice, id) order, of those the k with the fewest
 * mismatches among the ones with mismatchCount < Lsh::computeMismatchCountThresholdFromSimilarityThreshold
 * (src/Lsh.hpp:86-95), ascending (mismatch, id).  maxCheck 0 behaves as in the reference: no limit (:657 follows a push_back), except that
 * the walk ends at the first bucket that leaves the candidate list empty (:663).
 * Errors carry the reference's texts ("The slice lengths are not in decreasing order.", "Each slice length can be at
 * most 64 bits.").  Limits: k <= 4096, log2BucketCount <= 40, directly indexed slices <= 40 bits. */
int em2_find_similar_pairs7(const uint64_t* signatures, uint32_t cellCount, uint32_t lshCount, uint32_t k,
                            double similarityThreshold, const int32_t* sliceLengths, uint32_t sliceLengthCount,
                            uint32_t maxCheck, uint32_t log2BucketCount, em2_pair* pairs, uint32_t* usedCount);

/* ExpressionMatrixSubset + Lsh + findSimilarPairs4 in one call on host buffers (SURVEY.md 8(a) row a1 on the device:
 * src/ExpressionMatrixSubset.cpp:9-42 followed by src/Lsh.cpp:118-224 and src/ExpressionMatrixLsh.cpp:200-285): the
 * global CSR (CellExpressionCounts toc/data, global gene ids) restricted to the cells cellIds[0..cellCount) (NULL =
 * all cells in order) and to the genes with geneLocalIds[globalGeneId] != 0xffffffff (GeneSet-<name>-LocalIds), gene
 * ids remapped to those local ids; geneCount = size of the gene set = rows of `vectors`.  The restricted CSR never
 * exists on the host.  signatures may be NULL (not wanted); usedCount == NULL stops after the signatures
 * (computeLshSignatures), otherwise pairs / usedCount receive the SimilarPairs content as em2_find_similar_pairs4. */
int em2_subset_find_similar_pairs4(const uint64_t* globalToc, const em2_count* globalData, uint32_t globalCellCount,
                                   const uint32_t* cellIds, uint32_t cellCount, const uint32_t* geneLocalIds,
                                   uint32_t globalGeneCount, uint32_t geneCount, const double* vectors, uint32_t lshCount,
                                   uint64_t* signatures, uint32_t k, double similarityThreshold, em2_pair* pairs,
                                   uint32_t* usedCount);

/* ExpressionMatrix::findSimilarPairs5 (src/ExpressionMatrixLsh.cpp:355-496).  lshSliceLength must be in
 * [1,32] (the reference divides by zero for 0 and allocates 2^lshSliceLength vectors per slice). */
int em2_find_similar_pairs5(const uint64_t* signatures, uint32_t cellCount, uint32_t lshCount, uint32_t k,
                            double similarityThreshold, uint32_t lshSliceLength, uint64_t bucketOverflow,
                            em2_pair* pairs, uint32_t* usedCount);

/* ------------------------------------------------------------------------------------------------------
 * Device-resident entry points (multi-GPU sharding, benchmarking, callers that keep data in HBM).
 * ------------------------------------------------------------------------------------------------------ */

/* Bytes of device scratch em2_dev_compute_signatures needs. */
size_t em2_dev_compute_signatures_workspace(uint32_t cellCount, uint32_t lshCount);

/* Per-hyperplane-matrix auxiliary block (built once, reused for every shard / call): lshVectorsSums of
 * src/Lsh.cpp:137-144, the per-hyperplane maximum magnitude and a float copy of the matrix used by the screening
 * pass of em2_dev_compute_signatures. */
size_t em2_dev_vector_aux_bytes(uint32_t geneCount, uint32_t lshCount);
int em2_dev_prepare_vectors(const double* d_vectors, uint32_t geneCount, uint32_t lshCount, void* d_vectorAux,
                            void* stream);

/* As em2_compute_signatures, for the cellCount cells of a (shard of a) CSR in device memory.
 * d_vectorAux: the block em2_dev_prepare_vectors filled, or NULL.  With it (and lshCount a multiple of 4) most bits
 * are decided by a pass over the float copy of the hyperplanes under a rigorous error bound and only the undecided
 * 64-bit words are recomputed in the reference's sequential FP64 arithmetic; without it every bit is.  Both give
 * the same, reference-identical signatures. */
int em2_dev_compute_signatures(const uint64_t* d_toc, const em2_count* d_data, uint32_t cellCount,
                               uint32_t geneCount, const double* d_vectors, const void* d_vectorAux,
                               uint32_t lshCount, uint64_t* d_signatures, void* d_workspace,
                               size_t workspaceBytes, void* stream);

/* Which FIRST tier the last em2_dev_compute_signatures call on this workspace ran (the result does not depend on it; the time
 * does -- DESIGN.md 3.2).  The call waits for the whole device (hipDeviceSynchronize: it has no stream argument), and the
 * workspace must be untouched since that call -- the tier is read from a word the projection leaves in it.  haveVectorAux:
 * whether that call was given d_vectorAux.
 *   EM2_TIER_EXACT           the reference's arithmetic on every bit (no auxiliary block, or lshCount no multiple of 4)
 *   EM2_TIER_FLOAT           the float copy of the hyperplanes under its error bound (lshCount no multiple of 64)
 *   EM2_TIER_FIXED16_FLOAT   the 16-bit fixed-point copy, products summed in floating point (some count is no small integer)
 *   EM2_TIER_FIXED16_INTEGER the 16-bit fixed-point copy, products summed exactly in 32-bit integers (every count an integer of
 *                            at most 15 bits, every cell's sum of |count| at most 65 535: expression COUNTS)                    */
enum { EM2_TIER_EXACT = 0, EM2_TIER_FLOAT = 1, EM2_TIER_FIXED16_FLOAT = 2, EM2_TIER_FIXED16_INTEGER = 3 };
int em2_dev_compute_signatures_tier(const void* d_workspace, uint32_t cellCount, uint32_t lshCount, int haveVectorAux, int* tier);

/* Bytes of device scratch em2_dev_find_similar_pairs4 needs for rowCount rows. */
size_t em2_dev_find_similar_pairs4_workspace(uint32_t cellCount, uint32_t rowCount, uint32_t lshCount,
                                             uint32_t k);

/* ExpressionMatrixSubset on device-resident arrays (src/ExpressionMatrixSubset.cpp:9-42), two steps because the size
 * of the result is not known beforehand: _count writes d_toc[0..cellCount] (offsets of the restricted CSR, d_toc[cellCount]
 * = its entry count, read it back to size d_data), _fill writes the entries.  d_cellIds NULL = all cells in order. */
size_t em2_dev_subset_workspace(uint32_t cellCount);
int em2_dev_subset_count(const uint64_t* d_globalToc, const em2_count* d_globalData, const uint32_t* d_cellIds,
                         uint32_t cellCount, const uint32_t* d_geneLocalIds, uint32_t globalGeneCount, uint64_t* d_toc,
                         void* d_workspace, size_t workspaceBytes, void* stream);
int em2_dev_subset_fill(const uint64_t* d_globalToc, const em2_count* d_globalData, const uint32_t* d_cellIds,
                        uint32_t cellCount, const uint32_t* d_geneLocalIds, uint32_t globalGeneCount, const uint64_t* d_toc,
                        em2_count* d_data, void* stream);

/* Which form of the scan em2_dev_find_similar_pairs4 runs for this shape -- information for benchmarks and logs, the
 * results are identical.  0: every row of the launch is compared with every column (cellCount*rowCount ordered
 * comparisons).  1: symmetric form, used when one launch holds all rows of a large problem: every unordered pair is
 * evaluated once, as in the reference's own loop (src/ExpressionMatrixLsh.cpp:218-263), and offered to both cells. */
int em2_dev_find_similar_pairs4_form(uint32_t cellCount, uint32_t rowCount);

/* The same question with the signature width: for 129..2048 bits the symmetric form contracts its pairs as FP4 dot
 * products on the matrix cores (3; up to 1024 bits 0 / 1 operands, popcount(a & b) - (popcount(a) + popcount(b)) / 2 =
 * -mismatches / 2; above, +-1 operands, 2048 - 2 * mismatches; both exact in f32)
 * and starts at 32768 cells instead of 131072.  A launch that is not symmetric -- a shard of the rows, SURVEY 8(e)'s
 * partitioning across GPUs -- takes the rows form on the matrix cores (4) from 2^31 (row, column) pairs on: every row walks
 * all columns in ascending order, which is the per-cell contract of src/ExpressionMatrixLsh.cpp:200-285 as it stands.
 * The last_launch query below reports what actually ran. */
int em2_dev_find_similar_pairs4_form_for(uint32_t cellCount, uint32_t rowCount, uint32_t lshCount);

/* Facts about the calling thread's last em2_dev_find_similar_pairs4 launch, for benchmarks: values[0] form (as above),
 * [1] duration in ms of the scan kernel proper when the launcher measured it with HIP events on the launch stream
 * (symmetric form, which synchronises anyway), else -1, [2] (64-row wave, column) steps executed, [3] symmetric
 * form: inbox entries sorted and replayed, [4] column segments, [5] cells whose rows scanned all columns, [6] pairs
 * contracted on the matrix cores and [7] the duration in ms of that kernel alone (form 3, the matrix-core form of the
 * symmetric scan: FP4 contraction, signatures zero-extended to 1024 or 2048 bits), [8] the shader clock in GHz that
 * kernel ran at (sums over its blocks of s_memtime and s_memrealtime ticks; 0 when unknown). */
int em2_dev_find_similar_pairs4_last_launch(double* values, uint32_t valueCount);


/* findSimilarPairs4 for the rows [rowBegin,rowEnd) of the cell set against all cellCount cells: the shard
 * one rank owns.  d_signatures holds ALL cellCount signatures (after the all-gather).  d_pairs has
 * (rowEnd-rowBegin)*k slots and d_usedCount (rowEnd-rowBegin) entries, indexed by row-rowBegin.
 * The first call for a given (lshCount, similarityThreshold) on a device builds and caches small lookup
 * tables (one synchronous allocation + copy); later calls only enqueue kernels. */
int em2_dev_find_similar_pairs4(const uint64_t* d_signatures, uint32_t cellCount, uint32_t rowBegin,
                                uint32_t rowEnd, uint32_t lshCount, uint32_t k, double similarityThreshold,
                                em2_pair* d_pairs, uint32_t* d_usedCount, void* d_workspace,
                                size_t workspaceBytes, void* stream);

/* findSimilarPairs7 on device-resident signatures for the cells [rowBegin,rowEnd) (buckets over all cells; rows
 * shard over ranks like em2_dev_find_similar_pairs5).  sliceLengths is a host array.  Allocates its own scratch and
 * synchronises the stream. */
int em2_dev_find_similar_pairs7(const uint64_t* d_signatures, uint32_t cellCount, uint32_t rowBegin, uint32_t rowEnd,
                                uint32_t lshCount, uint32_t k, double similarityThreshold, const int32_t* sliceLengths,
                                uint32_t sliceLengthCount, uint32_t maxCheck, uint32_t log2BucketCount, em2_pair* d_pairs,
                                uint32_t* d_usedCount, void* stream);

/* ---- findSimilarPairs4 across GPUs with every unordered pair evaluated once (one process per GPU) ----
 * The 64-cell blocks of the problem are dealt round-robin to the ranks (block g: rank g % world).  Every rank holds
 * ALL signatures (after the all-gather of the projection shards) and calls the four phases in order with the same
 * arguments; between the phases the CALLER runs the collectives on views of the workspace:
 *     phase 0;  all_reduce(MAX) of snap = int32[cellCount] at snapOffset
 *     phase 1;  all_reduce(MAX) of snap
 *     phase 2;  em2_dev_fsp4_sharded_status -> own entry count; all ranks agree on maxUsed = max of the counts, fill
 *               pool[used, maxUsed) with ~0 (pool = uint64[poolCapacity] at poolOffset), all_gather pool[0, maxUsed)
 *               into gathered = uint64[world*maxUsed] at gatheredOffset
 *     phase 3 with gatheredCount = world*maxUsed
 * or, when world is a power of two, exchanging only what each rank needs:
 *     phase 2;  status -> used;  phase 4 with gatheredCount = used: the pool entries grouped by the rank that owns their
 *               target cell ((key >> ownerShift) & (world-1)) in sorted = uint64[..] at sortedOffset; all_to_all of those
 *               groups into gathered; phase 3 with gatheredCount = entries received
 * d_pairs [cellCount][k] and d_usedCount [cellCount] are indexed by GLOBAL cell id; phase 3 fills the rows of the
 * cells this rank owns.  If any rank reports overflow the result is unusable and the caller falls back to
 * em2_dev_find_similar_pairs4 on row shards.  em2_dev_fsp4_sharded_plan: values[0] eligible (0: shape too small, use
 * the row-shard call), [1] workspace bytes (256-byte aligned allocation), [2] snapOffset, [3] poolOffset,
 * [4] poolCapacity, [5] gatheredOffset, [6] gatheredCapacity, [7] prefix cells, [8] blocks owned, [9] blocks,
 * [10] sortedOffset, [11] ownerShift.
 * No reference counterpart (the reference is single-threaded); results are those of src/ExpressionMatrixLsh.cpp:155-290. */
int em2_dev_fsp4_sharded_plan(uint32_t cellCount, uint32_t lshCount, uint32_t k, uint32_t rank, uint32_t world,
                              uint64_t* values, uint32_t valueCount);
int em2_dev_fsp4_sharded_phase(int phase, const uint64_t* d_signatures, uint32_t cellCount, uint32_t lshCount, uint32_t k,
                               double similarityThreshold, uint32_t rank, uint32_t world, em2_pair* d_pairs,
                               uint32_t* d_usedCount, void* d_workspace, size_t workspaceBytes, uint64_t gatheredCount,
                               void* stream);
int em2_dev_fsp4_sharded_status(uint32_t cellCount, uint32_t k, uint32_t rank, uint32_t world, const void* d_workspace,
                                void* stream, uint64_t* usedEntries, uint32_t* overflow);

/* ---- findSimilarPairs4 across the GPUs of a node from C or C++ (csrc/em2_dist.hip) ----
 * One process per GPU.  Every rank calls with the same arguments and its own shard of the signatures -- the cells
 * [rank * shard, min(cellCount, (rank + 1) * shard)), shard = ceil(cellCount / world), the contiguous ranges of north_star --
 * and ends with the SimilarPairs rows of exactly those cells.  The function issues every collective itself:
 *   rows form       all_gather of the signature shards, then em2_dev_find_similar_pairs4 on the rank's rows;
 *   symmetric form  (large problems; every unordered pair once across the ranks) all_gather, phases 0-3 of
 *                   em2_dev_fsp4_sharded_phase with two all_reduce(MAX) of the snapshots, one small all_gather by which the
 *                   ranks agree on entry counts and overflow, the all_to_all of the deferred candidates (all_gather when
 *                   world is not a power of two), and an all_to_all that moves the finished rows from the block-cyclic
 *                   owners to the contiguous ranges.  A pool overflow on any rank sends all ranks to the rows form.
 * em2_dist_find_similar_pairs4_form: 2 if a problem of this shape takes the symmetric form, else 0 (EM2_SHARDED_SCAN=0 and
 * EM2_SHARDED_MIN_CELLS as in expressionmatrix2_amd/sharded.py, whose DevicePipeline is this choreography in Python).
 *   d_localSignatures  [shard][words] device: the rank's signatures (the last rank's unused tail is ignored)
 *   d_allSignatures    [shard * world][words] device: receives all signatures (rows [0, cellCount) are the cells)
 *   d_pairs / d_usedCount  [rows][k] / [rows] device, rows = the rank's range
 *   d_workspace        em2_dist_find_similar_pairs4_workspace(...) bytes, device
 *   stageMs            NULL, or EM2_DIST_MS_COUNT doubles that receive the wall time per kind of stage; asking for them
 *                      synchronises the stream after every stage (measurements), NULL leaves the call asynchronous up to
 *                      the read-backs the exchange needs
 * em2_dist_find_similar_pairs4 takes an RCCL communicator (ncclComm_t, passed as void*: this header does not include
 * rccl.h).  RCCL is bound at run time -- the nccl* symbols already in the process (the library that made the communicator),
 * else librccl.so.1 -- so libem2lsh.so has no link-time dependency on it.  em2_dist_find_similar_pairs4_with takes the
 * transport as a table instead (MPI, a test harness, ...): every function works on DEVICE buffers, is called by all ranks
 * in the same order, is ordered after earlier work on `stream` and before later work on it, and returns 0 or an error.
 * all_gather's send buffer may be the rank's own slot of the receive buffer.  all_to_all_v gets byte counts and byte
 * offsets per peer (host arrays of `world` entries).
 * A HIP or transport error on one rank ends that rank's call with an error while the others may wait in a collective:
 * abort the communicator (ncclCommAbort), as in any RCCL program.  No reference counterpart (the reference is one thread). */
typedef struct em2_collectives {
    void* context;
    int world;
    int rank;
    int (*all_gather)(void* context, const void* d_send, void* d_recv, size_t bytesPerRank, void* stream);
    int (*all_reduce_max_i32)(void* context, void* d_buffer, size_t count, void* stream);
    int (*all_to_all_v)(void* context, const void* d_send, const uint64_t* sendBytes, const uint64_t* sendOffsets,
                        void* d_recv, const uint64_t* recvBytes, const uint64_t* recvOffsets, void* stream);
} em2_collectives;
#define EM2_DIST_MS_GATHER_SIGNATURES 0
#define EM2_DIST_MS_SCAN 1
#define EM2_DIST_MS_ALL_REDUCE 2
#define EM2_DIST_MS_EXCHANGE 3
#define EM2_DIST_MS_REDISTRIBUTE 4
#define EM2_DIST_MS_COUNT 5
size_t em2_dist_find_similar_pairs4_workspace(uint32_t cellCount, uint32_t lshCount, uint32_t k, uint32_t rank, uint32_t world);
int em2_dist_find_similar_pairs4_form(uint32_t cellCount, uint32_t lshCount, uint32_t k, uint32_t world);
int em2_dist_find_similar_pairs4(void* ncclCommunicator, const uint64_t* d_localSignatures, uint32_t cellCount, uint32_t lshCount,
                                 uint32_t k, double similarityThreshold, uint64_t* d_allSignatures, em2_pair* d_pairs,
                                 uint32_t* d_usedCount, void* d_workspace, size_t workspaceBytes, void* stream, double* stageMs);
int em2_dist_find_similar_pairs4_with(const em2_collectives* collectives, const uint64_t* d_localSignatures, uint32_t cellCount,
                                      uint32_t lshCount, uint32_t k, double similarityThreshold, uint64_t* d_allSignatures,
                                      em2_pair* d_pairs, uint32_t* d_usedCount, void* d_workspace, size_t workspaceBytes,
                                      void* stream, double* stageMs);

/* Synchronises `stream` and reports whether the last em2_dev_find_similar_pairs4 on this workspace completed: the
 * scan hands per-row state from one column segment to the next between waves, and a hand-off wait that exceeds
 * ~4 s raises an error word instead of hanging the GPU (never observed).  rowCount and k as in that call. */
int em2_dev_find_similar_pairs4_status(const void* d_workspace, uint32_t rowCount, uint32_t k, void* stream);

/* findSimilarPairs5 for the cells [rowBegin,rowEnd) of the cell set (the bucket tables are built over all
 * cellCount signatures, which every rank holds after the all-gather).  Unlike the other dev entry points this one
 * sizes its scratch from the data (bucket sizes are only known after the sort), so it allocates and frees device
 * memory itself and returns after synchronising the stream. */
int em2_dev_find_similar_pairs5(const uint64_t* d_signatures, uint32_t cellCount, uint32_t rowBegin,
                                uint32_t rowEnd, uint32_t lshCount, uint32_t k, double similarityThreshold,
                                uint32_t lshSliceLength, uint64_t bucketOverflow, em2_pair* d_pairs,
                                uint32_t* d_usedCount, void* stream);

/* Facts about the calling thread's last em2_dev_find_similar_pairs5 / em2_find_similar_pairs5, for benchmarks:
 * values[0] candidate ids gathered from the buckets (duplicates and the cell itself included: each costs the filter
 * one look at the sorted list, each distinct one a gather of 8*W signature bytes), [1] cells queried, [2] slices
 * (lshCount / lshSliceLength, src/ExpressionMatrixLsh.cpp:355), [3] batches, [4] / [5] ms of the candidate filter and
 * of the selection, summed over the batches (HIP events on the launch stream), [6] the DISTINCT candidates of all cells
 * (the sizes of the duplicate-free unions, the cell itself included): the signatures the filter actually gathers. */
int em2_dev_find_similar_pairs5_last_launch(double* values, uint32_t valueCount);

/* findSimilarPairs5 keeps its device scratch (bucket tables, candidate ids, candidate lists: about 10 GB at a million cells x
 * 2048 bits) between calls of the process, and em2_subset_find_similar_pairs4 the device copy of its result and its scan
 * workspace (12 GB at a million cells: it starts with room for 512 deferred candidates per cell and takes the 19 GB of the
 * device-level call once a launch of the process has overflowed that), because allocating gigabytes costs anything between
 * 2 ms and 2.7 s per call depending on the state of the host; at most EM2_SCRATCH_CACHE_MB megabytes are kept (default: a
 * sixteenth of the device's memory -- 18 GB of an MI355X's 288 --, the oldest blocks making room for newer ones; 0 = none).
 * This call frees what is kept.  The reference has no counterpart (its tables are std::vectors of the call,
 * src/ExpressionMatrixLsh.cpp:377-389). */
void em2_dev_release_scratch(void);

/* ------------------------------------------------------------------------------------------------------
 * SURVEY.md 8(f), first "next" row: the consumer of SimilarPairs.
 * ------------------------------------------------------------------------------------------------------ */

/* CellGraph::CellGraph (src/CellGraph.cpp:33-117; reached from ExpressionMatrix::createCellGraph,
 * src/ExpressionMatrix.cpp:1795-1845): the edges of the k-NN cell similarity graph in the order the reference adds
 * them.  Vertex v is the v-th cell of graphCellSet (the order of add_vertex).  pairs / usedCount / k are the content
 * of a SimilarPairs object whose cell set (sorted global ids) is similarPairsCellSet.  maxConnectivity 0 means no
 * limit, as in the reference (the size test at :101 follows a push_back).  The three output arrays need room for
 * graphCellCount*min(maxConnectivity ? maxConnectivity : k, k) edges; *edgeCount receives the number written.
 * Host buffers. */
int em2_cell_graph_edges(const em2_pair* pairs, const uint32_t* usedCount, uint32_t similarPairsCellCount, uint32_t k,
                         const uint32_t* similarPairsCellSet, const uint32_t* graphCellSet, uint32_t graphCellCount,
                         double similarityThreshold, uint32_t maxConnectivity, uint32_t* edgeVertex0,
                         uint32_t* edgeVertex1, float* edgeSimilarity, uint64_t* edgeCount);

/* The same with the SimilarPairs content still on the device (d_pairs / d_usedCount as em2_dev_find_similar_pairs4
 * left them; the cell sets are host arrays as above; the three edge arrays may be host OR device memory): the consumer of a
 * device-resident findSimilarPairs4 does not move 8 * k * cells bytes over PCIe twice, and with device edge arrays handed on
 * to em2_cell_graph_label_propagation (which takes host or device edge arrays as well) the edges never leave the device. */
int em2_dev_cell_graph_edges(const em2_pair* d_pairs, const uint32_t* d_usedCount, uint32_t similarPairsCellCount, uint32_t k,
                             const uint32_t* similarPairsCellSet, const uint32_t* graphCellSet, uint32_t graphCellCount,
                             double similarityThreshold, uint32_t maxConnectivity, uint32_t* edgeVertex0,
                             uint32_t* edgeVertex1, float* edgeSimilarity, uint64_t* edgeCount);

/* ExpressionMatrix::analyzeLsh (src/ExpressionMatrixLsh.cpp:1244-1367; Python: src/PythonModule.cpp:940-944): the
 * quality of the LSH similarity against the exact one, over every unordered pair of cells of an expression matrix
 * subset.  SURVEY.md 8(f).
 *   toc / data      : the subset's counts (what em2_matrix_subset / em2_dev_subset_* produce: local gene ids ascending)
 *   geneCount       : size of the gene set -- the n of the correlation coefficient (src/ExpressionMatrixSubset.cpp:115)
 *   signatures      : the cells' LSH signatures (em2_compute_signatures), lshCount bits each
 *   globalCellIds   : the cell set (column 3 and 4 of the pairs csv)
 *   seed            : seeds the mt19937 that downsamples the pairs csv (one draw per pair, in pair order)
 *   pairsCsvPath / statisticsCsvPath : the reference writes "Lsh-analysis.csv" and "LSH-analysis-statistics.csv" into
 *                     the working directory; statisticsCsvPath may be NULL
 *   sum0 / sum1 / sum2 (each 200 entries, may be NULL): per bin of exact similarity the number of pairs, the sum of
 *                     (lsh - exact) and the sum of its square, accumulated in the reference's pair order
 *   exactSimilarity / lshSimilarity (cellCount * (cellCount - 1) / 2 entries, may be NULL): the values per pair
 * The scalar products of the pairs and the mismatch counts are computed on the device (bit-identical to the
 * reference's merge loop: float products, double sum, ascending gene); what the reference's pair order defines (the
 * bins' double sums, the random draws, the csv) is walked on the host in that order.  Bit-exact in all outputs.
 * Errors: EM2_ERROR_RUNTIME "bin < binCount" where the reference's CZI_ASSERT (:1322) throws (a pair with exact
 * similarity 1, or without variance); the files are then incomplete, as the reference's are. */
int em2_analyze_lsh(const uint64_t* toc, const em2_count* data, uint32_t cellCount, uint32_t geneCount,
                    const uint64_t* signatures, uint32_t lshCount, const uint32_t* globalCellIds, uint32_t seed,
                    double csvDownsample, const char* pairsCsvPath, const char* statisticsCsvPath,
                    uint64_t* sum0, double* sum1, double* sum2, double* exactSimilarity, double* lshSimilarity);

/* CellGraph::labelPropagationClustering (src/CellGraph.cpp:443-612, ClusterTable src/CellGraph.hpp:50-121; reached
 * from ExpressionMatrix::createClusterGraph, src/ExpressionMatrix.cpp:2145-2149) over the graph em2_cell_graph_edges
 * built.  SURVEY.md 8(f) row 2.  vertexCellIds[v] is the cell id of vertex v, in add_vertex order with removed
 * isolated vertices left out; edges index that array and are in add_edge order.  clusterIds[v] receives the cluster
 * of vertex v after the reference's renumbering (0 = largest; equal sizes by decreasing original label).
 * *iterationCount (may be NULL) receives the number of iterations that ran.  The reference's schedule is serial by
 * definition (each update reads labels written earlier in the same std::shuffle(std::mt19937(seed)) order, and
 * the float weights accumulate in that order); the device runs it with a schedule that reproduces exactly those labels
 * (csrc/em2_cluster.hip).  Host buffers.  em2_dev_cell_graph_label_propagation takes the three edge arrays in DEVICE memory,
 * as em2_dev_cell_graph_edges left them when given device output arrays (they are not validated again): with the pairs of
 * em2_dev_find_similar_pairs4 the chain findSimilarPairs4 -> createCellGraph -> labelPropagationClustering then keeps pairs
 * and edges on the device from end to end; vertexCellIds and clusterIds stay host arrays. */
int em2_cell_graph_label_propagation(const uint32_t* vertexCellIds, uint32_t vertexCount, const uint32_t* edgeVertex0,
                                     const uint32_t* edgeVertex1, const float* edgeSimilarity, uint64_t edgeCount,
                                     uint64_t seed, uint64_t stableIterationCountThreshold,
                                     uint64_t maxIterationCount, uint32_t* clusterIds, uint64_t* iterationCount);
int em2_dev_cell_graph_label_propagation(const uint32_t* vertexCellIds, uint32_t vertexCount, const uint32_t* d_edgeVertex0,
                                     const uint32_t* d_edgeVertex1, const float* d_edgeSimilarity, uint64_t edgeCount,
                                     uint64_t seed, uint64_t stableIterationCountThreshold,
                                     uint64_t maxIterationCount, uint32_t* clusterIds, uint64_t* iterationCount);

/* ------------------------------------------------------------------------------------------------------
 * ExpressionMatrix-level entry points: the methods the reference binds to Python (src/PythonModule.cpp),
 * operating by NAME on a data directory in the reference's memory-mapped formats.  Results are files in
 * that directory (SimilarPairs-<name>-{Info,Pairs,CellInfo}, Lsh-<name>-{Info,Signatures}), byte-compatible
 * with what the reference writes.  Errors the reference reports with std::runtime_error come back as
 * EM2_ERROR_RUNTIME with the reference's message text.
 * ------------------------------------------------------------------------------------------------------ */
typedef struct em2_matrix em2_matrix;

/* ExpressionMatrix(directoryName, allowReadOnly) for an existing directory (src/ExpressionMatrix.cpp:109-160);
 * opens only what the LSH path reads: CellExpressionCounts.{toc,data}, CellSet-*, GeneSet-*-{GlobalIds,LocalIds}. */
int em2_matrix_open(const char* directoryName, em2_matrix** matrix);
void em2_matrix_close(em2_matrix* matrix);

/* ExpressionMatrix::findSimilarPairs4 (src/ExpressionMatrixLsh.cpp:155-303; bound at src/PythonModule.cpp:802-824,
 * defaults AllGenes, AllCells, k=100, similarityThreshold=0.2, lshCount=1024, seed=231). */
int em2_matrix_find_similar_pairs4(em2_matrix* matrix, const char* geneSetName, const char* cellSetName,
                                   const char* similarPairsName, size_t k, double similarityThreshold,
                                   size_t lshCount, unsigned int seed);

/* ExpressionMatrix::computeLshSignatures (src/ExpressionMatrixLsh.cpp:1150-1192; src/PythonModule.cpp:945-953). */
int em2_matrix_compute_lsh_signatures(em2_matrix* matrix, const char* geneSetName, const char* cellSetName,
                                      const char* lshName, size_t lshCount, unsigned int seed);

/* ExpressionMatrix::analyzeLsh(geneSetName, cellSetName, lshCount, seed, csvDownsample) (src/ExpressionMatrixLsh.cpp:
 * 1244-1367, bound without defaults at src/PythonModule.cpp:940-944): subset + signatures + em2_analyze_lsh.  The
 * reference writes "Lsh-analysis.csv" and "LSH-analysis-statistics.csv" into the working directory; outputDirectory
 * (NULL or "": the working directory) says where. */
int em2_matrix_analyze_lsh(em2_matrix* matrix, const char* geneSetName, const char* cellSetName, size_t lshCount,
                           unsigned int seed, double csvDownsample, const char* outputDirectory);

/* ExpressionMatrix::findSimilarPairs5 (src/ExpressionMatrixLsh.cpp:312-501; src/PythonModule.cpp:852-865,
 * default bucketOverflow=1000). */
int em2_matrix_find_similar_pairs5(em2_matrix* matrix, const char* geneSetName, const char* cellSetName,
                                   const char* lshName, const char* similarPairsName, size_t k,
                                   double similarityThreshold, size_t lshSliceLength, size_t bucketOverflow);

/* ExpressionMatrix::findSimilarPairs7 (src/ExpressionMatrixLsh.cpp:507-703; bound at src/PythonModule.cpp:882-897). */
int em2_matrix_find_similar_pairs7(em2_matrix* matrix, const char* geneSetName, const char* cellSetName,
                                   const char* lshName, const char* similarPairsName, size_t k,
                                   double similarityThreshold, const int32_t* lshSliceLengths, uint32_t sliceLengthCount,
                                   uint32_t maxCheck, size_t log2BucketCount);

/* ExpressionMatrix::removeSimilarPairs (src/ExpressionMatrixFindSimilarPairs.cpp:126-135). */
int em2_matrix_remove_similar_pairs(em2_matrix* matrix, const char* similarPairsName);

/* ExpressionMatrixSubset (src/ExpressionMatrixSubset.cpp:9-42) as plain arrays, for drivers that shard the work
 * themselves: first call with toc == NULL to get the sizes, then with toc[cellCount+1] and data[nnz]. */
int em2_matrix_subset(em2_matrix* matrix, const char* geneSetName, const char* cellSetName, uint32_t* geneCount,
                      uint32_t* cellCount, uint64_t* nnz, uint64_t* toc, em2_count* data);

/* SimilarPairs files: constructor + copy (src/SimilarPairs.cpp:11-42,369-379) from already selected and
 * sorted pairs; and the existing-object constructor (:47-83) with its hash / length checks.  For the read,
 * pass pairs/usedCount == NULL to get only k and cellCount. */
int em2_similar_pairs_write(const char* directoryName, const char* similarPairsName, const char* geneSetName,
                            const char* cellSetName, size_t k, uint32_t cellCount, const em2_pair* pairs,
                            const uint32_t* usedCount);
int em2_similar_pairs_read(const char* directoryName, const char* similarPairsName, uint64_t* k,
                           uint64_t* cellCount, em2_pair* pairs, uint32_t* usedCount);

/* SimilarPairs::Info (src/SimilarPairs.hpp:188-198): k, cell count and the names of the gene / cell set the object was
 * built on; name buffers must hold 256 bytes.  Same consistency checks as em2_similar_pairs_read. */
int em2_similar_pairs_info(const char* directoryName, const char* similarPairsName, uint64_t* k, uint64_t* cellCount,
                           char* geneSetName, char* cellSetName);

/* A cell set of the data directory (CellSet-<name>, src/CellSets.hpp:15): pass ids == NULL to get the count. */
int em2_matrix_cell_set(em2_matrix* matrix, const char* cellSetName, uint32_t* count, uint32_t* ids);

/* Lsh files Lsh-<name>-{Info,Signatures} (src/Lsh.hpp:136-141, src/Lsh.cpp:26-28,48-64,148). */
int em2_lsh_write(const char* directoryName, const char* lshName, uint64_t cellCount, uint64_t lshCount,
                  const uint64_t* signatures);
int em2_lsh_read(const char* directoryName, const char* lshName, uint64_t* cellCount, uint64_t* lshCount,
                 uint64_t* signatures);

/* Tooling for tests and benchmarks -- NOT a reference API.  Creates a directory that holds exactly the files
 * the LSH path reads (the reference's own constructor needs more files than that). */
int em2_tool_create_directory(const char* directoryName, uint32_t geneCount, uint32_t cellCount,
                              const uint64_t* toc, const em2_count* data);
int em2_tool_add_gene_set(const char* directoryName, const char* name, const uint32_t* sortedGlobalIds,
                          uint32_t count);
int em2_tool_add_cell_set(const char* directoryName, const char* name, const uint32_t* sortedCellIds,
                          uint32_t count);

#ifdef __cplusplus
}
#endif

#endif
