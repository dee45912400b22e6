// em2_fsp7.hip -- findSimilarPairs7 on gfx950, bit-identical to
// src/ExpressionMatrixLsh.cpp:507-827 (findSimilarPairs7 + findSimilarPairs7AssignCellsToBuckets).
//
// Reference: for every slice length (decreasing) and every slice of that length, cells are bucketed by the slice
// value (first bit most significant, BitSet.hpp:108-119) or, when the slice has at least log2BucketCount bits, by
// MurmurHash64A(&value, 8, 231) & (2^log2BucketCount - 1) (:815-819); buckets list cells in ascending id.  For a
// cell the buckets are walked in (length, slice) order; cells not seen before and different from the cell itself
// become candidates until maxCheck of them exist (:636-668); candidates with mismatchCount < mismatchCountThreshold
// (Lsh.hpp:86-95) are neighbours; the k smallest (mismatch, id) pairs are kept and sorted (:675-676: keepBest with
// std::less on a total order, so the result is the k smallest, whatever nth_element does internally) and stored with
// float(similarityTable[mismatch]).
//
// Device formulation (HBM / L2 latency bound integer work):
//   1. one key per (table, cell), table = (length, slice): (table << 40) | bucket; a stable rocPRIM radix sort groups
//      every bucket contiguously with ascending cell ids (the reference's 2^b vector headers are never built);
//   2. run boundaries give the bucket of every (table, cell);
//   3. one wave per cell walks its buckets 64 members at a time.  "Seen" is a per-wave bitmap in HBM (cleared after
//      each cell from the candidate list, like the reference's cellMap); the maxCheck cut is exact because the fresh
//      members of a chunk get their sequence numbers from a ballot prefix.  Mismatch counts by one lane per
//      candidate; the k smallest keys by bisection on the key value, then a rank sort.

#include "em2_device.h"

#include <cstring>

#include <rocprim/rocprim.hpp>

#include <vector>

namespace em2 {
namespace {

constexpr uint32_t kBucketBits = 40;
constexpr uint32_t kMaxK = 4096;

__device__ __forceinline__ uint64_t murmur8(uint64_t value)
{
    const uint64_t m = 0xc6a4a7935bd1e995ULL;
    const int r = 47;
    uint64_t h = 231ULL ^ (8ULL * m);
    uint64_t k = value;
    k *= m;
    k ^= k >> r;
    k *= m;
    h ^= k;
    h *= m;
    h ^= h >> r;
    h *= m;
    h ^= h >> r;
    return h;
}

__device__ __forceinline__ uint64_t sliceValue64(const uint64_t* sig, uint32_t slice, uint32_t length)
{
    const uint32_t b0 = slice * length;
    const uint32_t w = b0 >> 6;
    const uint32_t o = b0 & 63u;
    uint64_t window = sig[w] << o;
    if (o + length > 64u) window |= sig[w + 1] >> (64u - o);
    return window >> (64u - length);
}

__global__ void __launch_bounds__(256)
tableKeysKernel(const uint64_t* __restrict__ sig, uint32_t cellCount, uint32_t words, const uint32_t* __restrict__ tableLength,
                const uint32_t* __restrict__ tableSlice, uint32_t tableCount, uint32_t log2BucketCount,
                uint64_t* __restrict__ keys, uint32_t* __restrict__ cells)
{
    const uint64_t total = uint64_t(tableCount) * cellCount;
    const uint64_t bucketMask = (1ULL << log2BucketCount) - 1ULL;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += uint64_t(gridDim.x) * blockDim.x) {
        const uint32_t t = uint32_t(i / cellCount);
        const uint32_t c = uint32_t(i % cellCount);
        const uint32_t length = tableLength[t];
        const uint64_t value = sliceValue64(sig + size_t(c) * words, tableSlice[t], length);
        const uint64_t bucket = length < log2BucketCount ? value : (murmur8(value) & bucketMask);
        keys[i] = (uint64_t(t) << kBucketBits) | bucket;
        cells[i] = c;
    }
}

__global__ void __launch_bounds__(256)
runFlagsKernel7(const uint64_t* __restrict__ sortedKeys, uint64_t total, uint32_t* __restrict__ flags)
{
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += uint64_t(gridDim.x) * blockDim.x) {
        flags[i] = (i == 0 || sortedKeys[i] != sortedKeys[i - 1]) ? 1u : 0u;
    }
}

__global__ void __launch_bounds__(256)
runTablesKernel7(const uint64_t* __restrict__ sortedKeys, const uint32_t* __restrict__ sortedCells,
                 const uint32_t* __restrict__ flags, const uint32_t* __restrict__ scan, uint64_t total, uint32_t cellCount,
                 uint32_t* __restrict__ runStart, uint32_t* __restrict__ runOfTableCell)
{
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += uint64_t(gridDim.x) * blockDim.x) {
        const uint32_t run = scan[i] - 1u;
        if (flags[i]) runStart[run] = uint32_t(i);
        const uint32_t t = uint32_t(sortedKeys[i] >> kBucketBits);
        runOfTableCell[size_t(t) * cellCount + sortedCells[i]] = run;
        if (i == total - 1) runStart[run + 1u] = uint32_t(total);
    }
}

__device__ __forceinline__ uint32_t lanesBelow7(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
}

__device__ __forceinline__ void waveFence7()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
    __builtin_amdgcn_wave_barrier();
}

// One wave per cell (grid-stride over the rows).
__global__ void __launch_bounds__(64)
traverseKernel(const uint64_t* __restrict__ sig, uint32_t words, uint32_t cellCount, uint32_t rowBegin, uint32_t rowEnd,
               const uint32_t* __restrict__ runOfTableCell, const uint32_t* __restrict__ runStart,
               const uint32_t* __restrict__ sortedCells, uint32_t tableCount, uint32_t maxCheck, bool stopIfEmpty,
               uint64_t mismatchThreshold,
               uint32_t capacity, uint32_t bitmapWords, uint32_t* bitmaps, uint32_t* candidates, uint64_t* neighbors,
               const uint32_t* __restrict__ keyOfMismatch, const float* __restrict__ keySimilarity, uint32_t k,
               PairOut* __restrict__ outPairs, uint32_t* __restrict__ outUsed)
{
    __shared__ uint64_t selected[kMaxK];
    const uint32_t lane = threadIdx.x;
    uint32_t* bitmap = bitmaps + size_t(blockIdx.x) * bitmapWords;
    uint32_t* cand = candidates + size_t(blockIdx.x) * capacity;
    uint64_t* neigh = neighbors + size_t(blockIdx.x) * capacity;
    for (uint32_t row = rowBegin + blockIdx.x; row < rowEnd; row += gridDim.x) {
        const uint64_t* mine = sig + size_t(row) * words;
        uint32_t count = 0, n = 0;
        bool done = false;
        for (uint32_t t = 0; t < tableCount && !done; ++t) {
            const uint32_t run = runOfTableCell[size_t(t) * cellCount + row];
            const uint32_t begin = runStart[run];
            const uint32_t size = runStart[run + 1u] - begin;
            for (uint32_t base = 0; base < size && !done; base += 64u) {
                const uint32_t i = base + lane;
                const bool valid = i < size;
                const uint32_t member = valid ? sortedCells[begin + i] : 0u;
                bool fresh = false;
                if (valid && member != row) {                                               // :639-644
                    const uint32_t word = __hip_atomic_load(bitmap + (member >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    fresh = ((word >> (member & 31u)) & 1u) == 0u;
                }
                const uint64_t mask = __builtin_amdgcn_ballot_w64(fresh);
                const uint32_t remaining = maxCheck - count;
                const uint32_t sequence = lanesBelow7(mask);
                const bool accept = fresh && sequence < remaining;
                bool pass = false;
                uint32_t m = 0;
                if (accept) {
                    __hip_atomic_fetch_or(bitmap + (member >> 5), 1u << (member & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    cand[count + sequence] = member;
                    const uint64_t* other = sig + size_t(member) * words;
                    for (uint32_t w = 0; w < words; ++w) m += uint32_t(__builtin_popcountll(mine[w] ^ other[w]));
                    pass = uint64_t(m) < mismatchThreshold;                                 // :649
                }
                const uint64_t passMask = __builtin_amdgcn_ballot_w64(pass);
                if (pass) neigh[n + lanesBelow7(passMask)] = (uint64_t(m) << 32) | member;
                n += uint32_t(__builtin_popcountll(passMask));
                const uint32_t freshCount = uint32_t(__builtin_popcountll(mask));
                count += freshCount < remaining ? freshCount : remaining;
                done = count == maxCheck;                                                   // :652-661
                waveFence7();       // the bitmap updates of this chunk are visible to the next chunk's loads
            }
            // maxCheck == 0: the test after the member loop (:663) holds while no candidate has been found at all
            if (stopIfEmpty && count == 0u) done = true;
        }
        waveFence7();

        // keepBest(neighbors, k, less) + sort (:675-676): the k smallest (mismatch, id) keys, ascending
        uint32_t kept = n;
        if (n > k) {
            uint64_t lo = 0, hi = ~0ull;
            while (lo < hi) {
                const uint64_t mid = lo + (hi - lo) / 2u;
                uint32_t below = 0;
                for (uint32_t base = 0; base < n; base += 64u) {
                    const uint32_t i = base + lane;
                    const bool in = i < n && __hip_atomic_load(neigh + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= mid;
                    below += uint32_t(__builtin_popcountll(__builtin_amdgcn_ballot_w64(in)));
                }
                if (below >= k) hi = mid;
                else lo = mid + 1u;
            }
            uint32_t out = 0;
            for (uint32_t base = 0; base < n; base += 64u) {
                const uint32_t i = base + lane;
                uint64_t key = 0;
                bool in = false;
                if (i < n) {
                    key = __hip_atomic_load(neigh + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    in = key <= lo;
                }
                const uint64_t inMask = __builtin_amdgcn_ballot_w64(in);
                if (in) selected[out + lanesBelow7(inMask)] = key;
                out += uint32_t(__builtin_popcountll(inMask));
            }
            kept = k;           // keys are distinct (distinct cell ids), so exactly k are <= the k-th smallest
        } else {
            for (uint32_t i = lane; i < n; i += 64u) selected[i] = __hip_atomic_load(neigh + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        PairOut* out = outPairs + size_t(row - rowBegin) * k;
        for (uint32_t i = lane; i < kept; i += 64u) {
            const uint64_t key = selected[i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < kept; ++j) rank += uint32_t(selected[j] < key);
            PairOut po;
            po.cell = uint32_t(key);
            po.similarity = keySimilarity[keyOfMismatch[uint32_t(key >> 32)]];             // :679-684
            out[rank] = po;
        }
        for (uint32_t i = kept + lane; i < k; i += 64u) {
            PairOut zero;
            zero.cell = 0u;
            zero.similarity = 0.0f;
            out[i] = zero;
        }
        if (lane == 0u) outUsed[row - rowBegin] = kept;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();

        // clear the "seen" bits of this cell (:687-689)
        for (uint32_t i = lane; i < count; i += 64u) {
            const uint32_t member = __hip_atomic_load(cand + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_and(bitmap + (member >> 5), ~(1u << (member & 31u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        waveFence7();
    }
}

struct Buffer7 {
    void* p = nullptr;
    ~Buffer7() { if (p) (void)hipFree(p); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; } }
    hipError_t allocate(size_t bytes) { release(); return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

uint32_t gridFor7(uint64_t n)
{
    const uint64_t blocks = (n + 255) / 256;
    return uint32_t(blocks > 16384 ? 16384 : (blocks ? blocks : 1));
}

#define EM2_TRY7(call)                       \
    do {                                     \
        hipError_t em2Err_ = (call);         \
        if (em2Err_ != hipSuccess) return em2Err_; \
    } while (0)

}  // namespace

uint32_t fsp7MaxK() { return kMaxK; }

// sliceLengths: validated by the caller (decreasing, 1..64).  mismatchThreshold as computed by
// Lsh::computeMismatchCountThresholdFromSimilarityThreshold (may be ~0).  Allocates its own scratch, synchronises.
hipError_t runFsp7(const uint64_t* d_sig, uint32_t cellCount, uint32_t rowBegin, uint32_t rowEnd, uint32_t lshCount, uint32_t k,
                   const int32_t* sliceLengths, uint32_t sliceLengthCount, uint32_t maxCheck, uint32_t log2BucketCount,
                   uint64_t mismatchThreshold, const DeviceTables& tables, PairOut* d_pairs, uint32_t* d_used, hipStream_t stream)
{
    const uint32_t rowCount = rowEnd - rowBegin;
    const uint32_t words = (lshCount - 1u) / 64u + 1u;
    if (rowCount == 0) return hipSuccess;
    if (k > kMaxK || log2BucketCount > kBucketBits) return hipErrorInvalidValue;
    EM2_TRY7(hipMemsetAsync(d_used, 0, size_t(rowCount) * sizeof(uint32_t), stream));
    if (k) EM2_TRY7(hipMemsetAsync(d_pairs, 0, size_t(rowCount) * k * sizeof(PairOut), stream));
    std::vector<uint32_t> tableLength, tableSlice;
    for (uint32_t li = 0; li < sliceLengthCount; ++li) {
        const uint32_t length = uint32_t(sliceLengths[li]);
        if (length < log2BucketCount && length > kBucketBits) return hipErrorInvalidValue;      // bucket ids must fit
        for (uint32_t si = 0; si < lshCount / length; ++si) {
            tableLength.push_back(length);
            tableSlice.push_back(si);
        }
    }
    const uint32_t tableCount = uint32_t(tableLength.size());
    if (tableCount == 0 || cellCount == 0 || k == 0) return hipStreamSynchronize(stream);
    // maxCheck == 0: the size test after the slice loop of a length (:667) also holds for a length that has no slice
    // at all (slice length above lshCount; such lengths come first) -- nothing has been found yet, the walk ends.
    if (maxCheck == 0 && sliceLengthCount > 0 && lshCount / uint32_t(sliceLengths[0]) == 0) return hipStreamSynchronize(stream);
    const uint64_t total = uint64_t(tableCount) * cellCount;
    if (total >= 0xffffffffULL || tableCount >= (1u << 23)) return hipErrorInvalidValue;

    Buffer7 dLength, dSlice, keysA, keysB, cellsA, cellsB, temp;
    EM2_TRY7(dLength.allocate(tableCount * sizeof(uint32_t)));
    EM2_TRY7(dSlice.allocate(tableCount * sizeof(uint32_t)));
    EM2_TRY7(hipMemcpyAsync(dLength.p, tableLength.data(), tableCount * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    EM2_TRY7(hipMemcpyAsync(dSlice.p, tableSlice.data(), tableCount * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    EM2_TRY7(keysA.allocate(total * sizeof(uint64_t)));
    EM2_TRY7(keysB.allocate(total * sizeof(uint64_t)));
    EM2_TRY7(cellsA.allocate(total * sizeof(uint32_t)));
    EM2_TRY7(cellsB.allocate(total * sizeof(uint32_t)));
    tableKeysKernel<<<gridFor7(total), 256, 0, stream>>>(d_sig, cellCount, words, dLength.as<uint32_t>(), dSlice.as<uint32_t>(), tableCount,
                                                          log2BucketCount, keysA.as<uint64_t>(), cellsA.as<uint32_t>());
    EM2_TRY7(hipGetLastError());
    size_t tempBytes = 0;
    EM2_TRY7(rocprim::radix_sort_pairs(nullptr, tempBytes, keysA.as<uint64_t>(), keysB.as<uint64_t>(), cellsA.as<uint32_t>(),
                                       cellsB.as<uint32_t>(), size_t(total), 0u, 64u, stream));
    EM2_TRY7(temp.allocate(tempBytes));
    EM2_TRY7(rocprim::radix_sort_pairs(temp.p, tempBytes, keysA.as<uint64_t>(), keysB.as<uint64_t>(), cellsA.as<uint32_t>(),
                                       cellsB.as<uint32_t>(), size_t(total), 0u, 64u, stream));
    const uint64_t* sortedKeys = keysB.as<uint64_t>();
    const uint32_t* sortedCells = cellsB.as<uint32_t>();

    Buffer7 flags, scan, runStart, runOf, scanTemp;
    EM2_TRY7(flags.allocate(total * sizeof(uint32_t)));
    EM2_TRY7(scan.allocate(total * sizeof(uint32_t)));
    EM2_TRY7(runStart.allocate((total + 1) * sizeof(uint32_t)));
    EM2_TRY7(runOf.allocate(total * sizeof(uint32_t)));
    runFlagsKernel7<<<gridFor7(total), 256, 0, stream>>>(sortedKeys, total, flags.as<uint32_t>());
    EM2_TRY7(hipGetLastError());
    size_t scanBytes = 0;
    EM2_TRY7(rocprim::inclusive_scan(nullptr, scanBytes, flags.as<uint32_t>(), scan.as<uint32_t>(), size_t(total), rocprim::plus<uint32_t>(), stream));
    EM2_TRY7(scanTemp.allocate(scanBytes));
    EM2_TRY7(rocprim::inclusive_scan(scanTemp.p, scanBytes, flags.as<uint32_t>(), scan.as<uint32_t>(), size_t(total), rocprim::plus<uint32_t>(), stream));
    runTablesKernel7<<<gridFor7(total), 256, 0, stream>>>(sortedKeys, sortedCells, flags.as<uint32_t>(), scan.as<uint32_t>(), total, cellCount,
                                                           runStart.as<uint32_t>(), runOf.as<uint32_t>());
    EM2_TRY7(hipGetLastError());
    EM2_TRY7(hipStreamSynchronize(stream));
    keysA.release(); cellsA.release(); flags.release(); scan.release(); scanTemp.release(); temp.release(); keysB.release();

    // per-wave scratch: "seen" bitmap, candidate list (for the clean-up), neighbour keys
    // maxCheck == 0 never equals a size after a push_back (:657), so nothing stops the member loops; the same test
    // after each bucket (:663, :667) does hold while the candidate list is still empty, and ends the walk there.
    const uint32_t limit = (maxCheck == 0 || maxCheck > cellCount) ? cellCount : maxCheck;
    const uint32_t effectiveMaxCheck = maxCheck == 0 ? 0xffffffffu : maxCheck;
    uint32_t waves = rowCount < 8192u ? rowCount : 8192u;
    const uint32_t bitmapWords = (cellCount + 31u) / 32u;
    while (waves > 64u && uint64_t(waves) * (uint64_t(limit) * 12u + uint64_t(bitmapWords) * 4u) > (8ull << 30)) waves /= 2u;
    Buffer7 bitmaps, candidates, neighbors;
    EM2_TRY7(bitmaps.allocate(size_t(waves) * bitmapWords * sizeof(uint32_t)));
    EM2_TRY7(candidates.allocate(size_t(waves) * limit * sizeof(uint32_t)));
    EM2_TRY7(neighbors.allocate(size_t(waves) * limit * sizeof(uint64_t)));
    EM2_TRY7(hipMemsetAsync(bitmaps.p, 0, size_t(waves) * bitmapWords * sizeof(uint32_t), stream));
    traverseKernel<<<waves, 64, 0, stream>>>(d_sig, words, cellCount, rowBegin, rowEnd, runOf.as<uint32_t>(), runStart.as<uint32_t>(),
                                             sortedCells, tableCount, effectiveMaxCheck, maxCheck == 0, mismatchThreshold, limit, bitmapWords,
                                             bitmaps.as<uint32_t>(), candidates.as<uint32_t>(), neighbors.as<uint64_t>(),
                                             tables.keyOfMismatch, tables.keySimilarity, k, d_pairs, d_used);
    EM2_TRY7(hipGetLastError());
    return hipStreamSynchronize(stream);
}

}  // namespace em2
