// em2_fsp5.hip -- findSimilarPairs5 on gfx950, bit-identical to src/ExpressionMatrixLsh.cpp:355-496.
//
// Reference algorithm: sliceCount = lshCount / lshSliceLength; slice s of a signature is bits
// [s*q, (s+1)*q), first bit most significant (BitSet.hpp:111-119); tables[s][value] lists, in ascending id
// order, the cells whose slice s equals value (:377-389).  For a cell c the candidates are the ascending,
// duplicate-free union (multipleSetUnion.hpp:44-76) of its sliceCount buckets, skipping buckets larger than
// bucketOverflow when that is non-zero (:414-431); candidates != c with similarityTable[mismatch] >
// similarityThreshold are collected in ascending id order (:436-445), cut once with keepBest(k) (:457), stored
// and sorted (:489-496).
//
// Device formulation (HBM-bound integer work; no matrix cores):
//   1. one key per (slice, cell): (slice << 32) | value.  A stable LSD radix sort (rocPRIM) over all keys
//      groups every bucket contiguously with ascending cell ids -- the 2^q-entry vector tables of the
//      reference (2.6 GB of vector headers at q = 20) are never materialised;
//   2. run boundaries -> bucket id of every (slice, cell); bucket sizes apply the overflow rule;
//   3. per batch of cells: gather the members of the cell's buckets, segmented radix sort per cell, then one
//      wave per cell removes duplicates, gathers the candidates' signatures, counts mismatches and keeps
//      m <= mGlobal in ascending id order (ballot + prefix popcount) -- filterKernel, no LDS, full occupancy --
//      and a second kernel runs the exact keepBest emulation in LDS and writes the sorted result (selectKernel).
// The sorts are library primitives (rocPRIM); everything specific to the path is hand-written here.

#include "em2_device.h"
#include "em2_select_wave.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>      // rocprim/iterator/texture_cache_iterator.hpp calls memset without including it

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <mutex>
#include <vector>

namespace em2 {
namespace {

constexpr uint32_t kSelectLdsEntries = 4096;        // lists up to this length are cut in 48 KB of LDS ...
constexpr uint32_t kSelectLdsEntriesBig = 12288;    // ... up to this length in 144 KB, longer ones in HBM

__device__ __forceinline__ uint32_t sliceValue(const uint64_t* sig, uint32_t slice, uint32_t q)
{
    const uint32_t b0 = slice * q;
    const uint32_t w = b0 >> 6;
    const uint32_t o = b0 & 63u;
    uint64_t window = sig[w] << o;
    if (o + q > 64u) window |= sig[w + 1] >> (64u - o);
    return uint32_t(window >> (64u - q));
}

__global__ void __launch_bounds__(256)
sliceKeysKernel(const uint64_t* __restrict__ sig, uint32_t cellCount, uint32_t words, uint32_t q, uint32_t sliceCount,
                uint64_t* __restrict__ keys, uint32_t* __restrict__ cells)
{
    const uint64_t total = uint64_t(sliceCount) * cellCount;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += uint64_t(gridDim.x) * blockDim.x) {
        const uint32_t s = uint32_t(i / cellCount);
        const uint32_t c = uint32_t(i % cellCount);
        keys[i] = (uint64_t(s) << 32) | sliceValue(sig + size_t(c) * words, s, q);
        cells[i] = c;
    }
}

__global__ void __launch_bounds__(256)
runFlagsKernel(const uint64_t* __restrict__ sortedKeys, uint64_t total, uint32_t* __restrict__ flags)
{
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += uint64_t(gridDim.x) * blockDim.x) {
        flags[i] = (i == 0 || sortedKeys[i] != sortedKeys[i - 1]) ? 1u : 0u;
    }
}

// runIndex[i] = (inclusive scan of flags)[i] - 1.  Records where each run starts and which run every
// (slice, cell) belongs to.
__global__ void __launch_bounds__(256)
runTablesKernel(const uint64_t* __restrict__ sortedKeys, const uint32_t* __restrict__ sortedCells,
                const uint32_t* __restrict__ flags, const uint32_t* __restrict__ scan, uint64_t total,
                uint32_t cellCount, uint32_t sliceCount, uint32_t* __restrict__ runStart, uint32_t* __restrict__ runOfSliceCell)
{
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += uint64_t(gridDim.x) * blockDim.x) {
        const uint32_t run = scan[i] - 1u;
        if (flags[i]) runStart[run] = uint32_t(i);
        const uint32_t s = uint32_t(sortedKeys[i] >> 32);
        runOfSliceCell[size_t(sortedCells[i]) * sliceCount + s] = run;        // [cell][slice]: a cell's descriptors are one contiguous read
        if (i == total - 1) runStart[run + 1u] = uint32_t(total);
    }
}

// Number of bucket members a cell will gather (before de-duplication), with the overflow rule applied.
__global__ void __launch_bounds__(256)
candidateCountKernel(const uint32_t* __restrict__ runOfSliceCell, const uint32_t* __restrict__ runStart,
                     uint32_t cellCount, uint32_t sliceCount, uint64_t bucketOverflow, uint64_t* __restrict__ counts)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cellCount) return;
    uint64_t n = 0;
    for (uint32_t s = 0; s < sliceCount; ++s) {
        const uint32_t run = runOfSliceCell[size_t(c) * sliceCount + s];
        const uint64_t size = runStart[run + 1u] - runStart[run];
        if (bucketOverflow == 0 || size <= bucketOverflow) n += size;       // ExpressionMatrixLsh.cpp:419
    }
    counts[c] = n;
}

// ---- the order in which the filter visits the cells of a batch (a schedule: it changes no result) ----
// The filter gathers the signatures of a cell's candidates, and the candidates of a cell are mostly the cells that are similar
// to it -- which are also the candidates of every other cell similar to it.  Visited in id order, the waves in flight at any
// time gather from all over the signature array (256 MB at a million cells x 2048 bits: the Infinity Cache's size); visited
// group by group, with every XCD working through groups of its own, the waves of an XCD gather from the few megabytes of
// signatures their group shares, which its L2 holds.  A group label needs no clustering: label(c) = the smallest cell id in
// any of c's buckets (bucket members ascend: the first member of each), followed through two rounds of pointer jumping
// (label(label(c)) twice), pulls the cells that are connected through shared buckets to the smallest ids of their
// neighbourhood -- a handful of values per cluster.  Any labels would give the same SimilarPairs.
__global__ void __launch_bounds__(256)
neighbourhoodLabelKernel(const uint32_t* __restrict__ runOfSliceCell, const uint32_t* __restrict__ runStart,
                         const uint32_t* __restrict__ sortedCells, uint32_t cellCount, uint32_t sliceCount, uint64_t bucketOverflow,
                         uint32_t* __restrict__ labels, uint64_t* __restrict__ counts)
{
    // one wave per cell, lane = slice (the dependent loads of a bucket's size and first member, 64 slices at a time); the same
    // walk gives the number of bucket members the cell will gather (candidateCountKernel's result; labels may be null)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= cellCount) return;
    uint32_t best = c;
    uint64_t members = 0;
    for (uint32_t s = lane; s < sliceCount; s += 64u) {
        const uint32_t run = runOfSliceCell[size_t(c) * sliceCount + s];
        const uint32_t begin = runStart[run];
        const uint64_t size = runStart[run + 1u] - begin;
        if (bucketOverflow != 0 && size > bucketOverflow) continue;                // ExpressionMatrixLsh.cpp:419
        members += size;
        if (labels) {
            const uint32_t first = sortedCells[begin];
            best = first < best ? first : best;
        }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t other = uint32_t(__shfl_xor(int(best), d, 64));
        best = other < best ? other : best;
        members += uint64_t(__shfl_xor((long long)members, d, 64));
    }
    if (lane == 0u) {
        if (labels) labels[c] = best;
        counts[c] = members;
    }
}

__global__ void __launch_bounds__(256)
jumpLabelsKernel(const uint32_t* __restrict__ in, uint32_t cellCount, uint32_t* __restrict__ out)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < cellCount) out[c] = in[in[c]];
}

__global__ void __launch_bounds__(256)
batchOrderKeysKernel(const uint32_t* __restrict__ labels, uint32_t batchBegin, uint32_t batchCells, uint32_t* __restrict__ keys,
                     uint32_t* __restrict__ locals)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batchCells) return;
    keys[i] = labels[batchBegin + i];
    locals[i] = i;
}

// One wave per cell of the batch: copy the members of its buckets into its segment.  The three dependent loads that
// describe a bucket (run of the slice's value, its begin, its end) are taken for 64 slices at once, lane = slice, and
// the copies go four buckets at a time (loads first, then stores): slice after slice the kernel sat in that latency,
// 102 times per cell (24 ms of the 398 at 1M cells x 2048 bits).
__global__ void __launch_bounds__(256)
gatherKernel(const uint32_t* __restrict__ runOfSliceCell, const uint32_t* __restrict__ runStart,
             const uint32_t* __restrict__ sortedCells, uint32_t cellCount, uint32_t sliceCount, uint64_t bucketOverflow,
             uint32_t batchBegin, uint32_t batchCells, const uint32_t* __restrict__ segmentBegin,
             uint32_t* __restrict__ candidates)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t local = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (local >= batchCells) return;
    const uint32_t c = batchBegin + local;
    uint32_t out = segmentBegin[local];
    for (uint32_t first = 0; first < sliceCount; first += 64u) {
        const uint32_t s = first + lane;
        uint32_t begin = 0, size = 0;
        if (s < sliceCount) {
            const uint32_t run = runOfSliceCell[size_t(c) * sliceCount + s];
            begin = runStart[run];
            size = runStart[run + 1u] - begin;
            if (bucketOverflow != 0 && uint64_t(size) > bucketOverflow) size = 0u;
        }
        // where each slice's bucket goes: exclusive prefix sum of the sizes, in slice order
        uint32_t inclusive = size;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t below = uint32_t(__shfl_up(int(inclusive), d, 64));
            if (lane >= uint32_t(d)) inclusive += below;
        }
        const uint32_t offset = out + inclusive - size;
        const uint32_t slices = sliceCount - first < 64u ? sliceCount - first : 64u;
        for (uint32_t j = 0; j < slices; j += 4u) {
            uint32_t from[4], count[4], to[4], value[4];
#pragma unroll
            for (uint32_t q = 0; q < 4u; ++q) {
                const uint32_t source = j + q < slices ? j + q : j;          // (uniform)
                from[q] = uint32_t(__builtin_amdgcn_readlane(int(begin), int(source)));
                count[q] = j + q < slices ? uint32_t(__builtin_amdgcn_readlane(int(size), int(source))) : 0u;
                to[q] = uint32_t(__builtin_amdgcn_readlane(int(offset), int(source)));
            }
#pragma unroll
            for (uint32_t q = 0; q < 4u; ++q) value[q] = lane < count[q] ? sortedCells[from[q] + lane] : 0u;
#pragma unroll
            for (uint32_t q = 0; q < 4u; ++q) {
                if (lane < count[q]) candidates[to[q] + lane] = value[q];
                for (uint32_t i = 64u + lane; i < count[q]; i += 64u) candidates[to[q] + i] = sortedCells[from[q] + i];     // (buckets beyond 64 members)
            }
        }
        out += uint32_t(__builtin_amdgcn_readlane(int(inclusive), 63));
    }
}


// The union of a cell's buckets without a sort (multipleSetUnion.hpp:44-76 asks for the ascending, duplicate-free union): a
// bitmap over the cell ids in LDS.  One block of 1024 threads per cell, two blocks per CU; the id space is covered in passes of
// kUnionWords * 32 ids (two passes at a million cells), so any cell count works.
//   * descriptors: thread s < sliceCount holds (begin, size) of the cell's bucket in slice s -- loaded one cell ahead, so the
//     two dependent loads behind them are not waited for -- and a prefix sum of the sizes numbers the bucket members 0 .. M-1;
//   * members: thread t takes members t, t + 1024, ... (binary search of the prefix sums for the bucket), the first
//     kUnionHeld of them with all loads in flight at once and kept in registers for both passes;
//   * per pass every member inside the pass's id range sets its bit (ds_or_b32); thread t then owns the words
//     [17 t, 17 t + 17) -- an odd stride, so the 64 lanes of a wave read 64 different banks -- takes them into registers,
//     zeroes them and counts its bits; one block-wide prefix sum gives every thread the position of its first id; the ids go
//     into the (now free) bitmap area in ascending order and from there to the cell's segment of `candidates` with
//     coalesced stores.
// The cell itself stays in the list (the reference takes the union first and drops the cell afterwards, :417-439: the filter
// does).  Replaces gatherKernel + rocPRIM's segmented radix sort (14 + 77 ms of the 390 at 1M cells x 2048 bits, q = 20);
// more than kUnionSlices slices keep that form.
constexpr uint32_t kUnionThreads = 1024;
constexpr uint32_t kUnionWordsPerThread = 17;
constexpr uint32_t kUnionWords = kUnionThreads * kUnionWordsPerThread;      // 17408 words = 557,056 ids per pass, 68 KB
constexpr uint32_t kUnionSlices = 256;
constexpr uint32_t kUnionHeld = 6;
constexpr uint32_t kUnionStaged = 2304;         // ids of a pass staged for coalesced stores (a pass of a million-cell problem has about 2000)
constexpr size_t kUnionLdsBytes = size_t(kUnionWords) * 4u + (2u * kUnionSlices + 1u) * 4u + 32u * 4u + kUnionStaged * 4u;

__global__ void __launch_bounds__(kUnionThreads, 8)          // two blocks per CU: at most 64 vector registers
unionKernel(const uint32_t* __restrict__ runOfCellSlice, const uint32_t* __restrict__ runStart,
            const uint32_t* __restrict__ sortedCells, uint32_t cellCount, uint32_t sliceCount, uint64_t bucketOverflow,
            uint32_t batchBegin, uint32_t batchCells, const uint32_t* __restrict__ segmentBegin,
            uint32_t* __restrict__ candidates, uint32_t* __restrict__ distinctCounts)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t unionLds[];
    uint32_t* bitmap = unionLds;                                    // kUnionWords, all zero between two uses
    uint32_t* bucketBegin = unionLds + kUnionWords;                 // kUnionSlices
    uint32_t* memberPrefix = bucketBegin + kUnionSlices;            // kUnionSlices + 1: members in the buckets before slice s
    uint32_t* waveTotals = memberPrefix + kUnionSlices + 1u;        // 16 (of 32)
    uint32_t* staging = waveTotals + 32u;                           // kUnionStaged
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    for (uint32_t w = t; w < kUnionWords; w += kUnionThreads) bitmap[w] = 0u;
    const uint32_t passes = (cellCount + kUnionWords * 32u - 1u) / (kUnionWords * 32u);
    // the descriptor this thread holds for the cell about to be processed
    uint32_t nextBegin = 0, nextSize = 0;
    auto loadDescriptor = [&](uint32_t local) {
        nextBegin = nextSize = 0u;
        if (local < batchCells && t < sliceCount) {
            const uint32_t run = runOfCellSlice[size_t(batchBegin + local) * sliceCount + t];
            nextBegin = runStart[run];
            nextSize = runStart[run + 1u] - nextBegin;
            if (bucketOverflow != 0 && uint64_t(nextSize) > bucketOverflow) nextSize = 0u;             // ExpressionMatrixLsh.cpp:419
        }
    };
    loadDescriptor(blockIdx.x);
    __syncthreads();
    for (uint32_t local = blockIdx.x; local < batchCells; local += gridDim.x) {
        uint32_t* out = candidates + segmentBegin[local];
        // ---- the buckets' begins and the prefix sums of their sizes (threads 0 .. 255 = waves 0 .. 3) ----
        uint32_t inclusiveSize = nextSize;
        if (t < kUnionSlices) {
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t below = uint32_t(__shfl_up(int(inclusiveSize), d, 64));
                if (lane >= uint32_t(d)) inclusiveSize += below;
            }
            if (lane == 63u) waveTotals[wave] = inclusiveSize;
            bucketBegin[t] = nextBegin;
        }
        __syncthreads();
        if (t < kUnionSlices) {
            uint32_t before = 0;
            for (uint32_t w = 0; w < wave; ++w) before += waveTotals[w];
            memberPrefix[t + 1u] = before + inclusiveSize;
            if (t == 0u) memberPrefix[0] = 0u;
        }
        __syncthreads();
        loadDescriptor(local + gridDim.x);              // (in flight until the next cell)
        const uint32_t members = memberPrefix[sliceCount];
        // ---- member f -> its id: bucket by binary search of the prefix sums ----
        auto memberAddress = [&](uint32_t f) {
            // (eight fixed steps, no branches: the searches of a thread's members are straight-line code whose LDS reads overlap)
            uint32_t lo = 0;                            // the last slice with memberPrefix[slice] <= f
#pragma unroll
            for (int step = 7; step >= 0; --step) {
                const uint32_t mid = lo + (1u << step);
                const uint32_t bound = memberPrefix[mid < sliceCount ? mid : sliceCount];
                lo = (mid < sliceCount && bound <= f) ? mid : lo;
            }
            return bucketBegin[lo] + (f - memberPrefix[lo]);
        };
        uint32_t held[kUnionHeld];
#pragma unroll
        for (uint32_t r = 0; r < kUnionHeld; ++r) {
            const uint32_t f = t + r * kUnionThreads;
            held[r] = f < members ? sortedCells[memberAddress(f)] : 0xffffffffu;
        }
        uint32_t written = 0;
        for (uint32_t pass = 0; pass < passes; ++pass) {
            const uint32_t lo = pass * kUnionWords * 32u;
            // ---- the members inside [lo, lo + kUnionWords * 32) set their bits ----
#pragma unroll
            for (uint32_t r = 0; r < kUnionHeld; ++r) {
                const uint32_t id = held[r] - lo;                   // (ids below lo, and the "no member" value, wrap out of range)
                if (held[r] != 0xffffffffu && id < kUnionWords * 32u) atomicOr(&bitmap[id >> 5], 1u << (id & 31u));
            }
            for (uint32_t f = t + kUnionHeld * kUnionThreads; f < members; f += kUnionThreads) {        // (long lists: reloaded per pass)
                const uint32_t id = sortedCells[memberAddress(f)] - lo;
                if (id < kUnionWords * 32u) atomicOr(&bitmap[id >> 5], 1u << (id & 31u));
            }
            __syncthreads();
            // ---- every thread counts the bits of its 17 words and notes which of them hold any (about two do) ----
            uint32_t count = 0, occupied = 0;
#pragma unroll
            for (uint32_t j = 0; j < kUnionWordsPerThread; ++j) {
                const uint32_t word = bitmap[t * kUnionWordsPerThread + j];
                count += uint32_t(__builtin_popcount(word));
                occupied |= (word != 0u ? 1u : 0u) << j;
            }
            // block-wide exclusive prefix sum of the counts
            uint32_t inclusive = count;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t below = uint32_t(__shfl_up(int(inclusive), d, 64));
                if (lane >= uint32_t(d)) inclusive += below;
            }
            if (lane == 63u) waveTotals[wave] = inclusive;
            __syncthreads();
            uint32_t before = 0, total = 0;
            for (uint32_t w = 0; w < kUnionThreads / 64u; ++w) {
                const uint32_t x = waveTotals[w];
                before += w < wave ? x : 0u;
                total += x;
            }
            uint32_t position = before + inclusive - count;
            // ---- ids in ascending order, the words left zero: through the staging area when they fit (coalesced stores), else
            // straight to the cell's segment ----
            const bool staged = total <= kUnionStaged;
            // (two loops, not one pointer chosen at run time: an LDS-or-global pointer becomes a flat access)
            if (staged) {
                while (occupied) {
                    const uint32_t j = uint32_t(__builtin_ctz(occupied));
                    occupied &= occupied - 1u;
                    uint32_t bits = bitmap[t * kUnionWordsPerThread + j];
                    bitmap[t * kUnionWordsPerThread + j] = 0u;
                    const uint32_t first = lo + (t * kUnionWordsPerThread + j) * 32u;
                    while (bits) {
                        const uint32_t bit = uint32_t(__builtin_ctz(bits));
                        bits &= bits - 1u;
                        staging[position++] = first + bit;
                    }
                }
            } else {
                while (occupied) {
                    const uint32_t j = uint32_t(__builtin_ctz(occupied));
                    occupied &= occupied - 1u;
                    uint32_t bits = bitmap[t * kUnionWordsPerThread + j];
                    bitmap[t * kUnionWordsPerThread + j] = 0u;
                    const uint32_t first = lo + (t * kUnionWordsPerThread + j) * 32u;
                    while (bits) {
                        const uint32_t bit = uint32_t(__builtin_ctz(bits));
                        bits &= bits - 1u;
                        out[written + position++] = first + bit;
                    }
                }
            }
            __syncthreads();
            if (staged) {
                for (uint32_t i = t; i < total; i += kUnionThreads) out[written + i] = staging[i];
            }
            written += total;
            __syncthreads();
        }
        if (t == 0u) distinctCounts[local] = written;
    }
}

__device__ __forceinline__ void waveFence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// One wave per cell: unique + mismatch filter in ascending id order (ExpressionMatrixLsh.cpp:436-445).  Uses no LDS
// and few registers on purpose: the kernel is a latency-bound gather of candidate signatures and wants every wave
// slot of the CU (the selection below, which stages lists in 48 KB of LDS, runs 3 waves per CU; as one fused kernel
// the gathers ran at that occupancy too and took 1.58 s of a 1.81 s run at 1M cells x 2048 bits).
__global__ void __launch_bounds__(256)
filterKernel(const uint64_t* __restrict__ sig, uint32_t words, uint32_t batchBegin, uint32_t batchCells,
             const uint32_t* __restrict__ segmentBegin, const uint32_t* __restrict__ sortedCandidates,
             Entry* __restrict__ lists, int32_t mGlobal, const uint32_t* __restrict__ keyOfMismatch,
             uint32_t* __restrict__ listCounts, const uint32_t* __restrict__ distinctCounts)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t local = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (local >= batchCells) return;
    const uint32_t c = batchBegin + local;
    const uint32_t begin = segmentBegin[local];
    const uint32_t end = distinctCounts ? begin + distinctCounts[local] : segmentBegin[local + 1u];
    const uint64_t* mine = sig + size_t(c) * words;
    Entry* list = lists + begin;                    // at most (end-begin) entries survive
    uint32_t n = 0;
    for (uint32_t base = begin; base < end; base += 64u) {
        const uint32_t i = base + lane;
        bool keep = false;
        uint32_t cand = 0, m = 0;
        if (i < end) {
            cand = sortedCandidates[i];
            const bool duplicate = i > begin && sortedCandidates[i - 1u] == cand;
            if (!duplicate && cand != c) {                                   // ExpressionMatrixLsh.cpp:437-439
                const uint64_t* other = sig + size_t(cand) * words;
                for (uint32_t w = 0; w < words; ++w) m += uint32_t(__builtin_popcountll(mine[w] ^ other[w]));
                keep = int32_t(m) <= mGlobal;                                // similarity > similarityThreshold (:441)
            }
        }
        const uint64_t mask = __builtin_amdgcn_ballot_w64(keep);
        if (keep) {
            const uint32_t before = __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
            Entry e;
            e.cell = cand;
            e.key = keyOfMismatch[m];
            list[n + before] = e;
        }
        n += uint32_t(__builtin_popcountll(mask));
    }
    if (lane == 0u) listCounts[local] = n;
}

// Default form of the filter: groups of lpc lanes (16 from 1024 bits on) read consecutive words of one candidate's
// signature -- one contiguous request per candidate instead of 64 scattered 8-byte reads per wave instruction -- and
// add their partial popcounts with a butterfly; only the candidates that need a count (not a duplicate, not the cell
// itself) are visited, looked up by rank through LDS.  rocprof at 1M cells x 2048 bits, q = 20: 210 ms against 730 ms
// for the one-lane-per-candidate filterKernel above (kept for signatures beyond 8192 bits and A/B runs).
__global__ void __launch_bounds__(256)
filterCooperativeKernel(const uint64_t* __restrict__ sig, uint32_t words, uint32_t batchBegin, uint32_t batchCells,
                        const uint32_t* __restrict__ segmentBegin, const uint32_t* __restrict__ sortedCandidates,
                        Entry* __restrict__ lists, int32_t mGlobal, const uint32_t* __restrict__ keyOfMismatch,
                        uint32_t* __restrict__ listCounts, const uint32_t* __restrict__ distinctCounts)
{
    __shared__ uint32_t candOfRankAll[4][64];
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* candOfRank = candOfRankAll[threadIdx.x >> 6];
    const uint32_t local = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (local >= batchCells) return;
    const uint32_t c = batchBegin + local;
    const uint32_t begin = segmentBegin[local];
    // (after unionKernel the segment holds the distinct candidates only, distinctCounts of them)
    const uint32_t end = distinctCounts ? begin + distinctCounts[local] : segmentBegin[local + 1u];
    Entry* list = lists + begin;
    uint32_t lpc = 16u;
    while (lpc > words) lpc >>= 1;
    if (lpc == 0u) lpc = 1u;
    const uint32_t perStep = 64u / lpc;
    const uint32_t sub = lane % lpc;
    const uint32_t slot = lane / lpc;
    // this lane's share of the cell's own signature (words sub, sub+lpc, ...), at most 8 registers (words <= 64... 128)
    uint64_t mine[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) mine[t] = (sub + t * lpc < words) ? sig[size_t(c) * words + sub + t * lpc] : 0ull;
    uint32_t n = 0;
    for (uint32_t base = begin; base < end; base += 64u) {
        const uint32_t i = base + lane;
        bool need = false;
        uint32_t cand = 0;
        if (i < end) {
            cand = sortedCandidates[i];
            const bool duplicate = i > begin && sortedCandidates[i - 1u] == cand;
            need = !duplicate && cand != c;
        }
        const uint64_t needMask = __builtin_amdgcn_ballot_w64(need);
        const uint32_t needCount = uint32_t(__builtin_popcountll(needMask));
        const uint32_t myRank = __builtin_amdgcn_mbcnt_hi(uint32_t(needMask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(needMask), 0u));
        if (need) candOfRank[myRank] = cand;
        waveFence();
        // step s computes the counts of the candidates of rank s*perStep .. s*perStep+perStep-1; the count of rank r is
        // parked in lane r of mOfRank (a register indexed by lane) through a shuffle-free trick: lane r reads it from LDS
        uint32_t m = 0;
        for (uint32_t first = 0; first < needCount; first += perStep) {
            const uint32_t rank = first + slot;
            uint32_t part = 0;
            if (rank < needCount) {
                const uint64_t* theirs = sig + size_t(candOfRank[rank]) * words + sub;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    if (sub + t * lpc < words) part += uint32_t(__builtin_popcountll(mine[t] ^ theirs[t * lpc]));
                }
            }
            for (uint32_t d = 1; d < lpc; d <<= 1) part += uint32_t(__shfl_xor(int(part), int(d), 64));
            const uint32_t got = uint32_t(__shfl(int(part), int(((myRank - first) % perStep) * lpc), 64));
            if (need && myRank >= first && myRank < first + perStep) m = got;
        }
        waveFence();
        const bool keep = need && int32_t(m) <= mGlobal;
        const uint64_t mask = __builtin_amdgcn_ballot_w64(keep);
        if (keep) {
            const uint32_t before = __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
            Entry e;
            e.cell = cand;
            e.key = keyOfMismatch[m];
            list[n + before] = e;
        }
        n += uint32_t(__builtin_popcountll(mask));
    }
    if (lane == 0u) listCounts[local] = n;
}

// The cooperative filter with 16-byte loads and four groups of candidates in flight (default for an even number of words up
// to 64, i.e. every multiple of 128 bits up to 4096): a group of lpc lanes reads ONE candidate's signature as consecutive
// 16-byte units -- at 2048 bits one 256-byte request per candidate, where filterCooperativeKernel issues two 128-byte requests
// of 8-byte lane loads (8-byte accesses reach 0.54-0.70 of the 16-byte rate: MI355X_MICROARCH.md, visibility table) -- and a
// lane has UNROLL such loads outstanding before the first popcount.  Everything else is filterCooperativeKernel.
// T: 16-byte units per lane -- 1 (narrow signatures, 16 lanes per candidate), 4 (2048 bits: four lanes per candidate), 2 (up to
// 4096 bits).  LPC: lanes per candidate as a compile-time constant (16, 8, 4), or 0 = decided at run time (narrow signatures); FULL: every lane holds a unit of the row
// (units == T * LPC: 2048 and 4096 bits).  Since the grouped visiting order the kernel
// is bound by its vector ALUs (VALU busy 85 %), so the inner loop is straight-line code: the loads are unconditional (a slot
// without a candidate reads the cell's own row and counts 0), one 16-byte load per unit, the 16-lane sums are four
// data-parallel-primitive adds, and a candidate's count goes to its rank's slot in LDS, from where the lane that holds the
// candidate picks it up once per 64 candidates (it was a shuffle per group of candidates).
template <int T, int LPC, bool FULL>
__global__ void __launch_bounds__(256)
filterWideKernel(const uint64_t* __restrict__ sig, uint32_t words, uint32_t batchBegin, uint32_t batchCells,
                 const uint32_t* __restrict__ segmentBegin, const uint32_t* __restrict__ sortedCandidates,
                 Entry* __restrict__ lists, int32_t mGlobal, const uint32_t* __restrict__ keyOfMismatch,
                 uint32_t* __restrict__ listCounts, const uint32_t* __restrict__ distinctCounts,
                 const uint32_t* __restrict__ order, uint32_t chunk)
{
    constexpr int UNROLL = 4;
    __shared__ uint32_t candOfRankAll[4][64];
    __shared__ uint32_t countOfRankAll[4][64];
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* candOfRank = candOfRankAll[threadIdx.x >> 6];
    uint32_t* countOfRank = countOfRankAll[threadIdx.x >> 6];
    uint32_t local = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (order) {
        // (see neighbourhoodLabelKernel) workgroups are dealt to the 8 XCDs round-robin: XCD x walks positions
        // [x * chunk, (x + 1) * chunk) of the batch's cells in the order of their labels
        const uint32_t xcd = blockIdx.x & 7u;
        const uint32_t position = xcd * chunk + (blockIdx.x >> 3) * (blockDim.x >> 6) + (threadIdx.x >> 6);
        const uint32_t chunkEnd = (xcd + 1u) * chunk < batchCells ? (xcd + 1u) * chunk : batchCells;
        if (position >= chunkEnd) return;
        local = order[position];
    }
    if (local >= batchCells) return;
    const uint32_t c = batchBegin + local;
    const uint32_t listBegin = segmentBegin[local];
    const uint32_t listEnd = distinctCounts ? listBegin + distinctCounts[local] : segmentBegin[local + 1u];
    Entry* list = lists + listBegin;
    const uint32_t begin = listBegin, end = listEnd;
    const uint32_t units = words / 2u;                      // 16-byte units of a signature
    uint32_t lpc = LPC ? uint32_t(LPC) : 1u;
    if (!LPC) {
        while (lpc < 16u && lpc * uint32_t(T) < units) lpc <<= 1;
    }
    const uint32_t perStep = 64u / lpc;
    const uint32_t sub = lane % lpc;
    const uint32_t slot = lane / lpc;
    const uint4* sig16 = reinterpret_cast<const uint4*>(sig);
    uint4 mine[T];
    uint32_t unit[T];          // the lane's 16-byte units of a row (a lane beyond the row re-reads unit 0 and counts nothing)
    bool active[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        active[t] = sub + uint32_t(t) * lpc < units;
        unit[t] = active[t] ? sub + uint32_t(t) * lpc : 0u;
        mine[t] = sig16[size_t(c) * units + unit[t]];
    }
    uint32_t n = 0;
    // (the candidate ids and the list entries are streams: non-temporal, so that they do not push the signatures the XCD's cells
    // share out of its L2.  The ids of the NEXT 64 candidates are loaded before this turn's rows are: a turn was two dependent
    // memory round trips, ids then rows.  A candidate's predecessor comes from the lane below, lane 0's from the turn before.)
    uint32_t nextCand = begin + lane < end ? __builtin_nontemporal_load(sortedCandidates + begin + lane) : 0u;
    uint32_t lastOfPreviousTurn = 0;
    for (uint32_t base = begin; base < end; base += 64u) {
        const uint32_t i = base + lane;
        bool need = false;
        const uint32_t cand = nextCand;
        nextCand = i + 64u < end ? __builtin_nontemporal_load(sortedCandidates + i + 64u) : 0u;
        {
            const uint32_t below = uint32_t(__shfl_up(int(cand), 1, 64));
            const uint32_t previous = lane ? below : lastOfPreviousTurn;
            lastOfPreviousTurn = uint32_t(__builtin_amdgcn_readlane(int(cand), 63));
            if (i < end) {
                const bool duplicate = i > listBegin && previous == cand;
                need = !duplicate && cand != c;                                  // ExpressionMatrixLsh.cpp:437-439
            }
        }
        const uint64_t needMask = __builtin_amdgcn_ballot_w64(need);
        const uint32_t needCount = uint32_t(__builtin_popcountll(needMask));
        const uint32_t myRank = __builtin_amdgcn_mbcnt_hi(uint32_t(needMask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(needMask), 0u));
        if (need) candOfRank[myRank] = cand;
        waveFence();
        for (uint32_t first = 0; first < needCount; first += perStep * uint32_t(UNROLL)) {
            uint4 theirs[UNROLL][T];
            uint32_t rank[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                rank[u] = first + uint32_t(u) * perStep + slot;
                const uint32_t rowCell = rank[u] < needCount ? candOfRank[rank[u]] : c;
                const uint4* row = sig16 + size_t(rowCell) * units;
#pragma unroll
                for (int t = 0; t < T; ++t) theirs[u][t] = row[unit[t]];
            }
            uint32_t part[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                part[u] = 0;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const uint32_t bits = uint32_t(__builtin_popcount(mine[t].x ^ theirs[u][t].x)) + uint32_t(__builtin_popcount(mine[t].y ^ theirs[u][t].y)) +
                                          uint32_t(__builtin_popcount(mine[t].z ^ theirs[u][t].z)) + uint32_t(__builtin_popcount(mine[t].w ^ theirs[u][t].w));
                    part[u] += (FULL || active[t]) ? bits : 0u;
                }
            }
            // sums over the lpc (<= 16) lanes of a candidate: data-parallel-primitive adds inside a row of 16 lanes, the four
            // candidates of a turn step by step (an add's result may not feed the next DPP read for two cycles)
            if (lpc >= 2u) {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) part[u] += uint32_t(__builtin_amdgcn_update_dpp(0, int(part[u]), 0xB1, 0xF, 0xF, false));      // quad_perm [1,0,3,2]
            }
            if (lpc >= 4u) {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) part[u] += uint32_t(__builtin_amdgcn_update_dpp(0, int(part[u]), 0x4E, 0xF, 0xF, false));      // quad_perm [2,3,0,1]
            }
            if (lpc >= 8u) {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) part[u] += uint32_t(__builtin_amdgcn_update_dpp(0, int(part[u]), 0x141, 0xF, 0xF, false));     // row_half_mirror
            }
            if (lpc >= 16u) {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) part[u] += uint32_t(__builtin_amdgcn_update_dpp(0, int(part[u]), 0x140, 0xF, 0xF, false));     // row_mirror
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                if (sub == 0u && rank[u] < needCount) countOfRank[rank[u]] = part[u];
            }
        }
        waveFence();
        const uint32_t m = need ? countOfRank[myRank] : 0u;
        const bool keep = need && int32_t(m) <= mGlobal;                     // similarity > similarityThreshold (:441)
        const uint64_t mask = __builtin_amdgcn_ballot_w64(keep);
        if (keep) {
            const uint32_t before = __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
            const uint64_t entry = uint64_t(cand) | (uint64_t(keyOfMismatch[m]) << 32);          // Entry {cell, key}
            __builtin_nontemporal_store(entry, reinterpret_cast<uint64_t*>(list + n + before));
        }
        n += uint32_t(__builtin_popcountll(mask));
        waveFence();          // (the next 64 candidates overwrite the two rank arrays)
    }
    if (lane == 0u) listCounts[local] = n;
}

// One wave per cell: keepBest(cellNeighbors, k) (:457), then SimilarPairs::copy + sort (:489-496), store.
// CAPACITY whole entries are staged in LDS (12 bytes each: the form for k beyond the packed tiers' 2048 and for more than
// 65534 key classes).  The kernel is launched once per tier and batch and takes the cells with ABOVE < candidates <= CAPACITY:
// the smaller the tier, the more waves a CU holds.
template <uint32_t CAPACITY, uint32_t ABOVE>
__global__ void __launch_bounds__(64)
selectKernel(uint32_t batchCells, const uint32_t* __restrict__ segmentBegin, Entry* __restrict__ lists,
             const uint32_t* __restrict__ listCounts, const float* __restrict__ keySimilarity, uint32_t k,
             PairOut* __restrict__ outPairs, uint32_t* __restrict__ outUsed)
{
    __shared__ Entry lds[CAPACITY];
    __shared__ uint16_t ldsL[CAPACITY];
    __shared__ uint16_t ldsR[CAPACITY];
    const uint32_t lane = threadIdx.x;
    const uint32_t local = blockIdx.x;
    if (local >= batchCells) return;
    Entry* list = lists + segmentBegin[local];
    uint32_t n = listCounts[local];
    if (n <= ABOVE || n > CAPACITY) return;             // another tier's cells (beyond the last tier: selectGlobalKernel's)
    // (eight loads in flight before the first LDS store: see selectPackedKernel)
    for (uint32_t base = 0; base < n; base += 64u * 8u) {
        Entry staged[8];
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) {
            const uint32_t i = base + j * 64u + lane;
            staged[j] = i < n ? list[i] : Entry();
        }
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) {
            const uint32_t i = base + j * 64u + lane;
            if (i < n) lds[i] = staged[j];
        }
    }
    waveFence();
    if (n > k) {
        nthElementWaveT<uint16_t, false, Entry, 4>(lds, ldsL, ldsR, int(k), int(n), lane);
        n = k;
        waveFence();
    }
    PairOut* out = outPairs + size_t(local) * k;
    for (uint32_t i = lane; i < n; i += 64u) {
        const Entry e = lds[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) {
            const Entry o = lds[j];
            rank += uint32_t((o.key < e.key) || (o.key == e.key && o.cell < e.cell));
        }
        PairOut po;
        po.cell = e.cell;
        po.similarity = keySimilarity[e.key];
        out[rank] = po;
    }
    for (uint32_t i = n + lane; i < k; i += 64u) {
        PairOut zero;
        zero.cell = 0u;
        zero.similarity = 0.0f;
        out[i] = zero;
    }
    if (lane == 0u) outUsed[local] = n;
}

// The LDS tiers again with 8 bytes per entry instead of 12: what is staged is {key, position in the list} in four bytes
// (keys are mismatch classes, at most lshCount + 1 of them; a tier holds fewer than 65536 entries), the selection permutes
// those exactly as it would permute the entries -- only keys are ever compared -- and the cells of the k survivors are
// fetched from the list in global memory afterwards.  A third more waves per CU at every tier (the kernel lives on LDS
// latency): 4096 entries take 32 KB (five waves per CU instead of three), 6144 take 48 KB (three), 8192 take 64 KB (two),
// 16384 take 128 KB (one).
struct PackedEntry {
    uint16_t key;
    uint16_t index;
};
#ifdef EM2_DIAG
// (diagnostic build, EM2_TIMING=1: wave cycles of selectPackedKernel by phase -- staging, selection, survivors + sort, output)
__device__ unsigned long long selectPhaseCycles[8];
#define EM2_SELECT_PHASE(i)                                                                       \
    do {                                                                                          \
        const unsigned long long now_ = __builtin_readcyclecounter();                             \
        if (lane == 0u) atomicAdd(&selectPhaseCycles[i], now_ - phaseStart_);                     \
        phaseStart_ = now_;                                                                       \
    } while (0)
#else
#define EM2_SELECT_PHASE(i) do { } while (0)
#endif
constexpr uint32_t kSelectPackedMaxK = 2048;        // the survivors are gathered into the smallest tier's position arrays

template <uint32_t CAPACITY, uint32_t ABOVE>
__global__ void __launch_bounds__(64)
selectPackedKernel(uint32_t batchCells, const uint32_t* __restrict__ segmentBegin, const Entry* __restrict__ lists,
                   const uint32_t* __restrict__ listCounts, const float* __restrict__ keySimilarity, uint32_t k,
                   PairOut* __restrict__ outPairs, uint32_t* __restrict__ outUsed)
{
    __shared__ PackedEntry lds[CAPACITY];
    __shared__ __attribute__((aligned(8))) uint16_t positions[2u * CAPACITY];         // the two position arrays of the selection
    uint16_t* ldsL = positions;
    uint16_t* ldsR = positions + CAPACITY;
    const uint32_t lane = threadIdx.x;
    const uint32_t local = blockIdx.x;
    if (local >= batchCells) return;
    uint32_t n = listCounts[local];
    if (n <= ABOVE || n > CAPACITY) return;             // another tier's cells
#ifdef EM2_DIAG
    unsigned long long phaseStart_ = __builtin_readcyclecounter();
#endif
    const Entry* list = lists + segmentBegin[local];
    // Staging: the list was written by the filter of this batch (gigabytes per batch: it comes from HBM), and a loop of one
    // load and one LDS store per 64 entries paid a memory latency per turn -- 67 turns for the 4300 entries of config D's
    // lists, most of the kernel's time.  Sixteen loads are in flight before the first store.
    for (uint32_t base = 0; base < n; base += 64u * 16u) {
        uint32_t keys[16];
#pragma unroll
        for (uint32_t j = 0; j < 16u; ++j) {
            const uint32_t i = base + j * 64u + lane;
            keys[j] = i < n ? list[i].key : 0u;
        }
#pragma unroll
        for (uint32_t j = 0; j < 16u; ++j) {
            const uint32_t i = base + j * 64u + lane;
            if (i < n) {
                PackedEntry e;
                e.key = uint16_t(keys[j]);
                e.index = uint16_t(i);
                lds[i] = e;
            }
        }
    }
    waveFence();
    EM2_SELECT_PHASE(0);
    if (n > k) {
        nthElementWaveT<uint16_t, false, PackedEntry, 4>(lds, ldsL, ldsR, int(k), int(n), lane);
        n = k;
        waveFence();
    }
    EM2_SELECT_PHASE(1);
    // the survivors' cells, then the rank sort of SimilarPairs::copy + sort (:489-496) on (key, cell)
    Entry* kept = reinterpret_cast<Entry*>(positions);     // (the position arrays are free now; k <= kSelectPackedMaxK entries of 8 bytes)
    static_assert(CAPACITY * 4u >= kSelectPackedMaxK * 8u, "the position arrays must hold k entries");
    for (uint32_t i = lane; i < n; i += 64u) {
        Entry e;
        e.key = lds[i].key;
        e.cell = list[lds[i].index].cell;
        kept[i] = e;
    }
    waveFence();
    EM2_SELECT_PHASE(2);
    PairOut* out = outPairs + size_t(local) * k;
    uint32_t padded = 1;
    while (padded < n) padded <<= 1;
    if (padded * 8u <= CAPACITY * 4u) {
        // (the bitonic network of em2_select_wave.h whenever the position arrays hold the padded list: k <= 1024 at every tier)
        sortListWave(kept, n, lane);
        EM2_SELECT_PHASE(3);
        for (uint32_t i = lane; i < n; i += 64u) {
            const Entry e = kept[i];
            PairOut po;
            po.cell = e.cell;
            po.similarity = keySimilarity[e.key];
            out[i] = po;
        }
    } else {
        for (uint32_t i = lane; i < n; i += 64u) {
            const Entry e = kept[i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < n; ++j) {
                const Entry o = kept[j];
                rank += uint32_t((o.key < e.key) || (o.key == e.key && o.cell < e.cell));
            }
            PairOut po;
            po.cell = e.cell;
            po.similarity = keySimilarity[e.key];
            out[rank] = po;
        }
    }
    for (uint32_t i = n + lane; i < k; i += 64u) {
        PairOut zero;
        zero.cell = 0u;
        zero.similarity = 0.0f;
        out[i] = zero;
    }
    if (lane == 0u) outUsed[local] = n;
    EM2_SELECT_PHASE(4);
}

// The same for lists of any length, left where the filter wrote them: one wave per cell runs the wave-parallel selection
// (em2_select_wave.h) on global memory; its two position arrays are the cell's segments of the candidate arrays, which the
// filter has finished with.  Takes the cells with
// more than `above` candidates.  The lists of a batch are L2-resident between the filter and this kernel, and unlike the
// LDS forms this one keeps every wave slot of a CU busy.
__global__ void __launch_bounds__(256)
selectGlobalKernel(uint32_t batchCells, const uint32_t* __restrict__ segmentBegin, Entry* __restrict__ lists,
                   const uint32_t* __restrict__ listCounts, const float* __restrict__ keySimilarity, uint32_t k, uint32_t above,
                   uint32_t* __restrict__ positionsL, uint32_t* __restrict__ positionsR, PairOut* __restrict__ outPairs,
                   uint32_t* __restrict__ outUsed)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t local = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (local >= batchCells) return;
    uint32_t n = listCounts[local];
    if (n <= above) return;
    const uint32_t begin = segmentBegin[local];
    Entry* work = lists + begin;
    if (n > k) {
        waveSyncGlobal();           // the filter's stores came from another kernel; start from clean lines all the same
        nthElementWaveT<uint32_t, true>(work, positionsL + begin, positionsR + begin, int(k), int(n), lane);
        n = k;
    }
    PairOut* out = outPairs + size_t(local) * k;
    for (uint32_t i = lane; i < n; i += 64u) {
        const Entry e = work[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) {
            const Entry o = work[j];
            rank += uint32_t((o.key < e.key) || (o.key == e.key && o.cell < e.cell));
        }
        PairOut po;
        po.cell = e.cell;
        po.similarity = keySimilarity[e.key];
        out[rank] = po;
    }
    for (uint32_t i = n + lane; i < k; i += 64u) {
        PairOut zero;
        zero.cell = 0u;
        zero.similarity = 0.0f;
        out[i] = zero;
    }
    if (lane == 0u) outUsed[local] = n;
}

// The scratch of a call (tables, candidate ids, lists: 10 GB at a million cells x 2048 bits) comes from a cache of device blocks
// the process keeps between calls: hipMalloc of gigabytes costs anything between 4 and 200 ms per call depending on the state of
// the box (measured: the same command, two leases), more than the tables' kernels.  A block goes back to the cache only when
// the call completed (its stream synchronised); a call that returns early with an error frees its blocks (hipFree waits for the
// device).  EM2_SCRATCH_CACHE_MB caps what is kept (default: an eighth of the device's memory; 0 = nothing is kept); em2_dev_release_scratch() frees it.
class ScratchCache {
public:
    void* take(size_t bytes, int device, size_t* got)
    {
        std::lock_guard<std::mutex> guard(mutex_);
        size_t best = blocks_.size();
        for (size_t i = 0; i < blocks_.size(); ++i) {
            const Block& b = blocks_[i];
            if (b.device != device || b.bytes < bytes || b.bytes > bytes + bytes / 2u + (size_t(1) << 20)) continue;
            if (best == blocks_.size() || b.bytes < blocks_[best].bytes) best = i;
        }
        if (best == blocks_.size()) return nullptr;
        void* p = blocks_[best].p;
        *got = blocks_[best].bytes;
        total_ -= blocks_[best].bytes;
        blocks_.erase(blocks_.begin() + long(best));
        return p;
    }
    void give(void* p, size_t bytes, int device)
    {
        std::vector<Block> evicted;
        bool kept = false;
        {
            std::lock_guard<std::mutex> guard(mutex_);
            const size_t cap = capBytes();
            if (bytes <= cap) {
                // a block that fits the cap by itself makes room for itself: the OLDEST blocks go first (what the cache is for
                // is the few large blocks of the last call -- the scan's workspace, the result -- not whatever arrived first)
                while (!blocks_.empty() && (total_ + bytes > cap || blocks_.size() >= 256u)) {
                    evicted.push_back(blocks_.front());
                    total_ -= blocks_.front().bytes;
                    blocks_.erase(blocks_.begin());
                }
                blocks_.push_back(Block{p, bytes, device});
                total_ += bytes;
                kept = true;
            }
        }
        for (const Block& b : evicted) (void)hipFree(b.p);
        if (!kept) (void)hipFree(p);
    }
    void clear()
    {
        std::vector<Block> freed;
        {
            std::lock_guard<std::mutex> guard(mutex_);
            freed.swap(blocks_);
            total_ = 0;
        }
        for (const Block& b : freed) (void)hipFree(b.p);
    }
    static ScratchCache& instance()
    {
        static ScratchCache* cache = new ScratchCache();          // (never destroyed: the HIP runtime may be gone at exit)
        return *cache;
    }

private:
    struct Block { void* p; size_t bytes; int device; };
    static size_t capBytes()
    {
        // what a host process can live with: a sixteenth of the device's memory (18 GB of an MI355X's 288) unless
        // EM2_SCRATCH_CACHE_MB says otherwise (0: nothing is kept).  One findSimilarPairs5 call's scratch at a million cells x
        // 2048 bits is 10 GB; the scan workspace of em2_subset_find_similar_pairs4 at a million cells is 12 GB (round 5: 27, and an
        // eighth of the memory to hold it), and it is the block that matters: its hipMalloc took 0.4 ms in 22 calls of 24 on one
        // box and 2.7 and 4.0 s in the other two.
        if (const char* v = getenv("EM2_SCRATCH_CACHE_MB")) return size_t(strtoull(v, nullptr, 10)) << 20;
        static size_t share = 0;
        if (!share) {
            size_t freeBytes = 0, totalBytes = 0;
            share = hipMemGetInfo(&freeBytes, &totalBytes) == hipSuccess && totalBytes ? totalBytes / 16u : size_t(4) << 30;
        }
        return share;
    }
    std::mutex mutex_;
    std::vector<Block> blocks_;
    size_t total_ = 0;
};

thread_local bool scratchCallCompleted = false;          // set right before a call's normal return: its buffers may be cached

struct Buffer {
    void* p = nullptr;
    size_t bytes = 0;
    int device = 0;
    ~Buffer() { drop(scratchCallCompleted); }
    // (explicit releases happen behind a synchronisation of the stream: the block is idle)
    void release() { drop(true); }
    void drop(bool idle)
    {
        if (!p) return;
        if (idle) ScratchCache::instance().give(p, bytes, device);
        else (void)hipFree(p);
        p = nullptr;
    }
    hipError_t allocate(size_t wanted)
    {
        drop(false);
        wanted = wanted ? wanted : 1;
        if (hipGetDevice(&device) != hipSuccess) device = 0;
        p = ScratchCache::instance().take(wanted, device, &bytes);
        if (p) return hipSuccess;
        bytes = wanted;
        const hipError_t e = hipMalloc(&p, wanted);
        if (e != hipSuccess) {
            // (memory held by the cache may be what is missing)
            (void)hipGetLastError();
            ScratchCache::instance().clear();
            return hipMalloc(&p, wanted);
        }
        return e;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

uint32_t gridFor(uint64_t n)
{
    const uint64_t blocks = (n + 255) / 256;
    return uint32_t(blocks > 16384 ? 16384 : (blocks ? blocks : 1));
}

uint64_t envBatchLog2()
{
    const char* v = getenv("EM2_FSP5_BATCH_LOG2");
    const int x = v ? atoi(v) : 29;
    return uint64_t(x < 20 ? 20 : (x > 31 ? 31 : x));
}

uint32_t bitsFor(uint64_t maxValue)
{
    uint32_t b = 1;
    while (b < 64 && (maxValue >> b) != 0) ++b;
    return b;
}

#define EM2_TRY(call)                        \
    do {                                     \
        hipError_t em2Err_ = (call);         \
        if (em2Err_ != hipSuccess) return em2Err_; \
    } while (0)

}  // namespace


// Host driver: allocates its own scratch and synchronises the stream.  The bucket tables are built over ALL
// cellCount cells; candidates are generated and selected for the cells [rowBegin,rowEnd) only (the shard one
// rank owns).  d_pairs / d_used are device arrays of (rowEnd-rowBegin)*k and (rowEnd-rowBegin) elements.
namespace {
thread_local Fsp5LaunchInfo lastFsp5Info = {0., 0., 0., 0., -1., -1., -1.};
}

Fsp5LaunchInfo fsp5LastLaunchInfo() { return lastFsp5Info; }

void fsp5ReleaseScratch() { ScratchCache::instance().clear(); }

// The cache for other host-buffer entry points of the library (csrc/em2_capi.hip: the result and the scan workspace of
// em2_subset_find_similar_pairs4 -- 6 GB at 1M cells, whose hipMalloc took 1.6-2.7 s in two calls of fourteen on one box and
// 1.7 ms otherwise).  scratchTake: a cached block of at least `bytes` on the current device, or nullptr; scratchGive: an IDLE block
// back (freed when the cache is full).
void* scratchTake(size_t bytes, size_t* got)
{
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return nullptr;
    return ScratchCache::instance().take(bytes ? bytes : 1, device, got);
}

void scratchGive(void* p, size_t bytes)
{
    int device = 0;
    if (!p) return;
    if (hipGetDevice(&device) != hipSuccess) {
        (void)hipFree(p);
        return;
    }
    ScratchCache::instance().give(p, bytes, device);
}

hipError_t runFsp5(const uint64_t* d_sig, uint32_t cellCount, uint32_t rowBegin, uint32_t rowEnd, uint32_t lshCount,
                   uint32_t k, uint32_t q, uint64_t bucketOverflow, const DeviceTables& tables, PairOut* d_pairs,
                   uint32_t* d_used, hipStream_t stream)
{
    const uint32_t rowCount = rowEnd - rowBegin;
    const uint32_t words = (lshCount - 1u) / 64u + 1u;
    const uint32_t sliceCount = lshCount / q;                       // ExpressionMatrixLsh.cpp:355
    if (rowCount == 0) return hipSuccess;
    scratchCallCompleted = false;
    // EM2_TIMING=1: wall time of the stages of one call on stderr (each mark synchronises the stream)
    const bool stageTiming = getenv("EM2_TIMING") && (getenv("EM2_TIMING")[0] == '1' || getenv("EM2_TIMING")[0] == '2');
    const bool stageSync = stageTiming && getenv("EM2_TIMING")[0] == '1';          // (2: host time only, nothing synchronised)
    auto stageClock = std::chrono::steady_clock::now();
    auto stage = [&](const char* name) {
        if (!stageTiming) return;
        if (stageSync) (void)hipStreamSynchronize(stream);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[em2 timing]   findSimilarPairs5: %s %.2f ms\n", name, std::chrono::duration<double, std::milli>(now - stageClock).count());
        stageClock = now;
    };
    EM2_TRY(hipMemsetAsync(d_used, 0, size_t(rowCount) * sizeof(uint32_t), stream));
    if (k) EM2_TRY(hipMemsetAsync(d_pairs, 0, size_t(rowCount) * k * sizeof(PairOut), stream));
    if (sliceCount == 0 || cellCount == 0) return hipStreamSynchronize(stream);
    const uint64_t total = uint64_t(sliceCount) * cellCount;
    if (total >= 0xffffffffULL) return hipErrorInvalidValue;

    // 1. keys and stable sort
    Buffer keysA, keysB, cellsA, cellsB, temp;
    EM2_TRY(keysA.allocate(total * sizeof(uint64_t)));
    EM2_TRY(keysB.allocate(total * sizeof(uint64_t)));
    EM2_TRY(cellsA.allocate(total * sizeof(uint32_t)));
    EM2_TRY(cellsB.allocate(total * sizeof(uint32_t)));
    sliceKeysKernel<<<gridFor(total), 256, 0, stream>>>(d_sig, cellCount, words, q, sliceCount, keysA.as<uint64_t>(), cellsA.as<uint32_t>());
    EM2_TRY(hipGetLastError());
    const uint32_t endBit = 32u + bitsFor(sliceCount - 1u);
    size_t tempBytes = 0;
    EM2_TRY(rocprim::radix_sort_pairs(nullptr, tempBytes, keysA.as<uint64_t>(), keysB.as<uint64_t>(), cellsA.as<uint32_t>(),
                                      cellsB.as<uint32_t>(), size_t(total), 0u, endBit, stream));
    EM2_TRY(temp.allocate(tempBytes));
    EM2_TRY(rocprim::radix_sort_pairs(temp.p, tempBytes, keysA.as<uint64_t>(), keysB.as<uint64_t>(), cellsA.as<uint32_t>(),
                                      cellsB.as<uint32_t>(), size_t(total), 0u, endBit, stream));
    const uint64_t* sortedKeys = keysB.as<uint64_t>();
    const uint32_t* sortedCells = cellsB.as<uint32_t>();

    stage("slice keys + stable sort");
    // 2. runs (= buckets)
    Buffer flags, scan, runStart, runOf, counts;
    EM2_TRY(flags.allocate(total * sizeof(uint32_t)));
    EM2_TRY(scan.allocate(total * sizeof(uint32_t)));
    EM2_TRY(runStart.allocate((total + 1) * sizeof(uint32_t)));
    EM2_TRY(runOf.allocate(total * sizeof(uint32_t)));
    EM2_TRY(counts.allocate(size_t(cellCount) * sizeof(uint64_t)));
    runFlagsKernel<<<gridFor(total), 256, 0, stream>>>(sortedKeys, total, flags.as<uint32_t>());
    EM2_TRY(hipGetLastError());
    size_t scanBytes = 0;
    EM2_TRY(rocprim::inclusive_scan(nullptr, scanBytes, flags.as<uint32_t>(), scan.as<uint32_t>(), size_t(total), rocprim::plus<uint32_t>(), stream));
    Buffer scanTemp;
    EM2_TRY(scanTemp.allocate(scanBytes));
    EM2_TRY(rocprim::inclusive_scan(scanTemp.p, scanBytes, flags.as<uint32_t>(), scan.as<uint32_t>(), size_t(total), rocprim::plus<uint32_t>(), stream));
    runTablesKernel<<<gridFor(total), 256, 0, stream>>>(sortedKeys, sortedCells, flags.as<uint32_t>(), scan.as<uint32_t>(), total,
                                                        cellCount, sliceCount, runStart.as<uint32_t>(), runOf.as<uint32_t>());
    EM2_TRY(hipGetLastError());
    stage("run tables");
    // the labels that order the filter's visits, and in the same walk over the cells' bucket descriptors the number of members
    // every cell will gather
    const bool grouped = true;
    Buffer labelsA, labelsB;
    if (grouped) {
        EM2_TRY(labelsA.allocate(size_t(cellCount) * sizeof(uint32_t)));
        EM2_TRY(labelsB.allocate(size_t(cellCount) * sizeof(uint32_t)));
    }
    neighbourhoodLabelKernel<<<dim3((cellCount + 3u) / 4u), 256, 0, stream>>>(runOf.as<uint32_t>(), runStart.as<uint32_t>(), sortedCells, cellCount, sliceCount,
                                                                             bucketOverflow, grouped ? labelsA.as<uint32_t>() : nullptr, counts.as<uint64_t>());
    EM2_TRY(hipGetLastError());
    if (grouped) {
        const dim3 labelGrid((cellCount + 255u) / 256u);
        jumpLabelsKernel<<<labelGrid, 256, 0, stream>>>(labelsA.as<uint32_t>(), cellCount, labelsB.as<uint32_t>());
        EM2_TRY(hipGetLastError());
        jumpLabelsKernel<<<labelGrid, 256, 0, stream>>>(labelsB.as<uint32_t>(), cellCount, labelsA.as<uint32_t>());
        EM2_TRY(hipGetLastError());
    }
    std::vector<uint64_t> hostCounts(cellCount);
    EM2_TRY(hipMemcpyAsync(hostCounts.data(), counts.p, size_t(cellCount) * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    EM2_TRY(hipStreamSynchronize(stream));
    stage("labels + counts to the host");
    // keys / flags / scan are no longer needed
    keysA.release(); flags.release(); scan.release(); scanTemp.release(); temp.release(); cellsA.release();

    // 3. batches of cells whose gathered candidates fit the budget.  The batches are planned first so that the scratch
    // is allocated once, at the size of the largest batch (allocating per batch cost more than the kernels).
    // (EM2_FSP5_BATCH_LOG2: A/B measurements -- the larger the batch, the longer the filter's grouped order stays on one
    // neighbourhood's signatures, and the more scratch: 16 bytes per candidate id)
    const uint64_t budgetLog2 = envBatchLog2();
    const uint64_t budget = 1ull << budgetLog2;      // 2^29 candidate ids per batch by default (8 GiB of scratch: ids twice, lists)
    struct Batch { uint32_t begin, end; std::vector<uint32_t> seg; };
    std::vector<Batch> batches;
    uint64_t maxTotal = 0;
    uint32_t maxCells = 0;
    for (uint32_t batchBegin = rowBegin; batchBegin < rowEnd;) {
        Batch batch;
        batch.begin = batchBegin;
        uint64_t sum = 0;
        uint32_t batchEnd = batchBegin;
        batch.seg.assign(1, 0u);
        while (batchEnd < rowEnd && batchEnd - batchBegin < (1u << 20)) {
            const uint64_t n = hostCounts[batchEnd];
            if (n >= 0xffffffffULL) return hipErrorInvalidValue;
            if (sum + n > budget && batchEnd > batchBegin) break;
            sum += n;
            ++batchEnd;
            batch.seg.push_back(uint32_t(sum));
            if (sum > budget) break;                 // a single huge cell: its own batch
        }
        if (sum >= 0xffffffffULL) return hipErrorInvalidValue;
        batch.end = batchEnd;
        if (sum > maxTotal) maxTotal = sum;
        if (batchEnd - batchBegin > maxCells) maxCells = batchEnd - batchBegin;
        batches.push_back(std::move(batch));
        batchBegin = batchEnd;
    }
    {
        double gathered = 0.;
        for (uint32_t c = rowBegin; c < rowEnd; c++) gathered += double(hostCounts[c]);
        lastFsp5Info.gatheredCandidates = gathered;
        lastFsp5Info.cells = double(rowCount);
        lastFsp5Info.sliceCount = double(sliceCount);
        lastFsp5Info.batches = double(batches.size());
        lastFsp5Info.filterMs = lastFsp5Info.selectMs = 0.;
        lastFsp5Info.distinctCandidates = -1.;
    }
    static thread_local hipEvent_t timing[3] = {nullptr, nullptr, nullptr};
    if (!timing[0]) {
        if (hipEventCreate(&timing[0]) != hipSuccess || hipEventCreate(&timing[1]) != hipSuccess ||
            hipEventCreate(&timing[2]) != hipSuccess) timing[0] = timing[1] = timing[2] = nullptr;
    }
    Buffer segBegin, candA, candB, lists, sortTemp, listCounts, distinctCounts;
    // (more than kUnionSlices slices: gatherKernel + a segmented sort, round 2's form)
    const bool useUnion = sliceCount <= kUnionSlices;
    uint32_t unionBlocks = 1;
    if (useUnion) {
        int device = 0, cuCount = 0, perCu = 0;
        EM2_TRY(hipGetDevice(&device));
        EM2_TRY(hipDeviceGetAttribute(&cuCount, hipDeviceAttributeMultiprocessorCount, device));
        EM2_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&unionKernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(kUnionLdsBytes)));
        EM2_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, reinterpret_cast<const void*>(&unionKernel), int(kUnionThreads), kUnionLdsBytes));
        unionBlocks = uint32_t(cuCount > 0 ? cuCount : 1) * uint32_t(perCu > 0 ? perCu : 1);
    }
    std::vector<uint32_t> hostDistinct;
    double distinctTotal = 0.;
    EM2_TRY(segBegin.allocate((size_t(maxCells) + 1u) * sizeof(uint32_t)));
    EM2_TRY(candA.allocate(size_t(maxTotal) * sizeof(uint32_t)));
    EM2_TRY(candB.allocate(size_t(maxTotal) * sizeof(uint32_t)));
    EM2_TRY(lists.allocate(size_t(maxTotal) * sizeof(Entry)));
    EM2_TRY(listCounts.allocate(size_t(maxCells) * sizeof(uint32_t)));
    EM2_TRY(distinctCounts.allocate(size_t(maxCells) * sizeof(uint32_t)));
    if (useUnion) hostDistinct.resize(maxCells);
    Buffer orderKeysA, orderKeysB, orderLocalsA, orderLocalsB, orderTemp;
    size_t orderTempBytes = 0;
    if (grouped) {
        EM2_TRY(orderKeysA.allocate(size_t(maxCells) * sizeof(uint32_t)));
        EM2_TRY(orderKeysB.allocate(size_t(maxCells) * sizeof(uint32_t)));
        EM2_TRY(orderLocalsA.allocate(size_t(maxCells) * sizeof(uint32_t)));
        EM2_TRY(orderLocalsB.allocate(size_t(maxCells) * sizeof(uint32_t)));
        EM2_TRY(rocprim::radix_sort_pairs(nullptr, orderTempBytes, orderKeysA.as<uint32_t>(), orderKeysB.as<uint32_t>(), orderLocalsA.as<uint32_t>(),
                                          orderLocalsB.as<uint32_t>(), size_t(maxCells), 0u, 32u, stream));
        EM2_TRY(orderTemp.allocate(orderTempBytes));
    }
    size_t sortTempBytes = 0;
    stage("batch plan + scratch allocation");
    const uint32_t idBits = bitsFor(cellCount - 1u);
    // the filter by shape: 16-byte loads for an even number of words up to 4096 bits, the 8-byte cooperative form for odd word
    // counts and up to 8192 bits, one lane per candidate beyond
    const bool cooperative = words <= 8u * 16u;
    const bool wide = cooperative && words % 2u == 0u && words <= 64u && reinterpret_cast<uintptr_t>(d_sig) % 16u == 0u;
    for (const Batch& batch : batches) {
        const uint32_t batchBegin = batch.begin;
        const uint32_t batchCells = batch.end - batch.begin;
        const uint32_t batchTotal = batch.seg.back();
        EM2_TRY(hipMemcpyAsync(segBegin.p, batch.seg.data(), batch.seg.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        const uint32_t* sorted = candA.as<uint32_t>();
        const uint32_t* distinct = nullptr;
        if (batchTotal && useUnion) {
            // the ascending, duplicate-free union of every cell's buckets through a bitmap in LDS (unionKernel)
            unionKernel<<<unionBlocks, kUnionThreads, kUnionLdsBytes, stream>>>(runOf.as<uint32_t>(), runStart.as<uint32_t>(), sortedCells, cellCount,
                                                                                sliceCount, bucketOverflow, batchBegin, batchCells,
                                                                                segBegin.as<uint32_t>(), candA.as<uint32_t>(),
                                                                                distinctCounts.as<uint32_t>());
            EM2_TRY(hipGetLastError());
            distinct = distinctCounts.as<uint32_t>();
        } else if (batchTotal) {
            // more slices than unionKernel has descriptors for: gather the buckets' members and sort each cell's segment
            gatherKernel<<<(batchCells + 3u) / 4u, 256, 0, stream>>>(runOf.as<uint32_t>(), runStart.as<uint32_t>(), sortedCells, cellCount,
                                                                     sliceCount, bucketOverflow, batchBegin, batchCells,
                                                                     segBegin.as<uint32_t>(), candA.as<uint32_t>());
            EM2_TRY(hipGetLastError());
            size_t segBytes = 0;
            EM2_TRY(rocprim::segmented_radix_sort_keys(nullptr, segBytes, candA.as<uint32_t>(), candB.as<uint32_t>(), batchTotal, batchCells,
                                                       segBegin.as<uint32_t>(), segBegin.as<uint32_t>() + 1, 0u, idBits, stream));
            if (segBytes > sortTempBytes) {
                EM2_TRY(hipStreamSynchronize(stream));
                EM2_TRY(sortTemp.allocate(segBytes));
                sortTempBytes = segBytes;
            }
            EM2_TRY(rocprim::segmented_radix_sort_keys(sortTemp.p, segBytes, candA.as<uint32_t>(), candB.as<uint32_t>(), batchTotal, batchCells,
                                                       segBegin.as<uint32_t>(), segBegin.as<uint32_t>() + 1, 0u, idBits, stream));
            sorted = candB.as<uint32_t>();
        }
        stage("  batch: union");
        // the filter's visiting order of this batch: its cells by label (stable: equal labels keep the id order)
        const uint32_t* order = nullptr;
        uint32_t orderChunk = 0;
        if (grouped && wide && batchCells > 64u) {
            batchOrderKeysKernel<<<(batchCells + 255u) / 256u, 256, 0, stream>>>(labelsA.as<uint32_t>(), batchBegin, batchCells,
                                                                                 orderKeysA.as<uint32_t>(), orderLocalsA.as<uint32_t>());
            EM2_TRY(hipGetLastError());
            size_t bytes = orderTempBytes;
            EM2_TRY(rocprim::radix_sort_pairs(orderTemp.p, bytes, orderKeysA.as<uint32_t>(), orderKeysB.as<uint32_t>(), orderLocalsA.as<uint32_t>(),
                                              orderLocalsB.as<uint32_t>(), size_t(batchCells), 0u, idBits, stream));
            order = orderLocalsB.as<uint32_t>();
            orderChunk = ((batchCells + 7u) / 8u + 3u) & ~3u;
        }
        const uint32_t filterBlocks = order ? 8u * (orderChunk / 4u) : (batchCells + 3u) / 4u;
        stage("  batch: order");
        if (timing[0]) (void)hipEventRecord(timing[0], stream);
        if (wide) {
            // (signature rows are 16-byte aligned: an even number of words, and the array itself as hipMalloc / the caller's
            // uint64 array provides it -- checked below)
            const uint32_t units = words / 2u;
            const uint32_t unitsPerLane = words <= 32u ? 1u : 2u;
            uint32_t lanesPerCandidate = 1u;
            while (lanesPerCandidate < 16u && lanesPerCandidate * unitsPerLane < units) lanesPerCandidate <<= 1;
#define EM2_FILTER_WIDE(TT, LL, FF)                                                                                                     \
            filterWideKernel<TT, LL, FF><<<filterBlocks, 256, 0, stream>>>(d_sig, words, batchBegin, batchCells, segBegin.as<uint32_t>(), sorted,  \
                                                                     lists.as<Entry>(), tables.mGlobal, tables.keyOfMismatch,         \
                                                                     listCounts.as<uint32_t>(), distinct, order, orderChunk)
            // (2048 bits: FOUR units per lane and four lanes per candidate -- two reduction steps instead of four and a quarter of
            // the per-candidate bookkeeping per unit: 66.3 -> 62.5 ms at 1M cells, same box; {1, 16} 66.3, {2, 8} 68.1, {8, 2} 77.2,
            // {16, 1} 196.5; eight loads in flight instead of four change nothing, two cost 10 ms: profiles/r05_fsp5_experiments.md)
            if (unitsPerLane == 1u && units == 16u) EM2_FILTER_WIDE(4, 4, true);
            else if (unitsPerLane == 1u && lanesPerCandidate == 16u) EM2_FILTER_WIDE(1, 16, false);
            else if (unitsPerLane == 1u && lanesPerCandidate == 8u) EM2_FILTER_WIDE(1, 8, false);
            else if (unitsPerLane == 1u) EM2_FILTER_WIDE(1, 0, false);
            else if (units == 32u) EM2_FILTER_WIDE(2, 16, true);
            else if (lanesPerCandidate == 16u) EM2_FILTER_WIDE(2, 16, false);
            else EM2_FILTER_WIDE(2, 0, false);
#undef EM2_FILTER_WIDE
        } else if (cooperative) {
            filterCooperativeKernel<<<(batchCells + 3u) / 4u, 256, 0, stream>>>(d_sig, words, batchBegin, batchCells, segBegin.as<uint32_t>(),
                                                                                sorted, lists.as<Entry>(), tables.mGlobal,
                                                                                tables.keyOfMismatch, listCounts.as<uint32_t>(), distinct);
        } else {
            filterKernel<<<(batchCells + 3u) / 4u, 256, 0, stream>>>(d_sig, words, batchBegin, batchCells, segBegin.as<uint32_t>(), sorted,
                                                                     lists.as<Entry>(), tables.mGlobal, tables.keyOfMismatch,
                                                                     listCounts.as<uint32_t>(), distinct);
        }
        EM2_TRY(hipGetLastError());
        if (timing[0]) (void)hipEventRecord(timing[1], stream);
        // keepBest + sort + store, by list length: LDS tiers of 4096 / 5120 / 6144 / 8192 / 16384 entries with {key, position} in
        // 4 bytes + two position arrays (8 bytes per entry: 5, 4, 3, 2 and 1 wave per CU), beyond that the same wave-parallel
        // selection on global memory.  k above 2048 or more than 65534 key classes: tiers of 4096 / 5120 / 6656 / 12288 whole
        // entries (12 bytes each).
        PairOut* outPairs = d_pairs + size_t(batchBegin - rowBegin) * k;
        uint32_t* outUsed = d_used + (batchBegin - rowBegin);
        uint32_t globalAbove = kSelectLdsEntriesBig;
        // (the packed tiers keep a key in 16 bits: lshCount + 1 key classes must fit)
        const bool packed = k <= kSelectPackedMaxK && lshCount < 65535u;
        if (packed) {
#define EM2_SELECT_PACKED(CAPACITY, ABOVE)                                                                                        \
            selectPackedKernel<CAPACITY, ABOVE><<<batchCells, 64, 0, stream>>>(batchCells, segBegin.as<uint32_t>(), lists.as<Entry>(), \
                                                                              listCounts.as<uint32_t>(), tables.keySimilarity, k, \
                                                                              outPairs, outUsed);                                 \
            EM2_TRY(hipGetLastError())
            EM2_SELECT_PACKED(4096u, 0u);
            EM2_SELECT_PACKED(5120u, 4096u);          // (40 KB: four waves per CU where 6144 entries allow three; config D's lists are 3600..4700 long)
            EM2_SELECT_PACKED(6144u, 5120u);
            EM2_SELECT_PACKED(8192u, 6144u);
            EM2_SELECT_PACKED(16384u, 8192u);
#undef EM2_SELECT_PACKED
            globalAbove = 16384u;
        } else {
#define EM2_SELECT_TIER(CAPACITY, ABOVE)                                                                                         \
            selectKernel<CAPACITY, ABOVE><<<batchCells, 64, 0, stream>>>(batchCells, segBegin.as<uint32_t>(), lists.as<Entry>(), \
                                                                        listCounts.as<uint32_t>(), tables.keySimilarity, k, outPairs, \
                                                                        outUsed);                                                \
            EM2_TRY(hipGetLastError())
            EM2_SELECT_TIER(kSelectLdsEntries, 0u);
            EM2_SELECT_TIER(5120u, kSelectLdsEntries);
            EM2_SELECT_TIER(6656u, 5120u);
            EM2_SELECT_TIER(kSelectLdsEntriesBig, 6656u);
#undef EM2_SELECT_TIER
        }
        {
            selectGlobalKernel<<<(batchCells + 3u) / 4u, 256, 0, stream>>>(
                batchCells, segBegin.as<uint32_t>(), lists.as<Entry>(), listCounts.as<uint32_t>(), tables.keySimilarity, k, globalAbove,
                candA.as<uint32_t>(), candB.as<uint32_t>(), outPairs, outUsed);
            EM2_TRY(hipGetLastError());
        }
        if (timing[0]) (void)hipEventRecord(timing[2], stream);
        if (distinct) EM2_TRY(hipMemcpyAsync(hostDistinct.data(), distinct, size_t(batchCells) * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        EM2_TRY(hipStreamSynchronize(stream));       // the batch's offsets (pageable host memory) and scratch are reused
        if (distinct) for (uint32_t i = 0; i < batchCells; i++) distinctTotal += double(hostDistinct[i]);
        stage("  batch: filter + selection");
        if (timing[0]) {
            float a = 0.f, b = 0.f;
            if (hipEventElapsedTime(&a, timing[0], timing[1]) == hipSuccess && hipEventElapsedTime(&b, timing[1], timing[2]) == hipSuccess) {
                lastFsp5Info.filterMs += double(a);
                lastFsp5Info.selectMs += double(b);
            }
        }
    }
    lastFsp5Info.distinctCandidates = useUnion ? distinctTotal : -1.;
    if (stageTiming) {
        candA.release(); candB.release(); lists.release();
        stage("release of the candidate scratch");
        labelsA.release(); labelsB.release(); orderKeysA.release(); orderKeysB.release(); orderLocalsA.release(); orderLocalsB.release(); orderTemp.release();
        stage("release of the order scratch");
        keysB.release(); cellsB.release(); runStart.release(); runOf.release(); counts.release();
        stage("release of the tables");
    }
#ifdef EM2_DIAG
    if (getenv("EM2_TIMING") && getenv("EM2_TIMING")[0] == '1') {
        unsigned long long cycles[8] = {0};
        if (hipMemcpyFromSymbol(cycles, HIP_SYMBOL(selectPhaseCycles), sizeof(cycles)) == hipSuccess) {
            double total = 0;
            for (int i = 0; i < 5; i++) total += double(cycles[i]);
            fprintf(stderr, "[em2 timing] selectPackedKernel wave cycles (cumulative): staging %.1f%% selection %.1f%% survivors %.1f%% sort %.1f%% output %.1f%% (%.3g cycles)\n",
                    100 * cycles[0] / total, 100 * cycles[1] / total, 100 * cycles[2] / total, 100 * cycles[3] / total, 100 * cycles[4] / total, total);
        }
    }
#endif
    scratchCallCompleted = true;          // (every batch ended with a synchronisation of the stream: the scratch is idle)
    return hipSuccess;
}

}  // namespace em2
