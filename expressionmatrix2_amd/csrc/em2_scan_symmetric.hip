// em2_scan_symmetric.hip -- findSimilarPairs4 with every unordered pair evaluated once: the symmetric scan of one
// GPU (fsp4ScanSymmetricKernel + inbox sort + fsp4InboxReplayKernel) and the sharded symmetric scan across GPUs
// (prefix phases with the same kernel, fsp4TileKernel for the deferred square, replay).  Bit-identical to
// src/ExpressionMatrixLsh.cpp:200-285 + src/SimilarPairs.cpp:369-405; see em2_scan.hip for the per-cell contract and
// the CDNA4 mapping of the column loop, em2_scan_common.h for the shared device code.

#include "em2_scan_common.h"
#include "em2_matrix_step_asm.h"

#include <vector>

#include <rocprim/rocprim.hpp>

namespace em2 {
namespace {

// =========================================================================================================
// Symmetric form: every unordered pair is counted ONCE (the reference's own accounting, N(N-1)/2), which halves
// the v_xor/v_bcnt work the scan is bound by.
//
// The reference gets away with one evaluation per pair because its 64x64 block order happens to offer the
// candidates of every cell in ascending id order.  The same contract is kept here as follows.  Row block b (64
// cells, one per lane) scans only the columns BELOW its rows; for the pair (row r, column c < r) with mismatch m
//   * the row side is the usual in-lane state machine: candidates c arrive in ascending order;
//   * the column side -- cell c must be offered candidate r, but only after all its candidates below r -- is
//     deferred: if m <= snap[c], the entry (c, r, m) is EMITTED to an inbox in HBM.  snap[c] is a cut-off cell c
//     held at some earlier point of its own sequence (published at its segment hand-offs); cut-offs only tighten,
//     so everything not emitted would have been rejected whenever it was offered.
// After the scan the inbox is sorted by (c, r) (rocPRIM radix sort) and a second kernel replays, per cell, its
// entries in ascending r through the exact state machine, then finishes the rows.  Cells below fullRowBlocks*64
// have too few lower candidates for a useful snapshot; their blocks scan all columns themselves ("full rows",
// snap = -1, nothing is emitted to them), which costs 2*c0/N extra work.
// Work items are (segment, row block) as in the persistent kernel, but a triangle block only has the segments up
// to its diagonal; tickets enumerate segment-major through segTable.  The last 64 columns of a triangle block are
// its own cells (diagonal): a plain loop with the extra test column < row.
// If the inbox pool overflows (adversarial similarity order), the launcher reruns the ordered scan.
// =========================================================================================================

typedef const __attribute__((address_space(4))) int32_t* ScalarIntPtr;

// Returns the new chunk as pos | end << 32; pos > end (1, 0) = emission disabled after an overflow.
__device__ __attribute__((noinline)) uint64_t refillInboxChunk(uint64_t* inbox, uint32_t* control, uint64_t capacity,
                                                               uint32_t chunk, uint32_t lane, uint32_t pos, uint32_t end)
{
    for (uint32_t i = pos + lane; i < end; i += 64u) inbox[i] = ~0ull;      // sentinels sort to the end
    unsigned long long base = 0;
    if (lane == 0u) {
        base = __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(control), (unsigned long long)chunk,
                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const uint32_t lo = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(base))));
    const uint32_t hi = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(base >> 32))));
    const uint64_t b = uint64_t(lo) | (uint64_t(hi) << 32);
    if (b + chunk > capacity) {
        if (lane == 0u) __hip_atomic_store(control + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return 1ull;
    }
    return b | ((b + chunk) << 32);
}

__device__ __forceinline__ void emitColumn(bool emit, uint32_t col, uint32_t row, uint32_t m, uint32_t lane,
                                           uint32_t& emitPos, uint32_t& emitEnd)
{
    const uint64_t mask = __builtin_amdgcn_ballot_w64(emit);
    if (mask == 0ull) return;
    uint32_t p = uint32_t(__builtin_amdgcn_readfirstlane(int(emitPos)));
    uint32_t e = uint32_t(__builtin_amdgcn_readfirstlane(int(emitEnd)));
    if (p > e) return;
    const uint32_t n = uint32_t(__builtin_popcountll(mask));
    ArgsPtr aux = kernelArgs();
    if (p + n > e) {
        const uint64_t fresh = refillInboxChunk(aux->inbox, aux->inboxControl, aux->inboxCapacity, aux->inboxChunk, lane, p, e);
        p = uint32_t(fresh);
        e = uint32_t(fresh >> 32);
        if (p > e) {
            emitPos = p;
            emitEnd = e;
            return;
        }
    }
    if (emit) {
        const uint32_t nb = aux->rowBits;
        aux->inbox[p + lanesBelow(mask)] = (uint64_t(col) << (13u + nb)) | (uint64_t(row) << 13u) | uint64_t(m);
    }
    emitPos = p + n;
    emitEnd = e;
}

// Room left in this wave's inbox chunk; "unlimited" once emission is disabled (pos > end after an overflow).
__device__ __forceinline__ uint32_t inboxRoom(uint32_t emitPos, uint32_t emitEnd)
{
    const uint32_t p = uint32_t(__builtin_amdgcn_readfirstlane(int(emitPos)));
    const uint32_t e = uint32_t(__builtin_amdgcn_readfirstlane(int(emitEnd)));
    return p > e ? 0xffffffffu : e - p;
}

// Makes sure the chunk has room for one more column's worth of entries (64).
__device__ __forceinline__ void ensureInboxRoom(uint32_t lane, uint32_t& emitPos, uint32_t& emitEnd)
{
    if (inboxRoom(emitPos, emitEnd) >= 64u) return;
    ArgsPtr aux = kernelArgs();
    const uint32_t p = uint32_t(__builtin_amdgcn_readfirstlane(int(emitPos)));
    const uint32_t e = uint32_t(__builtin_amdgcn_readfirstlane(int(emitEnd)));
    const uint64_t fresh = refillInboxChunk(aux->inbox, aux->inboxControl, aux->inboxCapacity, aux->inboxChunk, lane, p, e);
    emitPos = uint32_t(fresh);
    emitEnd = uint32_t(fresh >> 32);
}

// scanColumns for the strictly-lower part of a triangle block: every column is below every row of the wave.
// Two columns per loop iteration so that the snapshot registers alternate at compile time.
//
// The loop body contains NO calls: keeping a prefetched 32-dword chunk alive across a call needs more
// call-preserved SGPRs than exist, and the compiler then parks a chunk in VGPR lanes on every step (measured:
// +25% run time).  So the rare path only stores -- the inbox entries (the caller guarantees room for one
// column, ensureInboxRoom), the row candidates (SPECULATIVE: to the log; otherwise straight to the row lists) --
// and the scan RETURNS to its caller whenever something needs service: inbox room below 64, a full log, or a
// row list that reached 2k entries (the caller cuts it and re-enters).  Returns the first column not scanned.
template <int W32, bool IDENTITY, bool SPECULATIVE>
__device__ __forceinline__ uint32_t scanColumnsEmit(const uint32_t* __restrict__ sig32, const int32_t* snap,
                                                    uint32_t colBegin, uint32_t colEnd, const uint32_t (&r)[W32],
                                                    uint32_t row, bool rowValid, uint32_t lane,
                                                    Entry* myList, uint32_t twoK, uint32_t& count, int32_t mMax,
                                                    Entry* myLog, uint32_t logCapacity, uint32_t& logCount,
                                                    uint32_t& emitPos, uint32_t emitEnd)
{
    constexpr int CH = W32 < 32 ? W32 : 32;
    constexpr int H = W32 / CH;
    constexpr int U = 2 * H;
    if (colBegin >= colEnd) return colEnd;
    ScalarPtr p = (ScalarPtr)(uintptr_t)sig32 + size_t(colBegin) * W32;
    ScalarIntPtr sp = (ScalarIntPtr)(uintptr_t)snap + colBegin;
    uint32_t chunk[2][CH];
    int32_t snapCol[2];
#pragma unroll
    for (int w = 0; w < CH; ++w) chunk[0][w] = p[w];
    snapCol[0] = sp[0];
    snapCol[1] = 0;
    __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0)
    uint32_t m = 0;
    for (uint32_t colBase = colBegin; colBase < colEnd; colBase += 2u) {
#pragma unroll
        for (int s = 0; s < U; ++s) {
            const int part = s % H;
            const int ci = s / H;
            const uint32_t col = colBase + uint32_t(ci);
            if (col < colEnd) {
                __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0)
                __builtin_amdgcn_sched_barrier(0);
                const bool lastChunk = (col + 1u == colEnd) && (part == H - 1);
                ScalarPtr pn = lastChunk ? p : p + CH;
#pragma unroll
                for (int w = 0; w < CH; ++w) chunk[(s + 1) & 1][w] = pn[w];
                p = pn;
                if (part == H - 1) {
                    ScalarIntPtr spn = lastChunk ? sp : sp + 1;
                    snapCol[ci ^ 1] = spn[0];
                    sp = spn;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int w = 0; w < CH; ++w) {
                    if (part == 0 && w == 0) popcountFirst(m, r[0] ^ chunk[s & 1][0]);
                    else popcountAccumulate(m, r[part * CH + w] ^ chunk[s & 1][w]);
                }
                if (part == H - 1) {
                    // one compare in the steady state: m against the looser of the row's and the column's cut-off
                    // (the empty asm pins the v_max behind the popcounts; without it hipcc hoists it in front of them and
                    // the kernel measured 1.2% slower)
                    int32_t limit = mMax > snapCol[ci] ? mMax : snapCol[ci];
                    asm volatile("" : "+v"(limit));
                    if (__builtin_amdgcn_ballot_w64(int32_t(m) <= limit) != 0ull) {
                        const bool pass = int32_t(m) <= mMax;
                        const bool emit = rowValid && int32_t(m) <= snapCol[ci];
                        bool stop = false;
                        const uint64_t emitMask = __builtin_amdgcn_ballot_w64(emit);
                        if (emitMask != 0ull) {
                            const uint32_t at = uint32_t(__builtin_amdgcn_readfirstlane(int(emitPos)));
                            if (at <= uint32_t(__builtin_amdgcn_readfirstlane(int(emitEnd)))) {
                                if (emit) {
                                    ArgsPtr aux = kernelArgs();
                                    aux->inbox[at + lanesBelow(emitMask)] =
                                        (uint64_t(col) << (13u + aux->rowBits)) | (uint64_t(row) << 13u) | uint64_t(m);
                                }
                                emitPos = at + uint32_t(__builtin_popcountll(emitMask));
                                stop = inboxRoom(emitPos, emitEnd) < 64u;
                            }
                        }
                        if (__builtin_amdgcn_ballot_w64(pass) != 0ull) {
                            if (SPECULATIVE) {
                                if (pass) {
                                    storeEntry(myLog + logCount, col, m);
                                    ++logCount;
                                }
                                stop |= __builtin_amdgcn_ballot_w64(logCount == logCapacity) != 0ull;
                            } else {
                                if (pass) {
                                    uint32_t key = m;
                                    if (!IDENTITY) key = kernelArgs()->keyOfMismatch[m];
                                    storeEntry(myList + count, col, key);
                                    ++count;
                                }
                                stop |= __builtin_amdgcn_ballot_w64(count == twoK) != 0ull;
                            }
                        }
                        if (stop) return col + 1u;
                    }
                    m = 0;
                }
            }
        }
    }
    return colEnd;
}

// The diagonal columns of a triangle block (its own 64 cells): pair (row, col) belongs to the lane with row > col.
template <int W32, bool IDENTITY, bool SPECULATIVE>
__device__ __forceinline__ uint32_t scanDiagonal(const uint32_t* __restrict__ sig32, const int32_t* snap,
                                                 uint32_t colBegin, uint32_t colEnd, const uint32_t (&r)[W32],
                                                 uint32_t row, bool rowValid, uint32_t lane, uint32_t blockV,
                                                 Entry* myList, uint32_t twoK, uint32_t& count, int32_t& mMax,
                                                 Entry* myLog, uint32_t logCapacity, uint32_t& logCount,
                                                 uint32_t& emitPos, uint32_t& emitEnd, unsigned char* ldsRaw)
{
    for (uint32_t col = colBegin; col < colEnd; ++col) {
        ScalarPtr cp = (ScalarPtr)(uintptr_t)sig32 + size_t(col) * W32;      // wave-uniform: scalar loads
        uint32_t m = 0;
#pragma unroll
        for (int w = 0; w < W32; ++w) popcountAccumulate(m, r[w] ^ cp[w]);
        const int32_t snapCol = __hip_atomic_load(snap + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool lower = col < row;
        const bool pass = lower && int32_t(m) <= mMax;
        const bool emit = lower && rowValid && int32_t(m) <= snapCol;
        emitColumn(emit, col, row, m, lane, emitPos, emitEnd);
        if (__builtin_amdgcn_ballot_w64(pass) != 0ull) {
            if (SPECULATIVE) {
                if (pass) {
                    storeEntry(myLog + logCount, col, m);
                    ++logCount;
                }
                if (__builtin_amdgcn_ballot_w64(logCount == logCapacity) != 0ull) return col + 1u;
            } else {
                acceptColumn<IDENTITY>(pass, col, row, m, lane, uint32_t(__builtin_amdgcn_readfirstlane(int(blockV))),
                                       myList, twoK, count, mMax, ldsRaw);
            }
        }
    }
    return colEnd;
}

// Uniform values that are only needed between the scan loops are parked in VGPRs (the loops need their ~100 SGPRs
// for two 32-dword column chunks; a build that kept these values in SGPRs spilled a chunk to VGPR lanes INSIDE the
// loop and ran 25% slower) and read back with v_readfirstlane_b32 where they are used.
__device__ __forceinline__ uint32_t parkInVgpr(uint32_t x)
{
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ uint32_t unpark(uint32_t v)
{
    return uint32_t(__builtin_amdgcn_readfirstlane(int(v)));
}

constexpr uint32_t kItemTriangle = 1u, kItemLast = 2u, kItemSpeculate = 4u;

template <int W32, bool IDENTITY>
__global__ void __launch_bounds__(256)
fsp4ScanSymmetricKernel(Fsp4Args args)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t emitPos = 0, emitEnd = 0;      // no chunk yet: the first emission takes one

    for (;;) {
        uint32_t ticket = 0;
        if (lane == 0u) {
            ticket = __hip_atomic_fetch_add(kernelArgs()->control, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ticket = uint32_t(__builtin_amdgcn_readfirstlane(int(ticket)));

        // parked (VGPR) copies of the item's uniform values
        uint32_t colBeginV, colEndV, segV, blockV, flagsV;
        uint32_t row;
        uint32_t r[W32];
        int32_t mMax;
        uint32_t count = 0;
        Entry* myList;
        Entry* myLog;
        uint32_t twoK, logCapacity;
        uint32_t logCount = 0;
        bool rowValid;
        {
            ArgsPtr aux = kernelArgs();
            if (ticket >= aux->totalTickets) break;
            const uint32_t cellCount = aux->cellCount;
            const uint32_t segments = aux->segments;
            const uint32_t* table = aux->segTable;
            uint32_t seg = 0;
            {
                uint32_t lo = 0, hi = segments;           // table[lo] <= ticket < table[hi]
                while (hi - lo > 1u) {
                    const uint32_t mid = (lo + hi) / 2u;
                    if (table[mid] <= ticket) lo = mid;
                    else hi = mid;
                }
                seg = lo;
            }
            const uint32_t local = ticket - table[seg];
            const uint32_t fullBlocks = aux->fullRowBlocks;
            // slot = list / state slot of the launch; its 64 cells start at rowBase (block-cyclic in the sharded scan)
            const uint32_t relative = local < fullBlocks ? local : table[segments + 1u + seg] + (local - fullBlocks);
            const uint32_t block = aux->localBlockBase + relative;
            uint32_t flags = relative >= fullBlocks ? kItemTriangle : 0u;
            const uint32_t rowBase = (block * aux->rowBlockStride + aux->rowBlockOffset) * 64u;
            twoK = parkInVgpr(2u * aux->k);
            logCapacity = parkInVgpr(aux->logCapacity);
            myList = aux->buffers + (size_t(block) * 64u + lane) * twoK;
            myLog = aux->logs + (size_t(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64u + lane) * logCapacity;
            const uint32_t cps = aux->columnsPerSegment;
            const uint32_t columnLimit = aux->columnLimit;
            const uint32_t colBegin = seg * cps;
            uint32_t colEnd = colBegin + cps;
            if (colEnd > columnLimit || seg + 1u == segments) colEnd = columnLimit;
            if (seg + 1u == segments) flags |= kItemLast;
            if (flags & kItemTriangle) {
                uint32_t diagEnd = rowBase + 64u;
                if (diagEnd > columnLimit) diagEnd = columnLimit;
                if (diagEnd <= colEnd) {
                    colEnd = diagEnd;
                    flags |= kItemLast;
                }
            }
            row = rowBase + lane;
            rowValid = row < cellCount;
            const uint32_t* rp = aux->sig32 + size_t(rowValid ? row : rowBase) * W32;
#pragma unroll
            for (int w = 0; w < W32; ++w) r[w] = rp[w];
            mMax = rowValid ? aux->mMaxInitial : -1;
            if (seg != 0u) {
                const uint32_t done = uint32_t(__builtin_amdgcn_readfirstlane(
                    int(__hip_atomic_load(aux->segmentsDone + block, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))));
                if (done < seg) {
                    flags |= kItemSpeculate;
                    if (done != 0u) {
                        const uint64_t st = __hip_atomic_load(reinterpret_cast<const uint64_t*>(aux->rowState) + size_t(block) * 64u + lane,
                                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        mMax = rowValid ? int32_t(uint32_t(st >> 32)) : -1;
                    }
                }
            }
            colBeginV = parkInVgpr(colBegin);
            colEndV = parkInVgpr(colEnd);
            segV = parkInVgpr(seg);
            blockV = parkInVgpr(block);
            flagsV = parkInVgpr(flags);
        }

        // Scans [from, colEnd) of the item: full-row blocks with scanColumns; triangle blocks with the emitting scan
        // over the columns strictly below the block, then the diagonal columns.  Returns the first column not scanned
        // (speculative scans stop when a log fills up).
#define EM2_SCAN_ITEM(SPEC, from, result)                                                                                   \
        do {                                                                                                                \
            const uint32_t colEnd_ = unpark(colEndV);                                                                       \
            uint32_t at_ = (from);                                                                                          \
            if (!(unpark(flagsV) & kItemTriangle)) {                                                                        \
                at_ = scanColumns<W32, IDENTITY, SPEC>(kernelArgs()->sig32, at_, colEnd_, r, row, lane, blockV, myList, twoK, \
                                                       count, mMax, myLog, logCapacity, logCount, ldsRaw);                  \
            } else {                                                                                                        \
                for (;;) {                                                                                                  \
                    const uint32_t colEndT_ = unpark(colEndV);                                                              \
                    const uint32_t rowBaseT_ = (unpark(blockV) * kernelArgs()->rowBlockStride + kernelArgs()->rowBlockOffset) * 64u;                                                        \
                    const uint32_t triEnd_ = colEndT_ < rowBaseT_ ? colEndT_ : rowBaseT_;                                   \
                    if (at_ >= triEnd_) break;                                                                              \
                    ensureInboxRoom(lane, emitPos, emitEnd);                                                                \
                    at_ = scanColumnsEmit<W32, IDENTITY, SPEC>(kernelArgs()->sig32, kernelArgs()->snap, at_, triEnd_, r, row, \
                                                               rowValid, lane, myList, twoK, count, mMax, myLog,            \
                                                               logCapacity, logCount, emitPos, emitEnd);                    \
                    uint32_t atV_ = parkInVgpr(at_);                                                                        \
                    if (SPEC) {                                                                                             \
                        if (__builtin_amdgcn_ballot_w64(logCount == logCapacity) != 0ull) break;                            \
                    } else {                                                                                                \
                        /* cut the row lists that reached 2k entries (no new candidate: pass = false) */                    \
                        acceptColumn<IDENTITY>(false, 0u, row, 0u, lane, unpark(blockV), myList, twoK, count, mMax, ldsRaw); \
                    }                                                                                                       \
                    at_ = unpark(atV_);                                                                                     \
                }                                                                                                           \
                const uint32_t colEnd2_ = unpark(colEndV);                                                                  \
                const uint32_t rowBase2_ = (unpark(blockV) * kernelArgs()->rowBlockStride + kernelArgs()->rowBlockOffset) * 64u;                                                            \
                const uint32_t triEnd2_ = colEnd2_ < rowBase2_ ? colEnd2_ : rowBase2_;                                      \
                /* a log that filled up at the very last column below the block must not take more entries */               \
                const bool logFull_ = SPEC && __builtin_amdgcn_ballot_w64(logCount == logCapacity) != 0ull;                 \
                if (at_ >= triEnd2_ && !logFull_) {                                                                         \
                    const uint32_t colBegin2_ = unpark(colBeginV);                                                          \
                    const uint32_t diagBegin_ = colBegin2_ > rowBase2_ ? colBegin2_ : rowBase2_;                            \
                    at_ = scanDiagonal<W32, IDENTITY, SPEC>(kernelArgs()->sig32, kernelArgs()->snap,                        \
                                                            at_ > diagBegin_ ? at_ : diagBegin_, colEnd2_, r, row, rowValid, \
                                                            lane, blockV, myList, twoK, count, mMax, myLog, logCapacity,    \
                                                            logCount, emitPos, emitEnd, ldsRaw);                            \
                }                                                                                                           \
            }                                                                                                               \
            (result) = at_;                                                                                                 \
        } while (0)

        uint32_t resumeV = colBeginV;
        if (unpark(flagsV) & kItemSpeculate) {
            uint32_t resume;
            EM2_SCAN_ITEM(true, unpark(colBeginV), resume);
            resumeV = parkInVgpr(resume);
        }

        if (unpark(segV) != 0u) {
            ArgsPtr aux = kernelArgs();
            const uint32_t seg = unpark(segV);
            const uint32_t block = unpark(blockV);
            const uint32_t* flag = aux->segmentsDone + block;
            uint32_t error = 0;
            const uint64_t start = __builtin_amdgcn_s_memrealtime();         // 100 MHz
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seg) {
                __builtin_amdgcn_s_sleep(16);
                if (__builtin_amdgcn_s_memrealtime() - start > 400000000ull) {
                    error = 1;
                    break;
                }
            }
            if (error) {
                if (lane == 0u) __hip_atomic_store(aux->control + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const uint64_t st = __hip_atomic_load(reinterpret_cast<const uint64_t*>(aux->rowState) + size_t(block) * 64u + lane,
                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            count = uint32_t(st);
            mMax = rowValid ? int32_t(uint32_t(st >> 32)) : -1;
            if (unpark(flagsV) & kItemSpeculate) {
                for (uint32_t i = 0;; ++i) {
                    const bool active = i < logCount;
                    if (__builtin_amdgcn_ballot_w64(active) == 0ull) break;
                    uint32_t c = 0, m = 0;
                    if (active) {
                        const Entry e = myLog[i];
                        c = e.cell;
                        m = e.key;
                    }
                    const bool pass = active && int32_t(m) <= mMax;
                    if (__builtin_amdgcn_ballot_w64(pass) != 0ull) {
                        acceptColumn<IDENTITY>(pass, c, row, m, lane, unpark(blockV), myList, twoK, count, mMax, ldsRaw);
                    }
                }
            }
        }

        // ---- exact scan of whatever the speculation did not cover ----
        {
            uint32_t unused;
            EM2_SCAN_ITEM(false, unpark(resumeV), unused);
            (void)unused;
        }
#undef EM2_SCAN_ITEM

        // ---- full-row block at its last segment: finish; otherwise publish the state (for the next segment, for
        // the columns' snapshots and, at a triangle block's last segment, for the inbox replay) ----
        {
            ArgsPtr aux = kernelArgs();
            const uint32_t block = unpark(blockV);
            const uint32_t flags = unpark(flagsV);
            const uint32_t shardFlags = aux->shardFlags;
            if (!(flags & kItemTriangle) && (flags & kItemLast) && !(shardFlags & kShardNoFinish)) {
                finishRows(lane, block, count, ldsRaw);
            } else {
                const uint64_t st = uint64_t(count) | (uint64_t(uint32_t(mMax)) << 32);
                __hip_atomic_store(reinterpret_cast<uint64_t*>(aux->rowState) + size_t(block) * 64u + lane, st,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (((flags & kItemTriangle) || (shardFlags & kShardPublishAll)) && rowValid) {
                    __hip_atomic_store(aux->snap + row, mMax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0u && !(flags & kItemLast)) {
                    __hip_atomic_store(aux->segmentsDone + block, unpark(segV) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }

    // the unused tail of this wave's last inbox chunk becomes sentinels
    {
        const uint32_t p = uint32_t(__builtin_amdgcn_readfirstlane(int(emitPos)));
        const uint32_t e = uint32_t(__builtin_amdgcn_readfirstlane(int(emitEnd)));
        if (p <= e) {
            uint64_t* inbox = kernelArgs()->inbox;
            for (uint32_t i = p + lane; i < e; i += 64u) inbox[i] = ~0ull;
        }
    }
}

// =========================================================================================================
// Matrix-core form of the triangle part (1024-bit signatures).  A signature bit becomes the FP4 (E2M1) value +1 or -1;
// the dot product of two cells over the 1024 (padded) bits is 1024 - 2m, exact in the f32 accumulator, and
// v_mfma_scale_f32_32x32x64_f8f6f4 (scales 2^0) contracts 64 bits of 32 x 32 cells per instruction: 8 times the
// pairs per SIMD clock of the v_xor/v_bcnt loop at its instruction floor (tools/ubench_mfma_pairs.hip).
//
// The contract of the scan does not change, only who counts.  A block of 4 waves owns 4 consecutive triangle row
// blocks (a "quad", 256 cells); its waves walk the columns below the quad in lock step, 32 at a time: the tile's
// fragments (16 KB, stored in fragment order so the copy is linear) go through a double-buffered LDS image shared by
// the 4 waves; each wave contracts the tile with its own 64 rows, which it holds as the B operand (128 VGPRs), so a
// row of the result sits on lane & 31; 16 v_permlane32_swap turn the two 32x32 results into "lane = row, register =
// column", the layout of the v_xor/v_bcnt loop, and every column is tested against the looser of the row's and the
// column's bound with one v_min, one v_cmp and one branch, exactly as there.  What passes is rare and takes the old
// paths: the column side goes to the inbox, the row side to the wave's log (the lock step cannot stop for a list
// that fills up), which the wave replays through the exact state machine when the walk is over -- the speculative
// mode of the other kernels, always on.  The last columns of a quad (its own 256 cells: the band below each wave's
// rows and the diagonal) are done by the v_xor/v_bcnt code, each wave on its own.
// Items are (segment, quad); segment and full-row boundaries are multiples of 256 cells, so the 4 waves of a block
// always have the same columns.  Full-row blocks stay with fsp4ScanSymmetricKernel (a launch of their own).
// =========================================================================================================

typedef int FragmentWord4 __attribute__((ext_vector_type(4)));
typedef int FragmentWord8 __attribute__((ext_vector_type(8)));
typedef float Accumulator16 __attribute__((ext_vector_type(16)));

constexpr int kColumnsPerBranch = 4;                        // kernel ms at 1M cells with 1 / 2 / 4 / 8 / 16: 289* / 261 / 260 / 273 / 297 (* before the longer segments)
constexpr uint32_t kMatrixSteps = 16;                        // 1024 bits / 64 per MFMA
constexpr uint32_t kMatrixTileWords = kMatrixSteps * 64u;    // FragmentWord4 per 32-cell tile (16 KB)
constexpr float kMatrixBits = 1024.f;
// Narrowest padded width (dwords) that takes the matrix form by default.  Scan kernel ms at 1M cells, v_xor/v_bcnt form /
// matrix form: 512 bits 486 / 274, 256 bits 349 / 302 (the zero-extended fragments cost the full 16 k-steps).
constexpr uint32_t kMatrixMinPaddedDw = 8;

// sig32 [cell][2 * steps] -> fragments [cell / 32][k-step][lane]: lane l of k-step s holds cell (l & 31) of the block, bits
// s*64 + (l >> 5)*32 .. +31, one nibble per bit (0x2 = +1, 0xA = -1).  Cells past the end repeat the last one.
// steps = 16 (1024 bits) or 32 (2048 bits).
__global__ void __launch_bounds__(256)
expandFragmentsKernel(const uint32_t* __restrict__ sig32, uint32_t cellCount, uint32_t fragmentCount,
                      FragmentWord4* __restrict__ out, uint32_t steps = kMatrixSteps)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= fragmentCount) return;
    const uint32_t lane = i & 63u, step = (i >> 6) % steps, block = (i >> 6) / steps;
    uint32_t cell = block * 32u + (lane & 31u);
    if (cell >= cellCount) cell = cellCount - 1u;
    const uint32_t word = sig32[size_t(cell) * (2u * steps) + step * 2u + (lane >> 5)];
    FragmentWord4 v;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        uint32_t packed = 0;
#pragma unroll
        for (int n = 0; n < 8; n++) packed |= (((word >> (d * 8 + n)) & 1u) ? 0xAu : 0x2u) << (4 * n);
        v[d] = int(packed);
    }
    out[i] = v;
}

// sig32 [cell][paddedDw] -> [cell][32], zero-extended: what the matrix kernel's v_xor/v_bcnt parts and the fragment
// expansion read when the signatures are narrower than 1024 bits.
__global__ void __launch_bounds__(256)
widenSignaturesKernel(const uint32_t* __restrict__ sig32, uint32_t paddedDw, uint32_t cellCount, uint32_t* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cellCount * 32u) return;
    const uint32_t cell = i >> 5, word = i & 31u;
    out[i] = word < paddedDw ? sig32[size_t(cell) * paddedDw + word] : 0u;
}

// emitColumn for the walk below: the pool pointer and the key layout arrive in registers (emitColumn re-reads them
// from the kernarg segment on purpose, which costs the matrix kernel a scalar-load round trip per event); only a chunk
// that is used up goes through the out-of-line path.
__device__ __forceinline__ void emitColumnFast(bool emit, uint32_t target, uint32_t candidate, uint32_t m, uint32_t lane,
                                               uint32_t& emitPos, uint32_t& emitEnd, uint64_t* inbox, uint32_t rowBits)
{
    const uint64_t mask = __builtin_amdgcn_ballot_w64(emit);
    if (mask == 0ull) return;
    const uint32_t p = uint32_t(__builtin_amdgcn_readfirstlane(int(emitPos)));
    const uint32_t e = uint32_t(__builtin_amdgcn_readfirstlane(int(emitEnd)));
    const uint32_t n = uint32_t(__builtin_popcountll(mask));
    if (p > e || p + n > e) {
        emitColumn(emit, target, candidate, m, lane, emitPos, emitEnd);      // disabled after an overflow, or a new chunk
        return;
    }
    if (emit) inbox[p + lanesBelow(mask)] = (uint64_t(target) << (13u + rowBits)) | (uint64_t(candidate) << 13u) | uint64_t(m);
    emitPos = p + n;
}

// The lock-step walk over the tiles [colBegin, colEnd) (multiples of 32).  Returns the first column not scanned, the
// same in all waves of the block: the walk ends early, at a tile boundary, when some row's log could overflow in the
// next tile.  stopWords: 3 LDS words, zero on entry and on return.
// BOTH (the tile kernel of the sharded scan): the row side is deferred to the inbox as well, nothing is logged and the
// walk never stops early.
template <bool IDENTITY, bool BOTH = false>
__device__ __forceinline__ uint32_t scanTilesMatrix(const FragmentWord4* __restrict__ fragments, const int32_t* snap,
                                                    uint32_t colBegin, uint32_t colEnd, uint32_t rowFragmentBlock,
                                                    float rowDot, uint32_t row, bool rowValid, uint32_t lane,
                                                    Entry* myLog, uint32_t logCapacity, uint32_t& logCount,
                                                    uint32_t& emitPos, uint32_t& emitEnd, FragmentWord4* tiles,
                                                    volatile uint32_t* stopWords)
{
    const int scale = 0x7f7f7f7f;                // E8M0 127 = 2^0 in every byte
    uint64_t* const inbox = kernelArgs()->inbox;
    const uint32_t rowBits = kernelArgs()->rowBits;
    FragmentWord4 rows[2][kMatrixSteps];
#pragma unroll
    for (int t = 0; t < 2; t++) {
#pragma unroll
        for (int s = 0; s < int(kMatrixSteps); s++) {
            rows[t][s] = fragments[(size_t(rowFragmentBlock + uint32_t(t)) * kMatrixSteps + uint32_t(s)) * 64u + lane];
        }
    }
    // A tile travels global -> LDS without touching registers (global_load_lds_dwordx4: the LDS address is the wave's
    // base + lane * 16, which is exactly the fragment order).  Two tiles (64 columns) per barrier: the four waves may
    // drift by a tile, which absorbs the difference between a tile with events and one without; the pair after the one
    // being contracted is on its way meanwhile.
    const uint32_t waveSlot = uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6))) * 64u;
#define EM2_STAGE_TILE(tileIndex, buffer)                                                                                     \
    do {                                                                                                                      \
        const FragmentWord4* src_ = fragments + size_t(tileIndex) * kMatrixTileWords + threadIdx.x;                          \
        FragmentWord4* dst_ = tiles + (buffer) * kMatrixTileWords + waveSlot;                                                \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; j_++) {                                                                   \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_ + j_ * 256),              \
                                             (__attribute__((address_space(3))) void*)(dst_ + j_ * 256), 16, 0, 0);          \
        }                                                                                                                     \
    } while (0)
    EM2_STAGE_TILE(colBegin / 32u, 0u);
    if (colBegin + 32u < colEnd) EM2_STAGE_TILE(colBegin / 32u + 1u, 1u);
    __syncthreads();
    uint32_t iteration = 0;
    // The bounds of columns (lane & 31) of the two tiles, fetched one pair ahead like the fragments (the compiler sinks
    // a load placed in front of the MFMAs to its first use behind them, and the wave then sits out a global-load latency
    // per tile).  Any value a cell published earlier is valid: bounds only tighten.
    int32_t snapAhead[2];
    snapAhead[0] = snap[colBegin + (lane & 31u)];
    snapAhead[1] = colBegin + 32u < colEnd ? snap[colBegin + 32u + (lane & 31u)] : 0;
    for (uint32_t colBase = colBegin; colBase < colEnd; colBase += 64u, ++iteration) {
        const uint32_t pair = iteration & 1u;
        const int32_t snapLane[2] = {snapAhead[0], snapAhead[1]};
        if (colBase + 64u < colEnd) {
            EM2_STAGE_TILE(colBase / 32u + 2u, 2u * (pair ^ 1u));
            snapAhead[0] = snap[colBase + 64u + (lane & 31u)];
        }
        if (colBase + 96u < colEnd) {
            EM2_STAGE_TILE(colBase / 32u + 3u, 2u * (pair ^ 1u) + 1u);
            snapAhead[1] = snap[colBase + 96u + (lane & 31u)];
        }
#pragma unroll
        for (int sub = 0; sub < 2; sub++) {
            const uint32_t tileBase = colBase + 32u * uint32_t(sub);
            if (tileBase >= colEnd) break;
            const float columnDotLane = kMatrixBits - 2.f * float(snapLane[sub]);
            Accumulator16 acc0 = {}, acc1 = {};
            const FragmentWord4* tile = tiles + (2u * pair + uint32_t(sub)) * kMatrixTileWords;
            // column fragments four k-steps ahead of their MFMAs (the LDS latency of a read is two MFMA pairs long)
            FragmentWord4 ahead[4];
#pragma unroll
            for (int s = 0; s < 4; s++) ahead[s] = tile[s * 64 + int(lane)];
            __builtin_amdgcn_s_setprio(2);          // the SIMD's other wave is in its column tests: MFMAs first
#pragma unroll
            for (int s = 0; s < int(kMatrixSteps); s++) {
                const FragmentWord4 a = ahead[s & 3];
                if (s + 4 < int(kMatrixSteps)) ahead[s & 3] = tile[(s + 4) * 64 + int(lane)];
                const FragmentWord8 a8 = {a.x, a.y, a.z, a.w, 0, 0, 0, 0};
                const FragmentWord8 b0 = {rows[0][s].x, rows[0][s].y, rows[0][s].z, rows[0][s].w, 0, 0, 0, 0};
                const FragmentWord8 b1 = {rows[1][s].x, rows[1][s].y, rows[1][s].z, rows[1][s].w, 0, 0, 0, 0};
                acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b0, acc0, 4, 4, 0, scale, 0, scale);
                acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b1, acc1, 4, 4, 0, scale, 0, scale);
            }
            __builtin_amdgcn_s_setprio(0);
            // lane = row: acc0[i] <- column (i&3) + 8*(i>>2), acc1[i] <- that + 4
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const auto swapped = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc0[i]), __float_as_uint(acc1[i]), false, false);
                acc0[i] = __uint_as_float(swapped[0]);
                acc1[i] = __uint_as_float(swapped[1]);
            }
            // Four columns per branch: the per-column compares are OR-ed as lane masks on the scalar unit, and only a group
            // in which something passes looks at its columns one by one (a branch per column cost more than the compares:
            // 1.6 -> 2.4 * 10^12 pairs/s in tools/ubench_mfma_pairs.hip, where nothing ever passes and groups are 8 wide; here
            // about two events per tile make 4 the best width).
#pragma unroll
            for (int g = 0; g < 32 / kColumnsPerBranch; g++) {
                float columnDots[kColumnsPerBranch];
                bool any = false;
#pragma unroll
                for (int w = 0; w < kColumnsPerBranch; w++) {
                    const int c = kColumnsPerBranch * g + w;       // column c sits in acc0 / acc1 as the swaps left it
                    const float dot = (c & 7) < 4 ? acc0[4 * (c >> 3) + (c & 7)] : acc1[4 * (c >> 3) + (c & 7) - 4];
                    columnDots[w] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(columnDotLane), c));
                    // min(rowDot, columnDot) as one v_med3_f32 (fminf would canonicalise both inputs first)
                    any |= dot >= __builtin_amdgcn_fmed3f(rowDot, columnDots[w], -INFINITY);
                }
                if (__builtin_amdgcn_ballot_w64(any) == 0ull) continue;
#pragma unroll
                for (int w = 0; w < kColumnsPerBranch; w++) {
                    const int c = kColumnsPerBranch * g + w;       // column c sits in acc0 / acc1 as the swaps left it
                    const float dot = (c & 7) < 4 ? acc0[4 * (c >> 3) + (c & 7)] : acc1[4 * (c >> 3) + (c & 7) - 4];
                    const float columnDot = columnDots[w];
                    if (__builtin_amdgcn_ballot_w64(dot >= __builtin_amdgcn_fmed3f(rowDot, columnDot, -INFINITY)) != 0ull) {
                        const uint32_t col = tileBase + uint32_t(c);
                        const uint32_t m = uint32_t((kMatrixBits - dot) * 0.5f);
                        emitColumnFast(rowValid && dot >= columnDot, col, row, m, lane, emitPos, emitEnd, inbox, rowBits);
                        if (BOTH) {
                            emitColumnFast(rowValid && dot >= rowDot, row, col, m, lane, emitPos, emitEnd, inbox, rowBits);
                        } else if (dot >= rowDot) {
                            storeEntry(myLog + logCount, col, m);
                            ++logCount;
                        }
                    }
                }
            }
        }
        // a pair of tiles adds at most 64 entries to a row's log
        const bool full = !BOTH && __builtin_amdgcn_ballot_w64(logCount + 64u > logCapacity) != 0ull;
        const uint32_t slot = iteration % 3u;
        if (full && lane == 0u) stopWords[slot] = 1u;
        if (threadIdx.x == 0u) stopWords[(iteration + 1u) % 3u] = 0u;
        __syncthreads();
        if (stopWords[slot] != 0u) {
            __syncthreads();
            if (threadIdx.x == 0u) stopWords[slot] = 0u;
            __syncthreads();
            return colBase + 64u < colEnd ? colBase + 64u : colEnd;
        }
    }
#undef EM2_STAGE_TILE
    return colEnd;
}

// =========================================================================================================
// The same walk with the tile step in hand-scheduled assembly (em2_matrix_step_asm.h, written by
// tools/gen_matrix_step_asm.py) -- the form the kernels use; scanTilesMatrix above is kept for A/B runs
// (EM2_MATRIX_WALK=0).  What changes against it:
//  * the 32 MFMAs of a tile are fed through a four-deep ring of column fragments with counted lgkmcnt waits: one
//    wave alone keeps the matrix pipe of its SIMD busy (the compiler's schedule of the loop above waits for
//    lgkmcnt(0) in front of every second k-step);
//  * the results stay in the accumulator layout -- lane l, register i of accumulator a = row 32a + (l & 31), column
//    8 (i >> 2) + 4 (l >> 5) + (i & 3) -- and are tested there against min(row bound, column bound): no
//    v_permlane32_swap, no v_readlane; the column bounds of a tile travel through 128 bytes of LDS per wave and come
//    back as four 16-byte reads per lane half;
//  * two accumulator sets: the step of tile t carries the test of tile t-1 between its MFMAs (2 VALU + 1 SALU per
//    result), so a wave never leaves the matrix pipe idle for its column tests;
//  * a result that passes is only LOGGED by the step (one 8-byte record into the log of the lane and accumulator it
//    passed in); which side of the pair it is for, the exact state machine and the inbox are the replay's
//    (replayWalkLogs, drainWalkLogs), which handles many records per lane at a time instead of a few per step.
// Everything the step touches is pinned to physical registers (register map in the generator).  The walk may stop only
// at a pair boundary, where one tile is still untested: the stop rule keeps room for three tiles.
// =========================================================================================================

// LDS byte address of a pointer into the block's dynamic LDS
template <typename T>
__device__ __forceinline__ uint32_t ldsAddress(T* p)
{
    return uint32_t(uintptr_t((__attribute__((address_space(3))) char*)(p)));
}

typedef __attribute__((address_space(3))) float* LdsFloatPtr;
typedef __attribute__((address_space(3))) FragmentWord4* LdsFragmentPtr;
typedef volatile __attribute__((address_space(3))) uint32_t* LdsWordPtr;
typedef const __attribute__((address_space(1))) FragmentWord4* GlobalFragmentPtr;
typedef const __attribute__((address_space(1))) int32_t* GlobalIntPtr;
typedef __attribute__((address_space(1))) uint64_t* GlobalWord64Ptr;

constexpr uint32_t kMatrixLogMargin = 96u;
// inboxControl (256 bytes): words 0..3 inbox position / overflow, byte 32 the longest inbox, 40..63 and 64..127 the cycle
// counters of the diagnostic build, 128..143 the sums of shader-clock and 100 MHz wall-clock ticks of the matrix kernel's blocks
constexpr uint32_t kClockWordsOffset = 32u;     // in 32-bit words

typedef const __attribute__((address_space(3))) int32_t* LdsIntPtr;

// LDS byte address -> pointer (32 bits on the device; the detour keeps the host pass of the compiler quiet)
template <typename P>
__device__ __forceinline__ P ldsPointer(uint32_t address)
{
    return (P)(uintptr_t)address;
}

// Per-wave LDS block of the walk (byte offsets).
constexpr uint32_t kWalkRowDot = 0u;            // float[64]: bound of row r as a dot product (1024 - 2 mMax), read by the steps
constexpr uint32_t kWalkBounds = 256u;          // float[4][32]: column bounds of the four tile buffers (as dot products)
constexpr uint32_t kWalkSnapStage = 768u;       // int32[2][64]: the published cut-offs of a pair of tiles, as loaded
constexpr uint32_t kMatrixWalkLdsBytes = 1280u;

__device__ __forceinline__ uint32_t uniform(uint32_t x) { return uint32_t(__builtin_amdgcn_readfirstlane(int(x))); }

__device__ __forceinline__ uint64_t uniform64(uint64_t v)
{
    return uint64_t(uniform(uint32_t(v))) | (uint64_t(uniform(uint32_t(v >> 32))) << 32);
}

// The lane id.  (While the walk still handled its events between the steps this was a volatile asm, recomputed wherever it
// was used, so that nothing derived from it stayed alive across a step, where it would have needed one of the few
// registers the steps leave to the compiler; the walk's own code is small now and keeps it.)
__device__ __forceinline__ uint32_t laneId()
{
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// A record of the walk: the step's stub wrote {first column of the tile | 2i + a, dot} into the log of the LANE in which
// register i of accumulator a passed its test (em2_matrix_step_asm.h).  Lane s = 32 h + t holds rows t and 32 + t of the
// wave and the columns 8q + 4h + j of a tile (i = 4q + j): its records ascend in the column.
struct WalkRecord {
    uint32_t code;                  // tile's first column (a multiple of 32) | 2i + a
    float dot;                      // 1024 - 2 * mismatches
};
__device__ __forceinline__ uint32_t walkRecordColumn(uint32_t code, uint32_t half)
{
    const uint32_t i = (code & 31u) >> 1;
    return (code & ~31u) + 8u * (i >> 2) + 4u * half + (i & 3u);
}
// (the records of a lane are written by that lane and read by others: past the L1, like the row lists)
__device__ __forceinline__ WalkRecord loadWalkRecord(const Entry* log, uint32_t index)
{
    const Entry e = loadEntryCoherent(log + index);
    WalkRecord r;
    r.code = e.cell;
    r.dot = __uint_as_float(e.key);
    return r;
}

// The lock-step walk over the tiles [colBegin, colEnd) with the hand-scheduled steps.  Out of line: the steps own
// v28..v255, and inlined into the kernels the values that live across the walk compete with them; as a function of its
// own the walk keeps next to nothing in vector registers across a step, and nothing of the compiler's may ever sit at
// v64 or above (tools/check_matrix_walk_registers.py checks the compiled code).  Wave-uniform arguments arrive in vector
// registers and are moved to the scalar file first; pointers get their address spaces back (a generic pointer would make
// the compiler emit flat_ instructions, whose out-of-order completion would also break the counted LDS waits of the
// steps).  LDS arguments are byte addresses.
// The walk only LOGS what passes min(row bound, column bound): `waveLog` is the wave's log area, per lane two logs of
// logCapacity / 2 records, one per accumulator (walkLogOf); recordCount[a] (in / out) = the calling lane's number of
// records in its log of accumulator a.  It returns the first column not scanned, the same in all waves of the block: it
// ends early, at a pair boundary, when some log could overflow within the next three tiles (48 records: a tile has 16
// registers per accumulator).  The caller replays the logs (replayWalkLogs / drainWalkLogs) and calls again.
template <bool IDENTITY, bool BOTH = false, bool DIAG = false>
__device__ __attribute__((noinline)) uint32_t scanTilesMatrixPinned(const void* auxArg, const void* fragmentsArg, const void* snapArg,
                                                                    uint32_t colBeginArg, uint32_t colEndArg,
                                                                    uint32_t rowFragmentBlockArg, float rowDotArg,
                                                                    Entry* waveLogArg, uint32_t logCapacityArg, uint32_t* recordCount,
                                                                    uint32_t tilesLdsArg, uint32_t stopWordsLdsArg, uint32_t walkLdsArg)
{
    const GlobalFragmentPtr fragments = (GlobalFragmentPtr)uniform64(reinterpret_cast<uint64_t>(fragmentsArg));
    const GlobalIntPtr snap = (GlobalIntPtr)uniform64(reinterpret_cast<uint64_t>(snapArg));
    const uint32_t colBegin = uniform(colBeginArg), colEnd = uniform(colEndArg);
    const uint32_t rowFragmentBlock = uniform(rowFragmentBlockArg), logCapacity = uniform(logCapacityArg);
    const uint32_t tilesLds = uniform(tilesLdsArg);
    const LdsWordPtr stopWords = ldsPointer<LdsWordPtr>(uniform(stopWordsLdsArg));
    const uint32_t walkLds = uniform(walkLdsArg);
    const LdsFloatPtr boundScratch = ldsPointer<LdsFloatPtr>(walkLds + kWalkBounds);
    const LdsIntPtr snapStage = ldsPointer<LdsIntPtr>(walkLds + kWalkSnapStage);
    const uint64_t logBase = uniform64(reinterpret_cast<uint64_t>(waveLogArg));
    // (EM2_MATRIX_DIAG, measurements only: the walk that looks at the bits is an instantiation of its own, so that the
    // one that runs in production has none of their branches between its steps)
    const uint32_t diag = DIAG ? EM2_DIAG_WORD((ArgsPtr)uniform64(reinterpret_cast<uint64_t>(auxArg))) : 0u;
    const uint32_t halfCapacity = logCapacity / 2u;
    // byte offsets into the wave's log area: where the lane's two logs begin, where its next records go (kept in two
    // registers of the walk; the steps return them), and beyond which the walk has to stop
    const uint32_t firstOffset0 = laneId() * logCapacity * uint32_t(sizeof(Entry));
    const uint32_t firstOffset1 = firstOffset0 + halfCapacity * uint32_t(sizeof(Entry));
    const uint32_t stopRecords = halfCapacity > kMatrixLogMargin / 2u ? halfCapacity - kMatrixLogMargin / 2u : 0u;
    const uint32_t stopOffset0 = firstOffset0 + stopRecords * uint32_t(sizeof(Entry));
    const uint32_t stopOffset1 = firstOffset1 + stopRecords * uint32_t(sizeof(Entry));
    uint32_t recordOffset = firstOffset0 + recordCount[0] * uint32_t(sizeof(Entry));
    uint32_t recordOffset1 = firstOffset1 + recordCount[1] * uint32_t(sizeof(Entry));
    {
        const uint32_t lane = laneId();
        ldsPointer<LdsFloatPtr>(walkLds + kWalkRowDot)[lane] = rowDotArg;         // for the steps: float[64], lane = row
        asm volatile(EM2_MATRIX_SET_RECORD_OFFSETS : : "v"(recordOffset), "v"(recordOffset1) : EM2_MATRIX_OWNED_REGISTERS);
        // the B operand: the 2 x 16 fragments of the wave's rows straight into their registers (v128..v255)
        const uint64_t rowFragments = reinterpret_cast<uint64_t>(fragments) + size_t(rowFragmentBlock) * kMatrixTileWords * 16u;
        asm volatile(EM2_MATRIX_LOAD_ROWS : : "s"(rowFragments) : EM2_MATRIX_STEP_CLOBBERS);
    }
    const uint32_t waveSlot = uniform(threadIdx.x >> 6) * 64u;
    // A tile travels global -> LDS without touching registers (global_load_lds_dwordx4: LDS address = M0 + 16 * lane,
    // which is exactly the fragment order; this wave moves its quarter, 4 x 1 KB).  Issued from inline asm: the compiler
    // must not know of these transfers -- it orders every LDS access of its own behind an LDS-DMA it has seen with
    // s_waitcnt vmcnt(0).  What needs the tiles waits for them explicitly in front of the barrier.  The lane addresses are
    // 64-bit vector registers: the form with a scalar base and a 32-bit lane offset left 1 KB pieces of a tile stale now
    // and then (tests/test_gpu_fsp4.py::test_sharded_tile_walk_repeats_bit_identically).
#define EM2_STAGE_TILE(tileIndex, buffer)                                                                                     \
    do {                                                                                                                      \
        if (diag & 16u) break;                                                                                                \
        const uint64_t src_ = reinterpret_cast<uint64_t>(fragments) + (size_t(tileIndex) * kMatrixTileWords + waveSlot) * 16u + \
                              laneId() * 16u;                                                                                 \
        const uint32_t dst_ = tilesLds + ((buffer) * kMatrixTileWords + waveSlot) * 16u;                                     \
        /* (one address register per piece: the instruction's offset field would move the LDS address as well) */            \
        asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"                                      \
                     "s_add_u32 m0, %4, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"                              \
                     "s_add_u32 m0, %4, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"                              \
                     "s_add_u32 m0, %4, 0x3000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off"                                  \
                     :                                                                                                        \
                     : "v"(src_), "v"(src_ + 0x1000u), "v"(src_ + 0x2000u), "v"(src_ + 0x3000u), "s"(dst_)                   \
                     : "memory", "m0", "scc");                                                                                \
    } while (0)
    // The published cut-offs of the 64 columns of a pair of tiles (lane = column) travel the same way one pair ahead
    // (columns past the end repeat the last one: never tested).  Any value a cell published earlier is valid: bounds
    // only tighten.
#define EM2_STAGE_SNAP(firstColumn, buffer)                                                                                   \
    do {                                                                                                                      \
        uint32_t column_ = (firstColumn) + laneId();                                                                          \
        column_ = column_ < colEnd ? column_ : colEnd - 1u;                                                                   \
        const uint64_t address_ = reinterpret_cast<uint64_t>(snap) + uint64_t(column_) * 4u;                                 \
        const uint32_t dst_ = walkLds + kWalkSnapStage + (buffer) * 256u;                                                     \
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off"                                           \
                     :                                                                                                        \
                     : "v"(address_), "s"(dst_)                                                                               \
                     : "memory", "m0");                                                                                       \
    } while (0)
#define EM2_WAIT_STAGED() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
    EM2_STAGE_TILE(colBegin / 32u, 0u);
    if (colBegin + 32u < colEnd) EM2_STAGE_TILE(colBegin / 32u + 1u, 1u);
    EM2_STAGE_SNAP(colBegin, 0u);
    EM2_WAIT_STAGED();
    __syncthreads();
    bool tested = false;
    uint64_t passScratch[5];        // scalar pairs for the steps: pass masks in flight, saved exec
    bool pending = false, pendingInY = false;
    uint32_t pendingBase = 0, pendingSlot = 0;
    uint32_t iteration = 0, stopSlot = 0;
    uint32_t result = colEnd;
    // the staged cut-offs of the pair about to be walked (lane = column); those of the next pair are read right behind
    // the barrier that ends a pair, together with the stop word: one LDS round trip there instead of two
    int32_t stagedSnap = snapStage[laneId()];
    for (uint32_t colBase = colBegin; colBase < colEnd; colBase += 64u, ++iteration) {
        const uint32_t pair = iteration & 1u;
        boundScratch[pair * 64u + laneId()] = kMatrixBits - 2.f * float(stagedSnap);
        if (colBase + 64u < colEnd) {
            EM2_STAGE_TILE(colBase / 32u + 2u, 2u * (pair ^ 1u));
            EM2_STAGE_SNAP(colBase + 64u, pair ^ 1u);
        }
        if (colBase + 96u < colEnd) EM2_STAGE_TILE(colBase / 32u + 3u, 2u * (pair ^ 1u) + 1u);
        // ---- first tile of the pair -> X, under it the test of the pending tile (always in Y here) ----
        {
            const uint32_t tileBase = tilesLds + 2u * pair * (kMatrixTileWords * 16u);
            if (pending && !(diag & 32u)) {
                const uint32_t boundBase = walkLds + kWalkBounds + pendingSlot * 128u;
                asm volatile(EM2_MATRIX_STEP_X_TESTING_Y
                             : "=v"(recordOffset), "=v"(recordOffset1), "=&s"(passScratch[0]), "=&s"(passScratch[1]), "=&s"(passScratch[2]), "=&s"(passScratch[3]), "=&s"(passScratch[4])
                             : "s"(tileBase), "s"(boundBase), "s"(walkLds + kWalkRowDot), "s"(logBase), "s"(pendingBase)
                             : EM2_MATRIX_STEP_CLOBBERS);
                tested = true;
            } else {
                asm volatile(EM2_MATRIX_STEP_X : : "s"(tileBase) : EM2_MATRIX_STEP_CLOBBERS);
            }
            pending = true;
            pendingInY = false;
            pendingBase = colBase;
            pendingSlot = 2u * pair;
        }
        // ---- second tile -> Y, under it the test of the first ----
        if (colBase + 32u < colEnd && (diag & 32u)) {
            const uint32_t tileBase = tilesLds + (2u * pair + 1u) * (kMatrixTileWords * 16u);
            asm volatile(EM2_MATRIX_STEP_Y : : "s"(tileBase) : EM2_MATRIX_STEP_CLOBBERS);
            pendingInY = true;
        } else if (colBase + 32u < colEnd) {
            const uint32_t tileBase = tilesLds + (2u * pair + 1u) * (kMatrixTileWords * 16u);
            const uint32_t boundBase = walkLds + kWalkBounds + pendingSlot * 128u;
            asm volatile(EM2_MATRIX_STEP_Y_TESTING_X
                         : "=v"(recordOffset), "=v"(recordOffset1), "=&s"(passScratch[0]), "=&s"(passScratch[1]), "=&s"(passScratch[2]), "=&s"(passScratch[3]), "=&s"(passScratch[4])
                         : "s"(tileBase), "s"(boundBase), "s"(walkLds + kWalkRowDot), "s"(logBase), "s"(pendingBase)
                         : EM2_MATRIX_STEP_CLOBBERS);
            tested = true;
            pendingInY = true;
            pendingBase = colBase + 32u;
            pendingSlot = 2u * pair + 1u;
        }
        if (diag & 1u) {            // (measurements: the records are written, then dropped)
            asm volatile(EM2_MATRIX_SET_RECORD_OFFSETS : : "v"(firstOffset0), "v"(firstOffset1) : EM2_MATRIX_OWNED_REGISTERS);
            recordOffset = firstOffset0;
            recordOffset1 = firstOffset1;
        }
        // the untested tile and the next pair add at most 48 records to a log before the next chance to stop
        const bool full = __builtin_amdgcn_ballot_w64(recordOffset > stopOffset0 || recordOffset1 > stopOffset1) != 0ull;
        const uint32_t slot = stopSlot;
        stopSlot = stopSlot == 2u ? 0u : stopSlot + 1u;              // (iteration % 3, without the division)
        if (full && laneId() == 0u) stopWords[slot] = 1u;
        if (waveSlot == 0u && laneId() == 0u) stopWords[stopSlot] = 0u;
        if (!(diag & 128u)) EM2_WAIT_STAGED();          // (128: measurements only -- the tiles are used before they have arrived)
        if (!(diag & 64u)) __syncthreads();
        const uint32_t stop = stopWords[slot];
        stagedSnap = snapStage[(pair ^ 1u) * 64u + laneId()];
        if (stop != 0u) {
            __syncthreads();
            if (waveSlot == 0u && laneId() == 0u) stopWords[slot] = 0u;
            __syncthreads();
            result = colBase + 64u < colEnd ? colBase + 64u : colEnd;
            break;
        }
    }
#undef EM2_STAGE_TILE
#undef EM2_STAGE_SNAP
#undef EM2_WAIT_STAGED
    // ---- the tile still untested ----
    if (pending && !(diag & 32u)) {
        const uint32_t boundBase = walkLds + kWalkBounds + pendingSlot * 128u;
        if (pendingInY) {
            asm volatile(EM2_MATRIX_TEST_Y
                         : "=v"(recordOffset), "=v"(recordOffset1), "=&s"(passScratch[0]), "=&s"(passScratch[1]), "=&s"(passScratch[2]), "=&s"(passScratch[3]), "=&s"(passScratch[4])
                         : "s"(boundBase), "s"(walkLds + kWalkRowDot), "s"(logBase), "s"(pendingBase) : EM2_MATRIX_STEP_CLOBBERS);
        } else {
            asm volatile(EM2_MATRIX_TEST_X
                         : "=v"(recordOffset), "=v"(recordOffset1), "=&s"(passScratch[0]), "=&s"(passScratch[1]), "=&s"(passScratch[2]), "=&s"(passScratch[3]), "=&s"(passScratch[4])
                         : "s"(boundBase), "s"(walkLds + kWalkRowDot), "s"(logBase), "s"(pendingBase) : EM2_MATRIX_STEP_CLOBBERS);
        }
        tested = true;
    }
    // the records were stored by one lane and are read back by others: the stores must have left the wave before the
    // caller replays the logs (it reads past the L1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tested && !(diag & 1u)) {
        recordCount[0] = (recordOffset - firstOffset0) / uint32_t(sizeof(Entry));
        recordCount[1] = (recordOffset1 - firstOffset1) / uint32_t(sizeof(Entry));
    }
    return result;
}

// The same walk for 2048-bit signatures (EM2_MATRIX_WIDE_*: the registers hold 32 rows x 32 k-steps, a tile is 32 columns x
// 32 k-steps = two 16 KB slots side by side, a step is one tile).  One call walks the columns for ONE half of the wave's 64
// rows: rowHalf a = rows 32a .. 32a+31, whose fragments are the 32 KB block rowFragmentBlock (in 32-cell blocks), whose
// records go to the lanes' logs of accumulator a (recordCount[a]).  The walk may stop at any tile boundary, where one tile
// is still untested: a log needs room for two tiles (32 records).
template <bool IDENTITY>
__device__ __attribute__((noinline)) uint32_t scanTilesMatrixWide(const void* fragmentsArg, const void* snapArg, uint32_t colBeginArg,
                                                                  uint32_t colEndArg, uint32_t rowFragmentBlockArg, float rowDotArg,
                                                                  uint32_t rowHalfArg, Entry* waveLogArg, uint32_t logCapacityArg,
                                                                  uint32_t* recordCount, uint32_t tilesLdsArg, uint32_t stopWordsLdsArg,
                                                                  uint32_t walkLdsArg)
{
    const GlobalFragmentPtr fragments = (GlobalFragmentPtr)uniform64(reinterpret_cast<uint64_t>(fragmentsArg));
    const GlobalIntPtr snap = (GlobalIntPtr)uniform64(reinterpret_cast<uint64_t>(snapArg));
    const uint32_t colBegin = uniform(colBeginArg), colEnd = uniform(colEndArg);
    const uint32_t rowFragmentBlock = uniform(rowFragmentBlockArg), logCapacity = uniform(logCapacityArg);
    const uint32_t rowHalf = uniform(rowHalfArg);
    const uint32_t tilesLds = uniform(tilesLdsArg);
    const LdsWordPtr stopWords = ldsPointer<LdsWordPtr>(uniform(stopWordsLdsArg));
    const uint32_t walkLds = uniform(walkLdsArg);
    const LdsFloatPtr boundScratch = ldsPointer<LdsFloatPtr>(walkLds + kWalkBounds);
    const LdsIntPtr snapStage = ldsPointer<LdsIntPtr>(walkLds + kWalkSnapStage);
    const uint64_t logBase = uniform64(reinterpret_cast<uint64_t>(waveLogArg));
    const uint32_t halfCapacity = logCapacity / 2u;
    const uint32_t firstOffset = (laneId() * logCapacity + rowHalf * halfCapacity) * uint32_t(sizeof(Entry));
    const uint32_t stopRecords = halfCapacity > kMatrixLogMargin / 2u ? halfCapacity - kMatrixLogMargin / 2u : 0u;
    const uint32_t stopOffset = firstOffset + stopRecords * uint32_t(sizeof(Entry));
    uint32_t recordOffset = firstOffset + recordCount[rowHalf] * uint32_t(sizeof(Entry));
    uint32_t unusedOffset = 0;
    const uint32_t stateBase = walkLds + kWalkRowDot + 128u * rowHalf;       // the half's 32 row bounds
    {
        ldsPointer<LdsFloatPtr>(walkLds + kWalkRowDot)[laneId()] = rowDotArg;     // float[64], lane = row
        asm volatile(EM2_MATRIX_SET_RECORD_OFFSETS : : "v"(recordOffset), "v"(recordOffset) : EM2_MATRIX_OWNED_REGISTERS);
        const uint64_t rowFragments = reinterpret_cast<uint64_t>(fragments) + size_t(rowFragmentBlock) * (2u * kMatrixTileWords * 16u);
        asm volatile(EM2_MATRIX_LOAD_ROWS : : "s"(rowFragments) : EM2_MATRIX_STEP_CLOBBERS);
    }
    const uint32_t waveSlot = uniform(threadIdx.x >> 6) * 64u;
    // (global -> LDS as in scanTilesMatrixPinned, in units of 16 KB: unit u of the fragment array into slot `buffer`)
#define EM2_STAGE_UNIT(unitIndex, buffer)                                                                                     \
    do {                                                                                                                      \
        const uint64_t src_ = reinterpret_cast<uint64_t>(fragments) + (size_t(unitIndex) * kMatrixTileWords + waveSlot) * 16u + \
                              laneId() * 16u;                                                                                 \
        const uint32_t dst_ = tilesLds + ((buffer) * kMatrixTileWords + waveSlot) * 16u;                                     \
        asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"                                      \
                     "s_add_u32 m0, %4, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"                              \
                     "s_add_u32 m0, %4, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"                              \
                     "s_add_u32 m0, %4, 0x3000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off"                                  \
                     :                                                                                                        \
                     : "v"(src_), "v"(src_ + 0x1000u), "v"(src_ + 0x2000u), "v"(src_ + 0x3000u), "s"(dst_)                   \
                     : "memory", "m0", "scc");                                                                                \
    } while (0)
#define EM2_STAGE_WIDE_TILE(firstColumn, parity)                                                                              \
    do {                                                                                                                      \
        EM2_STAGE_UNIT((firstColumn) / 16u, 2u * (parity));                                                                   \
        EM2_STAGE_UNIT((firstColumn) / 16u + 1u, 2u * (parity) + 1u);                                                         \
        uint32_t column_ = (firstColumn) + (laneId() & 31u);                                                                  \
        column_ = column_ < colEnd ? column_ : colEnd - 1u;                                                                   \
        const uint64_t address_ = reinterpret_cast<uint64_t>(snap) + uint64_t(column_) * 4u;                                 \
        const uint32_t dstSnap_ = walkLds + kWalkSnapStage + (parity) * 256u;                                                 \
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off"                                           \
                     :                                                                                                        \
                     : "v"(address_), "s"(dstSnap_)                                                                           \
                     : "memory", "m0");                                                                                       \
    } while (0)
    EM2_STAGE_WIDE_TILE(colBegin, 0u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bool tested = false;
    uint64_t passScratch[5];
    bool pending = false;
    uint32_t pendingBase = 0, pendingParity = 0;
    uint32_t iteration = 0, stopSlot = 0;
    uint32_t result = colEnd;
    int32_t stagedSnap = snapStage[laneId()];
    for (uint32_t colBase = colBegin; colBase < colEnd; colBase += 32u, ++iteration) {
        const uint32_t parity = iteration & 1u;
        boundScratch[parity * 32u + (laneId() & 31u)] = 2.f * kMatrixBits - 2.f * float(stagedSnap);
        if (colBase + 32u < colEnd) EM2_STAGE_WIDE_TILE(colBase + 32u, parity ^ 1u);
        const uint32_t tileBase = tilesLds + 2u * parity * (kMatrixTileWords * 16u);
        const uint32_t boundBase = walkLds + kWalkBounds + pendingParity * 128u;
        const uint32_t tileCode = pendingBase | rowHalf;
        if (!pending) {
            asm volatile(EM2_MATRIX_WIDE_STEP_X : : "s"(tileBase) : EM2_MATRIX_STEP_CLOBBERS);        // (the first tile: parity 0)
        } else if (parity == 0u) {
            asm volatile(EM2_MATRIX_WIDE_STEP_X_TESTING_Y
                         : "=v"(recordOffset), "=v"(unusedOffset), "=&s"(passScratch[0]), "=&s"(passScratch[1]), "=&s"(passScratch[2]), "=&s"(passScratch[3]), "=&s"(passScratch[4])
                         : "s"(tileBase), "s"(boundBase), "s"(stateBase), "s"(logBase), "s"(tileCode)
                         : EM2_MATRIX_STEP_CLOBBERS);
            tested = true;
        } else {
            asm volatile(EM2_MATRIX_WIDE_STEP_Y_TESTING_X
                         : "=v"(recordOffset), "=v"(unusedOffset), "=&s"(passScratch[0]), "=&s"(passScratch[1]), "=&s"(passScratch[2]), "=&s"(passScratch[3]), "=&s"(passScratch[4])
                         : "s"(tileBase), "s"(boundBase), "s"(stateBase), "s"(logBase), "s"(tileCode)
                         : EM2_MATRIX_STEP_CLOBBERS);
            tested = true;
        }
        pending = true;
        pendingBase = colBase;
        pendingParity = parity;
        // the untested tile and the next one add at most 32 records to a log before the next chance to stop
        const bool full = __builtin_amdgcn_ballot_w64(recordOffset > stopOffset) != 0ull;
        const uint32_t slot = stopSlot;
        stopSlot = stopSlot == 2u ? 0u : stopSlot + 1u;
        if (full && laneId() == 0u) stopWords[slot] = 1u;
        if (waveSlot == 0u && laneId() == 0u) stopWords[stopSlot] = 0u;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const uint32_t stop = stopWords[slot];
        stagedSnap = snapStage[(parity ^ 1u) * 64u + laneId()];
        if (stop != 0u) {
            __syncthreads();
            if (waveSlot == 0u && laneId() == 0u) stopWords[slot] = 0u;
            __syncthreads();
            result = colBase + 32u;
            break;
        }
    }
#undef EM2_STAGE_WIDE_TILE
#undef EM2_STAGE_UNIT
    if (pending) {
        const uint32_t boundBase = walkLds + kWalkBounds + pendingParity * 128u;
        const uint32_t tileCode = pendingBase | rowHalf;
        if (pendingParity == 1u) {
            asm volatile(EM2_MATRIX_WIDE_TEST_Y
                         : "=v"(recordOffset), "=v"(unusedOffset), "=&s"(passScratch[0]), "=&s"(passScratch[1]), "=&s"(passScratch[2]), "=&s"(passScratch[3]), "=&s"(passScratch[4])
                         : "s"(boundBase), "s"(stateBase), "s"(logBase), "s"(tileCode) : EM2_MATRIX_STEP_CLOBBERS);
        } else {
            asm volatile(EM2_MATRIX_WIDE_TEST_X
                         : "=v"(recordOffset), "=v"(unusedOffset), "=&s"(passScratch[0]), "=&s"(passScratch[1]), "=&s"(passScratch[2]), "=&s"(passScratch[3]), "=&s"(passScratch[4])
                         : "s"(boundBase), "s"(stateBase), "s"(logBase), "s"(tileCode) : EM2_MATRIX_STEP_CLOBBERS);
        }
        tested = true;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tested) recordCount[rowHalf] = (recordOffset - firstOffset) / uint32_t(sizeof(Entry));
    return result;
}

// The log of `lane` for accumulator a in the wave's log area.
__device__ __forceinline__ const Entry* walkLogOf(const Entry* waveLog, uint32_t logCapacity, uint32_t lane, uint32_t a)
{
    return waveLog + size_t(lane) * logCapacity + a * (logCapacity / 2u);
}

// A log read a few records ahead of its use (the records were written by another lane: every load goes past the L1, and
// a replay that waited for each one -- the next record is needed to decide which stream to take from -- spent most of its
// time in that latency).
struct WalkLogReader {
    const Entry* log;
    uint32_t count, fetched, taken;
    WalkRecord ahead[4];
    __device__ __forceinline__ void start(const Entry* l, uint32_t n)
    {
        log = l;
        count = n;
        fetched = taken = 0u;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            ahead[i].code = 0u;
            ahead[i].dot = 0.f;
            if (uint32_t(i) < count) ahead[i] = loadWalkRecord(log, uint32_t(i));
        }
        fetched = count < 4u ? count : 4u;
    }
    __device__ __forceinline__ bool have() const { return taken < count; }
    __device__ __forceinline__ WalkRecord front() const { return ahead[0]; }
    __device__ __forceinline__ void pop()
    {
        ahead[0] = ahead[1];
        ahead[1] = ahead[2];
        ahead[2] = ahead[3];
        if (fetched < count) ahead[3] = loadWalkRecord(log, fetched);
        fetched += fetched < count ? 1u : 0u;
        ++taken;
    }
};

// The replay of the walk's logs for the rows of the wave (lane = row, as everywhere outside the walk).  Row r = 32a + t
// finds its records in the accumulator-a logs of lanes t (columns 8q .. 8q+3 of every group) and 32 + t (columns 8q+4 ..
// 8q+7); both ascend in the column, and the row's candidates must be offered in ascending order: a two-way merge, every
// lane its own, all lanes in step.  Per record: the row side through the exact state machine (acceptColumn), the column
// side -- unless the rows scan all columns themselves (full rows) -- to the inbox if it passes the column's published
// cut-off, read now (fresher than the one the walk tested against: fewer entries).
// recordCount[a] = the calling lane's number of records in its log of accumulator a.
template <bool IDENTITY, bool WIDE = false>
__device__ __forceinline__ void replayWalkLogs(const Entry* waveLog, uint32_t logCapacity, const uint32_t (&recordCount)[2], uint32_t lane,
                                               uint32_t row, bool rowValid, bool emitColumns, uint32_t listBlock, Entry* myList,
                                               uint32_t twoK, uint32_t& count, int32_t& mMax, uint32_t& emitPos, uint32_t& emitEnd,
                                               unsigned char* ldsRaw)
{
    const uint32_t t = lane & 31u, a = lane >> 5;
    // the counts of the two source logs: accumulator a of lanes t and 32 + t
    const uint32_t mine0 = uint32_t(__shfl(int(recordCount[0]), int(t), 64)), mine1 = uint32_t(__shfl(int(recordCount[1]), int(t), 64));
    const uint32_t theirs0 = uint32_t(__shfl(int(recordCount[0]), int(t + 32u), 64)), theirs1 = uint32_t(__shfl(int(recordCount[1]), int(t + 32u), 64));
    WalkLogReader lower, upper;
    lower.start(walkLogOf(waveLog, logCapacity, t, a), a ? mine1 : mine0);
    upper.start(walkLogOf(waveLog, logCapacity, t + 32u, a), a ? theirs1 : theirs0);
    const int32_t* snap = kernelArgs()->snap;
    for (;;) {
        const bool have0 = lower.have(), have1 = upper.have();
        const bool active = have0 || have1;
        if (__builtin_amdgcn_ballot_w64(active) == 0ull) break;
        const WalkRecord r0 = lower.front(), r1 = upper.front();
        const uint32_t col0 = have0 ? walkRecordColumn(r0.code, 0u) : 0xffffffffu;
        const uint32_t col1 = have1 ? walkRecordColumn(r1.code, 1u) : 0xffffffffu;
        const bool take0 = col0 < col1;                   // (the halves never hold the same column)
        const uint32_t col = take0 ? col0 : col1;
        const float dot = take0 ? r0.dot : r1.dot;
        const uint32_t m = uint32_t(((WIDE ? 2.f : 1.f) * kMatrixBits - dot) * 0.5f);
        if (active) {
            if (take0) lower.pop();
            else upper.pop();
        }
        const bool passRow = active && int32_t(m) <= mMax;
        if (__builtin_amdgcn_ballot_w64(passRow) != 0ull) {
            acceptColumn<IDENTITY>(passRow, col, row, m, lane, listBlock, myList, twoK, count, mMax, ldsRaw);
        }
        if (emitColumns) {
            const bool passColumn = active && rowValid && int32_t(m) <= __hip_atomic_load(snap + (active ? col : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            emitColumn(passColumn, col, row, m, lane, emitPos, emitEnd);
        }
    }
}

// The tile kernel of the sharded scan defers both sides: every lane empties its own two logs, order is irrelevant (the
// inbox is sorted).  rowBase = cell id of the wave's row 0.
template <bool WIDE = false>
__device__ __forceinline__ void drainWalkLogs(const Entry* waveLog, uint32_t logCapacity, const uint32_t (&recordCount)[2], uint32_t lane,
                                              uint32_t rowBase, uint32_t cellCount, uint32_t& emitPos, uint32_t& emitEnd)
{
    const int32_t* snap = kernelArgs()->snap;
#pragma unroll
    for (uint32_t a = 0; a < 2u; a++) {
        const Entry* log = walkLogOf(waveLog, logCapacity, lane, a);
        const uint32_t rowId = rowBase + 32u * a + (lane & 31u);
        const int32_t snapOfRow = rowId < cellCount ? snap[rowId] : -1;
        for (uint32_t i = 0;; ++i) {
            const bool active = i < recordCount[a];
            if (__builtin_amdgcn_ballot_w64(active) == 0ull) break;
            WalkRecord r;
            r.code = 0u;
            r.dot = 0.f;
            if (active) r = loadWalkRecord(log, i);
            const uint32_t col = walkRecordColumn(r.code, lane >> 5);
            const uint32_t m = uint32_t(((WIDE ? 2.f : 1.f) * kMatrixBits - r.dot) * 0.5f);
            const bool valid = active && rowId < cellCount;
            const int32_t snapCol = valid ? snap[col] : -1;
            emitColumn(valid && int32_t(m) <= snapCol, col, rowId, m, lane, emitPos, emitEnd);        // target col
            emitColumn(valid && int32_t(m) <= snapOfRow, rowId, col, m, lane, emitPos, emitEnd);      // target row
        }
    }
}

// TIMED (EM2_MATRIX_DIAG bit 2048, measurements only): every wave sums the shader-clock cycles it spends in the phases of
// its items -- ticket, set-up, walk, hand-off wait, replay, the quad's own columns, publication -- and adds them to eight
// 64-bit counters behind the inbox control words (printed by the launcher with EM2_SCAN_VERBOSE=1).
#define EM2_PHASE(index)                                                                                                       \
    do {                                                                                                                      \
        if (TIMED) {                                                                                                          \
            const uint64_t now_ = __builtin_readcyclecounter();                                                               \
            phaseCycles[index] += now_ - phaseStart;                                                                          \
            phaseStart = now_;                                                                                                \
        }                                                                                                                     \
    } while (0)
template <bool IDENTITY, bool PINNED, bool WIDE = false, bool TIMED = false>
__device__ __forceinline__ void scanMatrixBody(unsigned char* ldsRaw)
{
    uint64_t phaseCycles[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t phaseStart = TIMED ? __builtin_readcyclecounter() : 0ull;
    static_assert(!WIDE || PINNED, "the 2048-bit form has the hand-scheduled walk only");
    constexpr int W32 = WIDE ? 64 : 32;                          // dwords per signature as the v_xor/v_bcnt parts read them
    constexpr float bits = WIDE ? 2.f * kMatrixBits : kMatrixBits;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    FragmentWord4* tiles = reinterpret_cast<FragmentWord4*>(ldsRaw + kernelArgs()->matrixLdsOffset);
    volatile uint32_t* shared = reinterpret_cast<volatile uint32_t*>(ldsRaw + kernelArgs()->matrixLdsOffset + 4u * kMatrixTileWords * 16u);
    // The wave's LDS block of the walk is its selection area (never needed at the same time) when that is large enough
    // and 16-byte aligned, and sits behind the tiles otherwise (matrixWalkAliasesSelection on the host side).
    const uint32_t selectionStride = 2u * kernelArgs()->k * kLdsBytesPerEntrySlot;
    unsigned char* walkBlock = (selectionStride >= kMatrixWalkLdsBytes && selectionStride % 16u == 0u)
                                   ? ldsRaw + wave * selectionStride
                                   : ldsRaw + kernelArgs()->matrixLdsOffset + 4u * kMatrixTileWords * 16u + 64u + wave * kMatrixWalkLdsBytes;
    // shared[0..2] stop words of the walk, shared[3] the block's ticket
    if (threadIdx.x < 4u) shared[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t emitPos = 0, emitEnd = 0;
    // The clock this launch ran at: every block's first lane stamps the shader clock counter and the 100 MHz wall counter when
    // it starts and when it leaves; the launcher divides the sums (MI355X_MICROARCH.md, 'DVFS give-back' item 6).  Two scalar
    // reads per block and launch; the values go to two words of the control block that nothing else reads.
    const uint64_t clockStart = __builtin_amdgcn_s_memtime(), wallStart = __builtin_amdgcn_s_memrealtime();

    for (;;) {
        if (threadIdx.x == 0u) {
            // a hand-off that timed out anywhere ends the launch for everybody
            const bool broken = __hip_atomic_load(kernelArgs()->control + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            shared[3] = broken ? 0xffffffffu
                               : __hip_atomic_fetch_add(kernelArgs()->control, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const uint32_t ticket = uint32_t(__builtin_amdgcn_readfirstlane(int(shared[3])));
        __syncthreads();
        ArgsPtr aux = kernelArgs();
        EM2_PHASE(0);
        if (ticket >= aux->totalTickets) break;

        // ---- the item: segment seg, quad = 4 row blocks from quadBlock ----
        const uint32_t segments = aux->segments;
        const uint32_t* table = aux->segTable;
        // (binary search: the table has up to 256 segments, and a linear walk through it -- one dependent scalar load per
        // segment -- cost an item of a late segment some 25 us, as much as two dozen tiles)
        uint32_t seg = 0;
        {
            uint32_t lo = 0, hi = segments;               // table[lo] <= ticket < table[hi]
            while (hi - lo > 1u) {
                const uint32_t mid = (lo + hi) / 2u;
                if (table[mid] <= ticket) lo = mid;
                else hi = mid;
            }
            seg = lo;
        }
        // block = list / state slot of the launch; its 64 cells start at (block * stride + offset) * 64 (block-cyclic in
        // the sharded scan, where a quad is 4 slots whose cells are not adjacent -- but there every column lies below them)
        // The first fullQuads items of a segment are quads of full-row blocks (the first cells of the problem, or the prefix
        // blocks of the sharded scan): their rows take every column of the segment, emit nothing and finish themselves.
        const uint32_t local = ticket - table[seg];
        const uint32_t fullQuads = (aux->fullRowBlocks + 3u) / 4u;
        const bool fullRows = local < fullQuads;
        const uint32_t quadBlock = fullRows ? aux->localBlockBase + 4u * local : table[segments + 1u + seg] + 4u * (local - fullQuads);
        const uint32_t block = quadBlock + wave;
        const bool idle = block >= (fullRows ? aux->localBlockBase + aux->fullRowBlocks : aux->rowBlocks);      // a short last quad
        const uint32_t listBlock = idle ? aux->rowBlocks - 1u : block;
        const uint32_t quadRowBase = (quadBlock * aux->rowBlockStride + aux->rowBlockOffset) * 64u;
        const uint32_t rowBase = (block * aux->rowBlockStride + aux->rowBlockOffset) * 64u;
        const uint32_t rowFragmentBlock = 2u * (listBlock * aux->rowBlockStride + aux->rowBlockOffset);
        const uint32_t row = rowBase + lane;
        const bool rowValid = !idle && row < aux->cellCount;
        const uint32_t twoK = 2u * aux->k;
        const uint32_t minLogCapacity = PINNED ? kMatrixLogMargin : 64u;
        uint32_t logCapacity = aux->logCapacity < minLogCapacity ? minLogCapacity : aux->logCapacity;
        Entry* myList = aux->buffers + (size_t(listBlock) * 64u + lane) * twoK;
        Entry* myLog = aux->logs + (size_t(blockIdx.x * 4u + wave) * 64u + lane) * logCapacity;
        const uint32_t cps = aux->columnsPerSegment;
        const uint32_t colBegin = seg * cps;
        uint32_t colEnd = colBegin + cps;
        if (colEnd > aux->columnLimit || seg + 1u == segments) colEnd = aux->columnLimit;
        const bool last = !fullRows && quadRowBase < colEnd;    // the segment that holds the quad's own cells
        // (a full row's last segment may end at a cell count that is no multiple of 32: the walk takes whole tiles)
        const uint32_t commonEnd = last ? quadRowBase : (fullRows ? colEnd & ~31u : colEnd);
        int32_t mMax = rowValid ? aux->mMaxInitial : -1;
        uint32_t count = 0, logCount = 0;
        uint32_t recordCount[2] = {0u, 0u};          // (hand-scheduled walk: the calling lane's records per accumulator)
        bool haveState = seg == 0u;
        if (seg != 0u && !idle) {
            const uint32_t done = uint32_t(__builtin_amdgcn_readfirstlane(
                int(__hip_atomic_load(aux->segmentsDone + block, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))));
            if (done != 0u) {
                if (done >= seg && !(EM2_DIAG_WORD(aux) & 512u)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                const uint64_t st = __hip_atomic_load(reinterpret_cast<const uint64_t*>(aux->rowState) + size_t(block) * 64u + lane,
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                mMax = rowValid ? int32_t(uint32_t(st >> 32)) : -1;
                if (done >= seg) {
                    count = uint32_t(st);
                    haveState = true;
                }
            }
        }

        // ---- departures ----
        // The blocks take their items whenever they finish the previous one, so the 64 blocks of an XCD walk the same
        // segment's tiles at 64 different offsets, which its 4 MB L2 cannot hold together (47 % hits, 570 GB of fabric reads
        // per launch at 1M cells).  With departTicks set, a walk starts only within departWindow ticks of a multiple of
        // departTicks on the 100 MHz wall counter all blocks share: blocks that finish at about the same time leave together
        // and follow each other through the tiles closely enough for the L2 to serve all but the first.  A block waits
        // (asleep) for at most one period; order and content of everything it does are unchanged.
        if (PINNED && aux->departTicks != 0u && colBegin < commonEnd) {
            if (threadIdx.x == 0u) {
                const uint32_t period = aux->departTicks, window = aux->departWindow;
                for (uint32_t spins = 0; spins < (1u << 16); ++spins) {
                    if (uint32_t(__builtin_amdgcn_s_memrealtime() % period) < window) break;
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            __syncthreads();
        }
        bool failed = false;
        uint32_t at = colBegin;
        uint32_t rowHalf = 0;           // (2048 bits: the columns are walked once per half of the wave's rows)
        EM2_PHASE(1);
        for (;;) {
            if (at < commonEnd) {
                if (WIDE) {
                    at = scanTilesMatrixWide<IDENTITY>(aux->fragments, aux->snap, at, commonEnd, rowFragmentBlock + rowHalf,
                                                       bits - 2.f * float(mMax), rowHalf, myLog - size_t(lane) * logCapacity, logCapacity,
                                                       recordCount, ldsAddress(tiles), ldsAddress(const_cast<uint32_t*>(shared)),
                                                       ldsAddress(walkBlock));
                } else if (PINNED) {
                    if (EM2_DIAG_WORD(aux)) {
                        at = scanTilesMatrixPinned<IDENTITY, false, true>((const void*)(uintptr_t)aux, aux->fragments, aux->snap, at, commonEnd,
                                                         rowFragmentBlock, kMatrixBits - 2.f * float(mMax), myLog - size_t(lane) * logCapacity,
                                                         logCapacity, recordCount, ldsAddress(tiles), ldsAddress(const_cast<uint32_t*>(shared)),
                                                         ldsAddress(walkBlock));
                    } else {
                        at = scanTilesMatrixPinned<IDENTITY, false, false>((const void*)(uintptr_t)aux, aux->fragments, aux->snap, at, commonEnd,
                                                         rowFragmentBlock, kMatrixBits - 2.f * float(mMax), myLog - size_t(lane) * logCapacity,
                                                         logCapacity, recordCount, ldsAddress(tiles), ldsAddress(const_cast<uint32_t*>(shared)),
                                                         ldsAddress(walkBlock));
                    }
                } else {
                    at = scanTilesMatrix<IDENTITY>(static_cast<const FragmentWord4*>(aux->fragments), aux->snap, at, commonEnd,
                                                   rowFragmentBlock, kMatrixBits - 2.f * float(mMax), row, rowValid && !fullRows, lane, myLog,
                                                   logCapacity, logCount, emitPos, emitEnd, tiles, shared);
                }
            }
            EM2_PHASE(2);
            if (!haveState && !idle && !failed) {
                const uint32_t* flag = aux->segmentsDone + block;
                const uint64_t start = __builtin_amdgcn_s_memrealtime();         // 100 MHz
                while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seg) {
                    __builtin_amdgcn_s_sleep(16);
                    if (__builtin_amdgcn_s_memrealtime() - start > 400000000ull ||
                        __hip_atomic_load(aux->control + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                        failed = true;
                        break;
                    }
                }
                if (failed) {
                    if (lane == 0u) __hip_atomic_store(aux->control + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    if (!(EM2_DIAG_WORD(aux) & 512u)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    const uint64_t st = __hip_atomic_load(reinterpret_cast<const uint64_t*>(aux->rowState) + size_t(block) * 64u + lane,
                                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    count = uint32_t(st);
                    mMax = rowValid ? int32_t(uint32_t(st >> 32)) : -1;
                    haveState = true;
                }
            }
            EM2_PHASE(3);
            // replay the log through the exact state machine (ascending column order per row)
            if (PINNED) {
                if (!idle && !failed) {
                    replayWalkLogs<IDENTITY, WIDE>(myLog - size_t(lane) * logCapacity, logCapacity, recordCount, lane, row, rowValid, !fullRows,
                                                   listBlock, myList, twoK, count, mMax, emitPos, emitEnd, ldsRaw);
                }
                recordCount[0] = recordCount[1] = 0u;
            } else if (!idle && !failed) {
                for (uint32_t i = 0;; ++i) {
                    const bool active = i < logCount;
                    if (__builtin_amdgcn_ballot_w64(active) == 0ull) break;
                    uint32_t c = 0, m = 0;
                    if (active) {
                        const Entry e = myLog[i];
                        c = e.cell;
                        m = e.key;
                    }
                    const bool pass = active && int32_t(m) <= mMax;
                    if (__builtin_amdgcn_ballot_w64(pass) != 0ull) {
                        acceptColumn<IDENTITY>(pass, c, row, m, lane, listBlock, myList, twoK, count, mMax, ldsRaw);
                    }
                }
            }
            logCount = 0;
            EM2_PHASE(4);
            if (at >= commonEnd) {
                if (!WIDE || rowHalf == 1u || colBegin >= commonEnd) break;
                rowHalf = 1u;
                at = colBegin;
            }
        }
        // (a wave whose hand-off failed keeps walking with its block -- the barriers need it -- and the launch ends at
        // the next ticket)

        // ---- full rows: the columns of a last, partial tile ----
        if (fullRows && commonEnd < colEnd && !idle && !failed) {
            uint32_t r[W32];
            const uint32_t* rp = aux->sig32 + size_t(rowValid ? row : rowBase) * uint32_t(W32);
#pragma unroll
            for (int w = 0; w < W32; ++w) r[w] = rp[w];
            uint32_t unusedLogCount = 0;
            scanColumns<W32, IDENTITY, false>(kernelArgs()->sig32, commonEnd, colEnd, r, row, lane, listBlock, myList, twoK, count, mMax,
                                             myLog, logCapacity, unusedLogCount, ldsRaw);
        }

        // ---- the quad's own 256 columns: the band below this wave's rows and its diagonal, as in the other kernel ----
        if (last && !idle && !failed) {
            uint32_t r[W32];
            const uint32_t* rp = aux->sig32 + size_t(rowValid ? row : rowBase) * uint32_t(W32);
#pragma unroll
            for (int w = 0; w < W32; ++w) r[w] = rp[w];
            uint32_t diagEnd = rowBase + 64u;
            if (diagEnd > aux->columnLimit) diagEnd = aux->columnLimit;
            uint32_t from = quadRowBase;
            while (from < rowBase) {
                ensureInboxRoom(lane, emitPos, emitEnd);
                uint32_t unusedLogCount = 0;
                from = scanColumnsEmit<W32, IDENTITY, false>(kernelArgs()->sig32, kernelArgs()->snap, from, rowBase, r, row,
                                                            rowValid, lane, myList, twoK, count, mMax, myLog, logCapacity,
                                                            unusedLogCount, emitPos, emitEnd);
                acceptColumn<IDENTITY>(false, 0u, row, 0u, lane, listBlock, myList, twoK, count, mMax, ldsRaw);
            }
            uint32_t unusedLogCount = 0;
            scanDiagonal<W32, IDENTITY, false>(kernelArgs()->sig32, kernelArgs()->snap, rowBase, diagEnd, r, row, rowValid, lane,
                                               listBlock, myList, twoK, count, mMax, myLog, logCapacity, unusedLogCount, emitPos,
                                               emitEnd, ldsRaw);
        }

        EM2_PHASE(5);
        // ---- full rows at their last segment: finish; otherwise publish the state: for the next segment, for the
        // columns' snapshots, for the inbox replay ----
        if (!idle && !failed) {
            ArgsPtr aux2 = kernelArgs();
            const bool finalSegment = seg + 1u == aux2->segments;
            const uint32_t shardFlags = aux2->shardFlags;
            if (fullRows && finalSegment && !(shardFlags & kShardNoFinish)) {
                finishRows(lane, block, count, ldsRaw);
            } else {
                const uint64_t st = uint64_t(count) | (uint64_t(uint32_t(mMax)) << 32);
                __hip_atomic_store(reinterpret_cast<uint64_t*>(aux2->rowState) + size_t(block) * 64u + lane, st,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // a full row's snapshot stays "never emit to this column" unless the sharded scan asks for it
                if (rowValid && (!fullRows || (shardFlags & kShardPublishAll))) {
                    __hip_atomic_store(aux2->snap + row, mMax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // (EM2_MATRIX_DIAG bits 256 / 512, measurements only: no release / no acquire -- results may be wrong)
                if (!(EM2_DIAG_WORD(aux2) & 256u)) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0u && !(fullRows ? finalSegment : last)) {
                    __hip_atomic_store(aux2->segmentsDone + block, seg + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        EM2_PHASE(6);
    }

    // the unused tail of this wave's last inbox chunk becomes sentinels
    {
        const uint32_t p = uint32_t(__builtin_amdgcn_readfirstlane(int(emitPos)));
        const uint32_t e = uint32_t(__builtin_amdgcn_readfirstlane(int(emitEnd)));
        if (p <= e) {
            uint64_t* inbox = kernelArgs()->inbox;
            for (uint32_t i = p + lane; i < e; i += 64u) inbox[i] = ~0ull;
        }
    }
    if (TIMED && lane == 0u) {
        unsigned long long* counters = reinterpret_cast<unsigned long long*>(kernelArgs()->inboxControl + 16);
        for (int i = 0; i < 8; i++) atomicAdd(counters + i, (unsigned long long)phaseCycles[i]);
    }
    if (threadIdx.x == 0u) {
        unsigned long long* clockWords = reinterpret_cast<unsigned long long*>(kernelArgs()->inboxControl + kClockWordsOffset);
        atomicAdd(clockWords, (unsigned long long)(__builtin_amdgcn_s_memtime() - clockStart));
        atomicAdd(clockWords + 1, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - wallStart));
    }
}
#undef EM2_PHASE

// The two entry points.  The kernel of the hand-scheduled walk lets the compiler allocate 64 vector registers only
// (amdgpu_num_vgpr): v64..v255 hold the rows and the accumulators of the steps, which the compiler does not know
// of -- this is what keeps its own code, the events of a tile included, out of them (the kernel descriptor still
// asks for 256: the clobber lists of the steps count).
template <bool IDENTITY>
__global__ void __launch_bounds__(256, 2)
fsp4ScanMatrixKernel(Fsp4Args args)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
    scanMatrixBody<IDENTITY, false>(ldsRaw);
}

template <bool IDENTITY>
__global__ void __launch_bounds__(256, 2)
fsp4ScanMatrixPinnedKernel(Fsp4Args args)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
    scanMatrixBody<IDENTITY, true>(ldsRaw);
}

// 2048-bit signatures (scanTilesMatrixWide)
template <bool IDENTITY>
__global__ void __launch_bounds__(256, 2)
fsp4ScanMatrixWideKernel(Fsp4Args args)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
    scanMatrixBody<IDENTITY, true, true>(ldsRaw);
}

// (EM2_MATRIX_DIAG bit 2048: the same kernels with the phase timers, identity keys only)
template <bool WIDE>
__global__ void __launch_bounds__(256, 2)
fsp4ScanMatrixTimedKernel(Fsp4Args args)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
    scanMatrixBody<true, true, WIDE, true>(ldsRaw);
}

// Second phase of the symmetric scan: one wave per triangle row block replays the sorted inbox entries of its 64
// cells (ascending candidate id per cell) through the exact state machine and finishes the rows.
template <bool IDENTITY>
__global__ void __launch_bounds__(256)
fsp4InboxReplayKernel(Fsp4Args args, const uint64_t* __restrict__ sorted, uint64_t sortedCount)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
    const uint32_t lane = threadIdx.x & 63u;
    // slots [replayBegin, replayEnd) = [localBlockBase + fullRowBlocks, rowBlocks) (full-row blocks of the
    // one-GPU form are finished by the scan kernel itself)
    const uint32_t block = args.localBlockBase + args.fullRowBlocks + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (block >= args.rowBlocks) return;
    const uint32_t row = (block * args.rowBlockStride + args.rowBlockOffset) * 64u + lane;
    const bool rowValid = row < args.rowEnd;
    uint32_t twoK = 2u * args.k;
    Entry* myList = args.buffers + (size_t(block) * 64u + lane) * twoK;
    const uint64_t st = reinterpret_cast<const uint64_t*>(args.rowState)[size_t(block) * 64u + lane];
    uint32_t count = uint32_t(st);
    int32_t mMax = rowValid ? int32_t(uint32_t(st >> 32)) : -1;
    const uint32_t nb = args.rowBits;
    const uint64_t fieldMask = (1ull << (2u * nb)) - 1ull;
    const bool timed = (EM2_DIAG_WORD_OF(args) & 4096u) != 0u;
    uint64_t clock0 = timed ? __builtin_readcyclecounter() : 0ull, clock1 = 0, clock2 = 0;
    uint64_t bound[2];
    {
        // the two lower bounds in one loop: two independent chains of dependent loads instead of one after the other
        const uint64_t target0 = uint64_t(row) << nb, target1 = uint64_t(row + 1u) << nb;
        uint64_t lo0 = 0, hi0 = sortedCount, lo1 = 0, hi1 = sortedCount;
        while (__builtin_amdgcn_ballot_w64(lo0 < hi0 || lo1 < hi1) != 0ull) {
            const uint64_t mid0 = lo0 + (hi0 - lo0) / 2u, mid1 = lo1 + (hi1 - lo1) / 2u;
            const uint64_t v0 = lo0 < hi0 ? sorted[mid0] : 0ull, v1 = lo1 < hi1 ? sorted[mid1] : 0ull;
            if (lo0 < hi0) {
                if (((v0 >> 13u) & fieldMask) < target0) lo0 = mid0 + 1u;
                else hi0 = mid0;
            }
            if (lo1 < hi1) {
                if (((v1 >> 13u) & fieldMask) < target1) lo1 = mid1 + 1u;
                else hi1 = mid1;
            }
        }
        bound[0] = lo0;
        bound[1] = lo1;
    }
    if (!rowValid) bound[1] = bound[0];
    if (EM2_DIAG_WORD_OF(args) & 4096u) {        // (EM2_MATRIX_DIAG bit 4096: the longest inbox of a cell and of a wave's 64 cells, for EM2_SCAN_VERBOSE)
        uint32_t longest = uint32_t(bound[1] - bound[0]);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) longest = max(longest, uint32_t(__shfl_xor(int(longest), d, 64)));
        if (lane == 0u) atomicMax(args.inboxControl + 8, longest);
        clock1 = __builtin_readcyclecounter();
    }
    const uint32_t idMask = (1u << nb) - 1u;
    // Entries are fetched four at a time (independent loads in flight: the loop is latency-bound otherwise) and then
    // offered one by one, in order.
    constexpr int kAhead = 4;
    for (uint64_t i = bound[0];; i += kAhead) {
        if (__builtin_amdgcn_ballot_w64(i < bound[1]) == 0ull) break;
        uint64_t e[kAhead];
#pragma unroll
        for (int q = 0; q < kAhead; ++q) e[q] = (i + q < bound[1]) ? sorted[i + q] : 0ull;
#pragma unroll
        for (int q = 0; q < kAhead; ++q) {
            const bool active = i + q < bound[1];
            const uint32_t c = uint32_t(e[q] >> 13u) & idMask;
            const uint32_t m = uint32_t(e[q]) & 0x1fffu;
            const bool pass = active && int32_t(m) <= mMax;
            if (__builtin_amdgcn_ballot_w64(pass) != 0ull) {
                acceptColumn<IDENTITY>(pass, c, row, m, lane, block, myList, twoK, count, mMax, ldsRaw);
            }
        }
    }
    if (timed) clock2 = __builtin_readcyclecounter();
    finishRows(lane, block, count, ldsRaw);
    if (timed && lane == 0u) {
        unsigned long long* cycles = reinterpret_cast<unsigned long long*>(args.inboxControl + 10);
        atomicAdd(cycles, (unsigned long long)(clock1 - clock0));
        atomicAdd(cycles + 1, (unsigned long long)(clock2 - clock1));
        atomicAdd(cycles + 2, (unsigned long long)(__builtin_readcyclecounter() - clock2));
    }
}

// =========================================================================================================
// Sharded symmetric scan, third phase: the square of the non-prefix cells, [M,N) x [M,N), lower triangle.
//
// By now every cell holds a true snapshot of its cut-off (its state after the M prefix candidates, exchanged
// between the ranks), so BOTH sides of a pair can be deferred: a tile is 64 rows x one column segment, belongs to
// no cell in particular, keeps no per-row state and depends on nothing -- tiles are dealt round-robin to the ranks
// (tile L goes to rank L % world) and to the waves of a rank through a ticket counter.  A pair (r, c), c < r, with
// mismatch m emits (target c, candidate r) if m <= snap[c] and (target r, candidate c) if m <= snap[r].
// Kernel-argument reuse: columnLimit = M, rowBlocks = number of 64-cell blocks of the whole problem,
// rowBlockStride / rowBlockOffset = world / rank, segTable = first tile and first block of every column segment,
// segments / columnsPerSegment = the segmentation of [M,N), totalTickets = tiles of this rank.
// =========================================================================================================
template <int W32>
__device__ __forceinline__ uint32_t scanTileEmit(const uint32_t* __restrict__ sig32, const int32_t* snap, uint32_t colBegin,
                                                 uint32_t colEnd, const uint32_t (&r)[W32], uint32_t row, bool rowValid,
                                                 int32_t snapRow, uint32_t lane, uint32_t& emitPos, uint32_t emitEnd)
{
    constexpr int CH = W32 < 32 ? W32 : 32;
    constexpr int H = W32 / CH;
    constexpr int U = 2 * H;
    if (colBegin >= colEnd) return colEnd;
    ScalarPtr p = (ScalarPtr)(uintptr_t)sig32 + size_t(colBegin) * W32;
    ScalarIntPtr sp = (ScalarIntPtr)(uintptr_t)snap + colBegin;
    uint32_t chunk[2][CH];
    int32_t snapCol[2];
#pragma unroll
    for (int w = 0; w < CH; ++w) chunk[0][w] = p[w];
    snapCol[0] = sp[0];
    snapCol[1] = 0;
    __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0)
    uint32_t m = 0;
    for (uint32_t colBase = colBegin; colBase < colEnd; colBase += 2u) {
#pragma unroll
        for (int s = 0; s < U; ++s) {
            const int part = s % H;
            const int ci = s / H;
            const uint32_t col = colBase + uint32_t(ci);
            if (col < colEnd) {
                __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0)
                __builtin_amdgcn_sched_barrier(0);
                const bool lastChunk = (col + 1u == colEnd) && (part == H - 1);
                ScalarPtr pn = lastChunk ? p : p + CH;
#pragma unroll
                for (int w = 0; w < CH; ++w) chunk[(s + 1) & 1][w] = pn[w];
                p = pn;
                if (part == H - 1) {
                    ScalarIntPtr spn = lastChunk ? sp : sp + 1;
                    snapCol[ci ^ 1] = spn[0];
                    sp = spn;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int w = 0; w < CH; ++w) {
                    if (part == 0 && w == 0) popcountFirst(m, r[0] ^ chunk[s & 1][0]);
                    else popcountAccumulate(m, r[part * CH + w] ^ chunk[s & 1][w]);
                }
                if (part == H - 1) {
                    int32_t limit = snapRow > snapCol[ci] ? snapRow : snapCol[ci];
                    asm volatile("" : "+v"(limit));
                    if (__builtin_amdgcn_ballot_w64(int32_t(m) <= limit) != 0ull) {
                        const bool toCol = rowValid && int32_t(m) <= snapCol[ci];
                        const bool toRow = rowValid && int32_t(m) <= snapRow;
                        const uint64_t maskCol = __builtin_amdgcn_ballot_w64(toCol);
                        const uint64_t maskRow = __builtin_amdgcn_ballot_w64(toRow);
                        const uint32_t at = uint32_t(__builtin_amdgcn_readfirstlane(int(emitPos)));
                        if ((maskCol | maskRow) != 0ull && at <= uint32_t(__builtin_amdgcn_readfirstlane(int(emitEnd)))) {
                            ArgsPtr aux = kernelArgs();
                            const uint32_t nb = aux->rowBits;
                            const uint32_t nCol = uint32_t(__builtin_popcountll(maskCol));
                            if (toCol) {
                                aux->inbox[at + lanesBelow(maskCol)] =
                                    (uint64_t(col) << (13u + nb)) | (uint64_t(row) << 13u) | uint64_t(m);
                            }
                            if (toRow) {
                                aux->inbox[at + nCol + lanesBelow(maskRow)] =
                                    (uint64_t(row) << (13u + nb)) | (uint64_t(col) << 13u) | uint64_t(m);
                            }
                            emitPos = at + nCol + uint32_t(__builtin_popcountll(maskRow));
                            if (inboxRoom(emitPos, emitEnd) < 128u) return col + 1u;
                        }
                    }
                    m = 0;
                }
            }
        }
    }
    return colEnd;
}

// Makes sure the chunk has room for one more column's worth of tile entries (2 per lane).
__device__ __forceinline__ void ensureInboxRoomForTile(uint32_t lane, uint32_t& emitPos, uint32_t& emitEnd)
{
    if (inboxRoom(emitPos, emitEnd) >= 128u) return;
    ArgsPtr aux = kernelArgs();
    const uint32_t p = uint32_t(__builtin_amdgcn_readfirstlane(int(emitPos)));
    const uint32_t e = uint32_t(__builtin_amdgcn_readfirstlane(int(emitEnd)));
    const uint64_t fresh = refillInboxChunk(aux->inbox, aux->inboxControl, aux->inboxCapacity, aux->inboxChunk, lane, p, e);
    emitPos = uint32_t(fresh);
    emitEnd = uint32_t(fresh >> 32);
}

template <int W32>
__global__ void __launch_bounds__(256)
fsp4TileKernel(Fsp4Args args)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t emitPos = 0, emitEnd = 0;
    for (;;) {
        uint32_t ticket = 0;
        if (lane == 0u) {
            ticket = __hip_atomic_fetch_add(kernelArgs()->control, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ticket = uint32_t(__builtin_amdgcn_readfirstlane(int(ticket)));
        uint32_t colBeginV, colEndV, rowBaseV;
        uint32_t row;
        uint32_t r[W32];
        int32_t snapRow;
        bool rowValid;
        {
            ArgsPtr aux = kernelArgs();
            if (ticket >= aux->totalTickets) break;
            const uint32_t cellCount = aux->cellCount;
            const uint32_t segments = aux->segments;
            const uint32_t* table = aux->segTable;
            const uint32_t tile = ticket * aux->rowBlockStride + aux->rowBlockOffset;     // round-robin over the ranks
            uint32_t lo = 0, hi = segments;              // last segment whose first tile is <= tile
            while (hi - lo > 1u) {
                const uint32_t mid = (lo + hi) / 2u;
                if (table[mid] <= tile) lo = mid;
                else hi = mid;
            }
            const uint32_t seg = lo;
            const uint32_t block = table[segments + 1u + seg] + (tile - table[seg]);
            const uint32_t rowBase = block * 64u;
            const uint32_t colBegin = aux->columnLimit + seg * aux->columnsPerSegment;
            uint32_t colEnd = colBegin + aux->columnsPerSegment;
            uint32_t diagEnd = rowBase + 64u;
            if (diagEnd > cellCount) diagEnd = cellCount;
            if (colEnd > diagEnd) colEnd = diagEnd;
            row = rowBase + lane;
            rowValid = row < cellCount;
            const uint32_t* rp = aux->sig32 + size_t(rowValid ? row : rowBase) * W32;
#pragma unroll
            for (int w = 0; w < W32; ++w) r[w] = rp[w];
            snapRow = rowValid ? aux->snap[row] : -1;
            colBeginV = parkInVgpr(colBegin);
            colEndV = parkInVgpr(colEnd);
            rowBaseV = parkInVgpr(rowBase);
        }
        // columns strictly below the block
        uint32_t at = unpark(colBeginV);
        for (;;) {
            const uint32_t colEnd = unpark(colEndV);
            const uint32_t rowBase = unpark(rowBaseV);
            const uint32_t triEnd = colEnd < rowBase ? colEnd : rowBase;
            if (at >= triEnd) break;
            ensureInboxRoomForTile(lane, emitPos, emitEnd);
            at = scanTileEmit<W32>(kernelArgs()->sig32, kernelArgs()->snap, at, triEnd, r, row, rowValid, snapRow, lane,
                                   emitPos, emitEnd);
        }
        // the block's own cells: pair (row, col) belongs to the lane with row > col
        {
            const uint32_t colEnd = unpark(colEndV);
            const uint32_t rowBase = unpark(rowBaseV);
            const uint32_t colBegin = unpark(colBeginV);
            const uint32_t* sig32 = kernelArgs()->sig32;
            const int32_t* snap = kernelArgs()->snap;
            for (uint32_t col = colBegin > rowBase ? colBegin : rowBase; col < colEnd; ++col) {
                ScalarPtr cp = (ScalarPtr)(uintptr_t)sig32 + size_t(col) * W32;
                uint32_t m = 0;
#pragma unroll
                for (int w = 0; w < W32; ++w) popcountAccumulate(m, r[w] ^ cp[w]);
                const int32_t snapCol = snap[col];
                const bool lower = rowValid && col < row;
                emitColumn(lower && int32_t(m) <= snapCol, col, row, m, lane, emitPos, emitEnd);      // target col
                emitColumn(lower && int32_t(m) <= snapRow, row, col, m, lane, emitPos, emitEnd);      // target row
            }
        }
    }
    {
        const uint32_t p = uint32_t(__builtin_amdgcn_readfirstlane(int(emitPos)));
        const uint32_t e = uint32_t(__builtin_amdgcn_readfirstlane(int(emitEnd)));
        if (p <= e) {
            uint64_t* inbox = kernelArgs()->inbox;
            for (uint32_t i = p + lane; i < e; i += 64u) inbox[i] = ~0ull;
        }
    }
}

// fsp4TileKernel on the matrix cores (1024-bit signatures): tiles are (segment, quad of 4 row blocks), a block of 4
// waves walks the columns of the segment below the quad in lock step (scanTilesMatrix, both sides deferred); the
// quad's own 256 columns are done by the v_xor/v_bcnt code.  Prefix and segment lengths are multiples of 256 cells.
template <bool PINNED, bool WIDE = false>
__device__ __forceinline__ void tileMatrixBody(unsigned char* ldsRaw)
{
    static_assert(!WIDE || PINNED, "the 2048-bit form has the hand-scheduled walk only");
    constexpr int W32 = WIDE ? 64 : 32;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    FragmentWord4* tiles = reinterpret_cast<FragmentWord4*>(ldsRaw + kernelArgs()->matrixLdsOffset);
    volatile uint32_t* shared = reinterpret_cast<volatile uint32_t*>(ldsRaw + kernelArgs()->matrixLdsOffset + 4u * kMatrixTileWords * 16u);
    unsigned char* walkBlock = ldsRaw + kernelArgs()->matrixLdsOffset + 4u * kMatrixTileWords * 16u + 64u + wave * kMatrixWalkLdsBytes;
    if (threadIdx.x < 4u) shared[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t emitPos = 0, emitEnd = 0;
    for (;;) {
        if (threadIdx.x == 0u) {
            shared[3] = __hip_atomic_fetch_add(kernelArgs()->control, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const uint32_t ticket = uint32_t(__builtin_amdgcn_readfirstlane(int(shared[3])));
        __syncthreads();
        ArgsPtr aux = kernelArgs();
        if (ticket >= aux->totalTickets) break;
        const uint32_t cellCount = aux->cellCount;
        const uint32_t segments = aux->segments;
        const uint32_t* table = aux->segTable;
        const uint32_t tile = ticket * aux->rowBlockStride + aux->rowBlockOffset;         // round-robin over the ranks
        uint32_t lo = 0, hi = segments;
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) / 2u;
            if (table[mid] <= tile) lo = mid;
            else hi = mid;
        }
        const uint32_t seg = lo;
        const uint32_t quadBlock = table[segments + 1u + seg] + 4u * (tile - table[seg]);
        const uint32_t block = quadBlock + wave;
        const bool idle = block >= aux->rowBlocks;
        const uint32_t fragmentBlock = idle ? aux->rowBlocks - 1u : block;
        const uint32_t quadRowBase = quadBlock * 64u;
        const uint32_t rowBase = block * 64u;
        const uint32_t row = rowBase + lane;
        const bool rowValid = !idle && row < cellCount;
        const uint32_t colBegin = aux->columnLimit + seg * aux->columnsPerSegment;
        uint32_t colEnd = colBegin + aux->columnsPerSegment;
        if (colEnd > cellCount) colEnd = cellCount;
        const bool last = quadRowBase < colEnd;
        const uint32_t commonEnd = last ? quadRowBase : colEnd;
        const int32_t snapRow = rowValid ? aux->snap[row] : -1;
        if (colBegin < commonEnd) {
            uint32_t unusedLogCount = 0;
            if (PINNED) {
                // the walk logs what passes either bound; both sides of every record go to the inbox afterwards
                const uint32_t logCapacity = aux->logCapacity < kMatrixLogMargin ? kMatrixLogMargin : aux->logCapacity;
                Entry* waveLog = aux->logs + size_t(blockIdx.x * 4u + wave) * 64u * logCapacity;
                uint32_t at = colBegin;
                if (WIDE) {
                    // (2048 bits: the columns once per half of the wave's rows, see scanTilesMatrixWide)
                    for (uint32_t rowHalf = 0; rowHalf < 2u; ++rowHalf) {
                        at = colBegin;
                        while (at < commonEnd) {
                            uint32_t records[2] = {0u, 0u};
                            at = scanTilesMatrixWide<true>(aux->fragments, aux->snap, at, commonEnd, 2u * fragmentBlock + rowHalf,
                                                           2.f * kMatrixBits - 2.f * float(snapRow), rowHalf, waveLog, logCapacity, records,
                                                           ldsAddress(tiles), ldsAddress(const_cast<uint32_t*>(shared)), ldsAddress(walkBlock));
                            if (!idle) drainWalkLogs<true>(waveLog, logCapacity, records, lane, rowBase, cellCount, emitPos, emitEnd);
                        }
                    }
                }
                while (!WIDE && at < commonEnd) {
                    uint32_t records[2] = {0u, 0u};
                    if (EM2_DIAG_WORD(aux)) {
                        at = scanTilesMatrixPinned<true, true, true>((const void*)(uintptr_t)aux, aux->fragments, aux->snap, at, commonEnd,
                                                           2u * fragmentBlock, kMatrixBits - 2.f * float(snapRow), waveLog, logCapacity,
                                                           records, ldsAddress(tiles), ldsAddress(const_cast<uint32_t*>(shared)),
                                                           ldsAddress(walkBlock));
                    } else {
                        at = scanTilesMatrixPinned<true, true, false>((const void*)(uintptr_t)aux, aux->fragments, aux->snap, at, commonEnd,
                                                           2u * fragmentBlock, kMatrixBits - 2.f * float(snapRow), waveLog, logCapacity,
                                                           records, ldsAddress(tiles), ldsAddress(const_cast<uint32_t*>(shared)),
                                                           ldsAddress(walkBlock));
                    }
                    if (!idle) drainWalkLogs(waveLog, logCapacity, records, lane, rowBase, cellCount, emitPos, emitEnd);
                }
            } else {
                scanTilesMatrix<true, true>(static_cast<const FragmentWord4*>(aux->fragments), aux->snap, colBegin, commonEnd,
                                            2u * fragmentBlock, kMatrixBits - 2.f * float(snapRow), row, rowValid, lane, nullptr, 0u,
                                            unusedLogCount, emitPos, emitEnd, tiles, shared);
            }
        }
        if (last && !idle) {
            uint32_t r[W32];
            const uint32_t* rp = kernelArgs()->sig32 + size_t(rowValid ? row : rowBase) * uint32_t(W32);
#pragma unroll
            for (int w = 0; w < W32; ++w) r[w] = rp[w];
            uint32_t at = quadRowBase;
            while (at < rowBase) {
                ensureInboxRoomForTile(lane, emitPos, emitEnd);
                at = scanTileEmit<W32>(kernelArgs()->sig32, kernelArgs()->snap, at, rowBase, r, row, rowValid, snapRow, lane, emitPos,
                                      emitEnd);
            }
            uint32_t diagEnd = rowBase + 64u;
            if (diagEnd > cellCount) diagEnd = cellCount;
            const uint32_t* sig32 = kernelArgs()->sig32;
            const int32_t* snap = kernelArgs()->snap;
            for (uint32_t col = rowBase; col < diagEnd; ++col) {
                ScalarPtr cp = (ScalarPtr)(uintptr_t)sig32 + size_t(col) * uint32_t(W32);
                uint32_t m = 0;
#pragma unroll
                for (int w = 0; w < W32; ++w) popcountAccumulate(m, r[w] ^ cp[w]);
                const int32_t snapCol = snap[col];
                const bool lower = rowValid && col < row;
                emitColumn(lower && int32_t(m) <= snapCol, col, row, m, lane, emitPos, emitEnd);      // target col
                emitColumn(lower && int32_t(m) <= snapRow, row, col, m, lane, emitPos, emitEnd);      // target row
            }
        }
    }
    {
        const uint32_t p = uint32_t(__builtin_amdgcn_readfirstlane(int(emitPos)));
        const uint32_t e = uint32_t(__builtin_amdgcn_readfirstlane(int(emitEnd)));
        if (p <= e) {
            uint64_t* inbox = kernelArgs()->inbox;
            for (uint32_t i = p + lane; i < e; i += 64u) inbox[i] = ~0ull;
        }
    }
}

__global__ void __launch_bounds__(256, 2)
fsp4TileMatrixKernel(Fsp4Args args)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
    tileMatrixBody<false>(ldsRaw);
}

__global__ void __launch_bounds__(256, 2)
fsp4TileMatrixPinnedKernel(Fsp4Args args)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
    tileMatrixBody<true>(ldsRaw);
}

// 2048-bit signatures
__global__ void __launch_bounds__(256, 2)
fsp4TileMatrixWideKernel(Fsp4Args args)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
    tileMatrixBody<true, true>(ldsRaw);
}

// max over `count` arrays of `n` int32 laid out back to back (the emulation's stand-in for all_reduce(MAX))
__global__ void maxReduceKernel(int32_t* __restrict__ arrays, uint32_t n, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t best = arrays[i];
    for (uint32_t a = 1; a < count; ++a) best = arrays[size_t(a) * n + i] > best ? arrays[size_t(a) * n + i] : best;
    for (uint32_t a = 0; a < count; ++a) arrays[size_t(a) * n + i] = best;
}

}  // namespace


// ---- symmetric (triangle) scan: eligibility and workspace ----
// EM2_SCAN_MODE=triangle forces it wherever it is possible (all rows of the problem in one launch),
// EM2_SCAN_MODE=persistent / simple disable it; by default it is used from kSymmetricMinCells cells on.
// dynamic LDS of the matrix kernels behind matrixLdsOffset: four tiles, the stop words + ticket, the waves' walk blocks
constexpr size_t kMatrixLdsBytes = 4u * kMatrixTileWords * 16u + 64u + 4u * kMatrixWalkLdsBytes;

// fsp4ScanMatrixKernel: the waves' walk blocks alias their selection areas when those are large enough (see there)
static bool matrixWalkAliasesSelection(uint32_t k)
{
    const uint32_t selectionStride = 2u * k * kLdsBytesPerEntrySlot;
    return selectionStride >= kMatrixWalkLdsBytes && selectionStride % 16u == 0u;
}

static size_t scanMatrixLdsBytes(uint32_t k)
{
    return kMatrixLdsBytes - (matrixWalkAliasesSelection(k) ? 4u * kMatrixWalkLdsBytes : 0u);
}

// EM2_MATRIX_WALK=0 keeps the compiler-scheduled walk (scanTilesMatrix) for A/B runs.
// (bit 0: fsp4ScanMatrixKernel, bit 1: fsp4TileMatrixKernel; default both)
static bool matrixWalkPinned(uint32_t which = 1u) { return (envNumber("EM2_MATRIX_WALK", 3) & which) != 0; }

static const void* scanMatrixKernelFor(bool identity, bool wide = false)
{
    if (identity && (diagNumber("EM2_MATRIX_DIAG") & 2048u) && (wide || matrixWalkPinned())) {
        return wide ? reinterpret_cast<const void*>(&fsp4ScanMatrixTimedKernel<true>)
                    : reinterpret_cast<const void*>(&fsp4ScanMatrixTimedKernel<false>);
    }
    if (wide) {
        return identity ? reinterpret_cast<const void*>(&fsp4ScanMatrixWideKernel<true>)
                        : reinterpret_cast<const void*>(&fsp4ScanMatrixWideKernel<false>);
    }
    if (matrixWalkPinned()) {
        return identity ? reinterpret_cast<const void*>(&fsp4ScanMatrixPinnedKernel<true>)
                        : reinterpret_cast<const void*>(&fsp4ScanMatrixPinnedKernel<false>);
    }
    return identity ? reinterpret_cast<const void*>(&fsp4ScanMatrixKernel<true>)
                    : reinterpret_cast<const void*>(&fsp4ScanMatrixKernel<false>);
}

constexpr uint32_t kSymmetricMinCells = 131072;
constexpr uint32_t kSymmetricMatrixMinCells = 32768;
constexpr uint32_t kMaxSegments = 64;
constexpr uint32_t kMatrixMaxSegments = 1024;    // (room for short segments: 2048 columns of 2048-bit fragments are the 2 MB an XCD's L2 holds)
constexpr uint32_t kTableWords = 2u * kMatrixMaxSegments + 2u;
constexpr uint32_t kInboxChunk = 512;

// Which signature widths take the matrix-core form of the triangle.  The fragments are always 1024 bits wide: a
// narrower signature is zero-extended (a zero bit is +1 on both sides, so the dot product stays 1024 - 2 * mismatches),
// which costs the full 16 k-steps per tile whatever the width.  EM2_SCAN_MATRIX: 0 never, 1 (default) the widths it
// is faster for (129..1024 bits, kMatrixMinPaddedDw), 2 every width up to 1024 bits (tests).
static bool matrixFormWanted(uint32_t paddedDw)
{
    const uint64_t mode = envNumber("EM2_SCAN_MATRIX", 1);
    if (mode == 0 || paddedDw > 32u) return false;
    return mode >= 2 || paddedDw >= kMatrixMinPaddedDw;
}

// 1025..2048-bit signatures (64 dwords as the scan sees them): the 2048-bit form of the matrix kernel
// (fsp4ScanMatrixWideKernel: 32 rows per wave and pass, two passes).  EM2_SCAN_MATRIX=0 / EM2_SCAN_MATRIX_WIDE=0 keep
// the v_xor/v_bcnt form.
static bool matrixWideWanted(uint32_t paddedDw)
{
    return paddedDw == 64u && envNumber("EM2_SCAN_MATRIX", 1) != 0 && envNumber("EM2_SCAN_MATRIX_WIDE", 1) != 0;
}

bool symmetricEligible(uint32_t cellCount, uint32_t rowCount, uint32_t paddedDw)
{
    // inbox keys hold two cell ids and a mismatch count in 64 bits: 13 + 2 * bits(cellCount) <= 64
    if (rowCount != cellCount || cellCount < 128u || cellCount > (1u << 25)) return false;
    const char* v = getenv("EM2_SCAN_MODE");
    if (v && v[0] == 't') return true;
    if (v && (v[0] == 's' || v[0] == 'p')) return false;
    // 1024-bit signatures take the matrix-core form, which wins much earlier (scan ms ordered / symmetric-matrix:
    // 30k cells 2.5 / 2.4, 60k 8.2 / 4.2, 100k 20.1 / 7.5)
    const bool matrix = matrixFormWanted(paddedDw) || matrixWideWanted(paddedDw);
    return cellCount >= envNumber("EM2_SYMMETRIC_MIN_CELLS", matrix ? kSymmetricMatrixMinCells : kSymmetricMinCells);
}

static uint64_t inboxCapacity(uint32_t cellCount)
{
    // EM2_INBOX_CAPACITY (entries) is a test knob: tiny pools force the overflow -> ordered-scan fallback.
    const uint64_t forced = envNumber("EM2_INBOX_CAPACITY", 0);
    if (forced >= kInboxChunk) return forced < 0xfff00000ull ? forced : 0xfff00000ull;
    uint64_t cap = uint64_t(cellCount) * 1024u;
    const uint64_t floor = uint64_t(maxResidentWaves()) * kInboxChunk * 2u;      // every wave can hold a chunk
    if (cap < floor) cap = floor;
    if (cap > 0xfff00000ull) cap = 0xfff00000ull;
    return cap;
}

static size_t inboxSortTempBytes(uint64_t capacity)
{
    size_t bytes = 0;
    uint64_t* none = nullptr;
    if (rocprim::radix_sort_keys(nullptr, bytes, none, none, size_t(capacity), 0u, 64u, hipStream_t(nullptr)) != hipSuccess) return 0;
    return bytes;
}

struct SymmetricLayout {
    size_t snap, table, tableMatrix, control, poolA, poolB, temp, fragments, widened, total, tempBytes;
    uint64_t capacity;
};

static SymmetricLayout symmetricLayout(uint32_t cellCount, uint32_t paddedDw)
{
    SymmetricLayout l;
    l.capacity = inboxCapacity(cellCount);
    l.tempBytes = inboxSortTempBytes(l.capacity);
    size_t at = 0;
    l.snap = at;    at += align256(size_t(cellCount) * 4u);
    l.table = at;   at += align256(kTableWords * 4u);
    l.tableMatrix = at; at += align256(kTableWords * 4u);
    l.control = at; at += 256u;
    l.poolA = at;   at += align256(size_t(l.capacity) * 8u);
    l.poolB = at;   at += align256(size_t(l.capacity) * 8u);
    l.temp = at;    at += align256(l.tempBytes);
    // FP4 fragments, matrix form: 4 bits per signature bit
    l.fragments = at; at += align256(size_t((cellCount + 63u) / 64u) * 64u * (matrixWideWanted(paddedDw) ? 1024u : 512u));
    // signatures zero-extended to 1024 bits for the v_xor/v_bcnt parts of the matrix kernel (a quad's own 256 columns)
    l.widened = at;
    if (paddedDw < 32u && matrixFormWanted(paddedDw)) at += align256(size_t(cellCount) * 128u);
    l.total = at;
    return l;
}

bool fsp4MatrixFormWanted(uint32_t paddedDw)
{
    return matrixFormWanted(paddedDw) || matrixWideWanted(paddedDw);
}

bool fsp4UsesSymmetricScan(uint32_t cellCount, uint32_t rowCount, uint32_t paddedDw)
{
    return symmetricEligible(cellCount, rowCount, paddedDw);
}

size_t fsp4SymmetricBytes(uint32_t cellCount, uint32_t rowCount, uint32_t paddedDw)
{
    if (!symmetricEligible(cellCount, rowCount, paddedDw)) return 0;
    return symmetricLayout(cellCount, paddedDw).total;
}

// Resident waves of a persistent-style launch of `kernel` (min(occupancy, 4 waves per SIMD) x CUs).
static hipError_t residentWaveSlots(const void* kernel, uint32_t wavesPerBlock, size_t lds, uint32_t* slots)
{
    int device = 0, cuCount = 0, blocksPerCu = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    e = hipDeviceGetAttribute(&cuCount, hipDeviceAttributeMultiprocessorCount, device);
    if (e != hipSuccess) return e;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocksPerCu, kernel, int(64u * wavesPerBlock), lds);
    if (e != hipSuccess) return e;
    if (blocksPerCu < 1) blocksPerCu = 1;
    int wanted = int(16u / wavesPerBlock);
    if (wanted < 1) wanted = 1;
    const char* v = getenv("EM2_BLOCKS_PER_CU");
    if (v && atoi(v) >= 1) wanted = atoi(v);
    if (wanted < blocksPerCu) blocksPerCu = wanted;
    *slots = uint32_t(cuCount) * uint32_t(blocksPerCu) * wavesPerBlock;
    return hipSuccess;
}

// The symmetric scan (see fsp4ScanSymmetricKernel).  *done = false when the inbox pool overflowed: nothing usable
// was produced and the caller runs the ordered scan instead.  Synchronises the stream (the sort size is read back).
hipError_t launchFsp4ScanSymmetric(Fsp4Args args, uint32_t paddedDw, bool identity, uint32_t wavesPerBlock,
                                          size_t lds, void* control, void* symmetricWs, hipStream_t stream, bool* done)
{
    *done = false;
    const uint32_t cellCount = args.cellCount;
    const uint32_t rowBlocks = args.rowBlocks;
    const void* kernel = nullptr;
#define EM2_SYMMETRIC(W32) \
    (identity ? reinterpret_cast<const void*>(&fsp4ScanSymmetricKernel<W32, true>) \
              : reinterpret_cast<const void*>(&fsp4ScanSymmetricKernel<W32, false>))
    switch (paddedDw) {
    case 2: kernel = EM2_SYMMETRIC(2); break;
    case 4: kernel = EM2_SYMMETRIC(4); break;
    case 8: kernel = EM2_SYMMETRIC(8); break;
    case 16: kernel = EM2_SYMMETRIC(16); break;
    case 32: kernel = EM2_SYMMETRIC(32); break;
    case 64: kernel = EM2_SYMMETRIC(64); break;
    case 128: kernel = EM2_SYMMETRIC(128); break;
    default: return hipErrorInvalidValue;
    }
#undef EM2_SYMMETRIC
    uint32_t slots = 0;
    hipError_t e = residentWaveSlots(kernel, wavesPerBlock, lds, &slots);
    if (e != hipSuccess) return e;

    // Cells below c0 scan all columns themselves.  Default max(4096, 32k), at most 1/8 of the cells: the snapshots of
    // cells with fewer than ~k similar lower neighbours filter nothing.  Measured at 1M cells (64 clusters, k=100):
    // c0 = 0 / 4096 / 16384 / 65536 -> 1088* / 932 / 940 / 1170* ms (* before the call-free loop).  EM2_FULL_ROW_CELLS
    // overrides (tests use 0 .. everything).
    uint64_t fullCells = 32ull * args.k;
    if (fullCells < 4096) fullCells = 4096;
    if (fullCells > cellCount / 8u) fullCells = cellCount / 8u;
    fullCells = envNumber("EM2_FULL_ROW_CELLS", fullCells);
    uint32_t fullRowBlocks = uint32_t((fullCells + 63u) / 64u);
    if (fullRowBlocks > rowBlocks) fullRowBlocks = rowBlocks;

    // Matrix-core form of the triangle part (see fsp4ScanMatrixKernel): 1024-bit signatures, the plain single-GPU
    // launch.  EM2_SCAN_MATRIX=0 keeps the v_xor/v_bcnt form.  Full-row and segment boundaries become multiples of 256
    // cells so that the four waves of a block always walk the same columns.
    const bool wide = matrixWideWanted(paddedDw);
    bool matrix = (matrixFormWanted(paddedDw) || wide) && wavesPerBlock == 4u && args.rowBlockStride == 1u && args.rowBlockOffset == 0u &&
                  args.localBlockBase == 0u && args.shardFlags == 0u && args.columnLimit == cellCount && args.rowBegin == 0u &&
                  ((lds + 15u) & ~size_t(15)) + scanMatrixLdsBytes(args.k) <= 150u * 1024u;      // selection area + four tiles
    if (matrix) {
        fullRowBlocks = (fullRowBlocks + 3u) & ~3u;
        if (fullRowBlocks >= rowBlocks) {
            fullRowBlocks = rowBlocks;
            matrix = false;             // nothing left for the triangle
        }
    }

    // Segments: as many as the column-count floor allows, up to kMaxSegments (EM2_SEGMENTS overrides): short
    // segments keep the column snapshots fresh and even out the triangle.
    uint64_t minSegmentColumns = envNumber("EM2_MIN_SEGMENT_COLUMNS", 4096);
    if (minSegmentColumns < 1) minSegmentColumns = 1;
    const uint32_t maxSegments = matrix ? kMatrixMaxSegments : kMaxSegments;
    uint64_t segments = cellCount / minSegmentColumns;
    if (segments > maxSegments) segments = maxSegments;
    const uint64_t forcedSegments = envNumber("EM2_SEGMENTS", 0);
    if (forcedSegments >= 1 && forcedSegments <= maxSegments) segments = forcedSegments;
    if (segments < 1) segments = 1;
    uint32_t cps = uint32_t((uint64_t(cellCount) + segments - 1u) / segments);
    if (matrix) cps = (cps + 255u) & ~255u;
    segments = (uint64_t(cellCount) + cps - 1u) / cps;

    // Tickets of the v_xor/v_bcnt kernel: per segment the full-row blocks, then (unless the matrix kernel takes them)
    // the triangle blocks that reach into the segment.  Tickets of the matrix kernel: per segment the quads likewise.
    uint32_t table[kTableWords], tableMatrix[kTableWords];
    uint64_t tickets = 0, ticketsMatrix = 0;
    for (uint32_t sIdx = 0; sIdx < segments; ++sIdx) {
        uint32_t firstTriangle = uint32_t((uint64_t(sIdx) * cps) / 64u);
        if (firstTriangle < fullRowBlocks) firstTriangle = fullRowBlocks;
        if (firstTriangle > rowBlocks) firstTriangle = rowBlocks;
        table[sIdx] = uint32_t(tickets);
        table[segments + 1u + sIdx] = firstTriangle;
        tickets += matrix ? 0u : fullRowBlocks + (rowBlocks - firstTriangle);         // the matrix kernel takes everything
        if (tickets >= 0xffffffffull) return hipErrorInvalidValue;
    }
    table[segments] = uint32_t(tickets);
    // The matrix kernel has segments of its own, and longer ones: every item starts by loading 32 KB of row fragments per
    // wave and priming the tile pipeline, and its columns need not stay in one L2 (kernel ms at 1M cells with 4096 /
    // 8192 / 16384 / 32768 / 131072 columns per segment: 289 / 278 / 273 / 270 / 274; 2048: 335).  The test knobs apply
    // to both launches.
    uint64_t segmentsMatrix = segments;
    uint32_t cpsMatrix = cps;
    if (matrix) {
        uint64_t defaultColumns = cellCount / 24u;              // small problems keep enough items to fill the machine
        defaultColumns = defaultColumns < 4096 ? 4096 : (defaultColumns > 16384 ? 16384 : defaultColumns);
        uint64_t minColumns = envNumber("EM2_MIN_SEGMENT_COLUMNS", defaultColumns);
        if (minColumns < 1) minColumns = 1;
        segmentsMatrix = cellCount / minColumns;
        if (segmentsMatrix > kMatrixMaxSegments) segmentsMatrix = kMatrixMaxSegments;
        if (forcedSegments >= 1 && forcedSegments <= kMatrixMaxSegments) segmentsMatrix = forcedSegments;
        if (segmentsMatrix < 1) segmentsMatrix = 1;
        cpsMatrix = uint32_t((uint64_t(cellCount) + segmentsMatrix - 1u) / segmentsMatrix);
        cpsMatrix = (cpsMatrix + 255u) & ~255u;
        segmentsMatrix = (uint64_t(cellCount) + cpsMatrix - 1u) / cpsMatrix;
        for (uint32_t sIdx = 0; sIdx < segmentsMatrix; ++sIdx) {
            uint32_t firstQuad = uint32_t((uint64_t(sIdx) * cpsMatrix) / 64u);      // a multiple of 4
            if (firstQuad < fullRowBlocks) firstQuad = fullRowBlocks;
            if (firstQuad > rowBlocks) firstQuad = rowBlocks;
            tableMatrix[sIdx] = uint32_t(ticketsMatrix);
            tableMatrix[segmentsMatrix + 1u + sIdx] = firstQuad;
            ticketsMatrix += fullRowBlocks / 4u + (rowBlocks - firstQuad + 3u) / 4u;       // full-row quads, then the triangle's
            if (ticketsMatrix >= 0xffffffffull) return hipErrorInvalidValue;
        }
        tableMatrix[segmentsMatrix] = uint32_t(ticketsMatrix);
    }

    const SymmetricLayout layout = symmetricLayout(cellCount, paddedDw);
    char* ws = static_cast<char*>(symmetricWs);
    char* c = static_cast<char*>(control);
    const size_t stateBytes = align256(size_t(rowBlocks) * 64u * 8u);
    const size_t doneBytes = align256(size_t(rowBlocks) * 4u);
    args.rowState = reinterpret_cast<uint32_t*>(c);
    args.segmentsDone = reinterpret_cast<uint32_t*>(c + stateBytes);
    args.control = reinterpret_cast<uint32_t*>(c + stateBytes + doneBytes);
    args.logs = reinterpret_cast<Entry*>(c + stateBytes + doneBytes + 256u);
    args.logCapacity = kLogCapacity;
    if (const char* v = getenv("EM2_LOG_CAPACITY")) {
        if (atoi(v) >= 1 && uint32_t(atoi(v)) < kLogCapacity) args.logCapacity = uint32_t(atoi(v));
    }
    args.segments = uint32_t(segments);
    args.columnsPerSegment = cps;
    args.snap = reinterpret_cast<int32_t*>(ws + layout.snap);
    args.inbox = reinterpret_cast<uint64_t*>(ws + layout.poolA);
    args.inboxControl = reinterpret_cast<uint32_t*>(ws + layout.control);
    args.segTable = reinterpret_cast<const uint32_t*>(ws + layout.table);
    args.inboxCapacity = layout.capacity;
    args.inboxChunk = kInboxChunk;
    args.fullRowBlocks = fullRowBlocks;
    uint32_t rowBits = 1;
    while ((1ull << rowBits) < uint64_t(cellCount)) ++rowBits;
    args.rowBits = rowBits;
    args.totalTickets = uint32_t(tickets);

    e = hipMemsetAsync(c + stateBytes, 0, doneBytes + 256u, stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(ws + layout.control, 0, 256u, stream);
    if (e != hipSuccess) return e;
    const size_t fullCellsClamped = size_t(fullRowBlocks) * 64u < cellCount ? size_t(fullRowBlocks) * 64u : cellCount;
    if (fullCellsClamped) {
        e = hipMemsetAsync(args.snap, 0xff, fullCellsClamped * 4u, stream);
        if (e != hipSuccess) return e;
    }
    if (cellCount > fullCellsClamped) {
        e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(args.snap + fullCellsClamped), args.mMaxInitial,
                              cellCount - fullCellsClamped, stream);
        if (e != hipSuccess) return e;
    }
    e = hipMemcpyAsync(ws + layout.table, table, (2u * segments + 2u) * 4u, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;

    const dim3 block(64u * wavesPerBlock);
    static thread_local hipEvent_t timing[3] = {nullptr, nullptr, nullptr};
    if (!timing[0]) {
        if (hipEventCreate(&timing[0]) != hipSuccess || hipEventCreate(&timing[1]) != hipSuccess ||
            hipEventCreate(&timing[2]) != hipSuccess) timing[0] = timing[1] = timing[2] = nullptr;
    }
    if (timing[0]) (void)hipEventRecord(timing[0], stream);
    if (tickets) {
        uint64_t wavesWanted = tickets;
        if (wavesWanted > slots) wavesWanted = slots;
        if (wavesWanted > maxResidentWaves()) wavesWanted = maxResidentWaves();
        const dim3 grid(uint32_t((wavesWanted + wavesPerBlock - 1u) / wavesPerBlock));
        void* kernelArgsArray[] = {&args};
        e = hipLaunchKernel(kernel, grid, block, kernelArgsArray, lds, stream);
        if (e != hipSuccess) return e;
    }
    if (matrix) {
        // full-row quads and triangle quads in one launch of the matrix kernel
        e = hipMemcpyAsync(ws + layout.tableMatrix, tableMatrix, (2u * segmentsMatrix + 2u) * 4u, hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(args.control, 0, 4u, stream);                 // the ticket; the error word stays
        if (e != hipSuccess) return e;
        const uint32_t matrixSteps = wide ? 2u * kMatrixSteps : kMatrixSteps;
        const uint32_t fragmentCount = rowBlocks * 2u * matrixSteps * 64u;
        Fsp4Args matrixArgs = args;
        if (paddedDw < 32u) {
            uint32_t* widened = reinterpret_cast<uint32_t*>(ws + layout.widened);
            widenSignaturesKernel<<<dim3((cellCount * 32u + 255u) / 256u), dim3(256), 0, stream>>>(args.sig32, paddedDw, cellCount, widened);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
            matrixArgs.sig32 = widened;
        }
        expandFragmentsKernel<<<dim3((fragmentCount + 255u) / 256u), dim3(256), 0, stream>>>(
            matrixArgs.sig32, cellCount, fragmentCount, reinterpret_cast<FragmentWord4*>(ws + layout.fragments), matrixSteps);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        matrixArgs.segTable = reinterpret_cast<const uint32_t*>(ws + layout.tableMatrix);
        matrixArgs.segments = uint32_t(segmentsMatrix);
        matrixArgs.columnsPerSegment = cpsMatrix;
        matrixArgs.totalTickets = uint32_t(ticketsMatrix);
        matrixArgs.fragments = ws + layout.fragments;
        matrixArgs.matrixLdsOffset = uint32_t((lds + 15u) & ~size_t(15));
        // (EM2_MATRIX_DEPART_US / EM2_MATRIX_DEPART_WINDOW_US: period and window of the walks' departures, microseconds)
        matrixArgs.departTicks = uint32_t(envNumber("EM2_MATRIX_DEPART_US", 0) * 100u);
        matrixArgs.departWindow = uint32_t(envNumber("EM2_MATRIX_DEPART_WINDOW_US", envNumber("EM2_MATRIX_DEPART_US", 0) / 4u) * 100u);
        const size_t matrixLds = size_t(matrixArgs.matrixLdsOffset) + scanMatrixLdsBytes(args.k);
        const void* matrixKernel = scanMatrixKernelFor(identity, wide);
        int device = 0, cuCount = 0, blocksPerCu = 0;
        e = hipGetDevice(&device);
        if (e != hipSuccess) return e;
        e = hipDeviceGetAttribute(&cuCount, hipDeviceAttributeMultiprocessorCount, device);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(matrixKernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(matrixLds));       // more than 64 KB
        if (e != hipSuccess) return e;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocksPerCu, matrixKernel, 256, matrixLds);
        if (e != hipSuccess) return e;
        blocksPerCu = blocksPerCu < 1 ? 1 : (blocksPerCu > 2 ? 2 : blocksPerCu);
        if (const char* v = getenv("EM2_BLOCKS_PER_CU")) {
            if (atoi(v) >= 1 && atoi(v) < blocksPerCu) blocksPerCu = atoi(v);
        }
        uint64_t blocksWanted = uint64_t(cuCount) * uint64_t(blocksPerCu);
        if (blocksWanted * 4u > maxResidentWaves()) blocksWanted = maxResidentWaves() / 4u;      // the logs are sized for that
        if (blocksWanted > ticketsMatrix) blocksWanted = ticketsMatrix;
        if (const char* v = getenv("EM2_SCAN_VERBOSE")) {
            if (v[0] == '1') fprintf(stderr, "[em2] matrix kernel: %d blocks per CU, %llu blocks, %zu bytes of LDS, walk %s\n", blocksPerCu,
                                     (unsigned long long)blocksWanted, matrixLds, matrixWalkPinned() ? "pinned" : "compiler");
        }
        void* matrixArgsArray[] = {&matrixArgs};
        if (timing[0]) (void)hipEventRecord(timing[2], stream);
        e = hipLaunchKernel(matrixKernel, dim3(uint32_t(blocksWanted)), dim3(256), matrixArgsArray, matrixLds, stream);
        if (e != hipSuccess) return e;
    }
    if (timing[0]) (void)hipEventRecord(timing[1], stream);
    const uint64_t ticketsMatrixCount = matrix ? ticketsMatrix : 0u;

    // the number of inbox entries (incl. chunk tails), the overflow flag and the hand-off error word
    // (and, in the same copy, the two clock words the matrix kernel leaves kClockWordsOffset words further on)
    uint32_t inboxWords[kClockWordsOffset + 4u] = {0};
    uint32_t controlWords[2] = {0, 0};
    e = hipMemcpyAsync(inboxWords, ws + layout.control, sizeof(inboxWords), hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(controlWords, args.control, sizeof(controlWords), hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    if (controlWords[1] != 0u) {
        *done = true;       // a hand-off timed out: the error word stays set for readFsp4Error
        return hipSuccess;
    }
    const uint64_t used = uint64_t(inboxWords[0]) | (uint64_t(inboxWords[1]) << 32);
    if (inboxWords[2] != 0u || used > layout.capacity) return hipSuccess;      // overflow: *done stays false
    double matrixClockGHz = 0.0;
    if (matrix) {
        unsigned long long ticks[2] = {0, 0};
        std::memcpy(ticks, inboxWords + kClockWordsOffset, sizeof(ticks));
        if (ticks[1]) matrixClockGHz = double(ticks[0]) / double(ticks[1]) * 0.1;          // s_memrealtime counts at 100 MHz
    }
    if (matrix && (diagNumber("EM2_MATRIX_DIAG") & 2048u)) {
        unsigned long long cycles[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpy(cycles, ws + layout.control + 64u, sizeof(cycles), hipMemcpyDeviceToHost) == hipSuccess) {
            double total = 0;
            for (int i = 0; i < 7; i++) total += double(cycles[i]);
            const char* names[7] = {"ticket", "set-up", "walk", "hand-off wait", "replay", "own columns", "publication"};
            fprintf(stderr, "[em2] matrix kernel, wave cycles by phase (%llu items):", (unsigned long long)ticketsMatrixCount);
            for (int i = 0; i < 7; i++) fprintf(stderr, " %s %.1f%%", names[i], 100.0 * double(cycles[i]) / (total > 0 ? total : 1));
            fprintf(stderr, "; %.0f cycles per item and wave outside the walk\n",
                    (total - double(cycles[2])) / (4.0 * double(ticketsMatrixCount ? ticketsMatrixCount : 1)));
        }
    }
    {
        float ms = -1.0f;
        if (!timing[0] || hipEventElapsedTime(&ms, timing[0], timing[1]) != hipSuccess) ms = -1.0f;
        double steps = matrix ? 0.0 : double(fullRowBlocks) * double(cellCount);       // (wave, column) steps of the v_xor/v_bcnt code
        double matrixPairs = matrix ? 64.0 * double(fullRowBlocks) * double(cellCount) : 0.0;
        for (uint32_t b = fullRowBlocks; b < rowBlocks; ++b) {
            const uint64_t end = uint64_t(b) * 64u + 64u;
            const uint64_t quadBase = matrix ? uint64_t(b & ~3u) * 64u : 0u;      // the matrix cores take the columns below the quad
            steps += double((end < cellCount ? end : cellCount) - quadBase);
            matrixPairs += 64.0 * double(quadBase);
        }
        float matrixMs = -1.0f;
        if (matrix && (!timing[0] || hipEventElapsedTime(&matrixMs, timing[2], timing[1]) != hipSuccess)) matrixMs = -1.0f;
        lastLaunchInfo.matrixPairs = matrixPairs;
        lastLaunchInfo.matrixKernelMs = double(matrixMs);
        lastLaunchInfo.matrixClockGHz = matrixClockGHz;
        lastLaunchInfo.form = matrix ? 3 : 1;
        lastLaunchInfo.scanKernelMs = double(ms);
        lastLaunchInfo.waveColumnSteps = steps;
        lastLaunchInfo.inboxEntries = double(used);
        lastLaunchInfo.segments = double(segments);
        lastLaunchInfo.fullRowCells = double(fullCellsClamped);
    }
    if (const char* v = getenv("EM2_SCAN_VERBOSE")) {
        if (v[0] == '1') fprintf(stderr, "[em2] symmetric scan%s: %u segments x %u columns, %u full-row blocks, %llu + %llu tickets, %llu inbox slots\n",
                                 matrix ? " (matrix cores)" : "", uint32_t(matrix ? segmentsMatrix : segments), matrix ? cpsMatrix : cps, fullRowBlocks,
                                 (unsigned long long)tickets, (unsigned long long)(matrix ? ticketsMatrix : 0), (unsigned long long)used);
    }

    const uint64_t* sorted = args.inbox;
    if (used) {
        size_t tempBytes = layout.tempBytes;
        uint64_t* out = reinterpret_cast<uint64_t*>(ws + layout.poolB);
        e = rocprim::radix_sort_keys(ws + layout.temp, tempBytes, args.inbox, out, size_t(used), 13u, 13u + 2u * rowBits, stream);
        if (e != hipSuccess) return e;
        sorted = out;
    }
    if (rowBlocks > fullRowBlocks) {
        const uint32_t waves = rowBlocks - fullRowBlocks;
        const dim3 rgrid((waves + wavesPerBlock - 1u) / wavesPerBlock);
        if (identity) fsp4InboxReplayKernel<true><<<rgrid, block, lds, stream>>>(args, sorted, used);
        else fsp4InboxReplayKernel<false><<<rgrid, block, lds, stream>>>(args, sorted, used);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (diagNumber("EM2_MATRIX_DIAG") & 4096u) {
        uint32_t longest = 0;
        if (hipMemcpy(&longest, ws + layout.control + 32u, 4u, hipMemcpyDeviceToHost) == hipSuccess) {
            unsigned long long cycles[3] = {0, 0, 0};
            (void)hipMemcpy(cycles, ws + layout.control + 40u, sizeof(cycles), hipMemcpyDeviceToHost);
            const double waves = double(rowBlocks - fullRowBlocks);
            fprintf(stderr, "[em2] inbox replay: %llu entries, %.1f per cell on average, longest inbox of a cell %u; cycles per wave: "
                            "search %.0f, replay %.0f, finish %.0f\n",
                    (unsigned long long)used, double(used) / double(cellCount), longest, double(cycles[0]) / waves,
                    double(cycles[1]) / waves, double(cycles[2]) / waves);
        }
    }
    *done = true;
    return hipSuccess;
}

// =========================================================================================================
// Sharded symmetric scan (one process per GPU; the collectives between the phases are the caller's, see
// expressionmatrix2_amd/sharded.py; runFsp4ShardedEmulation below plays all ranks on one GPU for the tests).
//
// 64-cell blocks are dealt to the ranks round-robin (block g belongs to rank g % world, where it is list / state
// slot g / world), so every rank holds rows of every part of the triangle.  The first M = prefixBlocks*64 cells are
// the PREFIX.
//   phase 0  own prefix blocks x columns [0,M): ordered in-lane scan (every pair of prefix cells is evaluated from
//            both sides: M^2 instead of M^2/2, 2% of the job at M = N/5); snapshots snap[c], c < M.
//            -> all_reduce(MAX) of snap
//   phase 1  own other blocks x columns [0,M): in-lane scan of the rows (their first M candidates), entries
//            (target c < M, candidate r) filtered by snap[c];  snapshots snap[r], r >= M.
//            -> all_reduce(MAX) of snap
//   phase 2  tiles of [M,N)^2 dealt round-robin (fsp4TileKernel): both sides deferred, filtered by the snapshots.
//            -> all_gather of the ranks' entry pools
//   phase 3  sort all entries by (target, candidate), replay own slots, finish own rows (global output index).
// Every cell is offered its candidates in ascending order: in-lane part first (columns < M), then its inbox.
// =========================================================================================================

static uint64_t shardCapLocal(uint32_t cellCount, uint32_t world)
{
    const uint64_t forced = envNumber("EM2_INBOX_CAPACITY", 0);
    if (forced >= kInboxChunk) return forced;
    uint64_t cap = uint64_t(cellCount) * 1024u / world;
    cap += cap / 4u;
    const uint64_t floor = uint64_t(maxResidentWaves()) * kInboxChunk * 2u;
    if (cap < floor) cap = floor;
    if (cap > 0xfff00000ull) cap = 0xfff00000ull;
    return cap;
}

Fsp4ShardPlan fsp4ShardPlan(uint32_t cellCount, uint32_t k, uint32_t rank, uint32_t world)
{
    Fsp4ShardPlan p;
    memset(&p, 0, sizeof(p));
    p.cellCount = cellCount;
    p.world = world;
    p.rank = rank;
    p.k = k;
    p.blocks = (cellCount + 63u) / 64u;
    if (world == 0 || rank >= world || k == 0 || p.blocks < 4u * world || cellCount > (1u << 25)) return p;     // not eligible
    // prefix: EM2_PREFIX_PERMILLE of the cells (default 200), a positive multiple of `world` blocks
    uint64_t prefixBlocks = (uint64_t(p.blocks) * envNumber("EM2_PREFIX_PERMILLE", 200) / 1000u + world / 2u) / world * world;
    if (prefixBlocks < world) prefixBlocks = world;
    if (prefixBlocks > uint64_t(p.blocks) - world) prefixBlocks = (uint64_t(p.blocks) - world) / world * world;
    {
        // a multiple of 4 blocks (256 cells) as well where that fits: the matrix-core tile kernel wants it
        uint64_t unit = world;
        while (unit % 4u) unit += world;
        uint64_t rounded = (prefixBlocks + unit / 2u) / unit * unit;
        if (rounded < unit) rounded = unit;
        while (rounded > unit && rounded > uint64_t(p.blocks) - world) rounded -= unit;
        if (rounded <= uint64_t(p.blocks) - world) prefixBlocks = rounded;
    }
    p.prefixBlocks = uint32_t(prefixBlocks);
    p.prefixCells = p.prefixBlocks * 64u;
    p.ownBlocks = (p.blocks - rank + world - 1u) / world;
    p.maxOwnBlocks = (p.blocks + world - 1u) / world;
    p.ownPrefixBlocks = p.prefixBlocks / world;
    p.capLocal = shardCapLocal(cellCount, world);
    p.capGathered = p.capLocal * world;
    p.sortTempBytes = inboxSortTempBytes(p.capGathered);
    size_t at = 0;
    p.offLists = at;        at += align256(size_t(p.maxOwnBlocks) * 64u * 2u * k * sizeof(Entry));
    p.offControl = at;      at += align256(fsp4ControlBytes(p.maxOwnBlocks * 64u));
    p.offSnap = at;         at += align256(size_t(cellCount) * 4u);
    p.offTable = at;        at += align256(kTableWords * 4u);
    p.offInboxControl = at; at += 256u;
    p.offPool = at;         at += align256(size_t(p.capLocal) * 8u);
    p.offFragments = at;    at += align256(size_t(p.blocks) * 64u * 1024u);     // FP4 fragments of up to 2048 bits (matrix-core kernels)
    p.rankBytes = at;
    p.offGathered = at;     at += align256(size_t(p.capGathered) * 8u);
    p.offSorted = at;       at += align256(size_t(p.capGathered) * 8u);
    p.offTemp = at;         at += align256(p.sortTempBytes);
    p.totalBytes = at;
    p.eligible = true;
    return p;
}

static const void* symmetricKernelFor(uint32_t paddedDw, bool identity)
{
#define EM2_SYMMETRIC(W32) \
    (identity ? reinterpret_cast<const void*>(&fsp4ScanSymmetricKernel<W32, true>) \
              : reinterpret_cast<const void*>(&fsp4ScanSymmetricKernel<W32, false>))
    switch (paddedDw) {
    case 2: return EM2_SYMMETRIC(2);
    case 4: return EM2_SYMMETRIC(4);
    case 8: return EM2_SYMMETRIC(8);
    case 16: return EM2_SYMMETRIC(16);
    case 32: return EM2_SYMMETRIC(32);
    case 64: return EM2_SYMMETRIC(64);
    case 128: return EM2_SYMMETRIC(128);
    default: return nullptr;
    }
#undef EM2_SYMMETRIC
}

static const void* tileKernelFor(uint32_t paddedDw)
{
    switch (paddedDw) {
    case 2: return reinterpret_cast<const void*>(&fsp4TileKernel<2>);
    case 4: return reinterpret_cast<const void*>(&fsp4TileKernel<4>);
    case 8: return reinterpret_cast<const void*>(&fsp4TileKernel<8>);
    case 16: return reinterpret_cast<const void*>(&fsp4TileKernel<16>);
    case 32: return reinterpret_cast<const void*>(&fsp4TileKernel<32>);
    case 64: return reinterpret_cast<const void*>(&fsp4TileKernel<64>);
    case 128: return reinterpret_cast<const void*>(&fsp4TileKernel<128>);
    default: return nullptr;
    }
}

// rankWs = the rank part of the workspace (plan.rankBytes), exchangeWs = gathered / sorted / temp areas (in the real
// multi-GPU run both are one allocation: exchangeWs = rankWs; the emulation shares one exchange area).
// gatheredCount: phase 3 only, entries in the gathered area.  outPairs / outUsed are indexed by GLOBAL cell id.
hipError_t launchFsp4ShardPhase(const Fsp4ShardPlan& plan, int phase, const uint32_t* sig32, uint32_t paddedDw,
                                const DeviceTables& t, void* rankWs, void* exchangeWs, PairOut* outPairs, uint32_t* outUsed,
                                uint64_t gatheredCount, hipStream_t stream)
{
    if (!plan.eligible) return hipErrorInvalidValue;
    const uint32_t k = plan.k;
    if (k == 0 || k > fsp4MaxK()) return hipErrorInvalidValue;
    const uint32_t bytesPerWave = 2u * k * kLdsBytesPerEntrySlot;
    uint32_t wavesPerBlock = kLdsBytesPerBlock / bytesPerWave;
    if (wavesPerBlock > 4) wavesPerBlock = 4;
    const size_t lds = size_t(wavesPerBlock) * bytesPerWave;
    const dim3 block(64u * wavesPerBlock);
    char* ws = static_cast<char*>(rankWs);
    char* xs = static_cast<char*>(exchangeWs);
    const uint32_t cellCount = plan.cellCount;
    const uint32_t M = plan.prefixCells;

    Fsp4Args args;
    memset(&args, 0, sizeof(args));
    args.sig32 = sig32;
    args.cellCount = cellCount;
    args.mMaxInitial = t.mMaxInitial;
    args.keyOfMismatch = t.keyOfMismatch;
    args.acceptMaxByKey = t.acceptMaxByKey;
    args.keySimilarity = t.keySimilarity;
    args.buffers = reinterpret_cast<Entry*>(ws + plan.offLists);
    args.outPairs = outPairs;
    args.outUsed = outUsed;
    args.k = k;
    args.rowBegin = 0;
    args.rowEnd = cellCount;
    char* c = ws + plan.offControl;
    const size_t stateBytes = align256(size_t((plan.maxOwnBlocks * 64u + 63u) / 64u) * 64u * 8u);
    const size_t doneBytes = align256(size_t((plan.maxOwnBlocks * 64u + 63u) / 64u) * 4u);
    args.rowState = reinterpret_cast<uint32_t*>(c);
    args.segmentsDone = reinterpret_cast<uint32_t*>(c + stateBytes);
    args.control = reinterpret_cast<uint32_t*>(c + stateBytes + doneBytes);
    args.logs = reinterpret_cast<Entry*>(c + stateBytes + doneBytes + 256u);
    args.logCapacity = kLogCapacity;
    if (const char* v = getenv("EM2_LOG_CAPACITY")) {
        if (atoi(v) >= 1 && uint32_t(atoi(v)) < kLogCapacity) args.logCapacity = uint32_t(atoi(v));
    }
    args.snap = reinterpret_cast<int32_t*>(ws + plan.offSnap);
    args.inbox = reinterpret_cast<uint64_t*>(ws + plan.offPool);
    args.inboxControl = reinterpret_cast<uint32_t*>(ws + plan.offInboxControl);
    args.segTable = reinterpret_cast<const uint32_t*>(ws + plan.offTable);
    args.inboxCapacity = plan.capLocal;
    args.inboxChunk = kInboxChunk;
    uint32_t rowBits = 1;
    while ((1ull << rowBits) < uint64_t(cellCount)) ++rowBits;
    args.rowBits = rowBits;
    args.rowBlockStride = plan.world;
    args.rowBlockOffset = plan.rank;
    args.columnLimit = M;
    args.shardFlags = kShardNoFinish | kShardPublishAll | kShardGlobalOutput;

    hipError_t e = hipSuccess;
    if (phase == 0) {
        lastLaunchInfo.matrixPairs = 0.0;
        lastLaunchInfo.matrixKernelMs = -1.0;
        lastLaunchInfo.form = 2;
        lastLaunchInfo.scanKernelMs = -1.0;
        lastLaunchInfo.waveColumnSteps = 0.0;
        lastLaunchInfo.inboxEntries = 0.0;
        lastLaunchInfo.segments = 0.0;
        lastLaunchInfo.fullRowCells = double(M);
    }
    if (phase == 0 || phase == 1) {
        if (phase == 0) {
            e = hipMemsetAsync(args.snap, 0x80, size_t(cellCount) * 4u, stream);        // below every real cut-off
            if (e != hipSuccess) return e;
            e = hipMemsetAsync(args.inboxControl, 0, 256u, stream);
            if (e != hipSuccess) return e;
        }
        const uint32_t slotBase = phase == 0 ? 0u : plan.ownPrefixBlocks;
        const uint32_t slotCount = phase == 0 ? plan.ownPrefixBlocks : plan.ownBlocks - plan.ownPrefixBlocks;
        if (slotCount == 0) return hipSuccess;
        const size_t matrixLdsOffset = (lds + 15u) & ~size_t(15);
        const size_t matrixLds = matrixLdsOffset + scanMatrixLdsBytes(args.k);
        const bool wide = matrixWideWanted(paddedDw);
        if ((paddedDw == 32u || wide) && wavesPerBlock == 4u && M % 256u == 0u && matrixLds <= 150u * 1024u &&
            envNumber("EM2_SCAN_MATRIX", 1) != 0) {
            // Phase 1, the rows beyond the prefix against the prefix columns: all of it below the rows, so all of it for the
            // matrix cores (fsp4ScanMatrixKernel over quads of slots; no quad ever reaches its own columns here).  Phase 0,
            // the prefix rows against the prefix columns from both sides: the same kernel's full-row items.
            uint64_t segments = M / 16384u;          // long segments: an item starts with 32 KB of row fragments per wave
            if (segments > kMatrixMaxSegments) segments = kMatrixMaxSegments;
            if (segments < 1) segments = 1;
            uint32_t cps = uint32_t((uint64_t(M) + segments - 1u) / segments);
            cps = (cps + 255u) & ~255u;
            segments = (uint64_t(M) + cps - 1u) / cps;
            const uint32_t quads = (slotCount + 3u) / 4u;
            uint32_t table[kTableWords];
            for (uint32_t sIdx = 0; sIdx < segments; ++sIdx) {
                table[sIdx] = sIdx * quads;
                table[segments + 1u + sIdx] = slotBase;
            }
            const uint64_t tickets = segments * quads;
            if (tickets >= 0xffffffffull) return hipErrorInvalidValue;
            table[segments] = uint32_t(tickets);
            args.segments = uint32_t(segments);
            args.columnsPerSegment = cps;
            args.localBlockBase = slotBase;
            args.rowBlocks = slotBase + slotCount;
            args.fullRowBlocks = phase == 0 ? slotCount : 0u;
            args.totalTickets = uint32_t(tickets);
            e = hipMemsetAsync(c + stateBytes, 0, doneBytes + (phase == 0 ? 256u : 4u), stream);
            if (e != hipSuccess) return e;
            e = hipMemcpyAsync(ws + plan.offTable, table, (2u * segments + 2u) * 4u, hipMemcpyHostToDevice, stream);
            if (e != hipSuccess) return e;
            const uint32_t matrixSteps = wide ? 2u * kMatrixSteps : kMatrixSteps;
            const uint32_t fragmentCount = plan.blocks * 2u * matrixSteps * 64u;
            expandFragmentsKernel<<<dim3((fragmentCount + 255u) / 256u), dim3(256), 0, stream>>>(
                sig32, cellCount, fragmentCount, reinterpret_cast<FragmentWord4*>(ws + plan.offFragments), matrixSteps);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
            args.fragments = ws + plan.offFragments;
            args.matrixLdsOffset = uint32_t(matrixLdsOffset);
            lastLaunchInfo.matrixPairs += double(slotCount) * 64.0 * double(M);
            const void* matrixKernel = scanMatrixKernelFor(t.identityKeys, wide);
            int device = 0, cuCount = 0;
            e = hipGetDevice(&device);
            if (e != hipSuccess) return e;
            e = hipDeviceGetAttribute(&cuCount, hipDeviceAttributeMultiprocessorCount, device);
            if (e != hipSuccess) return e;
            uint64_t blocksWanted = uint64_t(cuCount) * 2u;
            if (const char* v = getenv("EM2_BLOCKS_PER_CU")) {
                if (atoi(v) == 1) blocksWanted = uint64_t(cuCount);
            }
            if (blocksWanted * 4u > maxResidentWaves()) blocksWanted = maxResidentWaves() / 4u;
            if (blocksWanted > tickets) blocksWanted = tickets;
            e = hipFuncSetAttribute(matrixKernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(matrixLds));
            if (e != hipSuccess) return e;
            void* matrixArgsArray[] = {&args};
            return hipLaunchKernel(matrixKernel, dim3(uint32_t(blocksWanted)), dim3(256), matrixArgsArray, matrixLds, stream);
        }
        const void* kernel = symmetricKernelFor(paddedDw, t.identityKeys);
        if (!kernel) return hipErrorInvalidValue;
        uint32_t slots = 0;
        e = residentWaveSlots(kernel, wavesPerBlock, lds, &slots);
        if (e != hipSuccess) return e;
        uint64_t minSegmentColumns = envNumber("EM2_MIN_SEGMENT_COLUMNS", 4096);
        if (minSegmentColumns < 1) minSegmentColumns = 1;
        // enough (segment, slot) items for an even finish (~32 per resident wave), at most kMaxSegments
        uint64_t segments = (32ull * slots + slotCount - 1u) / slotCount;
        if (segments > M / minSegmentColumns) segments = M / minSegmentColumns;
        if (segments > kMaxSegments) segments = kMaxSegments;
        if (segments < 1) segments = 1;
        const uint32_t cps = uint32_t((uint64_t(M) + segments - 1u) / segments);
        segments = (uint64_t(M) + cps - 1u) / cps;
        uint32_t table[2u * kMaxSegments + 2u];
        for (uint32_t sIdx = 0; sIdx < segments; ++sIdx) {
            table[sIdx] = sIdx * slotCount;
            table[segments + 1u + sIdx] = 0u;
        }
        const uint64_t tickets = segments * slotCount;
        if (tickets >= 0xffffffffull) return hipErrorInvalidValue;
        table[segments] = uint32_t(tickets);
        args.segments = uint32_t(segments);
        args.columnsPerSegment = cps;
        args.localBlockBase = slotBase;
        args.rowBlocks = slotBase + slotCount;
        args.fullRowBlocks = phase == 0 ? slotCount : 0u;
        args.totalTickets = uint32_t(tickets);
        // hand-off flags and the ticket counter start at zero; the error word survives from phase 0 to phase 1
        e = hipMemsetAsync(c + stateBytes, 0, doneBytes + (phase == 0 ? 256u : 4u), stream);
        if (e != hipSuccess) return e;
        e = hipMemcpyAsync(ws + plan.offTable, table, (2u * segments + 2u) * 4u, hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) return e;
        lastLaunchInfo.waveColumnSteps += double(slotCount) * double(M);
        uint64_t wavesWanted = tickets;
        if (wavesWanted > slots) wavesWanted = slots;
        if (wavesWanted > maxResidentWaves()) wavesWanted = maxResidentWaves();
        const dim3 grid(uint32_t((wavesWanted + wavesPerBlock - 1u) / wavesPerBlock));
        void* kernelArgsArray[] = {&args};
        return hipLaunchKernel(kernel, grid, block, kernelArgsArray, lds, stream);
    }
    if (phase == 2) {
        const void* kernel = tileKernelFor(paddedDw);
        if (!kernel) return hipErrorInvalidValue;
        // 1024-bit signatures and a prefix of whole quads: the tiles go to the matrix cores (EM2_SCAN_MATRIX=0: never)
        const bool wide = matrixWideWanted(paddedDw);
        const bool matrix = (paddedDw == 32u || wide) && M % 256u == 0u && envNumber("EM2_SCAN_MATRIX", 1) != 0;
        const uint32_t span = cellCount - M;
        uint64_t segments = span / (matrix ? 16384u : 1024u);
        if (segments > 256) segments = 256;
        const uint64_t forced = envNumber("EM2_TILE_SEGMENTS", 0);
        if (forced >= 1 && forced <= 256) segments = forced;
        if (segments < 1) segments = 1;
        uint32_t cps = uint32_t((uint64_t(span) + segments - 1u) / segments);
        if (matrix) cps = (cps + 255u) & ~255u;
        segments = (uint64_t(span) + cps - 1u) / cps;
        uint32_t table[2u * 256u + 2u];
        uint64_t tiles = 0;
        for (uint32_t sIdx = 0; sIdx < segments; ++sIdx) {
            const uint32_t firstBlock = (M + sIdx * cps) / 64u;
            table[sIdx] = uint32_t(tiles);
            table[segments + 1u + sIdx] = firstBlock;
            tiles += matrix ? (plan.blocks - firstBlock + 3u) / 4u : plan.blocks - firstBlock;
            if (tiles >= 0xffffffffull) return hipErrorInvalidValue;
        }
        table[segments] = uint32_t(tiles);
        const uint64_t own = tiles > plan.rank ? (tiles - plan.rank + plan.world - 1u) / plan.world : 0u;
        if (own == 0) return hipSuccess;
        {
            double steps = 0.0, matrixPairs = 0.0;      // this rank's share of the tiles' work
            for (uint32_t b = plan.prefixBlocks; b < plan.blocks; ++b) {
                const uint64_t end = uint64_t(b) * 64u + 64u;
                const uint64_t from = matrix ? uint64_t(b & ~3u) * 64u : M;     // v_xor/v_bcnt: the quad's own columns only
                steps += double((end < cellCount ? end : cellCount) - from);
                matrixPairs += 64.0 * double(from - M);
            }
            lastLaunchInfo.waveColumnSteps += steps / double(plan.world);
            lastLaunchInfo.matrixPairs += matrixPairs / double(plan.world);
        }
        args.segments = uint32_t(segments);
        args.columnsPerSegment = cps;
        args.rowBlocks = plan.blocks;
        args.totalTickets = uint32_t(own);
        e = hipMemsetAsync(c + stateBytes + doneBytes, 0, 4u, stream);         // ticket counter (the error word stays)
        if (e != hipSuccess) return e;
        e = hipMemcpyAsync(ws + plan.offTable, table, (2u * segments + 2u) * 4u, hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) return e;
        int device = 0, cuCount = 0;
        e = hipGetDevice(&device);
        if (e != hipSuccess) return e;
        e = hipDeviceGetAttribute(&cuCount, hipDeviceAttributeMultiprocessorCount, device);
        if (e != hipSuccess) return e;
        if (matrix) {
            const uint32_t matrixSteps = wide ? 2u * kMatrixSteps : kMatrixSteps;
            const uint32_t fragmentCount = plan.blocks * 2u * matrixSteps * 64u;
            expandFragmentsKernel<<<dim3((fragmentCount + 255u) / 256u), dim3(256), 0, stream>>>(
                sig32, cellCount, fragmentCount, reinterpret_cast<FragmentWord4*>(ws + plan.offFragments), matrixSteps);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
            args.fragments = ws + plan.offFragments;
            args.matrixLdsOffset = 0u;
            const size_t matrixLds = args.matrixLdsOffset + kMatrixLdsBytes;
            uint64_t blocksWanted = uint64_t(cuCount) * 2u;
            if (blocksWanted > own) blocksWanted = own;
            const void* tileMatrixKernel = wide ? reinterpret_cast<const void*>(&fsp4TileMatrixWideKernel)
                                                : (matrixWalkPinned(2u) ? reinterpret_cast<const void*>(&fsp4TileMatrixPinnedKernel)
                                                                        : reinterpret_cast<const void*>(&fsp4TileMatrixKernel));
            e = hipFuncSetAttribute(tileMatrixKernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(matrixLds));
            if (e != hipSuccess) return e;
            void* matrixArgsArray[] = {&args};
            return hipLaunchKernel(tileMatrixKernel, dim3(uint32_t(blocksWanted)), dim3(256), matrixArgsArray, matrixLds, stream);
        }
        uint64_t wavesWanted = own;
        const uint64_t resident = uint64_t(cuCount) * 16u;
        if (wavesWanted > resident) wavesWanted = resident;
        const dim3 tileBlock(256);
        const dim3 grid(uint32_t((wavesWanted + 3u) / 4u));
        void* kernelArgsArray[] = {&args};
        return hipLaunchKernel(kernel, grid, tileBlock, kernelArgsArray, 0, stream);
    }
    if (phase == 4) {
        // Groups this rank's pool entries by the rank that owns their target cell (block-cyclic: owner = (target / 64)
        // % world, a bit field of the key when world is a power of two), for an all_to_all instead of the all_gather:
        // a stable one-digit radix sort of pool[0, gatheredCount) into the sorted area.
        if (gatheredCount > plan.capLocal || (plan.world & (plan.world - 1u)) != 0u) return hipErrorInvalidValue;
        if (gatheredCount == 0 || plan.world == 1) {
            if (gatheredCount) {
                e = hipMemcpyAsync(xs + plan.offSorted - plan.rankBytes, ws + plan.offPool, size_t(gatheredCount) * 8u, hipMemcpyDeviceToDevice, stream);
            }
            return e;
        }
        uint32_t ownerBits = 0;
        while ((1u << ownerBits) < plan.world) ++ownerBits;
        const uint32_t ownerShift = 13u + rowBits + 6u;
        size_t tempBytes = plan.sortTempBytes;
        return rocprim::radix_sort_keys(xs + plan.offTemp - plan.rankBytes, tempBytes, reinterpret_cast<uint64_t*>(ws + plan.offPool),
                                        reinterpret_cast<uint64_t*>(xs + plan.offSorted - plan.rankBytes), size_t(gatheredCount),
                                        ownerShift, ownerShift + ownerBits, stream);
    }
    if (phase == 3) {
        if (gatheredCount > plan.capGathered) return hipErrorInvalidValue;
        lastLaunchInfo.inboxEntries = double(gatheredCount);
        const uint64_t* sorted = reinterpret_cast<const uint64_t*>(xs + plan.offGathered - plan.rankBytes);
        if (gatheredCount) {
            size_t tempBytes = plan.sortTempBytes;
            uint64_t* in = reinterpret_cast<uint64_t*>(xs + plan.offGathered - plan.rankBytes);
            uint64_t* out = reinterpret_cast<uint64_t*>(xs + plan.offSorted - plan.rankBytes);
            e = rocprim::radix_sort_keys(xs + plan.offTemp - plan.rankBytes, tempBytes, in, out, size_t(gatheredCount), 13u,
                                         13u + 2u * rowBits, stream);
            if (e != hipSuccess) return e;
            sorted = out;
        }
        args.localBlockBase = 0;
        args.fullRowBlocks = 0;
        args.rowBlocks = plan.ownBlocks;
        args.shardFlags = kShardGlobalOutput;
        if (plan.ownBlocks == 0) return hipSuccess;
        const dim3 rgrid((plan.ownBlocks + wavesPerBlock - 1u) / wavesPerBlock);
        if (t.identityKeys) fsp4InboxReplayKernel<true><<<rgrid, block, lds, stream>>>(args, sorted, gatheredCount);
        else fsp4InboxReplayKernel<false><<<rgrid, block, lds, stream>>>(args, sorted, gatheredCount);
        return hipGetLastError();
    }
    return hipErrorInvalidValue;
}

// Reads a rank's entry count and flags after phase 2 (synchronises): used (entries incl. chunk padding),
// overflow (pool too small: the caller must fall back to the ordered scan), error (a hand-off timed out).
hipError_t readFsp4ShardStatus(const Fsp4ShardPlan& plan, const void* rankWs, hipStream_t stream, uint64_t* used,
                               uint32_t* overflow, uint32_t* error)
{
    const char* ws = static_cast<const char*>(rankWs);
    uint32_t inboxWords[4] = {0, 0, 0, 0};
    uint32_t controlWords[2] = {0, 0};
    const size_t stateBytes = align256(size_t(plan.maxOwnBlocks) * 64u * 8u);
    const size_t doneBytes = align256(size_t(plan.maxOwnBlocks) * 4u);
    hipError_t e = hipMemcpyAsync(inboxWords, ws + plan.offInboxControl, sizeof(inboxWords), hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(controlWords, ws + plan.offControl + stateBytes + doneBytes, sizeof(controlWords), hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    *used = uint64_t(inboxWords[0]) | (uint64_t(inboxWords[1]) << 32);
    *overflow = (inboxWords[2] != 0u || *used > plan.capLocal) ? 1u : 0u;
    *error = controlWords[1];
    return hipSuccess;
}

// sorted[0, n) is grouped by owner = (key >> shift) & (world - 1), ascending; bounds[r] = first index whose owner >= r,
// bounds[world] = n (the emulation's copy of the kernel of csrc/em2_dist.hip).
__global__ void emulationOwnerBoundsKernel(const uint64_t* __restrict__ sorted, uint64_t n, uint32_t shift, uint32_t world,
                                           uint64_t* __restrict__ bounds)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > world) return;
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo) / 2u;
        if (uint32_t((sorted[mid] >> shift) & uint64_t(world - 1u)) < r) lo = mid + 1u;
        else hi = mid;
    }
    bounds[r] = lo;
}

// All ranks of the sharded scan played one after the other on this GPU (tests; EM2_SCAN_MODE=virtual with
// EM2_VIRTUAL_WORLD=P).  *done = false: not eligible or an entry pool overflowed; the caller runs the ordered scan.
hipError_t runFsp4ShardedEmulation(const uint32_t* sig32, uint32_t paddedDw, uint32_t cellCount, uint32_t k,
                                          const DeviceTables& t, PairOut* outPairs, uint32_t* outUsed, uint32_t world,
                                          hipStream_t stream, bool* done)
{
    *done = false;
    std::vector<Fsp4ShardPlan> plans;
    for (uint32_t r = 0; r < world; ++r) plans.push_back(fsp4ShardPlan(cellCount, k, r, world));
    if (!plans[0].eligible) return hipSuccess;
    const Fsp4ShardPlan& p0 = plans[0];
    const bool verbose = getenv("EM2_SCAN_VERBOSE") && getenv("EM2_SCAN_VERBOSE")[0] == '1';
    // rank parts back to back, except that the snap arrays are laid out contiguously ([world][cellCount]) at the
    // end so that one kernel can play all_reduce(MAX)
    char* base = nullptr;
    const size_t exchangeBytes = p0.totalBytes - p0.rankBytes;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&base), p0.rankBytes * world + exchangeBytes);
    if (e != hipSuccess) return e;
    struct Free { char* p; ~Free() { (void)hipFree(p); } } guard{base};
    char* exchange = base + p0.rankBytes * world;
    std::vector<hipEvent_t> events;
    auto mark = [&]() { hipEvent_t ev; (void)hipEventCreate(&ev); (void)hipEventRecord(ev, stream); events.push_back(ev); };
    auto reduceSnap = [&]() -> hipError_t {
        // gather the ranks' snap arrays, reduce, scatter back (the emulation's all_reduce)
        int32_t* tmp = reinterpret_cast<int32_t*>(exchange);      // the exchange area is free at this point
        for (uint32_t r = 0; r < world; ++r) {
            hipError_t ee = hipMemcpyAsync(tmp + size_t(r) * cellCount, base + p0.rankBytes * r + p0.offSnap, size_t(cellCount) * 4u,
                                           hipMemcpyDeviceToDevice, stream);
            if (ee != hipSuccess) return ee;
        }
        maxReduceKernel<<<dim3((cellCount + 255u) / 256u), dim3(256), 0, stream>>>(tmp, cellCount, world);
        for (uint32_t r = 0; r < world; ++r) {
            hipError_t ee = hipMemcpyAsync(base + p0.rankBytes * r + p0.offSnap, tmp + size_t(r) * cellCount, size_t(cellCount) * 4u,
                                           hipMemcpyDeviceToDevice, stream);
            if (ee != hipSuccess) return ee;
        }
        return hipGetLastError();
    };
    if (size_t(cellCount) * 4u * world > exchangeBytes) return hipSuccess;      // cannot happen with sane capacities
    for (int phase = 0; phase < 3; ++phase) {
        for (uint32_t r = 0; r < world; ++r) {
            mark();
            e = launchFsp4ShardPhase(plans[r], phase, sig32, paddedDw, t, base + p0.rankBytes * r, exchange, outPairs, outUsed, 0, stream);
            if (e != hipSuccess) return e;
        }
        mark();
        if (phase < 2) {
            e = reduceSnap();
            if (e != hipSuccess) return e;
        }
    }
    // all_gather of the pools: each rank's used entries, padded with sentinels to the common maximum
    std::vector<uint64_t> used(world, 0);
    uint64_t maxUsed = 0;
    for (uint32_t r = 0; r < world; ++r) {
        uint32_t overflow = 0, error = 0;
        e = readFsp4ShardStatus(plans[r], base + p0.rankBytes * r, stream, &used[r], &overflow, &error);
        if (e != hipSuccess) return e;
        if (overflow || error) return hipSuccess;      // *done stays false
        if (used[r] > maxUsed) maxUsed = used[r];
    }
    uint64_t* gathered = reinterpret_cast<uint64_t*>(exchange + p0.offGathered - p0.rankBytes);
    // The exchange of the deferred candidates, as the product does it (csrc/em2_dist.hip, expressionmatrix2_amd/sharded.py): with a
    // power-of-two world every rank groups its pool by the owner of the target cell (phase 4) and the groups travel by
    // all_to_all -- a rank receives, sorts and replays only the candidates of its own cells; otherwise (or with
    // EM2_SHARDED_EXCHANGE=gather) every rank gathers every pool.  The routed form is played with device-to-device copies
    // through one staging area per receiver; its grouping sort is timed as phase 4.
    const bool routed = (world & (world - 1u)) == 0u && world > 1u &&
                        !(getenv("EM2_SHARDED_EXCHANGE") && getenv("EM2_SHARDED_EXCHANGE")[0] == 'g');
    std::vector<uint64_t> receivedEntries(world, 0);
    std::vector<size_t> phase4Events;
    char* staging = nullptr;
    struct FreeStaging { char*& p; ~FreeStaging() { if (p) (void)hipFree(p); } } stagingGuard{staging};
    if (routed) {
        uint64_t total = 0;
        for (uint32_t r = 0; r < world; ++r) total += used[r];
        e = hipMalloc(reinterpret_cast<void**>(&staging), std::max<size_t>(size_t(total) * 8u + (world + 1u) * 8u, 64));
        if (e != hipSuccess) return e;
        uint64_t* bounds = reinterpret_cast<uint64_t*>(staging + size_t(total) * 8u);
        uint32_t rowBits = 1, ownerBits = 0;
        while ((1ull << rowBits) < uint64_t(cellCount)) ++rowBits;
        while ((1u << ownerBits) < world) ++ownerBits;
        const uint32_t ownerShift = 13u + rowBits + 6u;
        const uint64_t* sortedPool = reinterpret_cast<const uint64_t*>(exchange + p0.offSorted - p0.rankBytes);
        // first pass: the counts matrix (what the ranks learn from the small all_gather); second pass: the copies
        std::vector<std::vector<uint64_t>> counts(world, std::vector<uint64_t>(world, 0));
        std::vector<std::vector<uint64_t>> starts(world, std::vector<uint64_t>(world + 1u, 0));
        for (int pass = 0; pass < 2; ++pass) {
            std::vector<uint64_t> receiverBase(world, 0), receiverFill(world, 0);
            if (pass == 1) {
                uint64_t at = 0;
                for (uint32_t q = 0; q < world; ++q) {
                    receiverBase[q] = at;
                    for (uint32_t r = 0; r < world; ++r) receivedEntries[q] += counts[r][q];
                    at += receivedEntries[q];
                }
            }
            for (uint32_t r = 0; r < world; ++r) {
                if (pass == 0) {
                    phase4Events.push_back(events.size());
                    mark();
                }
                e = launchFsp4ShardPhase(plans[r], 4, sig32, paddedDw, t, base + p0.rankBytes * r, exchange, outPairs, outUsed, used[r], stream);
                if (e != hipSuccess) return e;
                if (pass == 0) {
                    mark();
                    emulationOwnerBoundsKernel<<<dim3(1), dim3(256), 0, stream>>>(sortedPool, used[r], ownerShift, world, bounds);
                    e = hipMemcpyAsync(starts[r].data(), bounds, (world + 1u) * 8u, hipMemcpyDeviceToHost, stream);
                    if (e != hipSuccess) return e;
                    e = hipStreamSynchronize(stream);
                    if (e != hipSuccess) return e;
                    for (uint32_t q = 0; q < world; ++q) counts[r][q] = starts[r][q + 1u] - starts[r][q];
                } else {
                    for (uint32_t q = 0; q < world; ++q) {
                        if (!counts[r][q]) continue;
                        e = hipMemcpyAsync(reinterpret_cast<uint64_t*>(staging) + receiverBase[q] + receiverFill[q], sortedPool + starts[r][q],
                                           size_t(counts[r][q]) * 8u, hipMemcpyDeviceToDevice, stream);
                        if (e != hipSuccess) return e;
                        receiverFill[q] += counts[r][q];
                    }
                }
            }
            if (pass == 1) {
                for (uint32_t q = 0; q < world; ++q) {
                    if (receivedEntries[q] > plans[q].capGathered) return hipSuccess;          // *done stays false: the callers fall back
                }
                for (uint32_t q = 0; q < world; ++q) {
                    if (receivedEntries[q]) {
                        e = hipMemcpyAsync(gathered, reinterpret_cast<uint64_t*>(staging) + receiverBase[q], size_t(receivedEntries[q]) * 8u,
                                           hipMemcpyDeviceToDevice, stream);
                        if (e != hipSuccess) return e;
                    }
                    mark();
                    e = launchFsp4ShardPhase(plans[q], 3, sig32, paddedDw, t, base + p0.rankBytes * q, exchange, outPairs, outUsed,
                                             receivedEntries[q], stream);
                    if (e != hipSuccess) return e;
                }
            }
        }
    } else {
        e = hipMemsetAsync(gathered, 0xff, size_t(maxUsed) * world * 8u, stream);
        if (e != hipSuccess) return e;
        for (uint32_t r = 0; r < world; ++r) {
            if (!used[r]) continue;
            e = hipMemcpyAsync(gathered + size_t(r) * maxUsed, base + p0.rankBytes * r + p0.offPool, size_t(used[r]) * 8u,
                               hipMemcpyDeviceToDevice, stream);
            if (e != hipSuccess) return e;
        }
        for (uint32_t r = 0; r < world; ++r) {
            mark();
            e = launchFsp4ShardPhase(plans[r], 3, sig32, paddedDw, t, base + p0.rankBytes * r, exchange, outPairs, outUsed,
                                     maxUsed * world, stream);
            if (e != hipSuccess) return e;
            receivedEntries[r] = maxUsed * world;
        }
    }
    mark();
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    if (verbose) {
        fprintf(stderr, "[em2] sharded emulation: world %u, prefix %u cells, entries per rank (max) %llu;", world, p0.prefixCells,
                (unsigned long long)maxUsed);
        // events: phases 0..2 are (world starts + one end) each; then, routed, a (start, end) pair per rank for the grouping
        // sort; then the world starts + one end of phase 3
        size_t at = 0;
        for (int phase = 0; phase < 3; ++phase) {
            fprintf(stderr, " phase %d ms:", phase);
            for (uint32_t r = 0; r < world; ++r) {
                float ms = 0;
                (void)hipEventElapsedTime(&ms, events[at], events[at + 1]);
                fprintf(stderr, " %.2f", ms);
                ++at;
            }
            ++at;
        }
        if (routed) {
            fprintf(stderr, " grouping by owner ms:");
            for (size_t first : phase4Events) {
                float ms = 0;
                (void)hipEventElapsedTime(&ms, events[first], events[first + 1]);
                fprintf(stderr, " %.2f", ms);
            }
            at += 2u * phase4Events.size();
        }
        fprintf(stderr, " phase 3 (%s, entries received", routed ? "all_to_all" : "all_gather");
        for (uint32_t r = 0; r < world; ++r) fprintf(stderr, " %llu", (unsigned long long)receivedEntries[r]);
        fprintf(stderr, ") ms:");
        for (uint32_t r = 0; r < world; ++r) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, events[at], events[at + 1]);
            fprintf(stderr, " %.2f", ms);
            ++at;
        }
        fprintf(stderr, "\n");
    }
    for (hipEvent_t ev : events) (void)hipEventDestroy(ev);
    *done = true;
    return hipSuccess;
}

}  // namespace em2
