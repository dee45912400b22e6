// em2_scan_symmetric_device.h -- device code and constants shared by the two translation units of the symmetric scan:
// em2_scan_symmetric.hip (the one-GPU symmetric scan: fsp4ScanSymmetricKernel, the matrix-core kernels, the inbox replay and
// their launcher) and em2_scan_sharded.hip (the sharded symmetric scan across GPUs: tile kernels, phases, the emulation).
// Here: the inbox emission helpers, the v_xor/v_bcnt column loops of the symmetric forms, and the matrix-core walks
// (scanTilesMatrix, the hand-scheduled scanTilesMatrixPinned / scanTilesMatrixWide of em2_matrix_step_asm.h, their logs).
// Everything is internal to the library (anonymous namespace: each unit compiles its own copy).
#ifndef EM2_SCAN_SYMMETRIC_DEVICE_H
#define EM2_SCAN_SYMMETRIC_DEVICE_H

#include "em2_scan_common.h"
#include "em2_matrix_step_asm.h"

#include <vector>

#include <rocprim/rocprim.hpp>

namespace em2 {
namespace {

typedef const __attribute__((address_space(4))) int32_t* ScalarIntPtr;

// Returns the new chunk as pos | end << 32; pos > end (1, 0) = emission disabled after an overflow.
__device__ __attribute__((noinline)) uint64_t refillInboxChunk(uint64_t* inbox, uint32_t* control, uint64_t capacity,
                                                               uint32_t chunk, uint32_t lane, uint32_t pos, uint32_t end)
{
    for (uint32_t i = pos + lane; i < end; i += 64u) storeGlobalWord(inbox + i, ~0ull);      // sentinels sort to the end
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
        storeGlobalWord(aux->inbox + p + lanesBelow(mask), (uint64_t(col) << (13u + nb)) | (uint64_t(row) << 13u) | uint64_t(m));
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
                                    storeGlobalWord(aux->inbox + at + lanesBelow(emitMask),
                                                    (uint64_t(col) << (13u + aux->rowBits)) | (uint64_t(row) << 13u) | uint64_t(m));
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


// =========================================================================================================
// Matrix-core form of the triangle part (1024-bit signatures).  A signature bit becomes the FP4 (E2M1) value 0 or 1
// (round 6; +1 / -1 before, and still in the 2048-bit form): the dot product of two cells over the 1024 (padded) bits is
// popcount(a & b) and m = popcount(a) + popcount(b) - 2 dot, so an accumulator that STARTS at -(popcount(a) + popcount(b)) / 2
// ends at -m / 2, exact in f32 (matrixBoundOf, mismatchesOfMatrixResult); v_mfma_f32_32x32x64_f8f6f4 contracts 64 bits of
// 32 x 32 cells per instruction: 8 times the pairs per SIMD clock of the v_xor/v_bcnt loop at its instruction floor
// (tools/ubench_mfma_pairs.hip).  On 0 / 1 operands the matrix pipe holds 2.37 GHz where +-1 held 2.25 (fewer products that
// change sign); the terms cost 32 vector instructions and four LDS reads per tile (profiles/r06_scan_experiments.md, 4).
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

// =========================================================================================================
// The walk: the tile step is hand-scheduled assembly (em2_matrix_step_asm.h, written by tools/gen_matrix_step_asm.py).
// Against round 1's compiler-scheduled walk (git history: scanTilesMatrix, 251 ms where this one takes 172):
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
typedef volatile __attribute__((address_space(3))) uint8_t* LdsBytePtr;

// LDS byte address -> pointer (32 bits on the device; the detour keeps the host pass of the compiler quiet)
template <typename P>
__device__ __forceinline__ P ldsPointer(uint32_t address)
{
    return (P)(uintptr_t)address;
}

// Per-wave LDS block of the walk (byte offsets).
constexpr uint32_t kWalkRowDot = 0u;            // float[64]: bound of row r in the accumulators' unit (matrixBoundOf), read by the steps
constexpr uint32_t kWalkBounds = 256u;          // float[4][32]: column bounds of the four tile buffers (same unit)
constexpr uint32_t kWalkSnapStage = 768u;       // int32[64] (1024-bit walk) / int32[2][64] (2048-bit walk): published cut-offs as loaded
// the 2048-bit walk (+-1 operands):
constexpr uint32_t kWalkWrapCounts = 1280u;     // uint32[2][64]: a walk that went around -- the lanes' records per accumulator when it did
// the 1024-bit walk (0 / 1 operands):
constexpr uint32_t kWalkTermRing = 1024u;       // float[3][64]: the column terms (-popcount / 2) of three pairs of tiles, as loaded
constexpr uint32_t kWalkWrapCounts8 = 1792u;    // uint8[2][64]: the same counts as kWalkWrapCounts (a log holds 128 records)
constexpr uint32_t kMatrixWalkLdsBytes = 1920u;
static_assert(kLogCapacity / 2u <= 255u, "kWalkWrapCounts8 holds a log's record count in a byte");

// A mismatch count as a bound / a result of the matrix steps.  1024-bit steps (EM2_MATRIX_ZERO_ONE: FP4 0 / 1 operands, an
// accumulator starts at -(popcount of the row + popcount of the column) / 2 and adds popcount(row & column)): -mismatches / 2.
// 2048-bit steps (+-1 operands, accumulators start at 0): bits - 2 mismatches.  -1 ("nothing passes") lies above every result
// in both.
template <bool WIDE>
__device__ __forceinline__ float matrixBoundOf(int32_t mismatches)
{
    if (WIDE || !EM2_MATRIX_ZERO_ONE) return (WIDE ? 2048.f : 1024.f) - 2.f * float(mismatches);
    return -0.5f * float(mismatches);
}
template <bool WIDE>
__device__ __forceinline__ uint32_t mismatchesOfMatrixResult(float result)
{
    if (WIDE || !EM2_MATRIX_ZERO_ONE) return uint32_t(((WIDE ? 2048.f : 1024.f) - result) * 0.5f);
    return uint32_t(result * -2.f);
}

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

// A record of the walk: the step's stub wrote {first column of the tile | 2i + a, dot, column bound} into the log of the LANE
// in which register i of accumulator a passed its test (em2_matrix_step_asm.h: 12 bytes, 16 apart).  Lane s = 32 h + t holds
// rows t and 32 + t of the wave and the columns 8q + 4h + j of a tile (i = 4q + j): its records ascend in the column.  The
// third word is the bound the walk tested the COLUMN side against (matrixBoundOf of the column's published cut-off as the walk
// staged it, at most two pairs of tiles old; any value a cell published is valid, bounds only tighten): dot >= bound is the
// column side's test, which the replay therefore decides without a load of its own per record.
struct __attribute__((aligned(16))) WalkRecord {
    uint32_t code;                  // tile's first column (a multiple of 32) | 2i + a
    float dot;                      // the accumulator: -mismatches / 2 (1024-bit steps), 2048 - 2 * mismatches (2048-bit steps)
    float bound;                    // matrixBoundOf(the column's cut-off) (of -1: a column that takes no candidates)
};
static_assert(sizeof(WalkRecord) == EM2_MATRIX_RECORD_BYTES, "the steps store their records 16 bytes apart");
__device__ __forceinline__ uint32_t walkRecordColumn(uint32_t code, uint32_t half)
{
    const uint32_t i = (code & 31u) >> 1;
    return (code & ~31u) + 8u * (i >> 2) + 4u * half + (i & 3u);
}
// (The records of a lane are written by that lane and read by others, from addresses that earlier replays of the same wave
// have read: the replay starts with an acquire at agent scope -- the L1's lines go -- and reads them with plain loads.)
typedef uint32_t WalkRecordWords __attribute__((ext_vector_type(3)));         // (16 bytes apart, like the records)
__device__ __forceinline__ WalkRecord loadWalkRecord(const WalkRecord* log, uint32_t index)
{
    // (global memory, said so: the log's address comes out of the kernel-argument block, see em2_scan_common.h)
    const WalkRecordWords w = ((const __attribute__((address_space(1))) WalkRecordWords*)reinterpret_cast<const WalkRecordWords*>(log))[index];
    WalkRecord r;
    r.code = w.x;
    r.dot = __uint_as_float(w.y);
    r.bound = __uint_as_float(w.z);
    return r;
}

// The convoy (scanMatrixBody): words 4..8 of the block's stop-word area.  [4] = where this item's walk starts (written by the
// block's first thread, read by all); [5] = 0 or what the walk publishes its position under: (segment + 1) << 20 | lap << 12,
// lap = how often the convoy has been around the segment when this part of the walk began; [6..7] = the address of the word
// it publishes to; [8] = the segment's first column >> 6.  A position is [5] + (column >> 6) - [8]: positions of one segment
// grow as the convoy moves on, later segments (of the same parity: they share the word) are larger still, so the word --
// an atomic maximum -- is where the convoy's HEAD is, and who starts there starts where the tiles are in the L2.
// [9], [10] = the columns [begin, end) that scanTilesMatrixPinned walks BEHIND the ones it is called for (the walk that goes around
// in one call; the walk clears the two words, so only the call that follows their writing sees them).
constexpr uint32_t kConvoyStartWord = 4u, kConvoyCodeWord = 5u, kConvoyAddressWord = 6u, kConvoyPairBaseWord = 8u;
constexpr uint32_t kWrapBeginWord = 9u, kWrapEndWord = 10u;
constexpr uint32_t kWalkInLowerColumns = 0x80000000u;        // flag in scanTilesMatrixPinned's result
constexpr uint32_t kConvoyWordsOffset = 40u;    // in 32-bit words from inboxControl: 8 groups x 2 segment parities
constexpr uint32_t kConvoyStopsWord = 56u;      // ... and the number of walks of this launch that stopped for their logs
constexpr uint32_t kConvoyMaxPairs = 4096u;     // pairs of tiles per segment that a position can tell apart
__device__ __forceinline__ void publishWalkPosition(uint64_t address, uint32_t value)
{
    // (a global_ instruction from inline asm: a flat_ one would count against lgkmcnt, which the steps wait for by number)
    asm volatile("global_atomic_umax %0, %1, off sc1" : : "v"(address), "v"(value) : "memory");
}

// Where the walk of an item starts (scanMatrixBody, tileMatrixBody): the first column of its segment -- or, with the convoy, the
// column at which the head of its XCD group's walks through this segment is; the walk then goes around.  Block-uniform; the
// block's first thread sets the convoy's stop words for the call of the walk that follows.  aux->convoy: 0 = off, 1 = follow
// the head, n >= 2 = 64 (n - 1) columns into the segment (tests).
// A walk that goes around and then stops for its logs in its higher columns has walked them for nothing (scanMatrixBody), so
// the convoy is for data whose walks do not stop: nobody joins while more than one in 64 of the launch's items so far
// (`ticket` of them) has stopped -- the count is at kConvoyStopsWord.  (1M cells: the bench's data never stop, -3.7 %; 64
// tight clusters, 0.5 stops per item: +4.5 % with everybody joining, nothing either way with this rule.)
__device__ __forceinline__ uint32_t convoyStartColumn(ArgsPtr aux, volatile uint32_t* shared, uint32_t seg, uint32_t colBegin,
                                                      uint32_t commonEnd, uint32_t ticket)
{
    const uint32_t mode = aux->convoy;
    if (mode == 0u || colBegin + 128u >= commonEnd) {
        // (no convoy for this item: the words of the block's previous item must not make its walk publish or go around)
        if (threadIdx.x == 0u) {
            shared[kConvoyCodeWord] = 0u;
            shared[kWrapBeginWord] = 0u;
            shared[kWrapEndWord] = 0u;
        }
        __syncthreads();
        return colBegin;
    }
    if (threadIdx.x == 0u) {
        uint32_t* word = aux->inboxControl + kConvoyWordsOffset + 2u * (blockIdx.x & 7u) + (seg & 1u);
        const uint32_t code = (seg + 1u) << 20;
        uint32_t from = colBegin, lap = 0u;
        bool publish = mode == 1u;
        if (mode == 1u) {
            const uint32_t seen = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t stops = __hip_atomic_load(aux->inboxControl + kConvoyStopsWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (uint64_t(stops) * 64u > uint64_t(ticket)) {
                publish = false;                    // (walks that stop: every one from its segment's first column)
            } else if ((seen & 0xfff00000u) == code) {
                lap = (seen >> 12) & 0xffu;
                from = colBegin + ((seen & 0xfffu) << 6);
                publish = false;            // (unless it joins: a walk on its own is no head to follow)
            }
        } else {
            from = colBegin + 64u * (mode - 1u);
        }
        // (whole pairs of tiles from the segment's begin, and something left to walk on either side)
        if (from <= colBegin || from + 64u >= commonEnd) from = colBegin;
        else publish = mode == 1u;
        shared[kConvoyStartWord] = from;
        shared[kConvoyCodeWord] = publish ? code | (lap << 12) : 0u;
        shared[kConvoyAddressWord] = uint32_t(reinterpret_cast<uintptr_t>(word));
        shared[kConvoyAddressWord + 1u] = uint32_t(uint64_t(reinterpret_cast<uintptr_t>(word)) >> 32);
        shared[kConvoyPairBaseWord] = colBegin >> 6;
        // (one call walks both parts: scanTilesMatrixPinned, scanTilesMatrixWide)
        shared[kWrapBeginWord] = from != colBegin ? colBegin : 0u;
        shared[kWrapEndWord] = from != colBegin ? from : 0u;
    }
    __syncthreads();
    const uint32_t from = uint32_t(__builtin_amdgcn_readfirstlane(int(shared[kConvoyStartWord])));
    __syncthreads();
    return from;
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
// With the stop words kWrapBeginWord / kWrapEndWord set (scanMatrixBody: a walk that goes around its segment) the walk
// continues with those columns, whole pairs of tiles, when it has reached colEnd -- same pipeline, no new start -- and leaves
// each lane's record counts of that moment at kWalkWrapCounts; the result then is a column of THAT range (or its end), marked
// with kWalkInLowerColumns.
template <bool IDENTITY, bool BOTH = false, bool DIAG = false>
__device__ __attribute__((noinline)) uint32_t scanTilesMatrixPinned(const void* auxArg, const void* fragmentsArg, const void* snapArg,
                                                                    const void* termsArg, uint32_t colBeginArg, uint32_t colEndArg,
                                                                    uint32_t rowFragmentBlockArg, float rowDotArg, float rowTermArg,
                                                                    WalkRecord* waveLogArg, uint32_t logCapacityArg, uint32_t* recordCount,
                                                                    uint32_t tilesLdsArg, uint32_t stopWordsLdsArg, uint32_t walkLdsArg)
{
    const GlobalFragmentPtr fragments = (GlobalFragmentPtr)uniform64(reinterpret_cast<uint64_t>(fragmentsArg));
    const GlobalIntPtr snap = (GlobalIntPtr)uniform64(reinterpret_cast<uint64_t>(snapArg));
    const uint64_t terms = uniform64(reinterpret_cast<uint64_t>(termsArg));
    const uint32_t colBegin = uniform(colBeginArg), colEnd = uniform(colEndArg);
    const uint32_t rowFragmentBlock = uniform(rowFragmentBlockArg), logCapacity = uniform(logCapacityArg);
    const uint32_t tilesLds = uniform(tilesLdsArg);
    const LdsWordPtr stopWords = ldsPointer<LdsWordPtr>(uniform(stopWordsLdsArg));
    const uint32_t walkLds = uniform(walkLdsArg);
    const LdsFloatPtr boundScratch = ldsPointer<LdsFloatPtr>(walkLds + kWalkBounds);
    const LdsIntPtr snapStage = ldsPointer<LdsIntPtr>(walkLds + kWalkSnapStage);
    const uint64_t logBase = uniform64(reinterpret_cast<uint64_t>(waveLogArg));
    uint32_t convoyCode = uniform(stopWords[kConvoyCodeWord]);
    const uint32_t convoyPairBase = uniform(stopWords[kConvoyPairBaseWord]);
    const uint64_t convoyAddress = uint64_t(uniform(stopWords[kConvoyAddressWord])) | (uint64_t(uniform(stopWords[kConvoyAddressWord + 1u])) << 32);
    const uint32_t wrapBegin = uniform(stopWords[kWrapBeginWord]), wrapEnd = uniform(stopWords[kWrapEndWord]);
    // (EM2_MATRIX_DIAG, measurements only: the walk that looks at the bits is an instantiation of its own, so that the
    // one that runs in production has none of their branches between its steps)
    const uint32_t diag = DIAG ? EM2_DIAG_WORD((ArgsPtr)uniform64(reinterpret_cast<uint64_t>(auxArg))) : 0u;
    const uint32_t halfCapacity = logCapacity / 2u;
    // byte offsets into the wave's log area: where the lane's two logs begin, where its next records go (kept in two
    // registers of the walk; the steps return them), and beyond which the walk has to stop
    const uint32_t firstOffset0 = laneId() * logCapacity * EM2_MATRIX_RECORD_BYTES;
    const uint32_t firstOffset1 = firstOffset0 + halfCapacity * EM2_MATRIX_RECORD_BYTES;
    const uint32_t stopRecords = halfCapacity > kMatrixLogMargin / 2u ? halfCapacity - kMatrixLogMargin / 2u : 0u;
    const uint32_t stopOffset0 = firstOffset0 + stopRecords * EM2_MATRIX_RECORD_BYTES;
    const uint32_t stopOffset1 = firstOffset1 + stopRecords * EM2_MATRIX_RECORD_BYTES;
    uint32_t recordOffset = firstOffset0 + recordCount[0] * EM2_MATRIX_RECORD_BYTES;
    uint32_t recordOffset1 = firstOffset1 + recordCount[1] * EM2_MATRIX_RECORD_BYTES;
    {
        const uint32_t lane = laneId();
        ldsPointer<LdsFloatPtr>(walkLds + kWalkRowDot)[lane] = rowDotArg;         // for the steps: float[64], lane = row
        asm volatile(EM2_MATRIX_SET_RECORD_OFFSETS : : "v"(recordOffset), "v"(recordOffset1) : EM2_MATRIX_OWNED_REGISTERS);

        // the B operand: the 2 x 16 fragments of the wave's rows straight into their registers (v128..v255)
        const uint64_t rowFragments = reinterpret_cast<uint64_t>(fragments) + size_t(rowFragmentBlock) * kMatrixTileWords * 16u;
        asm volatile(EM2_MATRIX_LOAD_ROWS : : "s"(rowFragments) : EM2_MATRIX_STEP_CLOBBERS);
    }
    const uint32_t waveSlot = uniform(threadIdx.x >> 6) * 64u;
    // the terms of the lane's two rows (lane l computes rows l & 31 and 32 + (l & 31)): operands of every step
    const float rowTerm0 = __shfl(rowTermArg, int(laneId() & 31u), 64), rowTerm1 = __shfl(rowTermArg, int(32u + (laneId() & 31u)), 64);
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
    // only tighten.  One buffer: the wave has the pair's values in a register (stagedSnap) before it issues the next load.
#define EM2_STAGE_SNAP(firstColumn, lastColumn)                                                                               \
    do {                                                                                                                      \
        uint32_t column_ = (firstColumn) + laneId();                                                                          \
        column_ = column_ < (lastColumn) ? column_ : (lastColumn) - 1u;                                                       \
        const uint64_t address_ = reinterpret_cast<uint64_t>(snap) + uint64_t(column_) * 4u;                                 \
        const uint32_t dst_ = walkLds + kWalkSnapStage;                                                                       \
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off"                                           \
                     :                                                                                                        \
                     : "v"(address_), "s"(dst_)                                                                               \
                     : "memory", "m0");                                                                                       \
    } while (0)
    // The column terms of a pair of tiles (lane = column; the array is padded past the last cell) into slot `ringSlot` of the
    // ring of three pairs: the set a step restarts computes the tile AFTER the step's own, so the terms run a pair ahead of the
    // cut-offs.
#define EM2_STAGE_TERMS(firstColumn, ringSlot)                                                                                \
    do {                                                                                                                      \
        const uint64_t address_ = terms + uint64_t((firstColumn) + laneId()) * 4u;                                            \
        const uint32_t dst_ = walkLds + kWalkTermRing + (ringSlot) * 256u;                                                    \
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off"                                           \
                     :                                                                                                        \
                     : "v"(address_), "s"(dst_)                                                                               \
                     : "memory", "m0");                                                                                       \
    } while (0)
    // the pair of tiles behind the pair at `base` of the columns [.., end): the next of these columns, or -- at their end -- the
    // first of the lower ones (a walk that goes around); false: there is none
#define EM2_FOLLOWING_PAIR(base, end, lower, nextBase_, nextEnd_, nextLower_)                                                 \
    ((nextBase_) = (base) + 64u, (nextEnd_) = (end), (nextLower_) = (lower),                                                  \
     ((nextBase_) >= (end) && !(lower) && wrapBegin < wrapEnd) ? ((nextBase_) = wrapBegin, (nextEnd_) = wrapEnd, (nextLower_) = true) : false, \
     (nextBase_) < (nextEnd_))
#define EM2_WAIT_STAGED() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define EM2_KEEP_WRAP_COUNTS()                                                                                                \
    do {                                                                                                                      \
        ldsPointer<LdsBytePtr>(walkLds + kWalkWrapCounts8)[laneId()] = uint8_t((recordOffset - firstOffset0) / EM2_MATRIX_RECORD_BYTES); \
        ldsPointer<LdsBytePtr>(walkLds + kWalkWrapCounts8)[64u + laneId()] = uint8_t((recordOffset1 - firstOffset1) / EM2_MATRIX_RECORD_BYTES); \
    } while (0)
    EM2_STAGE_TILE(colBegin / 32u, 0u);
    if (colBegin + 32u < colEnd) EM2_STAGE_TILE(colBegin / 32u + 1u, 1u);
    EM2_STAGE_SNAP(colBegin, colEnd);
    EM2_STAGE_TERMS(colBegin, 0u);
    {
        uint32_t secondBase, secondEnd;
        bool secondLower;
        if (EM2_FOLLOWING_PAIR(colBegin, colEnd, false, secondBase, secondEnd, secondLower)) EM2_STAGE_TERMS(secondBase, 1u);
    }
    EM2_WAIT_STAGED();
    __syncthreads();
    if (wrapBegin < wrapEnd && waveSlot == 0u && laneId() == 0u) stopWords[kWrapBeginWord] = stopWords[kWrapEndWord] = 0u;
    // the two sets start as the first two tiles' row + column terms; from here on every testing step restarts the set it tests
    asm volatile(EM2_MATRIX_INIT_X : : "s"(walkLds + kWalkTermRing), "v"(rowTerm0), "v"(rowTerm1) : EM2_MATRIX_STEP_CLOBBERS);
    asm volatile(EM2_MATRIX_INIT_Y : : "s"(walkLds + kWalkTermRing + 128u), "v"(rowTerm0), "v"(rowTerm1) : EM2_MATRIX_STEP_CLOBBERS);
    uint32_t ring = 0;              // iteration % 3: the slot of this pair's terms
    bool tested = false;
    uint64_t passScratch[5];        // scalar pairs for the steps: pass masks in flight, saved exec
    bool pending = false, pendingInY = false;
    uint32_t pendingBase = 0, pendingSlot = 0;
    uint32_t iteration = 0, stopSlot = 0;
    uint32_t rangeEnd = colEnd;     // of the columns being walked: colEnd, then wrapEnd
    bool lowerColumns = false;      // the walk has gone around: [wrapBegin, wrapEnd) now
    bool keepCounts = false;        // ... and the last tile of the higher columns is tested by the step that comes next
    uint32_t result = colEnd;
    // the staged cut-offs of the pair about to be walked (lane = column); those of the next pair are read right behind
    // the barrier that ends a pair, together with the stop word: one LDS round trip there instead of two
    int32_t stagedSnap = snapStage[laneId()];
    for (uint32_t colBase = colBegin;; ++iteration) {
        const uint32_t pair = iteration & 1u;
        boundScratch[pair * 64u + laneId()] = matrixBoundOf<false>(stagedSnap);
        const uint32_t ringNext = ring == 2u ? 0u : ring + 1u, ringSecond = ringNext == 2u ? 0u : ringNext + 1u;
        // the pair behind this one: the next of these columns, or -- at their end -- the first of the lower ones
        uint32_t nextBase = colBase + 64u, nextEnd = rangeEnd;
        const bool around = nextBase >= rangeEnd && !lowerColumns && wrapBegin < wrapEnd;
        if (around) {
            nextBase = wrapBegin;
            nextEnd = wrapEnd;
        }
        const bool more = nextBase < nextEnd;
        if (more) {
            EM2_STAGE_TILE(nextBase / 32u, 2u * (pair ^ 1u));
            EM2_STAGE_SNAP(nextBase, nextEnd);
            if (nextBase + 32u < nextEnd) EM2_STAGE_TILE(nextBase / 32u + 1u, 2u * (pair ^ 1u) + 1u);
            uint32_t secondBase, secondEnd;
            bool secondLower;
            if (EM2_FOLLOWING_PAIR(nextBase, nextEnd, lowerColumns || around, secondBase, secondEnd, secondLower)) {
                EM2_STAGE_TERMS(secondBase, ringSecond);
            }
        }
        // ---- first tile of the pair -> X, under it the test of the pending tile (always in Y here) ----
        {
            const uint32_t tileBase = tilesLds + 2u * pair * (kMatrixTileWords * 16u);
            if (pending && !(diag & 32u)) {
                const uint32_t boundBase = walkLds + kWalkBounds + pendingSlot * 128u;
                // (Y, tested here, computes this pair's second tile next)
                asm volatile(EM2_MATRIX_STEP_X_TESTING_Y
                             : "=v"(recordOffset), "=v"(recordOffset1), "=&s"(passScratch[0]), "=&s"(passScratch[1]), "=&s"(passScratch[2]), "=&s"(passScratch[3]), "=&s"(passScratch[4])
                             : "s"(tileBase), "s"(boundBase), "s"(walkLds + kWalkRowDot), "s"(logBase), "s"(pendingBase),
                               "s"(walkLds + kWalkTermRing + ring * 256u + 128u), "v"(rowTerm0), "v"(rowTerm1)
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
        if (keepCounts) {           // (all tiles of the higher columns have been tested now)
            EM2_KEEP_WRAP_COUNTS();
            keepCounts = false;
        }
        // ---- second tile -> Y, under it the test of the first ----
        if (colBase + 32u < rangeEnd && (diag & 32u)) {
            const uint32_t tileBase = tilesLds + (2u * pair + 1u) * (kMatrixTileWords * 16u);
            asm volatile(EM2_MATRIX_STEP_Y : : "s"(tileBase) : EM2_MATRIX_STEP_CLOBBERS);
            pendingInY = true;
        } else if (colBase + 32u < rangeEnd) {
            const uint32_t tileBase = tilesLds + (2u * pair + 1u) * (kMatrixTileWords * 16u);
            const uint32_t boundBase = walkLds + kWalkBounds + pendingSlot * 128u;
            // (X, tested here, computes the next pair's first tile next)
            asm volatile(EM2_MATRIX_STEP_Y_TESTING_X
                         : "=v"(recordOffset), "=v"(recordOffset1), "=&s"(passScratch[0]), "=&s"(passScratch[1]), "=&s"(passScratch[2]), "=&s"(passScratch[3]), "=&s"(passScratch[4])
                         : "s"(tileBase), "s"(boundBase), "s"(walkLds + kWalkRowDot), "s"(logBase), "s"(pendingBase),
                           "s"(walkLds + kWalkTermRing + ringNext * 256u), "v"(rowTerm0), "v"(rowTerm1)
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
        if (around) {
            // The higher columns end here.  Their last tile is tested by the next step if that can do it (a pending tile in Y),
            // and right here otherwise (an odd number of tiles: the pending one sits in X, which the next step fills).
            if (!pendingInY && !(diag & 32u)) {
                const uint32_t boundBase = walkLds + kWalkBounds + pendingSlot * 128u;
                asm volatile(EM2_MATRIX_TEST_X
                             : "=v"(recordOffset), "=v"(recordOffset1), "=&s"(passScratch[0]), "=&s"(passScratch[1]), "=&s"(passScratch[2]), "=&s"(passScratch[3]), "=&s"(passScratch[4])
                             : "s"(boundBase), "s"(walkLds + kWalkRowDot), "s"(logBase), "s"(pendingBase) : EM2_MATRIX_STEP_CLOBBERS);
                tested = true;
                pending = false;
                EM2_KEEP_WRAP_COUNTS();
                // (no step has restarted the sets for the lower columns' first pair: X was tested without one, Y was left
                // with the terms of a tile that does not exist)
                asm volatile(EM2_MATRIX_INIT_X : : "s"(walkLds + kWalkTermRing + ringNext * 256u), "v"(rowTerm0), "v"(rowTerm1) : EM2_MATRIX_STEP_CLOBBERS);
                asm volatile(EM2_MATRIX_INIT_Y : : "s"(walkLds + kWalkTermRing + ringNext * 256u + 128u), "v"(rowTerm0), "v"(rowTerm1) : EM2_MATRIX_STEP_CLOBBERS);
            } else {
                keepCounts = true;
            }
            lowerColumns = true;
            rangeEnd = nextEnd;
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
        stagedSnap = snapStage[laneId()];
        if (convoyCode != 0u && (iteration & 3u) == 0u && waveSlot == 0u && laneId() == 0u) {
            publishWalkPosition(convoyAddress, convoyCode + ((colBase >> 6) - convoyPairBase));
        }
        if (stop != 0u) {
            __syncthreads();
            if (waveSlot == 0u && laneId() == 0u) stopWords[slot] = 0u;
            __syncthreads();
            result = more ? nextBase : rangeEnd;
            if (around && !BOTH) {
                // Logs this full at the end of the higher columns leave the lower ones no room (a call must be able to add
                // the records of three tiles to what it finds): for the caller the walk stopped in its higher columns.
                // (BOTH, the tile kernels: their caller empties the logs whatever the order of the columns.)
                result = colBegin;
                lowerColumns = false;
                keepCounts = false;
            }
            break;
        }
        if (!more) {
            result = rangeEnd;
            break;
        }
        if (around && convoyCode != 0u && ((convoyCode >> 12) & 0xffu) < 255u) convoyCode += 1u << 12;       // the convoy's next lap
        colBase = nextBase;
        ring = ringNext;
    }
#undef EM2_STAGE_TILE
#undef EM2_STAGE_SNAP
#undef EM2_STAGE_TERMS
#undef EM2_FOLLOWING_PAIR
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
    if (keepCounts) EM2_KEEP_WRAP_COUNTS();         // (a walk that stopped at the very end of its higher columns)
#undef EM2_KEEP_WRAP_COUNTS
    // the records were stored by one lane and are read back by others: the stores must have left the wave before the
    // caller replays the logs (it reads past the L1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tested && !(diag & 1u)) {
        recordCount[0] = (recordOffset - firstOffset0) / EM2_MATRIX_RECORD_BYTES;
        recordCount[1] = (recordOffset1 - firstOffset1) / EM2_MATRIX_RECORD_BYTES;
    }
    return lowerColumns ? result | kWalkInLowerColumns : result;
}

// The same walk for 2048-bit signatures (EM2_MATRIX_WIDE_*: the registers hold 32 rows x 32 k-steps, a tile is 32 columns x
// 32 k-steps = two 16 KB slots side by side, a step is one tile).  One call walks the columns for ONE half of the wave's 64
// rows: rowHalf a = rows 32a .. 32a+31, whose fragments are the 32 KB block rowFragmentBlock (in 32-cell blocks), whose
// records go to the lanes' logs of accumulator a (recordCount[a]).  The walk may stop at any tile boundary, where one tile
// is still untested: a log needs room for two tiles (32 records).
template <bool IDENTITY, bool BOTH = false>
__device__ __attribute__((noinline)) uint32_t scanTilesMatrixWide(const void* fragmentsArg, const void* snapArg, uint32_t colBeginArg,
                                                                  uint32_t colEndArg, uint32_t rowFragmentBlockArg, float rowDotArg,
                                                                  uint32_t rowHalfArg, WalkRecord* waveLogArg, uint32_t logCapacityArg,
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
    uint32_t convoyCode = uniform(stopWords[kConvoyCodeWord]);
    const uint32_t convoyPairBase = uniform(stopWords[kConvoyPairBaseWord]);
    const uint64_t convoyAddress = uint64_t(uniform(stopWords[kConvoyAddressWord])) | (uint64_t(uniform(stopWords[kConvoyAddressWord + 1u])) << 32);
    const uint32_t wrapBegin = uniform(stopWords[kWrapBeginWord]), wrapEnd = uniform(stopWords[kWrapEndWord]);
    const uint32_t halfCapacity = logCapacity / 2u;
    const uint32_t firstOffset = (laneId() * logCapacity + rowHalf * halfCapacity) * EM2_MATRIX_RECORD_BYTES;
    const uint32_t stopRecords = halfCapacity > kMatrixLogMargin / 2u ? halfCapacity - kMatrixLogMargin / 2u : 0u;
    const uint32_t stopOffset = firstOffset + stopRecords * EM2_MATRIX_RECORD_BYTES;
    uint32_t recordOffset = firstOffset + recordCount[rowHalf] * EM2_MATRIX_RECORD_BYTES;
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
#define EM2_STAGE_WIDE_TILE(firstColumn, lastColumn, parity)                                                                  \
    do {                                                                                                                      \
        EM2_STAGE_UNIT((firstColumn) / 16u, 2u * (parity));                                                                   \
        EM2_STAGE_UNIT((firstColumn) / 16u + 1u, 2u * (parity) + 1u);                                                         \
        uint32_t column_ = (firstColumn) + (laneId() & 31u);                                                                  \
        column_ = column_ < (lastColumn) ? column_ : (lastColumn) - 1u;                                                       \
        const uint64_t address_ = reinterpret_cast<uint64_t>(snap) + uint64_t(column_) * 4u;                                 \
        const uint32_t dstSnap_ = walkLds + kWalkSnapStage + (parity) * 256u;                                                 \
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off"                                           \
                     :                                                                                                        \
                     : "v"(address_), "s"(dstSnap_)                                                                           \
                     : "memory", "m0");                                                                                       \
    } while (0)
    EM2_STAGE_WIDE_TILE(colBegin, colEnd, 0u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wrapBegin < wrapEnd && waveSlot == 0u && laneId() == 0u) stopWords[kWrapBeginWord] = stopWords[kWrapEndWord] = 0u;
#define EM2_KEEP_WRAP_COUNTS()                                                                                                \
    ldsPointer<LdsWordPtr>(walkLds + kWalkWrapCounts)[rowHalf * 64u + laneId()] = (recordOffset - firstOffset) / EM2_MATRIX_RECORD_BYTES
    bool tested = false;
    uint64_t passScratch[5];
    bool pending = false;
    uint32_t pendingBase = 0, pendingParity = 0;
    uint32_t iteration = 0, stopSlot = 0;
    uint32_t rangeEnd = colEnd;     // (the walk that goes around: as in scanTilesMatrixPinned)
    bool lowerColumns = false, keepCounts = false;
    uint32_t result = colEnd;
    int32_t stagedSnap = snapStage[laneId()];
    for (uint32_t colBase = colBegin;; ++iteration) {
        const uint32_t parity = iteration & 1u;
        boundScratch[parity * 32u + (laneId() & 31u)] = 2.f * kMatrixBits - 2.f * float(stagedSnap);
        uint32_t nextBase = colBase + 32u, nextEnd = rangeEnd;
        const bool around = nextBase >= rangeEnd && !lowerColumns && wrapBegin < wrapEnd;
        if (around) {
            nextBase = wrapBegin;
            nextEnd = wrapEnd;
        }
        const bool more = nextBase < nextEnd;
        if (more) EM2_STAGE_WIDE_TILE(nextBase, nextEnd, parity ^ 1u);
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
        if (keepCounts) {           // (all tiles of the higher columns have been tested now)
            EM2_KEEP_WRAP_COUNTS();
            keepCounts = false;
        }
        pending = true;
        pendingBase = colBase;
        pendingParity = parity;
        if (around) {
            keepCounts = true;
            lowerColumns = true;
            rangeEnd = nextEnd;
        }
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
        if (convoyCode != 0u && (iteration & 3u) == 0u && waveSlot == 0u && laneId() == 0u) {
            publishWalkPosition(convoyAddress, convoyCode + ((colBase >> 6) - convoyPairBase));
        }
        if (stop != 0u) {
            __syncthreads();
            if (waveSlot == 0u && laneId() == 0u) stopWords[slot] = 0u;
            __syncthreads();
            result = more ? nextBase : rangeEnd;
            if (around && !BOTH) {
                // Logs this full at the end of the higher columns leave the lower ones no room (a call must be able to add
                // the records of three tiles to what it finds): for the caller the walk stopped in its higher columns.
                // (BOTH, the tile kernels: their caller empties the logs whatever the order of the columns.)
                result = colBegin;
                lowerColumns = false;
                keepCounts = false;
            }
            break;
        }
        if (!more) {
            result = rangeEnd;
            break;
        }
        if (around && convoyCode != 0u && ((convoyCode >> 12) & 0xffu) < 255u) convoyCode += 1u << 12;       // the convoy's next lap
        colBase = nextBase;
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
    if (keepCounts) EM2_KEEP_WRAP_COUNTS();         // (a walk that stopped at the very end of its higher columns)
#undef EM2_KEEP_WRAP_COUNTS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tested) recordCount[rowHalf] = (recordOffset - firstOffset) / EM2_MATRIX_RECORD_BYTES;
    return lowerColumns ? result | kWalkInLowerColumns : result;
}

// The log of `lane` for accumulator a in the wave's log area.
__device__ __forceinline__ const WalkRecord* walkLogOf(const WalkRecord* waveLog, uint32_t logCapacity, uint32_t lane, uint32_t a)
{
    return waveLog + size_t(lane) * logCapacity + a * (logCapacity / 2u);
}

// An element of the replay's merge: a record as ONE word that sorts by the column -- (column - the item's first column) << 13
// | the column side's test << 12 | mismatches (at most 2048).  The columns of an item lie within 2^19 of its first one (the
// launchers keep a segment that short), the columns of a row's records are all different, so comparing the words compares the
// columns; 0xffffffff stands behind every record.  One word instead of (column, payload): a vector instruction of the replay
// waits for a gap between the matrix instructions of the wave it shares its SIMD with -- some 30 cycles each, measured -- so
// the replay's time is its count of vector instructions, and the merge network is most of them.
constexpr uint32_t kMergeColumnShift = 13u, kMergePassBit = 0x1000u, kMergeMismatchMask = 0xfffu;
constexpr uint32_t kMaxColumnsPerItem = 1u << 19;
__device__ __forceinline__ uint32_t mergeKeyOf(bool have, const WalkRecord& r, uint32_t half, float bits, uint32_t firstColumn)
{
    const uint32_t relative = walkRecordColumn(r.code, half) - firstColumn;
    // (bits == 0: a result of the 0 / 1 steps, -mismatches / 2)
    const uint32_t m = uint32_t(__builtin_fmaf(r.dot, bits == 0.f ? -2.f : -0.5f, 0.5f * bits));
    const uint32_t key = (relative << kMergeColumnShift) | m | (r.dot >= r.bound ? kMergePassBit : 0u);
    return have ? key : 0xffffffffu;
}
__device__ __forceinline__ void orderKeys(uint32_t& lo, uint32_t& hi)
{
    const uint32_t a = lo, b = hi;
    lo = a < b ? a : b;
    hi = a < b ? b : a;
}
// The lanes whose bit STRIDE is clear (they keep the smaller element of a compare-exchange), as a mask
template <uint32_t STRIDE> __device__ __forceinline__ constexpr uint64_t lowerLanesOf()
{
    return STRIDE == 32u ? 0x00000000ffffffffull : STRIDE == 16u ? 0x0000ffff0000ffffull : STRIDE == 8u ? 0x00ff00ff00ff00ffull
         : STRIDE == 4u ? 0x0f0f0f0f0f0f0f0full : STRIDE == 2u ? 0x3333333333333333ull : 0x5555555555555555ull;
}
// The key of lane (lane ^ STRIDE), without the LDS: quad permutes for 1 and 2, a mirror of eight lanes followed by a
// reversal of four for 4 (7 - i, then ^ 3: i ^ 4), a rotation of sixteen lanes by eight for 8, and the gfx950 row / half swaps
// for 16 and 32: with the same value as both operands v_permlane16_swap / v_permlane32_swap return the even rows' (the lower
// half's) value in every lane of their first result and the odd rows' (the upper half's) in the second
// (tools/probe/permlane_swap.hip).  ds_bpermute in their place made a merge a chain of LDS round trips behind the matrix
// walk's fragment reads.
template <uint32_t STRIDE>
__device__ __forceinline__ uint32_t partnerKey(uint32_t key)
{
    if (STRIDE == 1u) return uint32_t(__builtin_amdgcn_update_dpp(0, int(key), 0xB1, 0xf, 0xf, false));          // quad_perm:[1,0,3,2]
    if (STRIDE == 2u) return uint32_t(__builtin_amdgcn_update_dpp(0, int(key), 0x4E, 0xf, 0xf, false));          // quad_perm:[2,3,0,1]
    if (STRIDE == 4u) {
        const int mirrored = __builtin_amdgcn_update_dpp(0, int(key), 0x141, 0xf, 0xf, false);                    // row_half_mirror
        return uint32_t(__builtin_amdgcn_update_dpp(0, mirrored, 0x1B, 0xf, 0xf, false));                         // quad_perm:[3,2,1,0]
    }
    if (STRIDE == 8u) return uint32_t(__builtin_amdgcn_update_dpp(0, int(key), 0x128, 0xf, 0xf, false));          // row_ror:8
    const bool lower = __builtin_amdgcn_inverse_ballot_w64(lowerLanesOf<STRIDE>());
    if (STRIDE == 16u) {
        const auto both = __builtin_amdgcn_permlane16_swap(key, key, false, false);
        return lower ? uint32_t(both[1]) : uint32_t(both[0]);
    }
    const auto both = __builtin_amdgcn_permlane32_swap(key, key, false, false);
    return lower ? uint32_t(both[1]) : uint32_t(both[0]);
}
// One stage: the lanes whose bit STRIDE is clear keep the smaller key, the others the larger
template <uint32_t STRIDE, int R>
__device__ __forceinline__ void bitonicStage(uint32_t (&key)[R])
{
    if (STRIDE == 8u || STRIDE == 4u) {
        // Eight (four) lanes apart, the lanes that keep the smaller key are whole banks of four: the DPP instructions' own bank
        // masks say who takes the minimum and who the maximum -- two instructions and no lane mask in scalar registers, where
        // min, max and a select under such a mask were three (and the masks of five strides were spilled to a register's lanes
        // and read back at every turn of the replay).  (s_nop 4: a DPP operand written by the vector instruction in front needs two
        // wait states, an EXEC written by one five; nothing adds them inside an asm statement.)
#pragma unroll
        for (int i = 0; i < R; i++) {
            uint32_t out;
            if (STRIDE == 8u) {
                asm("s_nop 4\n\t"
                    "v_min_u32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                    "v_max_u32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc"
                    : "=&v"(out) : "v"(key[i]));
            } else {
                const uint32_t mirrored = uint32_t(__builtin_amdgcn_update_dpp(0, int(key[i]), 0x141, 0xf, 0xf, false));        // row_half_mirror: 7 - i
                asm("s_nop 4\n\t"
                    "v_min_u32_dpp %0, %1, %2 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0x5\n\t"
                    "v_max_u32_dpp %0, %1, %2 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xa"
                    : "=&v"(out) : "v"(mirrored), "v"(key[i]));
            }
            key[i] = out;
        }
        return;
    }
    const bool lower = __builtin_amdgcn_inverse_ballot_w64(lowerLanesOf<STRIDE>());
    if (STRIDE >= 16u) {
        // (the swap hands every lane both keys of its pair -- the even rows' / lower half's in the first result: their minimum
        // and maximum need no partner to be picked first)
#pragma unroll
        for (int i = 0; i < R; i++) {
            uint32_t first, second;
            if (STRIDE == 16u) {
                const auto both = __builtin_amdgcn_permlane16_swap(key[i], key[i], false, false);
                first = uint32_t(both[0]);
                second = uint32_t(both[1]);
            } else {
                const auto both = __builtin_amdgcn_permlane32_swap(key[i], key[i], false, false);
                first = uint32_t(both[0]);
                second = uint32_t(both[1]);
            }
            const uint32_t smaller = first < second ? first : second, larger = first < second ? second : first;
            key[i] = lower ? smaller : larger;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < R; i++) {
        const uint32_t other = partnerKey<STRIDE>(key[i]);
        const uint32_t smaller = other < key[i] ? other : key[i], larger = other < key[i] ? key[i] : other;
        key[i] = lower ? smaller : larger;
    }
}
// Two bitonic merges at once, one per half of the wave: 32 keys each, one per lane
__device__ __forceinline__ void bitonicMergeHalves(uint32_t& key)
{
    uint32_t one[1] = {key};
    bitonicStage<16u, 1>(one);
    bitonicStage<8u, 1>(one);
    bitonicStage<4u, 1>(one);
    bitonicStage<2u, 1>(one);
    bitonicStage<1u, 1>(one);
    key = one[0];
}
// The bitonic merge of 64 R keys -- key i of the sequence sits in key[i >> 6] of lane i & 63; the first half ascends, the
// second descends -- into ascending order: log2(64 R) compare-exchange stages, those of 64 keys and more apart between the
// registers of a lane, the others between lanes.
template <int R>
__device__ __forceinline__ void bitonicMergeWave(uint32_t (&key)[R])
{
    if (R == 4) {
        orderKeys(key[0], key[2]);
        orderKeys(key[1], key[3]);
        orderKeys(key[0], key[1]);
        orderKeys(key[2], key[3]);
    } else {
        orderKeys(key[0], key[1]);
    }
    bitonicStage<32u, R>(key);
    bitonicStage<16u, R>(key);
    bitonicStage<8u, R>(key);
    bitonicStage<4u, R>(key);
    bitonicStage<2u, R>(key);
    bitonicStage<1u, R>(key);
}

// (what the replay reads from the kernel-argument block, once per call instead of once per row: a scalar load there is two
// hundred cycles in front of whatever needs it)
struct ReplayArgs {
    Entry* lists;                       // the lists of the wave's 64 rows
    uint64_t* inbox;
    const int32_t* acceptMaxByKey;
    const uint32_t* keyOfMismatch;
    uint32_t k, twoK, columnShift;      // columnShift = 13 + rowBits: where an inbox entry's target cell begins
    uint32_t vectorMemoryIssued;        // vector-memory instructions the replay is certain to have issued so far (awaitVectorMemory)
    uint32_t firstColumn;               // of the item's columns: the merge keys hold columns relative to it
    bool selfPairs;                     // the rows' own cells may be among their columns (full rows): dropped here
};
// The column side of a batch of records: one compaction into the wave's chunk of the inbox (emitColumn, with the arguments at hand)
__device__ __forceinline__ void emitBatchToInbox(bool emit, uint32_t col, uint32_t rowId, uint32_t m, uint32_t lane, ReplayArgs& args,
                                                 uint32_t& emitPos, uint32_t& emitEnd)
{
    const uint64_t emitMask = __builtin_amdgcn_ballot_w64(emit);
    if (emitMask == 0ull) return;
    uint32_t p = uniform(emitPos), end = uniform(emitEnd);
    if (p > end) return;                // (emission disabled after an overflow)
    const uint32_t entries = uint32_t(__builtin_popcountll(emitMask));
    if (p + entries > end) {
        ArgsPtr aux = kernelArgs();
        const uint64_t fresh = refillInboxChunk(aux->inbox, aux->inboxControl, aux->inboxCapacity, aux->inboxChunk, lane, p, end);
        p = uint32_t(fresh);
        end = uint32_t(fresh >> 32);
    }
    if (p <= end) {
        if (emit) storeGlobalWord(args.inbox + p + lanesBelow(emitMask), (uint64_t(col) << args.columnShift) | (uint64_t(rowId) << 13u) | uint64_t(m));
        args.vectorMemoryIssued += 1u;
        p += entries;
    }
    emitPos = p;
    emitEnd = end;
}

// A batch of one row's candidates, lane by lane in ascending order of the candidate, through the row's exact state machine: the
// lanes test together against the row's cut-off, what passes is appended to the row's list behind a prefix count, and a list
// that reaches 2k entries inside the batch is cut right behind the candidate that filled it (src/ExpressionMatrixLsh.cpp:243-251:
// push, then keepBest at 2k), the lanes behind that candidate testing again against the new cut-off.
template <bool IDENTITY>
__device__ __forceinline__ void offerBatchToRow(bool active, uint32_t col, uint32_t m, uint32_t lane, Entry* listRow, ReplayArgs& args,
                                                uint32_t& count, int32_t& mMax, unsigned char* ldsRaw)
{
    const uint32_t twoK = args.twoK;
    for (;;) {
        const bool pass = active && int32_t(m) <= mMax;
        const uint64_t passMask = __builtin_amdgcn_ballot_w64(pass);
        if (passMask == 0ull) break;
        const uint32_t passes = uint32_t(__builtin_popcountll(passMask)), room = twoK - count;
        const uint32_t index = lanesBelow(passMask);
        if (pass && index < room) {
            uint32_t entryKey = m;
            if (!IDENTITY) entryKey = args.keyOfMismatch[m];
            storeEntry(listRow + count + index, col, entryKey);
        }
        args.vectorMemoryIssued += 1u;          // (some lane passed, and the list has room for one at least)
        if (passes < room) {
            count += passes;
            break;
        }
        // the list is full behind the candidate of index room - 1: keepBest, the new cut-off, and the lanes behind that candidate again
        Entry* lds = reinterpret_cast<Entry*>(ldsRaw + size_t(threadIdx.x >> 6) * twoK * kLdsBytesPerEntrySlot);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        const uint32_t backKey = cutListToBest((LdsEntryPtr)lds, listRow, twoK, args.k, lane, true);
        mMax = __builtin_amdgcn_readfirstlane(args.acceptMaxByKey[backKey]);
        count = args.k;
        waveLdsFence();
        const uint32_t filler = uint32_t(__builtin_ctzll(__builtin_amdgcn_ballot_w64(pass && index == room - 1u)));
        active = active && lane > filler;
    }
}

template <bool IDENTITY, int R>
__device__ __forceinline__ void replayRowElements(const uint32_t (&key)[R], uint32_t n, uint32_t lane, uint32_t rowId, bool emitRow,
                                                  Entry* listRow, ReplayArgs& args, uint32_t& count, int32_t& mMax, uint32_t& emitPos,
                                                  uint32_t& emitEnd, unsigned char* ldsRaw, uint64_t* timed = nullptr)
{
#pragma unroll
    for (int b = 0; b < R; b++) {
        if (64u * uint32_t(b) >= n) break;
        const uint32_t col = (key[b] >> kMergeColumnShift) + args.firstColumn, m = key[b] & kMergeMismatchMask;
        offerBatchToRow<IDENTITY>(64u * uint32_t(b) + lane < n && !(args.selfPairs && col == rowId), col, m, lane, listRow, args, count, mMax, ldsRaw);
        if (emitRow) {
            emitBatchToInbox(64u * uint32_t(b) + lane < n && (key[b] & kMergePassBit) != 0u, col, rowId, m, lane, args, emitPos, emitEnd);
        }
    }
}

// Two rows at once, one per half of the wave: what replayRowElements does for one, for rows whose merged records fit 32 lanes
// (two logs of 16 records at most: most rows of most replays).  Lane 32 h + j holds merged record j of row h; the halves'
// states are uniform within a half (X: lanes 0..31, Y: 32..63), the ballots are shared, and a half whose list fills up is cut --
// the whole wave works on that one list -- before its lanes behind the filling record test again.
template <bool IDENTITY>
__device__ __forceinline__ void replayPairElements(uint32_t key, uint32_t lane, uint32_t rowX, uint32_t rowY,
                                                   uint32_t rowOfWave, bool emitX, bool emitY, ReplayArgs& args, uint32_t& countX,
                                                   int32_t& mMaxX, uint32_t& countY, int32_t& mMaxY, uint32_t& emitPos, uint32_t& emitEnd,
                                                   unsigned char* ldsRaw, uint64_t* timed = nullptr)
{
    const uint64_t tq0 = timed ? __builtin_readcyclecounter() : 0ull;
    const uint32_t twoK = args.twoK;
    const bool upper = __builtin_amdgcn_inverse_ballot_w64(0xffffffff00000000ull);
    const uint32_t col = (key >> kMergeColumnShift) + args.firstColumn, m = key & kMergeMismatchMask;
    const uint32_t rowId = rowOfWave + (upper ? rowY : rowX);
    // (a sentinel's column is beyond every count: `have` needs no look at the counts)
    const bool have = key != 0xffffffffu;
    bool active = have && !(args.selfPairs && col == rowId);
    for (;;) {
        const bool pass = active && int32_t(m) <= (upper ? mMaxY : mMaxX);
        const uint64_t passMask = __builtin_amdgcn_ballot_w64(pass);
        if (passMask == 0ull) break;
        const uint32_t passesX = uint32_t(__builtin_popcount(uint32_t(passMask))), passesY = uint32_t(__builtin_popcount(uint32_t(passMask >> 32)));
        const uint32_t roomX = twoK - countX, roomY = twoK - countY;
        // the entry's place among the lists of the wave's rows (32 bits; one address computation at the store)
        const uint32_t placeX = rowX * twoK + countX, placeY = rowY * twoK + countY - passesX;
        const uint32_t index = lanesBelow(passMask);
        const uint32_t limit = upper ? passesX + roomY : roomX;
        if (pass && index < limit) {
            uint32_t entryKey = m;
            if (!IDENTITY) entryKey = args.keyOfMismatch[m];
            storeEntry(args.lists + ((upper ? placeY : placeX) + index), col, entryKey);
        }
        args.vectorMemoryIssued += 1u;          // (some lane passed, and its list has room for one at least)
        const bool fullX = passesX >= roomX, fullY = passesY >= roomY;
        if (!fullX) countX += passesX;
        if (!fullY) countY += passesY;
        if (!fullX && !fullY) break;
        // (a half that did not fill its list is through with this batch; a half that did goes on behind the record that filled it)
        uint32_t fillerX = 64u, fillerY = 64u;
        Entry* lds = reinterpret_cast<Entry*>(ldsRaw + size_t(threadIdx.x >> 6) * twoK * kLdsBytesPerEntrySlot);
        if (fullX) {
            fillerX = uint32_t(__builtin_ctzll(__builtin_amdgcn_ballot_w64(pass && !upper && index == roomX - 1u)));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            const uint32_t backKey = cutListToBest((LdsEntryPtr)lds, args.lists + size_t(rowX) * twoK, twoK, args.k, lane, true);
            mMaxX = __builtin_amdgcn_readfirstlane(args.acceptMaxByKey[backKey]);
            countX = args.k;
            waveLdsFence();
        }
        if (fullY) {
            fillerY = uint32_t(__builtin_ctzll(__builtin_amdgcn_ballot_w64(pass && upper && index == passesX + roomY - 1u)));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            const uint32_t backKey = cutListToBest((LdsEntryPtr)lds, args.lists + size_t(rowY) * twoK, twoK, args.k, lane, true);
            mMaxY = __builtin_amdgcn_readfirstlane(args.acceptMaxByKey[backKey]);
            countY = args.k;
            waveLdsFence();
        }
        active = active && (upper ? lane > fillerY && fullY : lane > fillerX && fullX);
    }
    const uint64_t tq1 = timed ? __builtin_readcyclecounter() : 0ull;
    if (emitX || emitY) {
        emitBatchToInbox(have && (key & kMergePassBit) != 0u && (upper ? emitY : emitX), col, rowId, m, lane, args, emitPos, emitEnd);
    }
    if (timed) {
        timed[2] += tq1 - tq0;
        timed[6] += __builtin_readcyclecounter() - tq1;
    }
}

// The replay of the walk's logs for the rows of the wave, ROW BY ROW, the lanes side by side on one row's records.
// Row r = 32a + t finds its records in the accumulator-a logs of lanes t (columns 8q .. 8q+3 of every group) and 32 + t (columns
// 8q+4 .. 8q+7); both ascend in the column, and the row's candidates must be offered in ascending order.  The wave loads the two
// logs side by side -- the first ascending over the lanes, the second descending -- and merges them with a bitonic network (one
// record per lane and log for logs of up to 64 records, two beyond: a log holds 128 at most, kLogCapacity / 2); replayRowElements
// does the rest.  The loads of four rows are in flight while one is replayed (the records come from the L2: a form in which
// every lane merged its own row's two logs record by record waited a round trip per record, as long as the fullest lane had
// records, and spent a third of the waves' time there on clustered data).  Rows with a long log follow, one at a time.
// Per record: the row side through the exact state machine, the column side -- unless the rows scan all columns themselves
// (full rows) -- to the inbox if the record's third word says that it passed the column's published cut-off.
// recordCount[a] = the calling lane's number of records in its log of accumulator a.  A walk that went around its segment:
// firstRecord[a] = the lane's number of records when the walk reached the segment's end (their columns are the higher ones and
// come last); all = false replays only the records from firstRecord on (the walk stopped for its logs in the lower columns:
// those are in order and go first, the others stay).
// lane = row for row / rowValid / myList / count / mMax, as everywhere outside the walk.
struct ReplayRowLoads {
    uint32_t row;                       // 0..63, 64 = none
    uint32_t nA, nB;                    // records to replay of the row's two logs
    uint32_t issuedAt;                  // ReplayArgs::vectorMemoryIssued behind the row's two transfers
};
// The records of a row travel global -> LDS without touching registers (global_load_lds_dwordx4: LDS address = M0 + 16 * lane),
// into a ring of four slots of two 1 KB pieces per wave -- lane j's record of the first log at 16 j, of the second at 4096 + 16 j -- and are
// waited for by COUNT, so that the loads of the next three rows stay in flight while a row is replayed.  Left to the compiler
// (loads into registers) the wait in front of a row's first use was vmcnt(0): the loop's stores -- the lists, the inbox -- come
// in numbers it cannot bound, and it then drains the queue, a round trip to the L2 per row; and registers that an asm statement
// loads into are copied by the compiler while the load is in flight.  The LDS has no such copies.  Every call issues exactly
// two transfers (replayWalkLogs: a slot without a row loads for nobody), memory operations complete in order, and whatever
// else the loop issues in between only makes the count more conservative.
// The ring lives in the block's tile buffers, which nobody reads between two calls of the walk (its last barrier is behind
// the last read of a tile, and its staging has been waited for): slot i of wave w in the 4 KB piece of tile buffer i >> 1 that
// only wave w's staging writes (scanTilesMatrixPinned: EM2_STAGE_TILE) -- a wave that is back in the walk while its block's
// other waves still replay writes nothing of theirs.  (A wave stages the 1 KB pieces 4 KB apart of every 16 KB buffer that begin
// 1 KB x its number into it: two of the four pieces of a buffer are a slot.)
constexpr uint32_t kReplayRingSecondLog = 4096u;
__device__ __forceinline__ uint32_t replayRingSlot(uint32_t tilesLds, uint32_t wave, uint32_t slot)
{
    return tilesLds + (slot >> 1) * (kMatrixTileWords * 16u) + (slot & 1u) * 8192u + wave * 1024u;
}
__device__ __forceinline__ void issueRecordLoads(uint32_t ringSlotLds, const WalkRecord* logA, uint32_t indexA, const WalkRecord* logB,
                                                 uint32_t indexB)
{
    const uint64_t addressA = reinterpret_cast<uint64_t>(logA + indexA), addressB = reinterpret_cast<uint64_t>(logB + indexB);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"
                 "s_add_u32 m0, %2, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                 :
                 : "v"(addressA), "v"(addressB), "s"(ringSlotLds)
                 : "memory", "m0", "scc");
}
__device__ __forceinline__ void issueRecordLoad(uint32_t ringSlotLds, uint64_t address)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(address), "s"(ringSlotLds) : "memory", "m0");
}
// Waits until at most `younger` of the wave's vector-memory operations are outstanding.  The operations complete in order, so
// a transfer is complete once no more operations are outstanding than were issued behind it.  The replay counts what it
// issues behind a transfer -- the later turns' transfers AND the stores of the turns in between (the lists, the inbox): with the
// transfers alone as the count, a turn waited for the stores of the turn before it, a round trip to the L2 again.  Counting too
// few is safe (a longer wait), so only stores that are certain to be issued are counted, and 15 stands for "15 or more".
__device__ __forceinline__ void awaitVectorMemory(uint32_t younger)
{
    switch (younger < 15u ? younger : 15u) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    }
}
typedef const volatile __attribute__((address_space(3))) WalkRecordWords* LdsRecordPtr;
__device__ __forceinline__ WalkRecord walkRecordOf(const WalkRecordWords& w)
{
    WalkRecord r;
    r.code = w.x;
    r.dot = __uint_as_float(w.y);
    r.bound = __uint_as_float(w.z);
    return r;
}
template <bool IDENTITY, bool WIDE = false>
__device__ __forceinline__ void replayWalkLogs(const WalkRecord* waveLog, uint32_t logCapacity, const uint32_t (&recordCount)[2],
                                               const uint32_t (&firstRecord)[2], bool all, uint32_t lane,
                                               uint32_t row, bool rowValid, bool emitColumns, uint32_t listBlock, Entry* myList,
                                               uint32_t twoK, uint32_t& count, int32_t& mMax, uint32_t& emitPos, uint32_t& emitEnd,
                                               unsigned char* ldsRaw, uint32_t tilesLds, uint32_t firstColumn, uint64_t* timed = nullptr)
{
    const uint64_t tr0 = timed ? __builtin_readcyclecounter() : 0ull;
    constexpr float bits = WIDE ? 2.f * kMatrixBits : EM2_MATRIX_ZERO_ONE ? 0.f : kMatrixBits;          // (mergeKeyOf)
    // (the caller's arrays live in scratch memory -- the walk takes them by address: one read each)
    const uint32_t stored0Mine = recordCount[0], stored1Mine = recordCount[1], first0Mine = firstRecord[0], first1Mine = firstRecord[1];
    // what this lane's two logs hold for the replay, and which rows have records at all (bit 32 a + t: row 32 a + t)
    const uint32_t mine[2] = {all ? stored0Mine : stored0Mine - first0Mine, all ? stored1Mine : stored1Mine - first1Mine};
    // rows by the length of their longer log: up to 16 records (two rows per turn), up to 32 (a row per turn, a record per lane),
    // up to 64 (a row per turn, a record per lane and log), beyond (two records per lane and log)
    uint64_t rowsWithShortLogs = 0, rowsWithMediumLogs = 0, rowsWithRecords = 0, rowsWithLongLogs = 0;
#pragma unroll
    for (uint32_t a = 0; a < 2u; a++) {
        const uint64_t some = __builtin_amdgcn_ballot_w64(mine[a] != 0u), over16 = __builtin_amdgcn_ballot_w64(mine[a] > 16u);
        const uint64_t over32 = __builtin_amdgcn_ballot_w64(mine[a] > 32u), longLog = __builtin_amdgcn_ballot_w64(mine[a] > 64u);
        const uint64_t rowsSome = (some | (some >> 32)) & 0xffffffffull, rowsOver16 = (over16 | (over16 >> 32)) & 0xffffffffull;
        const uint64_t rowsOver32 = (over32 | (over32 >> 32)) & 0xffffffffull, rowsLong = (longLog | (longLog >> 32)) & 0xffffffffull;
        rowsWithLongLogs |= rowsLong << (32u * a);
        rowsWithRecords |= (rowsOver32 & ~rowsLong) << (32u * a);
        rowsWithMediumLogs |= (rowsOver16 & ~rowsOver32) << (32u * a);
        rowsWithShortLogs |= (rowsSome & ~rowsOver16) << (32u * a);
    }
    if ((rowsWithShortLogs | rowsWithMediumLogs | rowsWithRecords | rowsWithLongLogs) == 0ull) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // (the logs: written by other lanes, read here with plain loads)
    const uint64_t validRows = __builtin_amdgcn_ballot_w64(rowValid);
    const uint32_t rowOfWave = uniform(row - lane);
    ReplayArgs args;
    {
        ArgsPtr aux = kernelArgs();
        args.lists = aux->buffers + size_t(listBlock) * 64u * twoK;
        args.inbox = aux->inbox;
        args.acceptMaxByKey = aux->acceptMaxByKey;
        args.keyOfMismatch = aux->keyOfMismatch;
        args.k = aux->k;
        args.twoK = twoK;
        args.columnShift = 13u + aux->rowBits;
        args.vectorMemoryIssued = 0u;
        args.firstColumn = firstColumn;
        args.selfPairs = !emitColumns;          // (rows that scan all columns themselves meet their own cells; a triangle's rows lie behind their columns)
    }
    Entry* const listOfWave = args.lists;
    // logical record i of a log that holds `stored` records and is read from record `first` on
    auto physical = [](uint32_t i, uint32_t first, uint32_t stored) { return first + i < stored ? first + i : first + i - stored; };
    auto countsOf = [&](uint32_t r, uint32_t h, uint32_t& n, uint32_t& first, uint32_t& stored) {
        const uint32_t owner = (r & 31u) + 32u * h;
        const uint32_t stored0 = uint32_t(__builtin_amdgcn_readlane(int(stored0Mine), int(owner)));
        const uint32_t stored1 = uint32_t(__builtin_amdgcn_readlane(int(stored1Mine), int(owner)));
        const uint32_t first0 = uint32_t(__builtin_amdgcn_readlane(int(first0Mine), int(owner)));
        const uint32_t first1 = uint32_t(__builtin_amdgcn_readlane(int(first1Mine), int(owner)));
        stored = r < 32u ? stored0 : stored1;
        first = r < 32u ? first0 : first1;
        n = all ? stored : stored - first;
    };
    auto issue = [&](ReplayRowLoads& slot, uint32_t ringSlotLds) {
        // (no branch around the loads, not even when no row is left -- the slot then loads row 0's first places for nobody: a
        // slot that kept its old registers on one path would be copied behind the loads of the other, and waited for there)
        const bool any = rowsWithRecords != 0ull;
        const uint32_t r = any ? uint32_t(__builtin_ctzll(rowsWithRecords)) : 0u;
        rowsWithRecords &= rowsWithRecords - 1ull;           // (0 stays 0)
        slot.row = any ? r : 64u;
        uint32_t firstA, storedA, firstB, storedB;
        countsOf(r, 0u, slot.nA, firstA, storedA);
        countsOf(r, 1u, slot.nB, firstB, storedB);
        slot.nA = any ? slot.nA : 0u;
        slot.nB = any ? slot.nB : 0u;
        const WalkRecord* logA = walkLogOf(waveLog, logCapacity, r & 31u, r >> 5);
        const WalkRecord* logB = walkLogOf(waveLog, logCapacity, (r & 31u) + 32u, r >> 5);
        // (every lane loads, lanes without a record the log's first place: a load under a condition is a value that must be
        // merged with its alternative right behind it, i.e. waited for at once -- and the point is to wait four rows later)
        issueRecordLoads(ringSlotLds, logA, lane < slot.nA ? physical(lane, firstA, storedA) : 0u, logB,
                         63u - lane < slot.nB ? physical(63u - lane, firstB, storedB) : 0u);
        args.vectorMemoryIssued += 2u;
        slot.issuedAt = args.vectorMemoryIssued;
    };
    auto rowState = [&](uint32_t r, uint32_t& countOfRow, int32_t& mMaxOfRow) {
        countOfRow = uint32_t(__builtin_amdgcn_readlane(int(count), int(r)));
        mMaxOfRow = __builtin_amdgcn_readlane(mMax, int(r));
    };
    auto keepRowState = [&](uint32_t r, uint32_t countOfRow, int32_t mMaxOfRow) {
        if (lane == r) {
            count = countOfRow;
            mMax = mMaxOfRow;
        }
    };
    auto process = [&](const ReplayRowLoads& slot, uint32_t ringSlotLds) {
        const uint32_t r = slot.row;
        const WalkRecordWords a0 = ldsPointer<LdsRecordPtr>(ringSlotLds)[lane], b0 = ldsPointer<LdsRecordPtr>(ringSlotLds + kReplayRingSecondLog)[lane];
        uint32_t e[2];
        e[0] = mergeKeyOf(lane < slot.nA, walkRecordOf(a0), 0u, bits, args.firstColumn);
        e[1] = mergeKeyOf(63u - lane < slot.nB, walkRecordOf(b0), 1u, bits, args.firstColumn);
        bitonicMergeWave<2>(e);
        uint32_t countOfRow;
        int32_t mMaxOfRow;
        rowState(r, countOfRow, mMaxOfRow);
        replayRowElements<IDENTITY, 2>(e, slot.nA + slot.nB, lane, rowOfWave + r, emitColumns && ((validRows >> r) & 1ull) != 0ull,
                                       listOfWave + size_t(r) * twoK, args, countOfRow, mMaxOfRow, emitPos, emitEnd, ldsRaw, timed);
        keepRowState(r, countOfRow, mMaxOfRow);
        if (timed) timed[4] += 1u;
    };
    const uint32_t wave = uniform(threadIdx.x >> 6);
    // ---- rows with short logs, two per turn: row X's 32 places in lanes 0..31, row Y's in lanes 32..63; of a row's places the
    // first 16 are its first log's, ascending, the other 16 its second log's, descending ----
    {
        struct PairLoads {
            uint32_t rowX, rowY;            // 64 = none
            uint32_t valid;                 // per lane: its place holds a record (kept in a register: four counts selected by the
                                            // lane's half and log were turned into a table in scratch memory, read -- and waited
                                            // for with everything else in flight -- at every turn)
            uint32_t issuedAt;              // args.vectorMemoryIssued behind the turn's transfer
        };
        const bool upper = lane >= 32u, second = (lane & 16u) != 0u;
        const uint32_t place = second ? 31u - (lane & 31u) : lane & 31u;          // logical index in the lane's log
        auto issuePair = [&](PairLoads& slot, uint32_t ringSlotLds) {
            const bool anyX = rowsWithShortLogs != 0ull;
            const uint32_t rX = anyX ? uint32_t(__builtin_ctzll(rowsWithShortLogs)) : 0u;
            rowsWithShortLogs &= rowsWithShortLogs - 1ull;
            const bool anyY = rowsWithShortLogs != 0ull;
            const uint32_t rY = anyY ? uint32_t(__builtin_ctzll(rowsWithShortLogs)) : 0u;
            rowsWithShortLogs &= rowsWithShortLogs - 1ull;
            slot.rowX = anyX ? rX : 64u;
            slot.rowY = anyY ? rY : 64u;
            uint32_t nAX, nBX, nAY, nBY, firstAX, storedAX, firstBX, storedBX, firstAY, storedAY, firstBY, storedBY;
            countsOf(rX, 0u, nAX, firstAX, storedAX);
            countsOf(rX, 1u, nBX, firstBX, storedBX);
            countsOf(rY, 0u, nAY, firstAY, storedAY);
            countsOf(rY, 1u, nBY, firstBY, storedBY);
            // the lane's log: row X or Y by the lane's half, first or second log by its quarter
            const uint32_t nX = second ? nBX : nAX, nY = second ? nBY : nAY;
            const uint32_t firstX = second ? firstBX : firstAX, firstY = second ? firstBY : firstAY;
            const uint32_t storedX = second ? storedBX : storedAX, storedY = second ? storedBY : storedAY;
            const uint32_t n = upper ? (anyY ? nY : 0u) : (anyX ? nX : 0u);
            const uint32_t first = upper ? firstY : firstX, stored = upper ? storedY : storedX;
            const uint32_t r = upper ? rY : rX;
            const bool valid = place < n;
            slot.valid = valid ? 1u : 0u;
            const WalkRecord* log = walkLogOf(waveLog, logCapacity, (r & 31u) + (second ? 32u : 0u), r >> 5);
            issueRecordLoad(ringSlotLds, reinterpret_cast<uint64_t>(log + (valid ? physical(place, first, stored) : 0u)));
            args.vectorMemoryIssued += 1u;
            slot.issuedAt = args.vectorMemoryIssued;
        };
        auto processPair = [&](const PairLoads& slot, uint32_t ringSlotLds) {
            const uint64_t tp0 = timed ? __builtin_readcyclecounter() : 0ull;
            const WalkRecordWords w = ldsPointer<LdsRecordPtr>(ringSlotLds)[lane];
            uint32_t e = mergeKeyOf(slot.valid != 0u, walkRecordOf(w), second ? 1u : 0u, bits, args.firstColumn);
            bitonicMergeHalves(e);
            if (timed) {
                asm volatile("" : "+v"(e));
                timed[1] += __builtin_readcyclecounter() - tp0;
                timed[3] += 1u;
            }
            const uint32_t rX = slot.rowX, rY = slot.rowY < 64u ? slot.rowY : slot.rowX;          // (no second row: nothing of it is active)
            uint32_t countX, countY;
            int32_t mMaxX, mMaxY;
            rowState(rX, countX, mMaxX);
            rowState(rY, countY, mMaxY);
            replayPairElements<IDENTITY>(e, lane, rX, rY, rowOfWave, emitColumns && ((validRows >> rX) & 1ull) != 0ull,
                                         emitColumns && slot.rowY < 64u && ((validRows >> rY) & 1ull) != 0ull, args, countX, mMaxX, countY,
                                         mMaxY, emitPos, emitEnd, ldsRaw, timed);
            keepRowState(rX, countX, mMaxX);
            if (slot.rowY < 64u) keepRowState(rY, countY, mMaxY);
        };
        const uint32_t ring0 = replayRingSlot(tilesLds, wave, 0u), ring1 = replayRingSlot(tilesLds, wave, 1u);
        const uint32_t ring2 = replayRingSlot(tilesLds, wave, 2u), ring3 = replayRingSlot(tilesLds, wave, 3u);
        PairLoads s0, s1, s2, s3;
        issuePair(s0, ring0);
        issuePair(s1, ring1);
        issuePair(s2, ring2);
        issuePair(s3, ring3);
        for (;;) {
            {
                const uint64_t tw0 = timed ? __builtin_readcyclecounter() : 0ull;
                awaitVectorMemory(uniform(args.vectorMemoryIssued - s0.issuedAt));
                if (timed) timed[0] += __builtin_readcyclecounter() - tw0;
            }
            if (s0.rowX >= 64u) break;
            processPair(s0, ring0);
            {
                const uint64_t ti0 = timed ? __builtin_readcyclecounter() : 0ull;
                issuePair(s0, ring0);
                if (timed) timed[7] += __builtin_readcyclecounter() - ti0;
            }
            {
                const uint64_t tw0 = timed ? __builtin_readcyclecounter() : 0ull;
                awaitVectorMemory(uniform(args.vectorMemoryIssued - s1.issuedAt));
                if (timed) timed[0] += __builtin_readcyclecounter() - tw0;
            }
            if (s1.rowX >= 64u) break;
            processPair(s1, ring1);
            {
                const uint64_t ti0 = timed ? __builtin_readcyclecounter() : 0ull;
                issuePair(s1, ring1);
                if (timed) timed[7] += __builtin_readcyclecounter() - ti0;
            }
            {
                const uint64_t tw0 = timed ? __builtin_readcyclecounter() : 0ull;
                awaitVectorMemory(uniform(args.vectorMemoryIssued - s2.issuedAt));
                if (timed) timed[0] += __builtin_readcyclecounter() - tw0;
            }
            if (s2.rowX >= 64u) break;
            processPair(s2, ring2);
            {
                const uint64_t ti0 = timed ? __builtin_readcyclecounter() : 0ull;
                issuePair(s2, ring2);
                if (timed) timed[7] += __builtin_readcyclecounter() - ti0;
            }
            {
                const uint64_t tw0 = timed ? __builtin_readcyclecounter() : 0ull;
                awaitVectorMemory(uniform(args.vectorMemoryIssued - s3.issuedAt));
                if (timed) timed[0] += __builtin_readcyclecounter() - tw0;
            }
            if (s3.rowX >= 64u) break;
            processPair(s3, ring3);
            {
                const uint64_t ti0 = timed ? __builtin_readcyclecounter() : 0ull;
                issuePair(s3, ring3);
                if (timed) timed[7] += __builtin_readcyclecounter() - ti0;
            }
        }
        awaitVectorMemory(0u);
    }
    // ---- rows whose logs hold 32 records at most each, one per turn: the first log's 32 places in lanes 0..31, ascending, the
    // second log's in lanes 32..63, descending; one transfer, one key per lane, six stages ----
    {
        struct MediumLoads {
            uint32_t row;                   // 64 = none
            uint32_t valid;                 // per lane: its place holds a record
            uint32_t n;                     // the row's records
            uint32_t issuedAt;
        };
        const bool second = lane >= 32u;
        const uint32_t place = second ? 63u - lane : lane;
        auto issueMedium = [&](MediumLoads& slot, uint32_t ringSlotLds) {
            const bool any = rowsWithMediumLogs != 0ull;
            const uint32_t r = any ? uint32_t(__builtin_ctzll(rowsWithMediumLogs)) : 0u;
            rowsWithMediumLogs &= rowsWithMediumLogs - 1ull;
            slot.row = any ? r : 64u;
            uint32_t nA, nB, firstA, storedA, firstB, storedB;
            countsOf(r, 0u, nA, firstA, storedA);
            countsOf(r, 1u, nB, firstB, storedB);
            slot.n = any ? nA + nB : 0u;
            const uint32_t n = any ? (second ? nB : nA) : 0u;
            const uint32_t first = second ? firstB : firstA, stored = second ? storedB : storedA;
            const bool valid = place < n;
            slot.valid = valid ? 1u : 0u;
            const WalkRecord* log = walkLogOf(waveLog, logCapacity, (r & 31u) + (second ? 32u : 0u), r >> 5);
            issueRecordLoad(ringSlotLds, reinterpret_cast<uint64_t>(log + (valid ? physical(place, first, stored) : 0u)));
            args.vectorMemoryIssued += 1u;
            slot.issuedAt = args.vectorMemoryIssued;
        };
        auto processMedium = [&](const MediumLoads& slot, uint32_t ringSlotLds) {
            const WalkRecordWords w = ldsPointer<LdsRecordPtr>(ringSlotLds)[lane];
            uint32_t e[1] = {mergeKeyOf(slot.valid != 0u, walkRecordOf(w), second ? 1u : 0u, bits, args.firstColumn)};
            bitonicStage<32u, 1>(e);
            bitonicStage<16u, 1>(e);
            bitonicStage<8u, 1>(e);
            bitonicStage<4u, 1>(e);
            bitonicStage<2u, 1>(e);
            bitonicStage<1u, 1>(e);
            const uint32_t r = slot.row;
            uint32_t countOfRow;
            int32_t mMaxOfRow;
            rowState(r, countOfRow, mMaxOfRow);
            replayRowElements<IDENTITY, 1>(e, slot.n, lane, rowOfWave + r, emitColumns && ((validRows >> r) & 1ull) != 0ull,
                                           listOfWave + size_t(r) * twoK, args, countOfRow, mMaxOfRow, emitPos, emitEnd, ldsRaw, timed);
            keepRowState(r, countOfRow, mMaxOfRow);
        };
        const uint32_t ring0 = replayRingSlot(tilesLds, wave, 0u), ring1 = replayRingSlot(tilesLds, wave, 1u);
        const uint32_t ring2 = replayRingSlot(tilesLds, wave, 2u), ring3 = replayRingSlot(tilesLds, wave, 3u);
        MediumLoads s0, s1, s2, s3;
        issueMedium(s0, ring0);
        issueMedium(s1, ring1);
        issueMedium(s2, ring2);
        issueMedium(s3, ring3);
        for (;;) {
            awaitVectorMemory(uniform(args.vectorMemoryIssued - s0.issuedAt));
            if (s0.row >= 64u) break;
            processMedium(s0, ring0);
            issueMedium(s0, ring0);
            awaitVectorMemory(uniform(args.vectorMemoryIssued - s1.issuedAt));
            if (s1.row >= 64u) break;
            processMedium(s1, ring1);
            issueMedium(s1, ring1);
            awaitVectorMemory(uniform(args.vectorMemoryIssued - s2.issuedAt));
            if (s2.row >= 64u) break;
            processMedium(s2, ring2);
            issueMedium(s2, ring2);
            awaitVectorMemory(uniform(args.vectorMemoryIssued - s3.issuedAt));
            if (s3.row >= 64u) break;
            processMedium(s3, ring3);
            issueMedium(s3, ring3);
        }
        awaitVectorMemory(0u);
    }
    {
        // four rows in flight: a row's two transfers have the six of the three rows behind it younger than themselves at its turn
        const uint32_t ring0 = replayRingSlot(tilesLds, wave, 0u), ring1 = replayRingSlot(tilesLds, wave, 1u);
        const uint32_t ring2 = replayRingSlot(tilesLds, wave, 2u), ring3 = replayRingSlot(tilesLds, wave, 3u);
        ReplayRowLoads s0, s1, s2, s3;
        issue(s0, ring0);
        issue(s1, ring1);
        issue(s2, ring2);
        issue(s3, ring3);
        for (;;) {
            awaitVectorMemory(uniform(args.vectorMemoryIssued - s0.issuedAt));
            if (s0.row >= 64u) break;
            process(s0, ring0);
            issue(s0, ring0);
            awaitVectorMemory(uniform(args.vectorMemoryIssued - s1.issuedAt));
            if (s1.row >= 64u) break;
            process(s1, ring1);
            issue(s1, ring1);
            awaitVectorMemory(uniform(args.vectorMemoryIssued - s2.issuedAt));
            if (s2.row >= 64u) break;
            process(s2, ring2);
            issue(s2, ring2);
            awaitVectorMemory(uniform(args.vectorMemoryIssued - s3.issuedAt));
            if (s3.row >= 64u) break;
            process(s3, ring3);
            issue(s3, ring3);
        }
        awaitVectorMemory(0u);          // (the transfers that were issued for nobody: the tile buffers go back to the walk)
    }
    const uint64_t tr1 = timed ? __builtin_readcyclecounter() : 0ull;
    if (timed) timed[5] += tr1 - tr0;
    // ---- rows with a log of more than 64 records: two records per lane and log ----
    while (rowsWithLongLogs != 0ull) {
        if (timed) timed[6] += 1u;
        const uint32_t r = uint32_t(__builtin_ctzll(rowsWithLongLogs));
        rowsWithLongLogs &= rowsWithLongLogs - 1ull;
        uint32_t nA, firstA, storedA, nB, firstB, storedB;
        countsOf(r, 0u, nA, firstA, storedA);
        countsOf(r, 1u, nB, firstB, storedB);
        const WalkRecord* logA = walkLogOf(waveLog, logCapacity, r & 31u, r >> 5);
        const WalkRecord* logB = walkLogOf(waveLog, logCapacity, (r & 31u) + 32u, r >> 5);
        const WalkRecord a0 = loadWalkRecord(logA, lane < nA ? physical(lane, firstA, storedA) : 0u);
        const WalkRecord a1 = loadWalkRecord(logA, 64u + lane < nA ? physical(64u + lane, firstA, storedA) : 0u);
        const WalkRecord b0 = loadWalkRecord(logB, 63u - lane < nB ? physical(63u - lane, firstB, storedB) : 0u);
        const WalkRecord b1 = loadWalkRecord(logB, 127u - lane < nB ? physical(127u - lane, firstB, storedB) : 0u);
        // the sequence: the first log's 128 places ascending, then the second log's 128 places descending
        uint32_t e[4];
        e[0] = mergeKeyOf(lane < nA, a0, 0u, bits, args.firstColumn);
        e[1] = mergeKeyOf(64u + lane < nA, a1, 0u, bits, args.firstColumn);
        e[2] = mergeKeyOf(127u - lane < nB, b1, 1u, bits, args.firstColumn);
        e[3] = mergeKeyOf(63u - lane < nB, b0, 1u, bits, args.firstColumn);
        bitonicMergeWave<4>(e);
        uint32_t countOfRow;
        int32_t mMaxOfRow;
        rowState(r, countOfRow, mMaxOfRow);
        replayRowElements<IDENTITY, 4>(e, nA + nB, lane, rowOfWave + r, emitColumns && ((validRows >> r) & 1ull) != 0ull,
                                       listOfWave + size_t(r) * twoK, args, countOfRow, mMaxOfRow, emitPos, emitEnd, ldsRaw, timed);
        keepRowState(r, countOfRow, mMaxOfRow);
    }
}

// The tile kernel of the sharded scan defers both sides: every lane empties its own two logs, order is irrelevant (the
// inbox is sorted).  rowBase = cell id of the wave's row 0.
template <bool WIDE = false>
__device__ __forceinline__ void drainWalkLogs(const WalkRecord* waveLog, uint32_t logCapacity, const uint32_t (&recordCount)[2], uint32_t lane,
                                              uint32_t rowBase, uint32_t cellCount, uint32_t& emitPos, uint32_t& emitEnd)
{
    const int32_t* snap = kernelArgs()->snap;
    if (__builtin_amdgcn_ballot_w64((recordCount[0] | recordCount[1]) != 0u) == 0ull) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // (plain loads of what this lane stored through the L2)
#pragma unroll
    for (uint32_t a = 0; a < 2u; a++) {
        const WalkRecord* log = walkLogOf(waveLog, logCapacity, lane, a);
        const uint32_t rowId = rowBase + 32u * a + (lane & 31u);
        const int32_t snapOfRow = rowId < cellCount ? snap[rowId] : -1;
        for (uint32_t i = 0;; ++i) {
            const bool active = i < recordCount[a];
            if (__builtin_amdgcn_ballot_w64(active) == 0ull) break;
            WalkRecord r = WalkRecord();
            if (active) r = loadWalkRecord(log, i);
            const uint32_t col = walkRecordColumn(r.code, lane >> 5);
            const uint32_t m = mismatchesOfMatrixResult<WIDE>(r.dot);
            const bool valid = active && rowId < cellCount;
            emitColumn(valid && r.dot >= r.bound, col, rowId, m, lane, emitPos, emitEnd);             // target col
            emitColumn(valid && int32_t(m) <= snapOfRow, rowId, col, m, lane, emitPos, emitEnd);      // target row
        }
    }
}


// ---- host-side constants and small helpers of both units ----

// entries of Fsp4Args::terms: one per cell, padded with the last cell's to whole pairs of tiles and one pair more (the walk
// stages the terms of a pair of tiles 64 lanes wide without looking at the end)
static size_t matrixTermCount(uint32_t cellCount) { return size_t((cellCount + 63u) / 64u) * 64u + 64u; }

// dynamic LDS of the matrix kernels behind matrixLdsOffset: four tiles, the stop words + ticket, the waves' walk blocks
constexpr size_t kMatrixLdsBytes = 4u * kMatrixTileWords * 16u + 64u + 4u * kMatrixWalkLdsBytes;

// The wave's walk block (16-byte aligned: the steps read it 16 bytes at a time) inside its selection area of `stride` bytes: at
// the area's begin, or 8 bytes in where an odd k puts that at 8 (mod 16) -- until round 6 an odd k kept the blocks apart, and
// from k = 93 on that cost the kernel its second block per CU.
__host__ __device__ static inline bool matrixWalkBlockInSelectionArea(uint32_t stride)
{
    return stride >= kMatrixWalkLdsBytes + (stride % 16u);
}
__host__ __device__ static inline uint32_t matrixWalkBlockOffsetInSelectionAreas(uint32_t stride, uint32_t wave)
{
    const uint32_t begin = wave * stride;
    return begin + (16u - begin % 16u) % 16u;
}

// fsp4ScanMatrixKernel: the waves' walk blocks alias their selection areas when those are large enough (see there)
static bool matrixWalkAliasesSelection(uint32_t k)
{
    return matrixWalkBlockInSelectionArea(2u * k * kLdsBytesPerEntrySlot);
}

static size_t scanMatrixLdsBytes(uint32_t k)
{
    return kMatrixLdsBytes - (matrixWalkAliasesSelection(k) ? 4u * kMatrixWalkLdsBytes : 0u);
}

constexpr uint32_t kSymmetricMinCells = 131072;
constexpr uint32_t kSymmetricMatrixMinCells = 32768;
constexpr uint32_t kMaxSegments = 64;
constexpr uint32_t kMatrixMaxSegments = 1024;    // (room for short segments: 2048 columns of 2048-bit fragments are the 2 MB an XCD's L2 holds)
constexpr uint32_t kTableWords = 2u * kMatrixMaxSegments + 2u;
constexpr uint32_t kInboxChunk = 512;

// Which signature widths take the matrix-core form of the triangle.  The fragments are always 1024 bits wide: a
// narrower signature is zero-extended (a zero bit adds nothing to either popcount or to the dot product),
// which costs the full 16 k-steps per tile whatever the width.  EM2_SCAN_MATRIX: 0 never, 1 (default) the widths it
// is faster for (129..1024 bits, kMatrixMinPaddedDw), 2 every width up to 1024 bits (tests), 3 = 1 without the 2048-bit form.
static bool matrixFormWanted(uint32_t paddedDw)
{
    const uint64_t mode = envNumber("EM2_SCAN_MATRIX", 1);
    if (mode == 0 || paddedDw > 32u) return false;
    return mode == 2 || paddedDw >= kMatrixMinPaddedDw;
}

// 1025..2048-bit signatures (64 dwords as the scan sees them): the 2048-bit form of the matrix kernel
// (fsp4ScanMatrixWideKernel: 32 rows per wave and pass, two passes).  EM2_SCAN_MATRIX=0 / 3 keep
// the v_xor/v_bcnt form.
static bool matrixWideWanted(uint32_t paddedDw)
{
    const uint64_t mode = envNumber("EM2_SCAN_MATRIX", 1);
    return paddedDw == 64u && mode != 0 && mode != 3;
}

// (the double-buffer form of the sort: the two pools are its two buffers, and its temporary storage is the digit counts alone --
// the keys-in / keys-out form asks for a third buffer of the pool's size on top)
static size_t inboxSortTempBytesDoubleBuffer(uint64_t capacity)
{
    size_t bytes = 0;
    uint64_t* none = nullptr;
    rocprim::double_buffer<uint64_t> keys(none, none);
    if (rocprim::radix_sort_keys(nullptr, bytes, keys, size_t(capacity), 0u, 64u, hipStream_t(nullptr)) != hipSuccess) return 0;
    return bytes;
}

static size_t inboxSortTempBytes(uint64_t capacity)
{
    size_t bytes = 0;
    uint64_t* none = nullptr;
    if (rocprim::radix_sort_keys(nullptr, bytes, none, none, size_t(capacity), 0u, 64u, hipStream_t(nullptr)) != hipSuccess) return 0;
    return bytes;
}

}  // namespace

// ---- what em2_scan_symmetric.hip exports to em2_scan_sharded.hip (kernels are taken by address, launched by the caller) ----
const void* fsp4SymmetricKernelFor(uint32_t paddedDw, bool identity);          // fsp4ScanSymmetricKernel<paddedDw, identity>
const void* scanMatrixKernelFor(bool identity, bool wide = false);             // the matrix-core scan kernel the settings select
hipError_t residentWaveSlots(const void* kernel, uint32_t wavesPerBlock, size_t lds, uint32_t* slots);
// (terms: matrixTermCount(cellCount) floats for the 1024-bit steps' row and column terms, or null)
hipError_t launchExpandFragments(const uint32_t* sig32, uint32_t cellCount, uint32_t fragmentCount, void* out, uint32_t steps,
                                 float* terms, hipStream_t stream);
hipError_t launchInboxReplay(bool identity, dim3 grid, dim3 block, size_t lds, hipStream_t stream, const Fsp4Args& args,
                             const uint64_t* sorted, uint64_t count);

}  // namespace em2

#endif
