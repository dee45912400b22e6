// em2_scan_common.h -- what the translation units of the findSimilarPairs4 scan share: the kernel-argument
// struct, the per-row state machine (append, keepBest cut, finish) and the ordered column scan as device code, and a
// few host helpers.  Device functions live in an anonymous namespace on purpose: each .hip file gets its own copy.
// em2_scan.hip holds the ordered kernels and the launcher, em2_scan_symmetric.hip the symmetric / sharded forms.
#ifndef EM2_SCAN_COMMON_H
#define EM2_SCAN_COMMON_H

#include "em2_device.h"
#include "em2_select_wave.h"

#include <cstdlib>
#include <cstring>

namespace em2 {

// Kernel arguments, passed by value as ONE struct so that the kernarg segment is exactly this struct.
// The steady-state loop reads only sig32 / cellCount / mMaxInitial.  Everything else is needed by the rare
// path and the epilogue only; they re-read it from the kernarg segment through a laundered pointer so that
// the loop keeps its SGPRs for the two column chunks (a build that kept these values live across the loop
// spilled SGPRs into VGPR lanes inside it).
struct Fsp4Args {
    const uint32_t* sig32;
    uint32_t cellCount;
    int32_t mMaxInitial;
    const uint32_t* keyOfMismatch;
    const int32_t* acceptMaxByKey;
    const float* keySimilarity;
    Entry* buffers;
    PairOut* outPairs;
    uint32_t* outUsed;
    uint32_t k;
    uint32_t rowBegin;
    uint32_t rowEnd;
    uint32_t pad0;
    // persistent (segment-chained) variant only
    uint32_t* rowState;         // [rowBlocks*64][2] = {count, mMax} handed from one column segment to the next
    uint32_t* segmentsDone;     // [rowBlocks] number of finished column segments of the row block
    uint32_t* control;          // [0] ticket counter, [1] error flag
    uint32_t rowBlocks;
    uint32_t segments;
    uint32_t columnsPerSegment;
    uint32_t logCapacity;       // entries per row of a wave's speculative log
    Entry* logs;                // [resident waves][64][logCapacity]
    // symmetric (each unordered pair once) variant only
    int32_t* snap;              // [cellCount] last published cut-off of every cell; -1 = never emit to this column
    uint64_t* inbox;            // pool of emitted (column, row, mismatch) keys, handed out in chunks
    uint32_t* inboxControl;     // [0..1] 64-bit chunk cursor (entries), [2] overflow flag
    const uint32_t* segTable;   // [0..segments] first ticket of each segment, [segments+1 .. 2*segments] its first triangle block
    uint64_t inboxCapacity;     // entries
    uint32_t inboxChunk;        // entries per chunk (>= 64)
    uint32_t fullRowBlocks;     // row blocks [0, fullRowBlocks) scan every column themselves
    uint32_t rowBits;           // bits of a cell id in an inbox key
    uint32_t totalTickets;
    // row-block mapping (sharded symmetric scan; 1 / 0 / 0 / cellCount / 0 everywhere else): list / state slot b of
    // this launch holds the 64 cells starting at rowBegin + (b * rowBlockStride + rowBlockOffset) * 64
    uint32_t rowBlockStride;
    uint32_t rowBlockOffset;
    uint32_t localBlockBase;    // first list / state slot of this launch (symmetric kernels)
    uint32_t columnLimit;       // columns [0, columnLimit) only (symmetric kernels)
    uint32_t shardFlags;        // kShardNoFinish | kShardPublishAll | kShardGlobalOutput
    // matrix-core form of the symmetric scan only (em2_scan_symmetric.hip)
    const void* fragments;      // the signatures as FP4 in MFMA fragment order: 0 / 1, 512 B per cell (1024-bit steps); +-1, 1 KB per cell (2048-bit steps)
    const float* terms;         // 1024-bit steps: -popcount / 2 of every cell's signature, padded with the last cell's to a multiple of 64 cells + 64
    uint32_t matrixLdsOffset;   // where the column tiles start in the block's dynamic LDS
    uint32_t pad1;
    uint32_t convoy;            // matrix form: 0 = every walk starts at its segment's first column; 1 = a walk joins the walks of its XCD
                                // where they are and wraps around (scanMatrixBody); n >= 2 (tests): every walk starts 64 (n - 1) columns in
    uint32_t rowFragmentBase;   // matrix form: the fragments of list / state slot 0's rows start this many 32-cell blocks into `fragments`
                                // (0 where the rows are cells of the column array itself; the rows form with an unaligned
                                // rowBegin expands its rows once more behind the columns' fragments)
    uint32_t pad2;              // diagnostic build only (EM2_DIAG_WORD below): the EM2_MATRIX_DIAG bits; 0 in the product
};

// Measurement knobs that switch parts of the kernels off (EM2_MATRIX_DIAG, EM2_PROJECTION_DIAG) give wrong results by
// design.  They exist only in the diagnostic build (`make diag` -> libem2lsh_diag.so, compiled with -DEM2_DIAG, loaded by
// the tools through EM2_LIBRARY); the product library neither reads the variables nor contains the branches.
#ifdef EM2_DIAG
#define EM2_DIAG_WORD(args) ((args)->pad2)
#define EM2_DIAG_WORD_OF(args) ((args).pad2)
#else
#define EM2_DIAG_WORD(args) 0u
#define EM2_DIAG_WORD_OF(args) 0u
#endif

constexpr uint32_t kShardNoFinish = 1u;       // full-row blocks publish their state instead of finishing the rows
constexpr uint32_t kShardPublishAll = 2u;     // full-row blocks publish snapshots as well
constexpr uint32_t kShardGlobalOutput = 4u;   // outPairs / outUsed are indexed by global cell id

namespace {

typedef const __attribute__((address_space(4))) uint32_t* ScalarPtr;

constexpr uint32_t kLdsBytesPerBlock = 64u * 1024u;
constexpr uint32_t kLdsBytesPerEntrySlot = uint32_t(sizeof(Entry)) + 2u * uint32_t(sizeof(uint16_t));   // entry + Lpos + Rpos

__device__ __forceinline__ void waveLdsFence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// Entries of a row list are written by one lane and read back by other lanes of the same wave: read them
// at agent scope (L2-served, bypassing the CU's L1) after the writer's vmcnt(0).
// (The lists, the logs and the inbox are global memory, but their addresses come out of the kernel-argument block, which the
// kernels read through a laundered pointer: the compiler cannot know the address space and would access them with flat_
// instructions, which count against lgkmcnt as well as vmcnt and make every LDS wait a wait for the memory traffic too.  The
// helpers below say "global".)
typedef __attribute__((address_space(1))) uint64_t* GlobalWordPtr;
typedef const __attribute__((address_space(1))) uint64_t* GlobalConstWordPtr;
__device__ __forceinline__ Entry loadEntryCoherent(const Entry* p)
{
    const uint64_t v = __hip_atomic_load((GlobalConstWordPtr)reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
    Entry e;
    e.cell = uint32_t(v);
    e.key = uint32_t(v >> 32);
    return e;
}

__device__ __forceinline__ void storeEntry(Entry* p, uint32_t cell, uint32_t key)
{
    *(GlobalWordPtr)reinterpret_cast<uint64_t*>(p) = uint64_t(cell) | (uint64_t(key) << 32);
}

// an inbox entry (or any 64-bit word of global memory)
__device__ __forceinline__ void storeGlobalWord(uint64_t* p, uint64_t v)
{
    *(GlobalWordPtr)p = v;
}

// m += popcount(x) as ONE instruction.  The compiler usually forms v_bcnt_u32_b32 with its free accumulate from
// __builtin_popcount(x) + m, but in some instantiations it reassociates the 32 additions into a v_add3_u32 tree
// (+16 VALU instructions per column, measured in the .s); the asm pins the chain.
__device__ __forceinline__ void popcountAccumulate(uint32_t& m, uint32_t x)
{
    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(m) : "v"(x));
}

// m = popcount(x): the first word of a column starts the sum from the inline constant 0, which saves the v_mov that
// would otherwise reset the accumulator once per column.
__device__ __forceinline__ void popcountFirst(uint32_t& m, uint32_t x)
{
    asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(m) : "v"(x));
}


typedef const __attribute__((address_space(4))) Fsp4Args* ArgsPtr;

__device__ __forceinline__ ArgsPtr kernelArgs()
{
    ArgsPtr p = (ArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// Cuts the list at g (n entries) to its best k exactly as keepBest does, staging it through this wave's LDS
// area.  Out of line on purpose: inlining the selection's nested loops into the scan kernel raised SGPR
// pressure enough to spill the column chunk registers inside the steady-state loop.
// Returns the key of the entry that ends at position k-1 (tmp.back(), ExpressionMatrixLsh.cpp:249,256).
// The LDS area arrives as an LDS pointer (address space 3): through a generic pointer -- all an out-of-line function can know
// of its argument -- every access to it was a flat_ instruction that waited for ALL of the wave's outstanding memory traffic
// (vmcnt(0) and lgkmcnt(0): the lists' and the inbox's stores, the replay's loads in flight), some 140 times per selection.
typedef __attribute__((address_space(3))) Entry* LdsEntryPtr;
__device__ __attribute__((noinline)) uint32_t cutListToBest(LdsEntryPtr ldsArea, Entry* g, uint32_t n, uint32_t k,
                                                            uint32_t lane, bool writeBack)
{
    Entry* lds = (Entry*)ldsArea;           // (a cast the compiler sees through: the accesses below are ds_ instructions)
    // The arguments of an out-of-line function arrive in vector registers, and everything derived from them counts as divergent:
    // the selection's loops became loops under EXEC masks.  The counts and the pointers are wave-uniform: say so.  (Both together:
    // the scan kernel on clustered data 235 -> 225 ms, profiles/r06_scan_experiments.md.)
    n = uint32_t(__builtin_amdgcn_readfirstlane(int(n)));
    k = uint32_t(__builtin_amdgcn_readfirstlane(int(k)));
    {
        const uint64_t address = reinterpret_cast<uint64_t>(g);
        g = reinterpret_cast<Entry*>(uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(address))))) |
                                     (uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(address >> 32))))) << 32));
        const uint32_t ldsAddress = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(uintptr_t(ldsArea)))));
        lds = (Entry*)(LdsEntryPtr)(uintptr_t)ldsAddress;
    }
    // (four loads in flight: one load and one LDS store per turn paid a memory round trip per 64 entries -- four of them for
    // the scan's 200-entry lists, a good part of a selection's time)
    for (uint32_t base = 0; base < n; base += 256u) {
        Entry e[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint32_t i = base + 64u * j + lane;
            e[j] = Entry();
            if (i < n) e[j] = loadEntryCoherent(g + i);
        }
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint32_t i = base + 64u * j + lane;
            if (i < n) lds[i] = e[j];
        }
    }
    waveLdsFence();
    // This wave's LDS area is [2k entries][2k uint16][2k uint16] (kLdsBytesPerEntrySlot each); n <= 2k.
    uint16_t* Lpos = reinterpret_cast<uint16_t*>(lds + 2u * k);
    nthElementWave(lds, Lpos, Lpos + 2u * k, int(k), int(n), lane);
    if (writeBack) {
        for (uint32_t i = lane; i < k; i += 64u) {
            const Entry e = lds[i];
            storeEntry(g + i, e.cell, e.key);
        }
    }
    const uint32_t backKey = lds[k - 1u].key;
    return backKey;
}

// ---- rare path: some row of this wave accepts column `col` (mismatch count m in each lane) ----
// Appends {col, key(m)} to the lists of the passing lanes, then cuts every list that reached 2k.
// The append itself touches no kernel argument and waits for nothing (one address computation + one store);
// only a list reaching 2k goes to the kernarg segment.  IDENTITY: float similarities of different mismatch
// counts are all different (true for every lshCount <= 4096 with glibc's cos; checked on the host), so the rank
// key of a mismatch count is the mismatch count itself and no table lookup is needed.
template <bool IDENTITY>
__device__ __forceinline__ void acceptColumn(bool pass, uint32_t col, uint32_t row, uint32_t m, uint32_t lane,
                                             uint32_t listBlock, Entry* myList, uint32_t twoK, uint32_t& count,
                                             int32_t& mMax, unsigned char* ldsRaw)
{
    if (pass && col != row) {
        uint32_t key = m;
        if (!IDENTITY) key = kernelArgs()->keyOfMismatch[m];
        storeEntry(myList + count, col, key);
        ++count;
    }
    uint64_t full = __builtin_amdgcn_ballot_w64(count == twoK);
    if (full != 0ull) {
        ArgsPtr aux = kernelArgs();
        const uint32_t k = aux->k;
        Entry* const waveBuffers = aux->buffers + size_t(listBlock) * 64u * twoK;
        Entry* lds = reinterpret_cast<Entry*>(ldsRaw + size_t(threadIdx.x >> 6) * twoK * kLdsBytesPerEntrySlot);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        do {
            const uint32_t src = uint32_t(__builtin_ctzll(full));
            full &= full - 1ull;
            Entry* g = waveBuffers + size_t(src) * twoK;
            const uint32_t backKey = cutListToBest((LdsEntryPtr)lds, g, twoK, k, lane, true);
            // readfirstlane: the table load completes HERE, so the scan loop never has to wait on vector memory
            const int32_t newMax = __builtin_amdgcn_readfirstlane(aux->acceptMaxByKey[backKey]);
            if (lane == src) {
                count = k;
                mMax = newMax;
            }
            waveLdsFence();
        } while (full != 0ull);
    }
}

// ---- epilogue: final keepBest (ExpressionMatrixLsh.cpp:265-269), SimilarPairs::copy + sort ----
__device__ __forceinline__ void finishRows(uint32_t lane, uint32_t waveIndex, uint32_t count, unsigned char* ldsRaw)
{
    ArgsPtr aux = kernelArgs();
    const uint32_t k = aux->k;
    const uint32_t twoK = 2u * k;
    const uint32_t rowEnd = aux->rowEnd;
    const uint32_t waveRowBase = aux->rowBegin + (waveIndex * aux->rowBlockStride + aux->rowBlockOffset) * 64u;
    // output slot of the wave's first row: its position in the launch, or its global id (sharded scan)
    const uint32_t outBase = (aux->shardFlags & kShardGlobalOutput) ? waveRowBase : waveIndex * 64u;
    Entry* const waveBuffers = aux->buffers + size_t(waveIndex) * 64u * twoK;
    Entry* lds = reinterpret_cast<Entry*>(ldsRaw + size_t(threadIdx.x >> 6) * twoK * kLdsBytesPerEntrySlot);
    const float* keySimilarity = aux->keySimilarity;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    for (uint32_t src = 0; src < 64u; ++src) {
        const uint32_t srow = waveRowBase + src;
        if (srow >= rowEnd) break;
        uint32_t n = uint32_t(__builtin_amdgcn_readlane(int(count), int(src)));
        Entry* g = waveBuffers + size_t(src) * twoK;
        if (n > k) {
            cutListToBest((LdsEntryPtr)lds, g, n, k, lane, false);
            n = k;
        } else {
            for (uint32_t i = lane; i < n; i += 64u) lds[i] = loadEntryCoherent(g + i);
            waveLdsFence();
        }
        PairOut* out = aux->outPairs + size_t(outBase + src) * k;
        sortListWave(lds, n, lane);
        for (uint32_t i = lane; i < n; i += 64u) {
            const Entry e = lds[i];
            PairOut po;
            po.cell = e.cell;
            po.similarity = keySimilarity[e.key];
            out[i] = po;
        }
        for (uint32_t i = n + lane; i < k; i += 64u) {
            PairOut zero;
            zero.cell = 0u;
            zero.similarity = 0.0f;
            out[i] = zero;
        }
        if (lane == 0u) aux->outUsed[outBase + src] = n;
        waveLdsFence();
    }
}

// Scans columns [colBegin, colEnd) for this wave's 64 rows.
//   SPECULATIVE == false: accepted candidates go through acceptColumn (row lists, keepBest); returns colEnd.
//   SPECULATIVE == true : candidates with m <= mMax (a snapshot) are logged per lane; returns the first column
//                         NOT scanned (colEnd, or earlier if some lane's log filled up).
template <int W32, bool IDENTITY, bool SPECULATIVE>
__device__ __forceinline__ uint32_t scanColumns(const uint32_t* __restrict__ sig32, uint32_t colBegin, uint32_t colEnd,
                                                const uint32_t (&r)[W32], uint32_t row, uint32_t lane, uint32_t ticket,
                                                Entry* myList, uint32_t twoK, uint32_t& count, int32_t& mMax,
                                                Entry* myLog, uint32_t logCapacity, uint32_t& logCount,
                                                unsigned char* ldsRaw)
{
    constexpr int CH = W32 < 32 ? W32 : 32;
    constexpr int H = W32 / CH;
    constexpr int U = H < 2 ? 2 : H;
    constexpr int COLS = U / H;
    if (colBegin >= colEnd) return colEnd;
    ScalarPtr p = (ScalarPtr)(uintptr_t)sig32 + size_t(colBegin) * W32;
    uint32_t chunk[2][CH];
#pragma unroll
    for (int w = 0; w < CH; ++w) chunk[0][w] = p[w];
    __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0)
    uint32_t m = 0;
    for (uint32_t colBase = colBegin; colBase < colEnd; colBase += COLS) {
#pragma unroll
        for (int s = 0; s < U; ++s) {
            const int part = s % H;
            const uint32_t col = colBase + uint32_t(s / H);
            if (col < colEnd) {
                __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0)
                __builtin_amdgcn_sched_barrier(0);
                const bool lastChunk = (col + 1u == colEnd) && (part == H - 1);
                ScalarPtr pn = lastChunk ? p : p + CH;
#pragma unroll
                for (int w = 0; w < CH; ++w) chunk[(s + 1) & 1][w] = pn[w];
                p = pn;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int w = 0; w < CH; ++w) {
                    if (part == 0 && w == 0) popcountFirst(m, r[0] ^ chunk[s & 1][0]);
                    else popcountAccumulate(m, r[part * CH + w] ^ chunk[s & 1][w]);
                }
                if (part == H - 1) {
                    const bool pass = int32_t(m) <= mMax;
                    if (__builtin_amdgcn_ballot_w64(pass) != 0ull) {
                        if (SPECULATIVE) {
                            if (pass && col != row) {
                                storeEntry(myLog + logCount, col, m);
                                ++logCount;
                            }
                            if (__builtin_amdgcn_ballot_w64(logCount == logCapacity) != 0ull) return col + 1u;
                        } else {
                            // the row block is recomputed from the ticket here: keeping it live across the
                            // loop cost SGPR spills inside the loop
                            acceptColumn<IDENTITY>(pass, col, row, m, lane, ticket % kernelArgs()->rowBlocks, myList,
                                                   twoK, count, mMax, ldsRaw);
                        }
                    }
                    m = 0;
                }
            }
        }
    }
    return colEnd;
}

}  // namespace

// ---- host helpers ----
constexpr uint32_t kLogCapacity = 256;       // speculative log entries per row (2 KB per row, 128 KB per wave)

// Upper bound of the waves a persistent launch keeps resident on the current device (4 per SIMD).
inline uint32_t maxResidentWaves()
{
    int device = 0, cuCount = 0;
    if (hipGetDevice(&device) != hipSuccess ||
        hipDeviceGetAttribute(&cuCount, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cuCount <= 0) {
        cuCount = 304;
    }
    return uint32_t(cuCount) * 16u;
}

inline size_t align256(size_t x) { return (x + 255u) & ~size_t(255u); }

inline uint64_t envNumber(const char* name, uint64_t fallback)
{
    const char* v = getenv(name);
    if (!v || !*v) return fallback;
    char* end = nullptr;
    const unsigned long long x = strtoull(v, &end, 10);
    return end == v ? fallback : uint64_t(x);
}


// EM2_TIMING set (1: stages synchronised, 2: nothing synchronised): one line per launch of the scan on stderr as well.
inline bool scanVerbose()
{
    const char* v = getenv("EM2_TIMING");
    return v && (v[0] == '1' || v[0] == '2');
}

// The value of a measurement knob: 0 in the product, the environment variable in the diagnostic build.
inline uint64_t diagNumber(const char* name)
{
#ifdef EM2_DIAG
    return envNumber(name, 0);
#else
    (void)name;
    return 0;
#endif
}

// What the last launch on the calling thread did (defined in em2_scan.hip).
extern thread_local Fsp4LaunchInfo lastLaunchInfo;

// em2_scan_symmetric.hip
bool symmetricEligible(uint32_t cellCount, uint32_t rowCount, uint32_t paddedDw);
hipError_t launchFsp4ScanSymmetric(Fsp4Args args, uint32_t paddedDw, bool identity, uint32_t wavesPerBlock, size_t lds,
                                   void* control, void* symmetricWs, hipStream_t stream, bool* done);
bool rowsMatrixEligible(uint32_t cellCount, uint32_t rowCount, uint32_t paddedDw);
hipError_t launchFsp4ScanRowsMatrix(Fsp4Args args, uint32_t paddedDw, bool identity, uint32_t wavesPerBlock, size_t lds,
                                    void* control, void* workspace, bool symmetricWorkspace, hipStream_t stream, bool* done);
hipError_t runFsp4ShardedEmulation(const uint32_t* sig32, uint32_t paddedDw, uint32_t cellCount, uint32_t k,
                                   const DeviceTables& t, PairOut* outPairs, uint32_t* outUsed, uint32_t world,
                                   hipStream_t stream, bool* done);

}  // namespace em2

#endif
