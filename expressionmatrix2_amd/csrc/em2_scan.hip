// em2_scan.hip -- findSimilarPairs4 on gfx950: all-pairs Hamming scan with the reference's per-cell
// candidate selection, bit-identical to src/ExpressionMatrixLsh.cpp:200-285 + src/SimilarPairs.cpp:369-405.
//
// Per-cell contract (equivalent to the reference's 64x64 blocked loop, which offers every
// unordered pair to both of its cells, because each cell sees the other cells in ascending id order either
// way): for a cell c, candidates o = 0..N-1, o != c, arrive in ascending order; a candidate with mismatch
// count m is accepted iff m <= mMax(c) (integer form of the two floating-point tests, em2_tables.h), appended
// to the cell's list; when the list holds 2k entries it is cut to k with std::nth_element semantics
// (em2_select.h) and mMax(c) is re-derived from the entry that ends at position k-1.
//
// Mapping to CDNA4:
//   * one LANE owns one row (cell c).  Its signature (W32 32-bit words) lives in VGPRs for the whole kernel.
//   * columns are wave-uniform: the column signature is streamed through the SCALAR unit (s_load_dwordx16 from
//     the cell-major signature array, which is one linear stream) and enters the vector ALU as the SGPR
//     operand of v_xor_b32; v_bcnt_u32_b32 accumulates the popcount.  2 VALU instructions per 32 bits per
//     (row, column) pair is the instruction floor for XOR+popcount on this ISA; there is no LDS traffic, no
//     vector memory traffic and no barrier in the steady-state loop.
//   * scalar loads are software-pipelined one chunk (<= 32 dwords) ahead behind an explicit lgkmcnt(0)
//     (SMEM returns out of order, so the wait precedes the next issue).
//   * a lane whose candidate passes appends {column, key} to its row's list in HBM; lists that reach 2k are
//     staged into LDS by the whole wave and cut by one lane running the exact introselect emulation.
//   * every wave walks the columns in the same order, so the 128 MB (1M cells x 1024 bit) stream is shared
//     through the scalar caches / L2 / Infinity Cache.

#include "em2_device.h"
#include "em2_select_wave.h"

#include <cstdlib>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

namespace em2 {
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
__device__ __forceinline__ Entry loadEntryCoherent(const Entry* p)
{
    const uint64_t v = __hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
    Entry e;
    e.cell = uint32_t(v);
    e.key = uint32_t(v >> 32);
    return e;
}

__device__ __forceinline__ void storeEntry(Entry* p, uint32_t cell, uint32_t key)
{
    *reinterpret_cast<uint64_t*>(p) = uint64_t(cell) | (uint64_t(key) << 32);
}

// m += popcount(x) as ONE instruction.  The compiler usually forms v_bcnt_u32_b32 with its free accumulate from
// __builtin_popcount(x) + m, but in some instantiations it reassociates the 32 additions into a v_add3_u32 tree
// (+16 VALU instructions per column, measured in the .s); the asm pins the chain.
__device__ __forceinline__ void popcountAccumulate(uint32_t& m, uint32_t x)
{
    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(m) : "v"(x));
}

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
    uint32_t pad;
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
};

constexpr uint32_t kShardNoFinish = 1u;       // full-row blocks publish their state instead of finishing the rows
constexpr uint32_t kShardPublishAll = 2u;     // full-row blocks publish snapshots as well
constexpr uint32_t kShardGlobalOutput = 4u;   // outPairs / outUsed are indexed by global cell id

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
__device__ __attribute__((noinline)) uint32_t cutListToBest(Entry* lds, Entry* g, uint32_t n, uint32_t k,
                                                            uint32_t lane, bool writeBack)
{
    for (uint32_t i = lane; i < n; i += 64u) lds[i] = loadEntryCoherent(g + i);
    waveLdsFence();
    // This wave's LDS area is [2k entries][2k uint16][2k uint16] (kLdsBytesPerEntrySlot each); n <= 2k.
    uint16_t* Lpos = reinterpret_cast<uint16_t*>(lds + 2u * k);
    nthElementWave(lds, Lpos, Lpos + 2u * k, int(k), int(n), lane);
    if (writeBack) {
        for (uint32_t i = lane; i < k; i += 64u) g[i] = lds[i];
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
            const uint32_t backKey = cutListToBest(lds, g, twoK, k, lane, true);
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
            cutListToBest(lds, g, n, k, lane, false);
            n = k;
        } else {
            for (uint32_t i = lane; i < n; i += 64u) lds[i] = loadEntryCoherent(g + i);
            waveLdsFence();
        }
        PairOut* out = aux->outPairs + size_t(outBase + src) * k;
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
        if (lane == 0u) aux->outUsed[outBase + src] = n;
        waveLdsFence();
    }
}


// =========================================================================================================
// The scan kernel.  W32 = dwords per signature, R = rows owned by each lane (the wave owns 64*R rows).
//
// Measured on MI355X (profiles/r01_pmc_1Mcells_scan_projection.json, profiles/r01_ubench_valu_xor_bcnt.txt):
//   * v_xor_b32 / v_bcnt_u32_b32 issue at one wave64 instruction per 4 clocks per SIMD (16 lanes/clk):
//     SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU quad-cycles, clock 2.38 GHz (GRBM_GUI_ACTIVE), so the
//     instruction roofline of this formulation is 256 CU x 4 SIMD x 16 lanes x 2.4 GHz / (4*W lane-ops per
//     comparison) = 6.1e11 ordered comparisons/s at 1024 bits; this kernel keeps the VALUs 89% busy at 1M cells.
//   * the scalar path is NOT the limiter: R = 1, 2, 4 (2x / 4x fewer scalar loads per comparison) run within
//     3% of each other, R = 1 fastest (most waves).  R stays a template parameter for experiments
//     (EM2_ROWS_PER_LANE); the product path uses R = 1.
//   * the other operand path that keeps the per-pair instruction count at the floor -- column in VGPRs,
//     broadcast with DPP row_newbcast -- was built and measured as well: v_xor_b32_dpp is slower than the
//     SGPR-operand form on gfx950 and the kernel came out 5% slower; removed.
// =========================================================================================================
template <int W32, int R, bool IDENTITY>
__global__ void __launch_bounds__(256)
fsp4ScanKernel(Fsp4Args args)
{
    const uint32_t* __restrict__ sig32 = args.sig32;
    const uint32_t cellCount = args.cellCount;
    constexpr int CH = W32 < 32 ? W32 : 32;      // dwords per scalar-load chunk
    constexpr int H = W32 / CH;                  // chunks per column
    constexpr int U = H < 2 ? 2 : H;             // chunk steps per loop iteration (even, multiple of H)
    constexpr int COLS = U / H;                  // columns per loop iteration

    extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waveIndex = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);

    uint32_t row[R];            // this lane's cell ids: waveRowBase + 64*j + lane
    uint32_t r[R][W32];         // their signatures
    int32_t mMax[R];
    uint32_t count[R];
    Entry* myList[R];           // their candidate lists
    uint32_t twoK;
    {
        ArgsPtr aux = kernelArgs();
        const uint32_t waveRowBase = aux->rowBegin + waveIndex * (64u * R);
        if (waveRowBase >= aux->rowEnd) return;
        twoK = 2u * aux->k;
        asm volatile("" : "+v"(twoK));      // keep it in a VGPR: the scan loop is short of SGPRs, not VGPRs
#pragma unroll
        for (int j = 0; j < R; ++j) {
            row[j] = waveRowBase + 64u * j + lane;
            myList[j] = aux->buffers + (size_t(waveIndex * R + j) * 64u + lane) * twoK;
            const bool rowValid = row[j] < aux->rowEnd;
            const uint32_t* rp = sig32 + size_t(rowValid ? row[j] : waveRowBase) * W32;
#pragma unroll
            for (int w = 0; w < W32; ++w) r[j][w] = rp[w];
            mMax[j] = rowValid ? args.mMaxInitial : -1;
            count[j] = 0;
        }
    }

    ScalarPtr p = (ScalarPtr)(uintptr_t)sig32;
    uint32_t chunk[2][CH];
#pragma unroll
    for (int w = 0; w < CH; ++w) chunk[0][w] = p[w];
    __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0): the row signatures have landed before the loop starts

    uint32_t m[R];
#pragma unroll
    for (int j = 0; j < R; ++j) m[j] = 0;
    for (uint32_t colBase = 0; colBase < cellCount; colBase += COLS) {
#pragma unroll
        for (int s = 0; s < U; ++s) {
            const int part = s % H;
            const uint32_t col = colBase + uint32_t(s / H);
            if (col < cellCount) {
                // The chunk for this step was requested one step ago; wait for it, then request the next
                // one so that its latency is covered by this step's 2*CH*R vector instructions.
                __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0)
                __builtin_amdgcn_sched_barrier(0);
                const bool lastChunk = (col + 1u == cellCount) && (part == H - 1);
                ScalarPtr pn = lastChunk ? p : p + CH;
#pragma unroll
                for (int w = 0; w < CH; ++w) chunk[(s + 1) & 1][w] = pn[w];
                p = pn;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < R; ++j) {
#pragma unroll
                    for (int w = 0; w < CH; ++w) popcountAccumulate(m[j], r[j][part * CH + w] ^ chunk[s & 1][w]);
                }
                if (part == H - 1) {
                    bool any = false;
#pragma unroll
                    for (int j = 0; j < R; ++j) any |= int32_t(m[j]) <= mMax[j];
                    if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
#pragma unroll
                        for (int j = 0; j < R; ++j) {
                            const bool pass = int32_t(m[j]) <= mMax[j];
                            if (__builtin_amdgcn_ballot_w64(pass) != 0ull) {
                                acceptColumn<IDENTITY>(pass, col, row[j], m[j], lane, waveIndex * R + j, myList[j],
                                                       twoK, count[j], mMax[j], ldsRaw);
                            }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < R; ++j) m[j] = 0;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) finishRows(lane, waveIndex * R + j, count[j], ldsRaw);
}

// =========================================================================================================
// Persistent, segment-chained form of the same scan, with speculative look-ahead.
//
// Problem it solves (measured): a wave of fsp4ScanKernel owns 64 rows for ALL columns, so the grid is
// rows/64 equal, indivisible work items: 100k rows = 1563 waves for 1024 SIMDs (half of them get one wave, half
// two: 30 ms against a balanced 16 ms); 125k rows per GPU (1M cells on 8 GPUs) = 1954 waves.
//
// Here the columns are cut into S segments and a work item is (segment s, row block b).  Resident waves (4 per
// SIMD, the measured optimum) take items from one ticket counter in segment-major order (t -> s = t / B,
// b = t % B), so all row blocks advance together.  The per-cell contract needs block b's segments IN ORDER:
//   * exact path: if item (s-1,b) has already published its state -- per-row {count,mMax} plus the row lists in
//     HBM, with an agent-scope release -- item (s,b) acquires it and scans its columns exactly like
//     fsp4ScanKernel (cdna_hip_programming.md Guideline 16: stores, vmcnt(0), release fence, flag; poll,
//     acquire fence, loads);
//   * speculative path: if it has not (fewer row blocks than resident waves), item (s,b) does not idle: it scans
//     its columns against a SNAPSHOT of the rows' cut-offs (the last published mMax, or the initial one) and
//     LOGS every (column, mismatch) that passes.  Cut-offs only ever tighten, so the log is a superset, in
//     ascending column order, of what the exact state machine can accept in this segment.  Once (s-1,b) has
//     published, the log is replayed through the exact state machine (same append / keepBest code); columns that
//     are not in the log would have been rejected anyway.  A log that fills up (capacity per row = logCapacity)
//     stops the speculation at that column; the rest of the segment is scanned exactly after the hand-off.
// No deadlock: item (s-1,b) holds a lower ticket, tickets are only taken by running waves, and the chain ends
// at s = 0 which waits for nothing.  Spins are bounded (~4 s) and raise an error flag instead of hanging.
// =========================================================================================================

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
                for (int w = 0; w < CH; ++w) popcountAccumulate(m, r[part * CH + w] ^ chunk[s & 1][w]);
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

template <int W32, bool IDENTITY>
__global__ void __launch_bounds__(256)
fsp4ScanPersistentKernel(Fsp4Args args)
{
    const uint32_t* __restrict__ sig32 = args.sig32;
    const uint32_t cellCount = args.cellCount;

    extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
    const uint32_t lane = threadIdx.x & 63u;

    for (;;) {
        // ---- take the next work item (one ticket per WAVE: a variant with one ticket per 16-wave workgroup, whose
        // waves then stream the same columns in step, measured 43% slower: waves in step stall on their scalar
        // loads together) ----
        uint32_t ticket = 0;
        if (lane == 0u) {
            ticket = __hip_atomic_fetch_add(kernelArgs()->control, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ticket = uint32_t(__builtin_amdgcn_readfirstlane(int(ticket)));

        uint32_t colBegin, colEnd, row;
        uint32_t r[W32];
        int32_t mMax;
        uint32_t count = 0;
        Entry* myList;
        Entry* myLog;
        uint32_t twoK, logCapacity;
        uint32_t logCount = 0;
        bool rowValid;
        bool speculate = false;
        {
            ArgsPtr aux = kernelArgs();
            const uint32_t rowBlocks = aux->rowBlocks;
            if (ticket >= rowBlocks * aux->segments) return;
            const uint32_t seg = ticket / rowBlocks;
            const uint32_t block = ticket - seg * rowBlocks;
            twoK = 2u * aux->k;
            logCapacity = aux->logCapacity;
            myList = aux->buffers + (size_t(block) * 64u + lane) * twoK;
            myLog = aux->logs + (size_t(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64u + lane) * logCapacity;
            asm volatile("" : "+v"(twoK));          // VGPRs: the scan loop is short of SGPRs, not VGPRs
            asm volatile("" : "+v"(logCapacity));
            colBegin = seg * aux->columnsPerSegment;
            colEnd = colBegin + aux->columnsPerSegment;
            if (colEnd > cellCount || seg + 1u == aux->segments) colEnd = cellCount;
            row = aux->rowBegin + block * 64u + lane;
            rowValid = row < aux->rowEnd;
            const uint32_t* rp = sig32 + size_t(rowValid ? row : aux->rowBegin + block * 64u) * W32;
#pragma unroll
            for (int w = 0; w < W32; ++w) r[w] = rp[w];
            mMax = rowValid ? args.mMaxInitial : -1;
            if (seg != 0u) {
                const uint32_t done = uint32_t(__builtin_amdgcn_readfirstlane(
                    int(__hip_atomic_load(aux->segmentsDone + block, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))));
                speculate = done < seg;
                if (speculate && done != 0u) {
                    // snapshot of the cut-offs some earlier segment published: a valid (looser or equal) bound
                    const uint64_t st = __hip_atomic_load(reinterpret_cast<const uint64_t*>(aux->rowState) + size_t(block) * 64u + lane,
                                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    mMax = rowValid ? int32_t(uint32_t(st >> 32)) : -1;
                }
            }
        }

        uint32_t resume = colBegin;
        if (speculate) {
            resume = scanColumns<W32, IDENTITY, true>(sig32, colBegin, colEnd, r, row, lane, ticket, myList, twoK, count, mMax,
                                                      myLog, logCapacity, logCount, ldsRaw);
        }

        if (ticket >= kernelArgs()->rowBlocks) {
            // ---- wait for the previous segment of this row block, then take over its exact state ----
            ArgsPtr aux = kernelArgs();
            const uint32_t seg = ticket / aux->rowBlocks;
            const uint32_t block = ticket - seg * aux->rowBlocks;
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
                return;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const uint64_t st = __hip_atomic_load(reinterpret_cast<const uint64_t*>(aux->rowState) + size_t(block) * 64u + lane,
                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            count = uint32_t(st);
            mMax = rowValid ? int32_t(uint32_t(st >> 32)) : -1;

            // ---- replay the speculative log through the exact state machine (ascending column order per row) ----
            if (speculate) {
                const uint32_t block2 = block;
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
                        acceptColumn<IDENTITY>(pass, c, row, m, lane, block2, myList, twoK, count, mMax, ldsRaw);
                    }
                }
            }
        }

        // ---- exact scan of whatever the speculation did not cover (all of the segment on the exact path) ----
        scanColumns<W32, IDENTITY, false>(sig32, resume, colEnd, r, row, lane, ticket, myList, twoK, count, mMax, myLog,
                                          logCapacity, logCount, ldsRaw);

        // ---- last segment: finish the rows; otherwise publish the state for the next segment ----
        {
            ArgsPtr aux = kernelArgs();
            const uint32_t seg = ticket / aux->rowBlocks;
            const uint32_t block = ticket - seg * aux->rowBlocks;
            if (seg + 1u == aux->segments) {
                finishRows(lane, block, count, ldsRaw);
            } else {
                const uint64_t st = uint64_t(count) | (uint64_t(uint32_t(mMax)) << 32);
                __hip_atomic_store(reinterpret_cast<uint64_t*>(aux->rowState) + size_t(block) * 64u + lane, st,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0u) {
                    __hip_atomic_store(aux->segmentsDone + block, seg + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

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
                for (int w = 0; w < CH; ++w) popcountAccumulate(m, r[part * CH + w] ^ chunk[s & 1][w]);
                if (part == H - 1) {
                    // one compare in the steady state: m against the looser of the row's and the column's cut-off
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
            while (ticket >= table[seg + 1u]) ++seg;
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
                if (at_ >= triEnd2_) {                                                                                      \
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
    uint64_t bound[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint64_t target = uint64_t(row + uint32_t(j)) << nb;
        uint64_t lo = 0, hi = sortedCount;
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo) / 2u;
            if (((sorted[mid] >> 13u) & fieldMask) < target) lo = mid + 1u;
            else hi = mid;
        }
        bound[j] = lo;
    }
    if (!rowValid) bound[1] = bound[0];
    const uint32_t idMask = (1u << nb) - 1u;
    for (uint64_t i = bound[0];; ++i) {
        const bool active = i < bound[1];
        if (__builtin_amdgcn_ballot_w64(active) == 0ull) break;
        uint32_t c = 0, m = 0;
        if (active) {
            const uint64_t e = sorted[i];
            c = uint32_t(e >> 13u) & idMask;
            m = uint32_t(e) & 0x1fffu;
        }
        const bool pass = active && int32_t(m) <= mMax;
        if (__builtin_amdgcn_ballot_w64(pass) != 0ull) {
            acceptColumn<IDENTITY>(pass, c, row, m, lane, block, myList, twoK, count, mMax, ldsRaw);
        }
    }
    finishRows(lane, block, count, ldsRaw);
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
                for (int w = 0; w < CH; ++w) popcountAccumulate(m, r[part * CH + w] ^ chunk[s & 1][w]);
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

// max over `count` arrays of `n` int32 laid out back to back (the emulation's stand-in for all_reduce(MAX))
__global__ void maxReduceKernel(int32_t* __restrict__ arrays, uint32_t n, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t best = arrays[i];
    for (uint32_t a = 1; a < count; ++a) best = arrays[size_t(a) * n + i] > best ? arrays[size_t(a) * n + i] : best;
    for (uint32_t a = 0; a < count; ++a) arrays[size_t(a) * n + i] = best;
}

__global__ void repackSignaturesKernel(const uint64_t* __restrict__ src, uint32_t cellCount, uint32_t wordCount,
                                       uint32_t* __restrict__ dst, uint32_t paddedDw)
{
    const uint64_t total = uint64_t(cellCount) * paddedDw;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
         i += uint64_t(gridDim.x) * blockDim.x) {
        const uint64_t cell = i / paddedDw;
        const uint32_t dw = uint32_t(i % paddedDw);
        uint32_t v = 0;
        if (dw < 2u * wordCount) {
            const uint64_t word = src[cell * wordCount + (dw >> 1)];
            v = (dw & 1u) ? uint32_t(word >> 32) : uint32_t(word);
        }
        dst[i] = v;
    }
}

}  // namespace


uint32_t paddedDwords(uint32_t lshCount)
{
    if (lshCount == 0) return 0;
    const uint32_t dw = 2u * ((lshCount - 1u) / 64u + 1u);
    uint32_t p = 2;
    while (p < dw) p <<= 1;
    return p <= 128u ? p : 0u;
}

// EM2_ROWS_PER_LANE=2 selects two rows per lane for 1024/2048-bit signatures (A/B measurements only).
static uint32_t forcedRowsPerLane()
{
    const char* v = getenv("EM2_ROWS_PER_LANE");
    return (v && atoi(v) == 2) ? 2u : 0u;
}

uint32_t fsp4MaxK()
{
    return kLdsBytesPerBlock / (2u * kLdsBytesPerEntrySlot);
}

hipError_t launchRepackSignatures(const uint64_t* src, uint32_t cellCount, uint32_t wordCount, uint32_t* dst,
                                  uint32_t paddedDw, hipStream_t stream)
{
    const uint64_t total = uint64_t(cellCount) * paddedDw;
    if (total == 0) return hipSuccess;
    uint64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    repackSignaturesKernel<<<dim3(uint32_t(blocks)), dim3(256), 0, stream>>>(src, cellCount, wordCount, dst,
                                                                              paddedDw);
    return hipGetLastError();
}

constexpr uint32_t kLogCapacity = 256;       // speculative log entries per row (2 KB per row, 128 KB per wave)

// Upper bound of the waves a persistent launch keeps resident on the current device (4 per SIMD).
static uint32_t maxResidentWaves()
{
    int device = 0, cuCount = 0;
    if (hipGetDevice(&device) != hipSuccess ||
        hipDeviceGetAttribute(&cuCount, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cuCount <= 0) {
        cuCount = 304;
    }
    return uint32_t(cuCount) * 16u;
}

static size_t align256(size_t x) { return (x + 255u) & ~size_t(255u); }

size_t fsp4ControlBytes(uint32_t rowCount)
{
    const size_t rowBlocks = (size_t(rowCount) + 63u) / 64u;
    // rowState (8 B per row of every block) + segmentsDone (4 B per block) + control words + speculative logs
    return align256(rowBlocks * 64u * 8u) + align256(rowBlocks * 4u) + 256u +
           align256(size_t(maxResidentWaves()) * 64u * kLogCapacity * sizeof(Entry));
}

// EM2_SCAN_MODE=simple selects the one-wave-per-row-block kernel (A/B measurements); default is the persistent
// segment-chained kernel.
static bool scanModeIsSimple()
{
    const char* v = getenv("EM2_SCAN_MODE");
    return v && v[0] == 's';
}

// ---- symmetric (triangle) scan: eligibility and workspace ----
// EM2_SCAN_MODE=triangle forces it wherever it is possible (all rows of the problem in one launch),
// EM2_SCAN_MODE=persistent / simple disable it; by default it is used from kSymmetricMinCells cells on.
constexpr uint32_t kSymmetricMinCells = 131072;
constexpr uint32_t kMaxSegments = 64;
constexpr uint32_t kInboxChunk = 512;

static uint64_t envNumber(const char* name, uint64_t fallback)
{
    const char* v = getenv(name);
    if (!v || !*v) return fallback;
    char* end = nullptr;
    const unsigned long long x = strtoull(v, &end, 10);
    return end == v ? fallback : uint64_t(x);
}

static bool symmetricEligible(uint32_t cellCount, uint32_t rowCount)
{
    if (rowCount != cellCount || cellCount < 128u) return false;
    const char* v = getenv("EM2_SCAN_MODE");
    if (v && v[0] == 't') return true;
    if (v && (v[0] == 's' || v[0] == 'p')) return false;
    return cellCount >= envNumber("EM2_SYMMETRIC_MIN_CELLS", kSymmetricMinCells);
}

static uint64_t inboxCapacity(uint32_t cellCount)
{
    // EM2_INBOX_CAPACITY (entries) is a test knob: tiny pools force the overflow -> ordered-scan fallback.
    const uint64_t forced = envNumber("EM2_INBOX_CAPACITY", 0);
    if (forced >= kInboxChunk) return forced < 0xfff00000ull ? forced : 0xfff00000ull;
    uint64_t cap = uint64_t(cellCount) * envNumber("EM2_INBOX_PER_CELL", 1024);
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
    size_t snap, table, control, poolA, poolB, temp, total, tempBytes;
    uint64_t capacity;
};

static SymmetricLayout symmetricLayout(uint32_t cellCount)
{
    SymmetricLayout l;
    l.capacity = inboxCapacity(cellCount);
    l.tempBytes = inboxSortTempBytes(l.capacity);
    size_t at = 0;
    l.snap = at;    at += align256(size_t(cellCount) * 4u);
    l.table = at;   at += align256((2u * kMaxSegments + 2u) * 4u);
    l.control = at; at += 256u;
    l.poolA = at;   at += align256(size_t(l.capacity) * 8u);
    l.poolB = at;   at += align256(size_t(l.capacity) * 8u);
    l.temp = at;    at += align256(l.tempBytes);
    l.total = at;
    return l;
}

bool fsp4UsesSymmetricScan(uint32_t cellCount, uint32_t rowCount)
{
    return symmetricEligible(cellCount, rowCount);
}

size_t fsp4SymmetricBytes(uint32_t cellCount, uint32_t rowCount)
{
    if (!symmetricEligible(cellCount, rowCount)) return 0;
    return symmetricLayout(cellCount).total;
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

// What the last launch on this thread did (benchmarks and logs): see em2_dev_find_similar_pairs4_last_launch.
static thread_local Fsp4LaunchInfo lastLaunchInfo = {0, -1.0, 0.0, 0.0, 0.0, 0.0};

Fsp4LaunchInfo fsp4LastLaunchInfo() { return lastLaunchInfo; }

// The symmetric scan (see fsp4ScanSymmetricKernel).  *done = false when the inbox pool overflowed: nothing usable
// was produced and the caller runs the ordered scan instead.  Synchronises the stream (the sort size is read back).
static hipError_t launchFsp4ScanSymmetric(Fsp4Args args, uint32_t paddedDw, bool identity, uint32_t wavesPerBlock,
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

    // Segments: as many as the column-count floor allows, up to kMaxSegments (EM2_SEGMENTS overrides): short
    // segments keep the column snapshots fresh and even out the triangle.
    uint64_t minSegmentColumns = envNumber("EM2_MIN_SEGMENT_COLUMNS", 4096);
    if (minSegmentColumns < 1) minSegmentColumns = 1;
    uint64_t segments = cellCount / minSegmentColumns;
    if (segments > kMaxSegments) segments = kMaxSegments;
    const uint64_t forcedSegments = envNumber("EM2_SEGMENTS", 0);
    if (forcedSegments >= 1 && forcedSegments <= kMaxSegments) segments = forcedSegments;
    if (segments < 1) segments = 1;
    const uint32_t cps = uint32_t((uint64_t(cellCount) + segments - 1u) / segments);
    segments = (uint64_t(cellCount) + cps - 1u) / cps;

    uint32_t table[2u * kMaxSegments + 2u];
    uint64_t tickets = 0;
    for (uint32_t sIdx = 0; sIdx < segments; ++sIdx) {
        uint32_t firstTriangle = uint32_t((uint64_t(sIdx) * cps) / 64u);
        if (firstTriangle < fullRowBlocks) firstTriangle = fullRowBlocks;
        table[sIdx] = uint32_t(tickets);
        table[segments + 1u + sIdx] = firstTriangle;
        tickets += fullRowBlocks + (rowBlocks - firstTriangle);
        if (tickets >= 0xffffffffull) return hipErrorInvalidValue;
    }
    table[segments] = uint32_t(tickets);

    const SymmetricLayout layout = symmetricLayout(cellCount);
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
    if (getenv("EM2_DEBUG_NO_EMIT")) {      // timing experiment only: results are wrong
        e = hipMemsetAsync(args.snap, 0xff, size_t(cellCount) * 4u, stream);
        if (e != hipSuccess) return e;
    }
    e = hipMemcpyAsync(ws + layout.table, table, (2u * segments + 2u) * 4u, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;

    uint64_t wavesWanted = tickets;
    if (wavesWanted > slots) wavesWanted = slots;
    if (wavesWanted > maxResidentWaves()) wavesWanted = maxResidentWaves();
    const dim3 block(64u * wavesPerBlock);
    const dim3 grid(uint32_t((wavesWanted + wavesPerBlock - 1u) / wavesPerBlock));
    void* kernelArgsArray[] = {&args};
    static thread_local hipEvent_t timing[2] = {nullptr, nullptr};
    if (!timing[0]) {
        if (hipEventCreate(&timing[0]) != hipSuccess || hipEventCreate(&timing[1]) != hipSuccess) timing[0] = timing[1] = nullptr;
    }
    if (timing[0]) (void)hipEventRecord(timing[0], stream);
    e = hipLaunchKernel(kernel, grid, block, kernelArgsArray, lds, stream);
    if (e != hipSuccess) return e;
    if (timing[0]) (void)hipEventRecord(timing[1], stream);

    // the number of inbox entries (incl. chunk tails), the overflow flag and the hand-off error word
    uint32_t inboxWords[4] = {0, 0, 0, 0};
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
    {
        float ms = -1.0f;
        if (!timing[0] || hipEventElapsedTime(&ms, timing[0], timing[1]) != hipSuccess) ms = -1.0f;
        double steps = double(fullRowBlocks) * double(cellCount);       // (wave, column) steps of the scan kernel
        for (uint32_t b = fullRowBlocks; b < rowBlocks; ++b) {
            const uint64_t end = uint64_t(b) * 64u + 64u;
            steps += double(end < cellCount ? end : cellCount);
        }
        lastLaunchInfo.form = 1;
        lastLaunchInfo.scanKernelMs = double(ms);
        lastLaunchInfo.waveColumnSteps = steps;
        lastLaunchInfo.inboxEntries = double(used);
        lastLaunchInfo.segments = double(segments);
        lastLaunchInfo.fullRowCells = double(fullCellsClamped);
    }
    if (const char* v = getenv("EM2_SCAN_VERBOSE")) {
        if (v[0] == '1') fprintf(stderr, "[em2] symmetric scan: %u segments x %u columns, %u full-row blocks, %llu tickets, %llu inbox slots\n",
                                 uint32_t(segments), cps, fullRowBlocks, (unsigned long long)tickets, (unsigned long long)used);
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
    uint64_t cap = uint64_t(cellCount) * envNumber("EM2_INBOX_PER_CELL", 1024) / world;
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
    if (world == 0 || rank >= world || k == 0 || p.blocks < 4u * world) return p;     // not eligible: too small
    // prefix: EM2_PREFIX_PERMILLE of the cells (default 200), a positive multiple of `world` blocks
    uint64_t prefixBlocks = (uint64_t(p.blocks) * envNumber("EM2_PREFIX_PERMILLE", 200) / 1000u + world / 2u) / world * world;
    if (prefixBlocks < world) prefixBlocks = world;
    if (prefixBlocks > uint64_t(p.blocks) - world) prefixBlocks = (uint64_t(p.blocks) - world) / world * world;
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
    p.offTable = at;        at += align256((2u * 256u + 2u) * 4u);
    p.offInboxControl = at; at += 256u;
    p.offPool = at;         at += align256(size_t(p.capLocal) * 8u);
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
        const uint32_t span = cellCount - M;
        uint64_t segments = span / 1024u;
        if (segments > 256) segments = 256;
        const uint64_t forced = envNumber("EM2_TILE_SEGMENTS", 0);
        if (forced >= 1 && forced <= 256) segments = forced;
        if (segments < 1) segments = 1;
        const uint32_t cps = uint32_t((uint64_t(span) + segments - 1u) / segments);
        segments = (uint64_t(span) + cps - 1u) / cps;
        uint32_t table[2u * 256u + 2u];
        uint64_t tiles = 0;
        for (uint32_t sIdx = 0; sIdx < segments; ++sIdx) {
            const uint32_t firstBlock = (M + sIdx * cps) / 64u;
            table[sIdx] = uint32_t(tiles);
            table[segments + 1u + sIdx] = firstBlock;
            tiles += plan.blocks - firstBlock;
            if (tiles >= 0xffffffffull) return hipErrorInvalidValue;
        }
        table[segments] = uint32_t(tiles);
        const uint64_t own = tiles > plan.rank ? (tiles - plan.rank + plan.world - 1u) / plan.world : 0u;
        if (own == 0) return hipSuccess;
        {
            double steps = 0.0;         // this rank's share of the tiles' (wave, column) steps
            for (uint32_t b = plan.prefixBlocks; b < plan.blocks; ++b) {
                const uint64_t end = uint64_t(b) * 64u + 64u;
                steps += double((end < cellCount ? end : cellCount) - M);
            }
            lastLaunchInfo.waveColumnSteps += steps / double(plan.world);
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

// All ranks of the sharded scan played one after the other on this GPU (tests; EM2_SCAN_MODE=virtual with
// EM2_VIRTUAL_WORLD=P).  *done = false: not eligible or an entry pool overflowed; the caller runs the ordered scan.
static hipError_t runFsp4ShardedEmulation(const uint32_t* sig32, uint32_t paddedDw, uint32_t cellCount, uint32_t k,
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
    }
    mark();
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    if (verbose) {
        fprintf(stderr, "[em2] sharded emulation: world %u, prefix %u cells, entries per rank (max) %llu;", world, p0.prefixCells,
                (unsigned long long)maxUsed);
        size_t at = 0;
        for (int phase = 0; phase < 4; ++phase) {
            fprintf(stderr, " phase %d ms:", phase);
            for (uint32_t r = 0; r < world; ++r) {
                float ms = 0;
                (void)hipEventElapsedTime(&ms, events[at], events[at + 1]);
                fprintf(stderr, " %.2f", ms);
                ++at;
            }
            ++at;
        }
        fprintf(stderr, "\n");
    }
    for (hipEvent_t ev : events) (void)hipEventDestroy(ev);
    *done = true;
    return hipSuccess;
}

hipError_t launchFsp4Scan(const uint32_t* sig32, uint32_t paddedDw, uint32_t cellCount, uint32_t rowBegin,
                          uint32_t rowEnd, uint32_t k, const DeviceTables& t, Entry* buffers, PairOut* outPairs,
                          uint32_t* outUsed, void* control, hipStream_t stream, void* symmetricWs)
{
    if (rowEnd <= rowBegin) return hipSuccess;
    if (k == 0 || k > fsp4MaxK()) return hipErrorInvalidValue;
    const uint32_t bytesPerWave = 2u * k * kLdsBytesPerEntrySlot;
    uint32_t wavesPerBlock = kLdsBytesPerBlock / bytesPerWave;
    if (wavesPerBlock > 4) wavesPerBlock = 4;
    const uint32_t rows = rowEnd - rowBegin;
    const uint32_t rowBlocks = (rows + 63u) / 64u;

    const bool identity = t.identityKeys;
    Fsp4Args args;
    args.sig32 = sig32;
    args.cellCount = cellCount;
    args.mMaxInitial = t.mMaxInitial;
    args.keyOfMismatch = t.keyOfMismatch;
    args.acceptMaxByKey = t.acceptMaxByKey;
    args.keySimilarity = t.keySimilarity;
    args.buffers = buffers;
    args.outPairs = outPairs;
    args.outUsed = outUsed;
    args.k = k;
    args.rowBegin = rowBegin;
    args.rowEnd = rowEnd;
    args.pad = 0;
    args.rowState = nullptr;
    args.segmentsDone = nullptr;
    args.control = nullptr;
    args.rowBlocks = rowBlocks;
    args.segments = 1;
    args.columnsPerSegment = cellCount;
    args.logCapacity = 0;
    args.logs = nullptr;
    args.snap = nullptr;
    args.inbox = nullptr;
    args.inboxControl = nullptr;
    args.segTable = nullptr;
    args.inboxCapacity = 0;
    args.inboxChunk = 0;
    args.fullRowBlocks = 0;
    args.rowBits = 0;
    args.totalTickets = 0;
    args.rowBlockStride = 1;
    args.rowBlockOffset = 0;
    args.localBlockBase = 0;
    args.columnLimit = cellCount;
    args.shardFlags = 0;

    {
        // EM2_SCAN_MODE=virtual + EM2_VIRTUAL_WORLD=P: the multi-GPU symmetric scan with all ranks played on this GPU
        const char* mode = getenv("EM2_SCAN_MODE");
        if (mode && mode[0] == 'v' && rowBegin == 0 && rows == cellCount) {
            uint64_t world = envNumber("EM2_VIRTUAL_WORLD", 2);
            if (world < 1) world = 1;
            bool done = false;
            const hipError_t ev = runFsp4ShardedEmulation(sig32, paddedDw, cellCount, k, t, outPairs, outUsed, uint32_t(world), stream, &done);
            if (ev != hipSuccess) return ev;
            if (done) return hipSuccess;
        }
    }

    if (control && symmetricWs && rowBegin == 0 && symmetricEligible(cellCount, rows)) {
        bool done = false;
        const hipError_t es = launchFsp4ScanSymmetric(args, paddedDw, identity, wavesPerBlock,
                                                      size_t(wavesPerBlock) * bytesPerWave, control, symmetricWs, stream, &done);
        if (es != hipSuccess) return es;
        if (done) return hipSuccess;
        // inbox overflow: fall through to the ordered scan, which starts from scratch
    }

    lastLaunchInfo.form = 0;
    lastLaunchInfo.scanKernelMs = -1.0;
    lastLaunchInfo.waveColumnSteps = double(rowBlocks) * double(cellCount);
    lastLaunchInfo.inboxEntries = 0.0;
    lastLaunchInfo.segments = 1.0;
    lastLaunchInfo.fullRowCells = double(rows);

    if (scanModeIsSimple() || !control) {
        uint32_t rowsPerLane = forcedRowsPerLane();
        if (rowsPerLane == 0 || paddedDw >= 128) rowsPerLane = 1;
        const uint32_t waves = (rows + 64u * rowsPerLane - 1u) / (64u * rowsPerLane);
        if (wavesPerBlock > waves) wavesPerBlock = waves;
        const dim3 grid((waves + wavesPerBlock - 1u) / wavesPerBlock);
        const dim3 block(64u * wavesPerBlock);
        const size_t lds = size_t(wavesPerBlock) * bytesPerWave;
#define EM2_LAUNCH_SCAN(W32, RR)                                                          \
    do {                                                                                  \
        if (identity) fsp4ScanKernel<W32, RR, true><<<grid, block, lds, stream>>>(args);  \
        else fsp4ScanKernel<W32, RR, false><<<grid, block, lds, stream>>>(args);          \
    } while (0)
#define EM2_LAUNCH_SCAN_R(W32)                                  \
    do {                                                        \
        if (rowsPerLane == 2) EM2_LAUNCH_SCAN(W32, 2);          \
        else EM2_LAUNCH_SCAN(W32, 1);                           \
    } while (0)
        switch (paddedDw) {
        case 2: EM2_LAUNCH_SCAN(2, 1); break;
        case 4: EM2_LAUNCH_SCAN(4, 1); break;
        case 8: EM2_LAUNCH_SCAN(8, 1); break;
        case 16: EM2_LAUNCH_SCAN(16, 1); break;
        case 32: EM2_LAUNCH_SCAN_R(32); break;
        case 64: EM2_LAUNCH_SCAN_R(64); break;
        case 128: EM2_LAUNCH_SCAN(128, 1); break;
        default: return hipErrorInvalidValue;
        }
#undef EM2_LAUNCH_SCAN
#undef EM2_LAUNCH_SCAN_R
        return hipGetLastError();
    }

    // ---- persistent, segment-chained launch ----
    const dim3 block(64u * wavesPerBlock);
    const size_t lds = size_t(wavesPerBlock) * bytesPerWave;
    const void* kernel = nullptr;
#define EM2_PERSISTENT(W32) \
    (identity ? reinterpret_cast<const void*>(&fsp4ScanPersistentKernel<W32, true>) \
              : reinterpret_cast<const void*>(&fsp4ScanPersistentKernel<W32, false>))
    switch (paddedDw) {
    case 2: kernel = EM2_PERSISTENT(2); break;
    case 4: kernel = EM2_PERSISTENT(4); break;
    case 8: kernel = EM2_PERSISTENT(8); break;
    case 16: kernel = EM2_PERSISTENT(16); break;
    case 32: kernel = EM2_PERSISTENT(32); break;
    case 64: kernel = EM2_PERSISTENT(64); break;
    case 128: kernel = EM2_PERSISTENT(128); break;
    default: return hipErrorInvalidValue;
    }
#undef EM2_PERSISTENT
    int device = 0, cuCount = 0, blocksPerCu = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    e = hipDeviceGetAttribute(&cuCount, hipDeviceAttributeMultiprocessorCount, device);
    if (e != hipSuccess) return e;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocksPerCu, kernel, int(block.x), lds);
    if (e != hipSuccess) return e;
    if (blocksPerCu < 1) blocksPerCu = 1;
    {
        // Resident workgroups per CU.  Measured at 1M cells x 1024 bit (scan ms): 2 -> 2025, 3 -> 1755, 4 -> 1729,
        // 5 -> 1936: a fifth wave per SIMD costs more in the scalar-load path than it hides, so the default is
        // min(occupancy limit, 4 waves per SIMD).  EM2_BLOCKS_PER_CU overrides (measurements only).
        int wanted = int(16u / wavesPerBlock);
        if (wanted < 1) wanted = 1;
        const char* v = getenv("EM2_BLOCKS_PER_CU");
        if (v && atoi(v) >= 1) wanted = atoi(v);
        if (wanted < blocksPerCu) blocksPerCu = wanted;
    }
    const uint32_t slots = uint32_t(cuCount) * uint32_t(blocksPerCu) * wavesPerBlock;     // resident waves
    // Segments: enough work items (~32 per resident wave) for an even finish, at least 4096 columns each.
    uint64_t segments = (32ull * slots + rowBlocks - 1u) / rowBlocks;
    // Test knobs (tests/test_gpu_fsp4.py drives the hand-off, speculation and log-overflow paths at small sizes
    // with them): EM2_MIN_SEGMENT_COLUMNS (default 4096), EM2_LOG_CAPACITY (default, and maximum, kLogCapacity).
    uint64_t minSegmentColumns = 4096;
    if (const char* v = getenv("EM2_MIN_SEGMENT_COLUMNS")) {
        if (atoi(v) >= 1) minSegmentColumns = uint64_t(atoi(v));
    }
    const uint64_t maxByColumns = cellCount / minSegmentColumns;
    if (segments > maxByColumns) segments = maxByColumns;
    if (segments > 64) segments = 64;
    if (segments < 1) segments = 1;
    const uint32_t columnsPerSegment = uint32_t((uint64_t(cellCount) + segments - 1u) / segments);
    segments = (uint64_t(cellCount) + columnsPerSegment - 1u) / columnsPerSegment;
    if (uint64_t(rowBlocks) * segments >= 0xffffffffull) return hipErrorInvalidValue;

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
    args.columnsPerSegment = columnsPerSegment;
    lastLaunchInfo.segments = double(segments);
    e = hipMemsetAsync(c + stateBytes, 0, doneBytes + 256u, stream);
    if (e != hipSuccess) return e;

    uint64_t wavesWanted = uint64_t(rowBlocks) * segments;
    if (wavesWanted > slots) wavesWanted = slots;
    if (wavesWanted > maxResidentWaves()) wavesWanted = maxResidentWaves();       // the log area is sized for this
    const dim3 grid(uint32_t((wavesWanted + wavesPerBlock - 1u) / wavesPerBlock));
    void* kernelArgsArray[] = {&args};
    e = hipLaunchKernel(kernel, grid, block, kernelArgsArray, lds, stream);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

// Reads the error word a persistent launch raises when a hand-off wait timed out (never observed; a bounded
// spin is the alternative to a hung GPU).  Synchronises the stream.
hipError_t readFsp4Error(const void* control, uint32_t rowCount, hipStream_t stream, uint32_t* error)
{
    const size_t rowBlocks = (size_t(rowCount) + 63u) / 64u;
    const size_t stateBytes = align256(rowBlocks * 64u * 8u);
    const size_t doneBytes = align256(rowBlocks * 4u);
    uint32_t words[2] = {0, 0};
    hipError_t e = hipMemcpyAsync(words, static_cast<const char*>(control) + stateBytes + doneBytes, sizeof(words),
                                  hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(stream);
    *error = words[1];
    return e;
}

}  // namespace em2
