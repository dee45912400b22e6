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

//
// This file: the ordered kernels (one wave per row block; persistent segment-chained) and the launcher.  The shared
// device code is in em2_scan_common.h, the symmetric and sharded forms in em2_scan_symmetric.hip.

#include "em2_scan_common.h"

namespace em2 {
namespace {

// =========================================================================================================
// The scan kernel.  W32 = dwords per signature, R = rows owned by each lane (the wave owns 64*R rows).
//
// Measured on MI355X (profiles/r01_pmc_1Mcells_scan_projection.json, profiles/r01_ubench_valu_xor_bcnt.txt):
//   * v_xor_b32 / v_bcnt_u32_b32 issue at one wave64 instruction per 4 clocks per SIMD (16 lanes/clk):
//     SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU quad-cycles, clock 2.38 GHz (GRBM_GUI_ACTIVE), so the
//     instruction roofline of this formulation is 256 CU x 4 SIMD x 16 lanes x 2.4 GHz / (4*W lane-ops per
//     comparison) = 6.1e11 ordered comparisons/s at 1024 bits; this kernel keeps the VALUs 89% busy at 1M cells.
//   * the scalar path is NOT the limiter: R = 1, 2, 4 (2x / 4x fewer scalar loads per comparison) run within
//     3% of each other, R = 1 fastest (most waves).  R stays a template parameter (narrow signatures use R > 1); for
//     1024 bits and more the product path uses R = 1.
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

// (scanColumns, the column loop of this kernel, is in em2_scan_common.h: the symmetric kernel uses it for its full rows.)

template <int W32, bool IDENTITY>
__global__ void __launch_bounds__(256)
fsp4ScanPersistentKernel(Fsp4Args args)
{
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
        // the ticket and the column range are only needed between the scan loops: parked in VGPRs, so that the loops
        // keep their SGPRs for the column chunks
        uint32_t ticketV = ticket;
        asm volatile("" : "+v"(ticketV));

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
            if (colEnd > aux->cellCount || seg + 1u == aux->segments) colEnd = aux->cellCount;
            row = aux->rowBegin + block * 64u + lane;
            rowValid = row < aux->rowEnd;
            const uint32_t* rp = aux->sig32 + size_t(rowValid ? row : aux->rowBegin + block * 64u) * W32;
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

        uint32_t colEndV = colEnd;
        asm volatile("" : "+v"(colEndV));
        uint32_t resume = colBegin;
        if (speculate) {
            resume = scanColumns<W32, IDENTITY, true>(kernelArgs()->sig32, colBegin, colEnd, r, row, lane, ticketV, myList, twoK,
                                                      count, mMax, myLog, logCapacity, logCount, ldsRaw);
        }
        uint32_t resumeV = resume;
        asm volatile("" : "+v"(resumeV));
        ticket = uint32_t(__builtin_amdgcn_readfirstlane(int(ticketV)));

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
        scanColumns<W32, IDENTITY, false>(kernelArgs()->sig32, uint32_t(__builtin_amdgcn_readfirstlane(int(resumeV))),
                                          uint32_t(__builtin_amdgcn_readfirstlane(int(colEndV))), r, row, lane, ticketV, myList,
                                          twoK, count, mMax, myLog, logCapacity, logCount, ldsRaw);
        ticket = uint32_t(__builtin_amdgcn_readfirstlane(int(ticketV)));

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

// (two rows per lane for 1024/2048-bit signatures was an A/B knob of round 1: within 3 % of one row per lane, never faster)
static uint32_t forcedRowsPerLane() { return 0u; }

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

// What the last launch on this thread did (benchmarks and logs): see em2_dev_find_similar_pairs4_last_launch.
thread_local Fsp4LaunchInfo lastLaunchInfo = {0, -1.0, 0.0, 0.0, 0.0, 0.0, 0.0, -1.0, 0.0};

Fsp4LaunchInfo fsp4LastLaunchInfo() { return lastLaunchInfo; }


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
    args.convoy = 0;
    args.rowFragmentBase = 0;
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
    args.fragments = nullptr;
    args.matrixLdsOffset = 0;
#ifdef EM2_DIAG
    args.pad2 = uint32_t(envNumber("EM2_MATRIX_DIAG", 0));       // measurements only (fsp4ScanMatrixKernel)
#else
    args.pad2 = 0;
#endif

    {
        // EM2_SCAN_MODE=virtual:P -- the multi-GPU symmetric scan with all P ranks played on this GPU ("virtual" alone: 2)
        const char* mode = getenv("EM2_SCAN_MODE");
        if (mode && mode[0] == 'v' && rowBegin == 0 && rows == cellCount) {
            const char* colon = strchr(mode, ':');
            uint64_t world = colon ? strtoull(colon + 1, nullptr, 10) : 2u;
            if (world < 1) world = 1;
            bool done = false;
            const hipError_t ev = runFsp4ShardedEmulation(sig32, paddedDw, cellCount, k, t, outPairs, outUsed, uint32_t(world), stream, &done);
            if (ev != hipSuccess) return ev;
            if (done) return hipSuccess;
        }
    }

    if (control && symmetricWs && rowBegin == 0 && symmetricEligible(cellCount, rows, paddedDw)) {
        bool done = false;
        const hipError_t es = launchFsp4ScanSymmetric(args, paddedDw, identity, wavesPerBlock,
                                                      size_t(wavesPerBlock) * bytesPerWave, control, symmetricWs, stream, &done);
        if (es != hipSuccess) return es;
        if (done) return hipSuccess;
        // Inbox overflow: every row scans all columns itself, from scratch -- on the matrix cores where the signatures have
        // a matrix form (the symmetric workspace holds everything that launch needs), with the ordered scan below otherwise.
        const hipError_t er = launchFsp4ScanRowsMatrix(args, paddedDw, identity, wavesPerBlock, size_t(wavesPerBlock) * bytesPerWave,
                                                       control, symmetricWs, true, stream, &done);
        if (er != hipSuccess) return er;
        if (done) return hipSuccess;
    } else if (control && symmetricWs && rowsMatrixEligible(cellCount, rows, paddedDw)) {
        // a shard of the rows (north_star's partitioning across GPUs) against all columns, on the matrix cores
        bool done = false;
        const hipError_t er = launchFsp4ScanRowsMatrix(args, paddedDw, identity, wavesPerBlock, size_t(wavesPerBlock) * bytesPerWave,
                                                       control, symmetricWs, false, stream, &done);
        if (er != hipSuccess) return er;
        if (done) return hipSuccess;
    }

    lastLaunchInfo.form = 0;
    lastLaunchInfo.scanKernelMs = -1.0;
    lastLaunchInfo.waveColumnSteps = double(rowBlocks) * double(cellCount);
    lastLaunchInfo.inboxEntries = 0.0;
    lastLaunchInfo.segments = 1.0;
    lastLaunchInfo.fullRowCells = double(rows);
    lastLaunchInfo.matrixPairs = 0.0;
    lastLaunchInfo.matrixKernelMs = -1.0;
    lastLaunchInfo.matrixClockGHz = 0.0;

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
