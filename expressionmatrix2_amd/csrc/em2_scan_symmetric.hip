// em2_scan_symmetric.hip -- findSimilarPairs4 with every unordered pair evaluated once: the symmetric scan of one
// GPU (fsp4ScanSymmetricKernel / the matrix-core kernels + inbox sort + fsp4InboxReplayKernel).  The sharded symmetric scan
// across GPUs (prefix phases with the same kernels, tile kernels for the deferred square, replay) is em2_scan_sharded.hip;
// the device code the two share is em2_scan_symmetric_device.h.  Bit-identical to
// src/ExpressionMatrixLsh.cpp:200-285 + src/SimilarPairs.cpp:369-405; see em2_scan.hip for the per-cell contract and
// the CDNA4 mapping of the column loop, em2_scan_common.h for the shared device code.

#include "em2_scan_symmetric_device.h"

namespace em2 {
namespace {

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

// sig32 [cell][2 * steps] -> fragments [cell / 32][k-step][lane]: lane l of k-step s holds cell (l & 31) of the block, bits
// s*64 + (l >> 5)*32 .. +31, one FP4 nibble per bit.  Cells past the end repeat the last one.
// steps = 16 (1024 bits): a bit is 0x0 = 0 / 0x2 = 1 (the steps count popcount(row & column): em2_matrix_step_asm.h,
// EM2_MATRIX_ZERO_ONE); steps = 32 (2048 bits): 0x2 = +1 / 0xA = -1.
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
    const bool zeroOne = EM2_MATRIX_ZERO_ONE && steps == kMatrixSteps;
    const uint32_t zero = zeroOne ? 0x0u : 0x2u, one = zeroOne ? 0x2u : 0xAu;
    FragmentWord4 v;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        uint32_t packed = 0;
#pragma unroll
        for (int n = 0; n < 8; n++) packed |= (((word >> (d * 8 + n)) & 1u) ? one : zero) << (4 * n);
        v[d] = int(packed);
    }
    out[i] = v;
}

// The column (and row) terms of the 0 / 1 steps: -popcount / 2 of every cell's signature (sig32 [cell][dwords]); entries past
// the last cell repeat it, as the fragments do.
__global__ void __launch_bounds__(256)
signatureTermsKernel(const uint32_t* __restrict__ sig32, uint32_t dwords, uint32_t cellCount, uint32_t termCount, float* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= termCount) return;
    const uint32_t cell = i < cellCount ? i : cellCount - 1u;
    uint32_t bits = 0;
    for (uint32_t w = 0; w < dwords; ++w) bits += uint32_t(__builtin_popcount(sig32[size_t(cell) * dwords + w]));
    out[i] = -0.5f * float(bits);
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

// TIMED (EM2_MATRIX_DIAG bit 2048, measurements only): every wave sums the shader-clock cycles it spends in the phases of
// its items -- ticket, set-up, walk, hand-off wait, replay, the quad's own columns, publication -- and adds them to eight
// 64-bit counters behind the inbox control words (printed by the launcher with EM2_TIMING set).
#define EM2_PHASE(index)                                                                                                       \
    do {                                                                                                                      \
        if (TIMED) {                                                                                                          \
            const uint64_t now_ = __builtin_readcyclecounter();                                                               \
            phaseCycles[index] += now_ - phaseStart;                                                                          \
            phaseStart = now_;                                                                                                \
        }                                                                                                                     \
    } while (0)
template <bool IDENTITY, bool WIDE = false, bool TIMED = false>
__device__ __forceinline__ void scanMatrixBody(unsigned char* ldsRaw)
{
    uint64_t phaseCycles[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t phaseStart = TIMED ? __builtin_readcyclecounter() : 0ull;
    uint64_t walkRecords = 0, replayCalls = 0;      // (TIMED: this lane's records over the launch; replays of the wave)
    uint64_t replayTimed[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // (TIMED: the replay's split, see the launcher's line)
    constexpr int W32 = WIDE ? 64 : 32;                          // dwords per signature as the v_xor/v_bcnt parts read them
    constexpr float bits = WIDE ? 2.f * kMatrixBits : kMatrixBits;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    FragmentWord4* tiles = reinterpret_cast<FragmentWord4*>(ldsRaw + kernelArgs()->matrixLdsOffset);
    volatile uint32_t* shared = reinterpret_cast<volatile uint32_t*>(ldsRaw + kernelArgs()->matrixLdsOffset + 4u * kMatrixTileWords * 16u);
    // The wave's LDS block of the walk is its selection area (never needed at the same time) when that is large enough
    // and 16-byte aligned, and sits behind the tiles otherwise (matrixWalkAliasesSelection on the host side).
    const uint32_t selectionStride = 2u * kernelArgs()->k * kLdsBytesPerEntrySlot;
    unsigned char* walkBlock = matrixWalkBlockInSelectionArea(selectionStride)
                                   ? ldsRaw + matrixWalkBlockOffsetInSelectionAreas(selectionStride, wave)
                                   : ldsRaw + kernelArgs()->matrixLdsOffset + 4u * kMatrixTileWords * 16u + 64u + wave * kMatrixWalkLdsBytes;
    // shared[0..2] stop words of the walk, shared[3] the block's ticket, shared[4..8] the convoy (kConvoyStartWord ...)
    if (threadIdx.x < 12u) shared[threadIdx.x] = 0u;
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
        // (rowBegin / rowFragmentBase: 0 everywhere but in the rows form -- launchFsp4ScanRowsMatrix -- whose slots are the 64-cell
        // blocks of [rowBegin, rowEnd) and whose row fragments may be a copy behind the columns' fragments)
        const uint32_t quadRowBase = aux->rowBegin + (quadBlock * aux->rowBlockStride + aux->rowBlockOffset) * 64u;
        const uint32_t rowBase = aux->rowBegin + (block * aux->rowBlockStride + aux->rowBlockOffset) * 64u;
        const uint32_t rowFragmentBlock = aux->rowFragmentBase + 2u * (listBlock * aux->rowBlockStride + aux->rowBlockOffset);
        const uint32_t row = rowBase + lane;
        const bool rowValid = !idle && row < aux->rowEnd;
        const uint32_t twoK = 2u * aux->k;
        const uint32_t minLogCapacity = kMatrixLogMargin;
        uint32_t logCapacity = aux->logCapacity < minLogCapacity ? minLogCapacity : aux->logCapacity;
        Entry* myList = aux->buffers + (size_t(listBlock) * 64u + lane) * twoK;
        // (the walk's records are 16 bytes apart: its waves -- at most eight per CU -- share the area the v_xor/v_bcnt kernels'
        // sixteen waves per CU have for their 8-byte log entries)
        WalkRecord* waveLog = reinterpret_cast<WalkRecord*>(aux->logs) + size_t(blockIdx.x * 4u + wave) * 64u * logCapacity;
        Entry* myLog = reinterpret_cast<Entry*>(waveLog + size_t(lane) * logCapacity);       // (an argument the exact v_xor/v_bcnt loops below never use)
        const uint32_t cps = aux->columnsPerSegment;
        const uint32_t colBegin = seg * cps;
        uint32_t colEnd = colBegin + cps;
        if (colEnd > aux->columnLimit || seg + 1u == segments) colEnd = aux->columnLimit;
        const bool last = !fullRows && quadRowBase < colEnd;    // the segment that holds the quad's own cells
        // (a full row's last segment may end at a cell count that is no multiple of 32: the walk takes whole tiles)
        const uint32_t commonEnd = last ? quadRowBase : (fullRows ? colEnd & ~31u : colEnd);
        int32_t mMax = rowValid ? aux->mMaxInitial : -1;
        uint32_t count = 0;
        uint32_t recordCount[2] = {0u, 0u};          // (the calling lane's records per accumulator)
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

        // ---- the convoy ----
        // The departures showed what the L2s can do when the walks of an XCD stay together, and what waiting for that costs.
        // Nobody waits here: a walk STARTS WHERE THE OTHERS ARE and goes around -- from its starting column to the segment's
        // end, then from the segment's begin to its starting column.  Every walk publishes its position now and then in a
        // word of its XCD group (blockIdx & 7: the blocks of a launch are dealt to the XCDs in turn) and of its segment's
        // parity -- an atomic maximum, so the word is where the group's HEAD is (em2_scan_symmetric_device.h) -- and a block
        // that takes an item of the same segment starts there.  All walks move at the same pace, so the group
        // circles through the segment as one stream of tiles that the first to ask fetches over the fabric and the others find
        // in the L2.  The logs of such a walk hold the higher columns in front of the lower ones: the replay takes them in
        // the order of the columns (WalkLogReader), so the lists see the candidates exactly as before.  A walk whose logs fill
        // up before it has reached the segment's end cannot replay them yet (the lower columns come first): it drops them
        // and starts again at the segment's begin, as a walk without convoy (in the lower columns a stop is harmless: those
        // records are in order, they are replayed and removed).  Never at the bench's sizes (0 of 120 337 items stop at all).
        auto convoyStart = [&]() -> uint32_t {
            return convoyStartColumn(aux, shared, seg, colBegin, commonEnd, ticket);
        };
        bool failed = false;
        uint32_t start = convoyStart();         // (start == colBegin: a walk as ever)
        uint32_t at = start, rangeEnd = commonEnd;
        bool lower = false;                     // the walk has gone around: [colBegin, start) now
        uint32_t firstRecord[2] = {0u, 0u};     // the lane's records when it did
        uint32_t rowHalf = 0;           // (2048 bits: the columns are walked once per half of the wave's rows)
        // (1024 bits: -popcount / 2 of the lane's row; a row that is none starts so low that its results pass no bound)
        const float rowTerm = WIDE ? 0.f : rowValid ? aux->terms[row] : -4096.f;
        EM2_PHASE(1);
        for (;;) {
            if (at < rangeEnd) {
                if (WIDE) {
                    at = scanTilesMatrixWide<IDENTITY>(aux->fragments, aux->snap, at, rangeEnd, rowFragmentBlock + rowHalf,
                                                       bits - 2.f * float(mMax), rowHalf, waveLog, logCapacity,
                                                       recordCount, ldsAddress(tiles), ldsAddress(const_cast<uint32_t*>(shared)),
                                                       ldsAddress(walkBlock));
                } else {
                    if (EM2_DIAG_WORD(aux)) {
                        at = scanTilesMatrixPinned<IDENTITY, false, true>((const void*)(uintptr_t)aux, aux->fragments, aux->snap, aux->terms, at, rangeEnd,
                                                         rowFragmentBlock, matrixBoundOf<false>(mMax), rowTerm, waveLog,
                                                         logCapacity, recordCount, ldsAddress(tiles), ldsAddress(const_cast<uint32_t*>(shared)),
                                                         ldsAddress(walkBlock));
                    } else {
                        at = scanTilesMatrixPinned<IDENTITY, false, false>((const void*)(uintptr_t)aux, aux->fragments, aux->snap, aux->terms, at, rangeEnd,
                                                         rowFragmentBlock, matrixBoundOf<false>(mMax), rowTerm, waveLog,
                                                         logCapacity, recordCount, ldsAddress(tiles), ldsAddress(const_cast<uint32_t*>(shared)),
                                                         ldsAddress(walkBlock));
                    }
                }
            }
            EM2_PHASE(2);
            if (TIMED) {
                walkRecords += recordCount[0] + recordCount[1];
                replayCalls += 1u;
            }
            if (TIMED && start != colBegin && !lower) phaseCycles[7] += 1u << 20;           // (walks that joined a convoy)
            if (start != colBegin && !lower && (at & kWalkInLowerColumns) != 0u) {
                // the walk went around within the call: it is in its lower columns (or through with them)
                at &= ~kWalkInLowerColumns;
                lower = true;
                rangeEnd = start;
                if (WIDE) {
                    const volatile uint32_t* counts = reinterpret_cast<const volatile uint32_t*>(walkBlock + kWalkWrapCounts);
                    firstRecord[0] = rowHalf != 0u ? 0u : counts[lane];
                    firstRecord[1] = rowHalf == 0u ? 0u : counts[64u + lane];
                } else {
                    const volatile uint8_t* counts = walkBlock + kWalkWrapCounts8;
                    firstRecord[0] = counts[lane];
                    firstRecord[1] = counts[64u + lane];
                }
            }
            const bool stopped = at < rangeEnd;                           // (for its logs; the same in all waves of the block)
            if (TIMED && stopped) phaseCycles[7] += 1u;
            if (stopped && threadIdx.x == 0u && aux->convoy == 1u) {
                __hip_atomic_fetch_add(aux->inboxControl + kConvoyStopsWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (start != colBegin && !lower) {
                // the logs filled up in the higher columns (see above): once more, from the segment's begin, publishing nothing
                recordCount[0] = recordCount[1] = 0u;
                start = at = colBegin;
                if (threadIdx.x == 0u) shared[kConvoyCodeWord] = 0u;
                __syncthreads();
                continue;
            }
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
            {
                // (a walk that went around and stopped in its lower columns: those records only, the others stay)
                const bool all = !(lower && stopped);
                if (!idle && !failed) {
                    replayWalkLogs<IDENTITY, WIDE>(waveLog, logCapacity, recordCount, firstRecord, all, lane, row, rowValid, !fullRows,
                                                   listBlock, myList, twoK, count, mMax, emitPos, emitEnd, ldsRaw, ldsAddress(tiles), colBegin, TIMED ? replayTimed : nullptr);
                }
                recordCount[0] = all ? 0u : firstRecord[0];
                recordCount[1] = all ? 0u : firstRecord[1];
                // The rows' cut-offs as they stand, for whoever walks these cells as columns meanwhile: any value a cell held at
                // some point of its sequence is a valid snapshot, and one published at every replay instead of at the item's end
                // is a segment fresher when thresholds fall fast (threshold 0: half of all pairs pass a cell's first bound, and
                // with the cut-offs of an item's start the launch's first items alone filled the pool of deferred candidates).
                if (haveState && !fullRows && rowValid && !failed) {
                    __hip_atomic_store(aux->snap + row, mMax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            EM2_PHASE(4);
            if (at >= rangeEnd) {
                if (!WIDE || rowHalf == 1u || colBegin >= commonEnd) break;
                rowHalf = 1u;
                start = convoyStart();
                at = start;
                rangeEnd = commonEnd;
                lower = false;
                firstRecord[0] = firstRecord[1] = 0u;
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
    if (TIMED) {
        // (bytes 144..159 of the control block: the launch's records and its calls of the replay)
        unsigned long long* more = reinterpret_cast<unsigned long long*>(kernelArgs()->inboxControl + 36);
        atomicAdd(more, (unsigned long long)walkRecords);
        if (lane == 0u) atomicAdd(more + 1, (unsigned long long)replayCalls);
        // (the replay's split: the convoy's words, bytes 160..223 -- a timed launch that wants it runs with EM2_MATRIX_CONVOY=0)
        if (lane == 0u && kernelArgs()->convoy == 0u) {
            unsigned long long* split = reinterpret_cast<unsigned long long*>(kernelArgs()->inboxControl + 40);
            for (int i = 0; i < 8; i++) atomicAdd(split + i, (unsigned long long)replayTimed[i]);
        }
    }
    if (threadIdx.x == 0u) {
        unsigned long long* clockWords = reinterpret_cast<unsigned long long*>(kernelArgs()->inboxControl + kClockWordsOffset);
        atomicAdd(clockWords, (unsigned long long)(__builtin_amdgcn_s_memtime() - clockStart));
        atomicAdd(clockWords + 1, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - wallStart));
    }
}
#undef EM2_PHASE

// The entry points.  The compiler's own code in the walk (scanTilesMatrixPinned) lives in v0..v27: v28..v255 hold the rows,
// the accumulators and the fragment ring of the steps, which the compiler does not know of (the kernel descriptor still asks for
// 256 registers: the clobber lists of the steps count).
template <bool IDENTITY>
__global__ void __launch_bounds__(256, 2)
fsp4ScanMatrixPinnedKernel(Fsp4Args args)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
    scanMatrixBody<IDENTITY>(ldsRaw);
}

// 2048-bit signatures (scanTilesMatrixWide)
template <bool IDENTITY>
__global__ void __launch_bounds__(256, 2)
fsp4ScanMatrixWideKernel(Fsp4Args args)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
    scanMatrixBody<IDENTITY, true>(ldsRaw);
}

// (EM2_MATRIX_DIAG bit 2048: the same kernels with the phase timers, identity keys only)
template <bool WIDE>
__global__ void __launch_bounds__(256, 2)
fsp4ScanMatrixTimedKernel(Fsp4Args args)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
    scanMatrixBody<true, WIDE, true>(ldsRaw);
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
    if (EM2_DIAG_WORD_OF(args) & 4096u) {        // (EM2_MATRIX_DIAG bit 4096: the longest inbox of a cell and of a wave's 64 cells, for the launcher's line)
        uint32_t longest = uint32_t(bound[1] - bound[0]);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) longest = max(longest, uint32_t(__shfl_xor(int(longest), d, 64)));
        if (lane == 0u) atomicMax(args.inboxControl + 8, longest);
        clock1 = __builtin_readcyclecounter();
    }
    const uint32_t idMask = (1u << nb) - 1u;
    // Row by row, the lanes side by side on ONE cell's entries, 64 at a time (they ascend in the candidate): the cells' inboxes are
    // of very different lengths -- 240 entries on average at the bench's data and 4 000 for the longest, 930 and 52 000 on 8 tight
    // clusters -- and with a lane per cell, an entry per turn, the wave took as many turns (each a round trip to the L2) as
    // its longest inbox had entries: the longest cell of the launch alone was 12 to 38 ms on clustered data.
    {
        ReplayArgs rargs;
        rargs.lists = args.buffers + size_t(block) * 64u * twoK;
        rargs.inbox = nullptr;
        rargs.acceptMaxByKey = args.acceptMaxByKey;
        rargs.keyOfMismatch = args.keyOfMismatch;
        rargs.k = args.k;
        rargs.twoK = twoK;
        rargs.columnShift = 0u;
        rargs.vectorMemoryIssued = 0u;
        rargs.firstColumn = 0u;
        rargs.selfPairs = false;
        const uint64_t begin = bound[0];
        const uint32_t mine = uint32_t(bound[1] - bound[0]);
        uint64_t rowsWithEntries = __builtin_amdgcn_ballot_w64(mine != 0u);
        while (rowsWithEntries != 0ull) {
            const uint32_t r = uint32_t(__builtin_ctzll(rowsWithEntries));
            rowsWithEntries &= rowsWithEntries - 1ull;
            const uint32_t n = uint32_t(__builtin_amdgcn_readlane(int(mine), int(r)));
            const uint64_t first = uint64_t(uint32_t(__builtin_amdgcn_readlane(int(uint32_t(begin)), int(r)))) |
                                   (uint64_t(uint32_t(__builtin_amdgcn_readlane(int(uint32_t(begin >> 32)), int(r)))) << 32);
            uint32_t countOfRow = uint32_t(__builtin_amdgcn_readlane(int(count), int(r)));
            int32_t mMaxOfRow = __builtin_amdgcn_readlane(mMax, int(r));
            Entry* listRow = rargs.lists + size_t(r) * twoK;
            const uint64_t* entries = sorted + first;
            // (four batches at a time, the next four loaded under them -- every lane loads, past the end the last entry again:
            // one batch ahead, a batch waited for most of a round trip to the L2)
            uint64_t current[4], ahead[4];
#pragma unroll
            for (uint32_t j = 0; j < 4u; j++) {
                const uint32_t index = 64u * j + lane;
                current[j] = entries[index < n ? index : n - 1u];
            }
            for (uint32_t at = 0; at < n; at += 256u) {
#pragma unroll
                for (uint32_t j = 0; j < 4u; j++) {
                    const uint32_t index = at + 256u + 64u * j + lane;
                    ahead[j] = entries[index < n ? index : n - 1u];
                }
#pragma unroll
                for (uint32_t j = 0; j < 4u; j++) {
                    if (at + 64u * j >= n) break;
                    offerBatchToRow<IDENTITY>(at + 64u * j + lane < n, uint32_t(current[j] >> 13u) & idMask, uint32_t(current[j]) & 0x1fffu, lane,
                                              listRow, rargs, countOfRow, mMaxOfRow, ldsRaw);
                }
#pragma unroll
                for (uint32_t j = 0; j < 4u; j++) current[j] = ahead[j];
            }
            if (lane == r) {
                count = countOfRow;
                mMax = mMaxOfRow;
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

}  // namespace

const void* scanMatrixKernelFor(bool identity, bool wide)
{
    if (identity && (diagNumber("EM2_MATRIX_DIAG") & 2048u)) {
        return wide ? reinterpret_cast<const void*>(&fsp4ScanMatrixTimedKernel<true>)
                    : reinterpret_cast<const void*>(&fsp4ScanMatrixTimedKernel<false>);
    }
    if (wide) {
        return identity ? reinterpret_cast<const void*>(&fsp4ScanMatrixWideKernel<true>)
                        : reinterpret_cast<const void*>(&fsp4ScanMatrixWideKernel<false>);
    }
    return identity ? reinterpret_cast<const void*>(&fsp4ScanMatrixPinnedKernel<true>)
                    : reinterpret_cast<const void*>(&fsp4ScanMatrixPinnedKernel<false>);
}



// ---- symmetric (triangle) scan: eligibility and workspace ----
// EM2_SCAN_MODE=triangle forces it wherever it is possible (all rows of the problem in one launch),
// EM2_SCAN_MODE=persistent / simple disable it; by default it is used from kSymmetricMinCells cells on.



bool symmetricEligible(uint32_t cellCount, uint32_t rowCount, uint32_t paddedDw)
{
    // inbox keys hold two cell ids and a mismatch count in 64 bits: 13 + 2 * bits(cellCount) <= 64
    if (rowCount != cellCount || cellCount < 128u || cellCount > (1u << 25)) return false;
    const char* v = getenv("EM2_SCAN_MODE");
    if (v && v[0] == 't') return true;
    if (v && (v[0] == 's' || v[0] == 'p' || v[0] == 'r')) return false;
    // 1024-bit signatures take the matrix-core form, which wins much earlier (scan ms ordered / symmetric-matrix:
    // 30k cells 2.5 / 2.4, 60k 8.2 / 4.2, 100k 20.1 / 7.5)
    const bool matrix = matrixFormWanted(paddedDw) || matrixWideWanted(paddedDw);
    return cellCount >= (matrix ? kSymmetricMatrixMinCells : kSymmetricMinCells);
}

// Deferred candidates per cell the pool has room for: 1024 unless the calling thread says otherwise (the facade of
// em2_subset_find_similar_pairs4, which allocates the workspace itself, starts with less and comes back for the full pool if the
// launch overflowed: fsp4SetInboxEntriesPerCell).
thread_local uint32_t inboxEntriesPerCell = 0;
void fsp4SetInboxEntriesPerCell(uint32_t entries) { inboxEntriesPerCell = entries; }

static uint64_t inboxCapacity(uint32_t cellCount)
{
    // EM2_INBOX_CAPACITY (entries) is a test knob: tiny pools force the overflow -> ordered-scan fallback.
    const uint64_t forced = envNumber("EM2_INBOX_CAPACITY", 0);
    if (forced >= kInboxChunk) return forced < 0xfff00000ull ? forced : 0xfff00000ull;
    uint64_t cap = uint64_t(cellCount) * (inboxEntriesPerCell ? inboxEntriesPerCell : 1024u);
    const uint64_t floor = uint64_t(maxResidentWaves()) * kInboxChunk * 2u;      // every wave can hold a chunk
    if (cap < floor) cap = floor;
    if (cap > 0xfff00000ull) cap = 0xfff00000ull;
    return cap;
}


struct SymmetricLayout {
    size_t snap, table, tableMatrix, control, poolA, poolB, temp, fragments, terms, widened, total, tempBytes;
    uint64_t capacity;
};

static SymmetricLayout symmetricLayout(uint32_t cellCount, uint32_t paddedDw)
{
    SymmetricLayout l;
    l.capacity = inboxCapacity(cellCount);
    l.tempBytes = inboxSortTempBytesDoubleBuffer(l.capacity);
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
    l.terms = at;     at += align256(matrixTermCount(cellCount) * 4u);
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

// ---- the rows form on the matrix cores: rows [rowBegin, rowEnd) x all columns (launchFsp4ScanRowsMatrix) ----
// The partitioning SURVEY 8(e) prescribes for several GPUs (a rank's contiguous rows against every column, which is
// exactly the per-cell contract of src/ExpressionMatrixLsh.cpp:200-285) and what a symmetric scan whose inbox
// overflowed falls back on.  From kRowsMatrixMinPairs (row, column) pairs on; EM2_SCAN_MODE=rows takes it for every
// launch it can serve (tests), persistent / simple / triangle never.
constexpr uint64_t kRowsMatrixMinPairs = 1ull << 31;

bool rowsMatrixEligible(uint32_t cellCount, uint32_t rowCount, uint32_t paddedDw)
{
    if (!(matrixFormWanted(paddedDw) || matrixWideWanted(paddedDw))) return false;
    if (cellCount < 64u || rowCount == 0u) return false;
    const char* v = getenv("EM2_SCAN_MODE");
    if (v && v[0] == 'r') return true;
    if (v && (v[0] == 's' || v[0] == 'p' || v[0] == 't' || v[0] == 'v')) return false;
    return uint64_t(rowCount) * uint64_t(cellCount) >= kRowsMatrixMinPairs;
}

struct RowsMatrixLayout {
    size_t snap, table, control, fragments, terms, rowFragments, widened, total;
};

// (rows that are not all cells get room for a copy of their fragments: a rowBegin that is no multiple of 32 cannot
// address the columns' array in 32-cell blocks)
static RowsMatrixLayout rowsMatrixLayout(uint32_t cellCount, uint32_t rowCount, uint32_t paddedDw)
{
    RowsMatrixLayout l;
    const size_t bytesPerCell = matrixWideWanted(paddedDw) ? 1024u : 512u;
    size_t at = 0;
    l.snap = at;      at += align256(size_t(cellCount) * 4u);
    l.table = at;     at += align256(kTableWords * 4u);
    l.control = at;   at += 256u;
    l.fragments = at; at += size_t((cellCount + 63u) / 64u) * 64u * bytesPerCell;
    l.rowFragments = at;          // (whole 32-cell blocks behind the columns' fragments: Fsp4Args::rowFragmentBase counts them)
    if (rowCount < cellCount) at += size_t((rowCount + 63u) / 64u) * 64u * bytesPerCell;
    l.widened = at;
    if (paddedDw < 32u && matrixFormWanted(paddedDw)) at += align256(size_t(cellCount) * 128u);
    l.terms = at;     at += align256(matrixTermCount(cellCount) * 4u);
    l.total = at;
    return l;
}

bool fsp4UsesRowsMatrixScan(uint32_t cellCount, uint32_t rowCount, uint32_t paddedDw)
{
    return !symmetricEligible(cellCount, rowCount, paddedDw) && rowsMatrixEligible(cellCount, rowCount, paddedDw);
}

size_t fsp4SymmetricBytes(uint32_t cellCount, uint32_t rowCount, uint32_t paddedDw)
{
    if (symmetricEligible(cellCount, rowCount, paddedDw)) return symmetricLayout(cellCount, paddedDw).total;
    if (rowsMatrixEligible(cellCount, rowCount, paddedDw)) return rowsMatrixLayout(cellCount, rowCount, paddedDw).total;
    return 0;
}

// Resident waves of a persistent-style launch of `kernel` (min(occupancy, 4 waves per SIMD) x CUs).
hipError_t residentWaveSlots(const void* kernel, uint32_t wavesPerBlock, size_t lds, uint32_t* slots)
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

    // Segments: as many as the column-count floor allows, up to kMaxSegments: short
    // segments keep the column snapshots fresh and even out the triangle.
    uint64_t minSegmentColumns = envNumber("EM2_MIN_SEGMENT_COLUMNS", 4096);
    if (minSegmentColumns < 1) minSegmentColumns = 1;
    const uint32_t maxSegments = matrix ? kMatrixMaxSegments : kMaxSegments;
    uint64_t segments = cellCount / minSegmentColumns;
    if (segments > maxSegments) segments = maxSegments;
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
    // 8192 / 16384 / 32768 / 131072 columns per segment: 289 / 278 / 273 / 270 / 274; 2048: 335).  With the convoy, which
    // keeps the L2s out of the question (1M cells, same box, 16384 / 20480 / 24576 / 28672 / 32768 / 49152 columns: 185.4 /
    // 184.3 / 184.1 / 183.9 ms, and 178.0 / - / 176.8 / - / 177.5 / 180.3 on another; 2048 bits: 8192 / 16384 / 32768:
    // 396.1 / 394.2 / 399.2), the 1024-bit form takes 24576.  The test knobs apply to both launches.
    uint64_t segmentsMatrix = segments;
    uint32_t cpsMatrix = cps;
    if (matrix) {
        uint64_t defaultColumns = cellCount / 24u;              // small problems keep enough items to fill the machine
        const uint64_t longest = wide ? 16384u : 24576u;
        defaultColumns = defaultColumns < 4096 ? 4096 : (defaultColumns > longest ? longest : defaultColumns);
        uint64_t minColumns = envNumber("EM2_MIN_SEGMENT_COLUMNS", defaultColumns);
        if (minColumns < 1) minColumns = 1;
        segmentsMatrix = cellCount / minColumns;
        if (segmentsMatrix > kMatrixMaxSegments) segmentsMatrix = kMatrixMaxSegments;
        if (segmentsMatrix < 1) segmentsMatrix = 1;
        cpsMatrix = uint32_t((uint64_t(cellCount) + segmentsMatrix - 1u) / segmentsMatrix);
        cpsMatrix = (cpsMatrix + 255u) & ~255u;
        if (cpsMatrix > kMaxColumnsPerItem) {           // (the replay's merge keys hold a column relative to its item's first in 19 bits)
            segmentsMatrix = (uint64_t(cellCount) + kMaxColumnsPerItem - 1u) / kMaxColumnsPerItem;
            cpsMatrix = (uint32_t((uint64_t(cellCount) + segmentsMatrix - 1u) / segmentsMatrix) + 255u) & ~255u;
        }
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
        e = launchExpandFragments(matrixArgs.sig32, cellCount, fragmentCount, ws + layout.fragments, matrixSteps,
                                  reinterpret_cast<float*>(ws + layout.terms), stream);
        if (e != hipSuccess) return e;
        matrixArgs.terms = reinterpret_cast<const float*>(ws + layout.terms);
        matrixArgs.segTable = reinterpret_cast<const uint32_t*>(ws + layout.tableMatrix);
        matrixArgs.segments = uint32_t(segmentsMatrix);
        matrixArgs.columnsPerSegment = cpsMatrix;
        matrixArgs.totalTickets = uint32_t(ticketsMatrix);
        matrixArgs.fragments = ws + layout.fragments;
        matrixArgs.matrixLdsOffset = uint32_t((lds + 15u) & ~size_t(15));
        // (EM2_MATRIX_CONVOY: 0 = every walk from its segment's first column, 1 = the walks of an XCD go around together,
        // n >= 2: every walk starts 64 (n - 1) columns into its segment -- the tests' way to the same code)
        matrixArgs.convoy = uint32_t(envNumber("EM2_MATRIX_CONVOY", 1));
        if (cpsMatrix / 64u > kConvoyMaxPairs && matrixArgs.convoy == 1u) matrixArgs.convoy = 0u;       // (positions have 12 bits)
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
        if (blocksWanted * 8u > maxResidentWaves()) blocksWanted = maxResidentWaves() / 8u;      // the logs are sized for that
        if (blocksWanted > ticketsMatrix) blocksWanted = ticketsMatrix;
        if (scanVerbose()) {
            fprintf(stderr, "[em2] matrix kernel: %d blocks per CU, %llu blocks, %zu bytes of LDS\n", blocksPerCu,
                                     (unsigned long long)blocksWanted, matrixLds);
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
            fprintf(stderr, "; %.0f cycles per item and wave outside the walk; %.0f walks stopped for their logs\n",
                    (total - double(cycles[2])) / (4.0 * double(ticketsMatrixCount ? ticketsMatrixCount : 1)), double(cycles[7] & 0xfffffu) / 4.0);
            fprintf(stderr, "[em2] matrix kernel: %.0f walks started inside their segment\n", double(cycles[7] >> 20) / 4.0);
            unsigned long long more[2] = {0, 0};
            if (hipMemcpy(more, ws + layout.control + 144u, sizeof(more), hipMemcpyDeviceToHost) == hipSuccess) {
                fprintf(stderr, "[em2] matrix kernel: %llu records logged (%.2f per wave and tile), %llu replays of a wave's logs\n", more[0],
                        double(more[0]) / (double(cellCount) * double(cellCount) / 4096.0), more[1]);
            }
            unsigned long long split[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (envNumber("EM2_MATRIX_CONVOY", 1) == 0 && hipMemcpy(split, ws + layout.control + 160u, sizeof(split), hipMemcpyDeviceToHost) == hipSuccess) {
                const double t = total > 0 ? total : 1;
                fprintf(stderr, "[em2] matrix kernel, replay (%% of the waves' cycles): %llu pair turns: await %.2f, read + merge %.2f, rows %.2f, emit %.2f, "
                                "issue %.2f; all short rows %.2f; %llu rows of their own, %llu long rows\n", split[3], 100.0 * double(split[0]) / t,
                        100.0 * double(split[1]) / t, 100.0 * double(split[2]) / t, 100.0 * double(split[6]) / t, 100.0 * double(split[7]) / t,
                        100.0 * double(split[5]) / t, split[4], 0ull);
            }
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
    if (scanVerbose()) {
        fprintf(stderr, "[em2] symmetric scan%s: %u segments x %u columns, %u full-row blocks, %llu + %llu tickets, %llu inbox slots\n",
                                 matrix ? " (matrix cores)" : "", uint32_t(matrix ? segmentsMatrix : segments), matrix ? cpsMatrix : cps, fullRowBlocks,
                                 (unsigned long long)tickets, (unsigned long long)(matrix ? ticketsMatrix : 0), (unsigned long long)used);
    }

    const uint64_t* sorted = args.inbox;
    if (used) {
        size_t tempBytes = layout.tempBytes;
        uint64_t* out = reinterpret_cast<uint64_t*>(ws + layout.poolB);
        rocprim::double_buffer<uint64_t> keys(args.inbox, out);
        e = rocprim::radix_sort_keys(ws + layout.temp, tempBytes, keys, size_t(used), 13u, 13u + 2u * rowBits, stream);
        if (e != hipSuccess) return e;
        sorted = keys.current();
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


// The rows form on the matrix cores.  Every 64-row block of [rowBegin, rowEnd) is a full-row block of scanMatrixBody: its
// quad walks ALL columns of every segment in ascending order (speculatively against the cut-offs its rows published at
// the previous hand-off, then replayed in column order through the exact state machine -- the same walk, logs and replay
// as the triangle's), the self pair is dropped at the replay (acceptColumn), nothing is deferred: no inbox, no sort, no
// second kernel; the rows finish themselves at the last segment.  The column cut-offs the steps test against are all -1
// ("never"), so only the row side of a pair can pass.  ws: the rows layout, or -- symmetricWorkspace, the fallback of a
// symmetric scan whose inbox overflowed -- the symmetric layout, whose areas serve as well.
// *done = false: this launch cannot take the form (k too large for four waves per block, no room in the LDS).
hipError_t launchFsp4ScanRowsMatrix(Fsp4Args args, uint32_t paddedDw, bool identity, uint32_t wavesPerBlock, size_t lds,
                                    void* control, void* ws_, bool symmetricWorkspace, hipStream_t stream, bool* done)
{
    *done = false;
    const uint32_t cellCount = args.cellCount;
    const uint32_t rowBlocks = args.rowBlocks;
    const uint32_t rows = args.rowEnd - args.rowBegin;
    const bool wide = matrixWideWanted(paddedDw);
    if (!(matrixFormWanted(paddedDw) || wide) || wavesPerBlock != 4u ||
        ((lds + 15u) & ~size_t(15)) + scanMatrixLdsBytes(args.k) > 150u * 1024u) return hipSuccess;
    char* ws = static_cast<char*>(ws_);
    char *snapArea, *tableArea, *controlArea, *fragmentArea, *rowFragmentArea, *widenedArea, *termArea;
    if (symmetricWorkspace) {
        const SymmetricLayout l = symmetricLayout(cellCount, paddedDw);
        snapArea = ws + l.snap; tableArea = ws + l.tableMatrix; controlArea = ws + l.control;
        fragmentArea = ws + l.fragments; rowFragmentArea = nullptr; widenedArea = ws + l.widened;
        termArea = ws + l.terms;
        if (args.rowBegin != 0u) return hipErrorInvalidValue;
    } else {
        const RowsMatrixLayout l = rowsMatrixLayout(cellCount, rows, paddedDw);
        snapArea = ws + l.snap; tableArea = ws + l.table; controlArea = ws + l.control;
        fragmentArea = ws + l.fragments; rowFragmentArea = rows < cellCount ? ws + l.rowFragments : nullptr; widenedArea = ws + l.widened;
        termArea = ws + l.terms;
    }
    const void* matrixKernel = scanMatrixKernelFor(identity, wide);
    const size_t matrixLdsOffset = (lds + 15u) & ~size_t(15);
    const size_t matrixLds = matrixLdsOffset + scanMatrixLdsBytes(args.k);
    int device = 0, cuCount = 0, blocksPerCu = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    e = hipDeviceGetAttribute(&cuCount, hipDeviceAttributeMultiprocessorCount, device);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(matrixKernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(matrixLds));
    if (e != hipSuccess) return e;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocksPerCu, matrixKernel, 256, matrixLds);
    if (e != hipSuccess) return e;
    blocksPerCu = blocksPerCu < 1 ? 1 : (blocksPerCu > 2 ? 2 : blocksPerCu);
    if (const char* v = getenv("EM2_BLOCKS_PER_CU")) {
        if (atoi(v) >= 1 && atoi(v) < blocksPerCu) blocksPerCu = atoi(v);
    }
    uint64_t blocksWanted = uint64_t(cuCount) * uint64_t(blocksPerCu);
    if (blocksWanted * 8u > maxResidentWaves()) blocksWanted = maxResidentWaves() / 8u;      // the logs are sized for that

    // Segments as in the symmetric launch (24 576 / 16 384 columns at large sizes); a launch of few rows takes shorter
    // ones, down to 2048 columns, until there are some four items per resident block.
    const uint32_t quads = (rowBlocks + 3u) / 4u;
    uint64_t defaultColumns = cellCount / 24u;
    const uint64_t longest = wide ? 16384u : 24576u;
    defaultColumns = defaultColumns < 4096 ? 4096 : (defaultColumns > longest ? longest : defaultColumns);
    {
        const uint64_t segmentsForItems = (4u * blocksWanted + quads - 1u) / quads;
        uint64_t columnsForItems = cellCount / (segmentsForItems ? segmentsForItems : 1u);
        if (columnsForItems < 2048) columnsForItems = 2048;
        if (columnsForItems < defaultColumns) defaultColumns = columnsForItems;
    }
    uint64_t minColumns = envNumber("EM2_MIN_SEGMENT_COLUMNS", defaultColumns);
    if (minColumns < 1) minColumns = 1;
    uint64_t segments = cellCount / minColumns;
    if (segments > kMatrixMaxSegments) segments = kMatrixMaxSegments;
    if (segments < 1) segments = 1;
    uint32_t cps = uint32_t((uint64_t(cellCount) + segments - 1u) / segments);
    cps = (cps + 255u) & ~255u;
    if (cps > kMaxColumnsPerItem) {             // (the replay's merge keys hold a column relative to its item's first in 19 bits)
        segments = (uint64_t(cellCount) + kMaxColumnsPerItem - 1u) / kMaxColumnsPerItem;
        cps = (uint32_t((uint64_t(cellCount) + segments - 1u) / segments) + 255u) & ~255u;
    }
    segments = (uint64_t(cellCount) + cps - 1u) / cps;
    const uint64_t tickets = segments * quads;
    if (tickets >= 0xffffffffull) return hipErrorInvalidValue;
    uint32_t table[kTableWords];
    for (uint32_t sIdx = 0; sIdx < segments; ++sIdx) {
        table[sIdx] = sIdx * quads;
        table[segments + 1u + sIdx] = rowBlocks;        // (no triangle quads)
    }
    table[segments] = uint32_t(tickets);
    if (blocksWanted > tickets) blocksWanted = tickets;

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
    args.snap = reinterpret_cast<int32_t*>(snapArea);
    args.inbox = nullptr;
    args.inboxControl = reinterpret_cast<uint32_t*>(controlArea);
    args.segTable = reinterpret_cast<const uint32_t*>(tableArea);
    args.inboxCapacity = 0;
    args.inboxChunk = kInboxChunk;
    args.fullRowBlocks = rowBlocks;
    uint32_t rowBits = 1;
    while ((1ull << rowBits) < uint64_t(cellCount)) ++rowBits;
    args.rowBits = rowBits;
    args.totalTickets = uint32_t(tickets);
    args.rowBlockStride = 1;
    args.rowBlockOffset = 0;
    args.localBlockBase = 0;
    args.columnLimit = cellCount;
    args.shardFlags = 0;
    args.fragments = fragmentArea;
    args.matrixLdsOffset = uint32_t(matrixLdsOffset);
    args.convoy = uint32_t(envNumber("EM2_MATRIX_CONVOY", 1));
    if (cps / 64u > kConvoyMaxPairs && args.convoy == 1u) args.convoy = 0u;       // (positions have 12 bits)

    e = hipMemsetAsync(c + stateBytes, 0, doneBytes + 256u, stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(controlArea, 0, 256u, stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(snapArea, 0xff, size_t(cellCount) * 4u, stream);        // -1: no column ever takes a candidate
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(tableArea, table, (2u * segments + 2u) * 4u, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;

    static thread_local hipEvent_t timing[2] = {nullptr, nullptr};
    if (!timing[0]) {
        if (hipEventCreate(&timing[0]) != hipSuccess || hipEventCreate(&timing[1]) != hipSuccess) timing[0] = timing[1] = nullptr;
    }
    const uint32_t matrixSteps = wide ? 2u * kMatrixSteps : kMatrixSteps;
    const size_t bytesPerCell = wide ? 1024u : 512u;
    const uint32_t signatureDwords = 2u * matrixSteps;          // of a cell as the expansion and the v_xor/v_bcnt parts read it
    if (paddedDw < 32u) {
        uint32_t* widened = reinterpret_cast<uint32_t*>(widenedArea);
        widenSignaturesKernel<<<dim3((cellCount * 32u + 255u) / 256u), dim3(256), 0, stream>>>(args.sig32, paddedDw, cellCount, widened);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        args.sig32 = widened;
    }
    {
        const uint32_t columnBlocks = (cellCount + 63u) / 64u;
        const uint32_t fragmentCount = columnBlocks * 2u * matrixSteps * 64u;
        e = launchExpandFragments(args.sig32, cellCount, fragmentCount, fragmentArea, matrixSteps, reinterpret_cast<float*>(termArea), stream);
        if (e != hipSuccess) return e;
        args.terms = reinterpret_cast<const float*>(termArea);
    }
    // The rows' fragments are addressed in place -- inside the columns' array -- only when every 64-row block of the launch
    // lies within that array: a shard that begins at 32 (mod 64) and ends with the cells would otherwise read 32 cells past
    // it, into workspace nobody wrote (expandFragmentsKernel pads with copies of the last cell, so that a padded lane only
    // ever holds the operands of a signature and cannot pass a bound; stale FP4 magnitudes could).
    const bool rowsInPlace = args.rowBegin % 32u == 0u &&
                             uint64_t(args.rowBegin) + 64ull * rowBlocks <= uint64_t((cellCount + 63u) / 64u) * 64u;
    if (rowsInPlace) {
        args.rowFragmentBase = args.rowBegin / 32u;             // the rows are whole blocks of the columns' array
    } else {
        if (!rowFragmentArea) return hipErrorInvalidValue;
        const uint32_t fragmentCount = rowBlocks * 2u * matrixSteps * 64u;
        expandFragmentsKernel<<<dim3((fragmentCount + 255u) / 256u), dim3(256), 0, stream>>>(
            args.sig32 + size_t(args.rowBegin) * signatureDwords, rows, fragmentCount, reinterpret_cast<FragmentWord4*>(rowFragmentArea), matrixSteps);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        args.rowFragmentBase = uint32_t(size_t(rowFragmentArea - fragmentArea) / (32u * bytesPerCell));
    }
    if (scanVerbose()) {
        fprintf(stderr, "[em2] rows form on the matrix cores: rows [%u, %u), %u segments x %u columns, %llu tickets, %llu blocks, "
                                         "row fragments %s\n", args.rowBegin, args.rowEnd, uint32_t(segments), cps, (unsigned long long)tickets,
                                 (unsigned long long)blocksWanted, rowsInPlace ? "in the columns' array" : "expanded once more");
    }
    void* argsArray[] = {&args};
    if (timing[0]) (void)hipEventRecord(timing[0], stream);
    e = hipLaunchKernel(matrixKernel, dim3(uint32_t(blocksWanted)), dim3(256), argsArray, matrixLds, stream);
    if (e != hipSuccess) return e;
    if (timing[0]) (void)hipEventRecord(timing[1], stream);

    // the clock words of the launch (the hand-off error word stays in the control block for readFsp4Error)
    uint32_t inboxWords[kClockWordsOffset + 4u] = {0};
    e = hipMemcpyAsync(inboxWords, controlArea, sizeof(inboxWords), hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    double clockGHz = 0.0;
    {
        unsigned long long ticks[2] = {0, 0};
        std::memcpy(ticks, inboxWords + kClockWordsOffset, sizeof(ticks));
        if (ticks[1]) clockGHz = double(ticks[0]) / double(ticks[1]) * 0.1;
    }
    float ms = -1.0f;
    if (!timing[0] || hipEventElapsedTime(&ms, timing[0], timing[1]) != hipSuccess) ms = -1.0f;
    lastLaunchInfo.form = 4;
    lastLaunchInfo.scanKernelMs = double(ms);
    lastLaunchInfo.waveColumnSteps = double(rowBlocks) * double(cellCount & 31u);
    lastLaunchInfo.inboxEntries = 0.0;
    lastLaunchInfo.segments = double(segments);
    lastLaunchInfo.fullRowCells = double(rows);
    lastLaunchInfo.matrixPairs = 64.0 * double(rowBlocks) * double(cellCount & ~31u);
    lastLaunchInfo.matrixKernelMs = double(ms);
    lastLaunchInfo.matrixClockGHz = clockGHz;
    *done = true;
    return hipSuccess;
}


// ---- for em2_scan_sharded.hip ----

const void* fsp4SymmetricKernelFor(uint32_t paddedDw, bool identity)
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

hipError_t launchExpandFragments(const uint32_t* sig32, uint32_t cellCount, uint32_t fragmentCount, void* out, uint32_t steps,
                                 float* terms, hipStream_t stream)
{
    expandFragmentsKernel<<<dim3((fragmentCount + 255u) / 256u), dim3(256), 0, stream>>>(sig32, cellCount, fragmentCount,
                                                                                         static_cast<FragmentWord4*>(out), steps);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !terms) return e;
    // (matrixTermCount(cellCount) entries; the 2048-bit steps never read them)
    const uint32_t termCount = uint32_t(matrixTermCount(cellCount));
    signatureTermsKernel<<<dim3((termCount + 255u) / 256u), dim3(256), 0, stream>>>(sig32, 2u * steps, cellCount, termCount, terms);
    return hipGetLastError();
}

hipError_t launchInboxReplay(bool identity, dim3 grid, dim3 block, size_t lds, hipStream_t stream, const Fsp4Args& args,
                             const uint64_t* sorted, uint64_t count)
{
    if (identity) fsp4InboxReplayKernel<true><<<grid, block, lds, stream>>>(args, sorted, count);
    else fsp4InboxReplayKernel<false><<<grid, block, lds, stream>>>(args, sorted, count);
    return hipGetLastError();
}

}  // namespace em2
