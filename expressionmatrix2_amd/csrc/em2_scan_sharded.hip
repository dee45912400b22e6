// em2_scan_sharded.hip -- the sharded symmetric scan of findSimilarPairs4 across GPUs (one process per GPU; the collectives
// between the phases are the caller's: csrc/em2_dist.hip, expressionmatrix2_amd/sharded.py): the tile kernels of the deferred
// square (v_xor/v_bcnt and matrix-core forms), the plan, the phases, their status, and the emulation that plays all ranks
// on one GPU for the tests.  The prefix phases run the kernels of em2_scan_symmetric.hip (taken by address through
// em2_scan_symmetric_device.h, which also holds the device code both units share).

#include "em2_scan_symmetric_device.h"

namespace em2 {
namespace {

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
template <bool WIDE = false>
__device__ __forceinline__ void tileMatrixBody(unsigned char* ldsRaw)
{
    constexpr int W32 = WIDE ? 64 : 32;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    FragmentWord4* tiles = reinterpret_cast<FragmentWord4*>(ldsRaw + kernelArgs()->matrixLdsOffset);
    volatile uint32_t* shared = reinterpret_cast<volatile uint32_t*>(ldsRaw + kernelArgs()->matrixLdsOffset + 4u * kMatrixTileWords * 16u);
    unsigned char* walkBlock = ldsRaw + kernelArgs()->matrixLdsOffset + 4u * kMatrixTileWords * 16u + 64u + wave * kMatrixWalkLdsBytes;
    if (threadIdx.x < 12u) shared[threadIdx.x] = 0u;         // (words 4..10: the convoy, see scanMatrixBody)
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
        // (1024 bits: -popcount / 2 of the lane's row; a row that is none starts so low that its results pass no bound)
        const float rowTerm = WIDE ? 0.f : rowValid ? aux->terms[row] : -4096.f;
        if (colBegin < commonEnd) {
            {
                // the walk logs what passes either bound; both sides of every record go to the inbox afterwards
                const uint32_t logCapacity = aux->logCapacity < kMatrixLogMargin ? kMatrixLogMargin : aux->logCapacity;
                WalkRecord* waveLog = reinterpret_cast<WalkRecord*>(aux->logs) + size_t(blockIdx.x * 4u + wave) * 64u * logCapacity;
                // The walk may start where the other walks of the XCD are and go around (the convoy, scanMatrixBody); here the
                // order of the columns means nothing -- both sides of every record go to the inbox, which is sorted -- so the
                // logs are emptied after every call, wherever it stopped.  range = [at, end), then [colBegin, start).
                for (uint32_t rowHalf = 0; rowHalf < (WIDE ? 2u : 1u); ++rowHalf) {
                    const uint32_t start = convoyStartColumn(aux, shared, seg, colBegin, commonEnd, ~0u);      // (no stops are counted here)
                    uint32_t at = start, end = commonEnd;
                    bool around = start == colBegin;            // (nothing left below the starting column)
                    for (;;) {
                        if (at >= end) {
                            if (around) break;
                            around = true;                      // (the call stopped before it could go around itself)
                            at = colBegin;
                            end = start;
                            continue;
                        }
                        uint32_t records[2] = {0u, 0u};
                        uint32_t next;
                        if (WIDE) {
                            next = scanTilesMatrixWide<true, true>(aux->fragments, aux->snap, at, end, 2u * fragmentBlock + rowHalf,
                                                                   2.f * kMatrixBits - 2.f * float(snapRow), rowHalf, waveLog, logCapacity, records,
                                                                   ldsAddress(tiles), ldsAddress(const_cast<uint32_t*>(shared)), ldsAddress(walkBlock));
                        } else if (EM2_DIAG_WORD(aux)) {
                            next = scanTilesMatrixPinned<true, true, true>((const void*)(uintptr_t)aux, aux->fragments, aux->snap, aux->terms, at, end,
                                                               2u * fragmentBlock, matrixBoundOf<false>(snapRow), rowTerm, waveLog, logCapacity,
                                                               records, ldsAddress(tiles), ldsAddress(const_cast<uint32_t*>(shared)),
                                                               ldsAddress(walkBlock));
                        } else {
                            next = scanTilesMatrixPinned<true, true, false>((const void*)(uintptr_t)aux, aux->fragments, aux->snap, aux->terms, at, end,
                                                               2u * fragmentBlock, matrixBoundOf<false>(snapRow), rowTerm, waveLog, logCapacity,
                                                               records, ldsAddress(tiles), ldsAddress(const_cast<uint32_t*>(shared)),
                                                               ldsAddress(walkBlock));
                        }
                        if ((next & kWalkInLowerColumns) != 0u) {
                            around = true;
                            end = start;
                            next &= ~kWalkInLowerColumns;
                        }
                        at = next;
                        if (!idle) {
                            if (WIDE) drainWalkLogs<true>(waveLog, logCapacity, records, lane, rowBase, cellCount, emitPos, emitEnd);
                            else drainWalkLogs(waveLog, logCapacity, records, lane, rowBase, cellCount, emitPos, emitEnd);
                        }
                    }
                }
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
fsp4TileMatrixPinnedKernel(Fsp4Args args)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
    tileMatrixBody<false>(ldsRaw);
}

// 2048-bit signatures
__global__ void __launch_bounds__(256, 2)
fsp4TileMatrixWideKernel(Fsp4Args args)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
    tileMatrixBody<true>(ldsRaw);
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
    p.offTerms = at;        at += align256(matrixTermCount(p.blocks * 64u) * 4u);       // ... and the 1024-bit steps' terms
    p.rankBytes = at;
    p.offGathered = at;     at += align256(size_t(p.capGathered) * 8u);
    p.offSorted = at;       at += align256(size_t(p.capGathered) * 8u);
    p.offTemp = at;         at += align256(p.sortTempBytes);
    p.totalBytes = at;
    p.eligible = true;
    return p;
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
            e = launchExpandFragments(sig32, cellCount, fragmentCount, ws + plan.offFragments, matrixSteps,
                                      reinterpret_cast<float*>(ws + plan.offTerms), stream);
            if (e != hipSuccess) return e;
            args.fragments = ws + plan.offFragments;
            args.terms = reinterpret_cast<const float*>(ws + plan.offTerms);
            args.matrixLdsOffset = uint32_t(matrixLdsOffset);
            // No convoy here unless a test forces one (EM2_MATRIX_CONVOY >= 2): these launches are a few items per block, which
            // start together and stay together (measured at 1M cells, four ranks: 16.2 ms per launch without, 16.8 with it).
            args.convoy = uint32_t(envNumber("EM2_MATRIX_CONVOY", 0));
            if (args.convoy == 1u) args.convoy = 0u;
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
            if (blocksWanted * 8u > maxResidentWaves()) blocksWanted = maxResidentWaves() / 8u;
            if (blocksWanted > tickets) blocksWanted = tickets;
            e = hipFuncSetAttribute(matrixKernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(matrixLds));
            if (e != hipSuccess) return e;
            void* matrixArgsArray[] = {&args};
            return hipLaunchKernel(matrixKernel, dim3(uint32_t(blocksWanted)), dim3(256), matrixArgsArray, matrixLds, stream);
        }
        const void* kernel = fsp4SymmetricKernelFor(paddedDw, t.identityKeys);
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
            e = launchExpandFragments(sig32, cellCount, fragmentCount, ws + plan.offFragments, matrixSteps,
                                      reinterpret_cast<float*>(ws + plan.offTerms), stream);
            if (e != hipSuccess) return e;
            args.fragments = ws + plan.offFragments;
            args.terms = reinterpret_cast<const float*>(ws + plan.offTerms);
            args.matrixLdsOffset = 0u;
            // the tiles' walks go around their segments in convoys (scanMatrixBody; 30.5 -> 29.8 ms per launch at 1M cells, four
            // ranks); the position words start at zero
            args.convoy = uint32_t(envNumber("EM2_MATRIX_CONVOY", 1));
            if (cps / 64u > kConvoyMaxPairs && args.convoy == 1u) args.convoy = 0u;
            e = hipMemsetAsync(args.inboxControl + kConvoyWordsOffset, 0, 64u, stream);
            if (e != hipSuccess) return e;
            const size_t matrixLds = args.matrixLdsOffset + kMatrixLdsBytes;
            uint64_t blocksWanted = uint64_t(cuCount) * 2u;
            if (blocksWanted > own) blocksWanted = own;
            const void* tileMatrixKernel = wide ? reinterpret_cast<const void*>(&fsp4TileMatrixWideKernel)
                                                : reinterpret_cast<const void*>(&fsp4TileMatrixPinnedKernel);
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
        return launchInboxReplay(t.identityKeys, rgrid, block, lds, stream, args, sorted, gatheredCount);
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
// EM2_SCAN_MODE=virtual:P).  *done = false: not eligible or an entry pool overflowed; the caller runs the ordered scan.
hipError_t runFsp4ShardedEmulation(const uint32_t* sig32, uint32_t paddedDw, uint32_t cellCount, uint32_t k,
                                          const DeviceTables& t, PairOut* outPairs, uint32_t* outUsed, uint32_t world,
                                          hipStream_t stream, bool* done)
{
    *done = false;
    std::vector<Fsp4ShardPlan> plans;
    for (uint32_t r = 0; r < world; ++r) plans.push_back(fsp4ShardPlan(cellCount, k, r, world));
    if (!plans[0].eligible) return hipSuccess;
    const Fsp4ShardPlan& p0 = plans[0];
    const bool verbose = scanVerbose();
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
    // all_to_all -- a rank receives, sorts and replays only the candidates of its own cells; otherwise every rank gathers every pool.  The routed form is played with device-to-device copies
    // through one staging area per receiver; its grouping sort is timed as phase 4.
    const bool routed = (world & (world - 1u)) == 0u && world > 1u;
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

