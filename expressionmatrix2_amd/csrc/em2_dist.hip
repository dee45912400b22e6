// em2_dist.hip -- findSimilarPairs4 across the GPUs of a node from C: em2_dist_find_similar_pairs4 (RCCL communicator) and
// em2_dist_find_similar_pairs4_with (any transport, as a table of four collectives).  One process per GPU; every rank
// calls with the same arguments and its own shard of the signatures, and ends with the SimilarPairs rows of its own
// contiguous cell range (the partitioning of north_star).  Results are those of
// src/ExpressionMatrixLsh.cpp:155-290 on all cells.
//
// Two forms, agreed between the ranks:
//   rows       all_gather of the signature shards; every rank scans its own rows against all columns
//              (em2_dev_find_similar_pairs4).  One collective, no exchange of results.
//   symmetric  every unordered pair once ACROSS the ranks (em2_dev_fsp4_sharded_*): all_gather of the signatures, phase
//              0, all_reduce(MAX) of the snapshots, phase 1, all_reduce(MAX), phase 2, the ranks agree on entry counts and
//              overflow (all_gather of one small vector, the one host read-back of the exchange), the deferred candidates
//              travel to the owners of their target cells (all_to_all; all_gather when world is not a power of two),
//              phase 3 finishes the cells of the blocks a rank owns (64-cell blocks dealt round-robin), and one more
//              all_to_all with sizes known from arithmetic alone moves the finished rows to the ranks of their contiguous
//              ranges.  A pool overflow on any rank sends all ranks to the rows form together.
// The choreography is what expressionmatrix2_amd/sharded.py (DevicePipeline) does through torch.distributed; this file is
// the same for a C or C++ host (depth 2).
//
// RCCL is bound at run time: the nccl* entry points are looked up in the process first (the library the caller created
// the communicator with), then in librccl.so.1 / librccl.so -- libem2lsh.so itself has no link-time dependency on it.

#include "../../include/em2_lsh.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" void em2_internal_set_last_error(const char* message);

namespace {

int fail(int code, const std::string& message)
{
    em2_internal_set_last_error(message.c_str());
    return code;
}

#define EM2_DIST_HIP(call)                                                                                  \
    do {                                                                                                    \
        hipError_t em2HipError_ = (call);                                                                   \
        if (em2HipError_ != hipSuccess) return fail(EM2_ERROR_HIP, std::string(#call) + ": " + hipGetErrorString(em2HipError_)); \
    } while (0)

#define EM2_DIST_OK(call)                      \
    do {                                       \
        const int em2Rc_ = (call);             \
        if (em2Rc_ != EM2_OK) return em2Rc_;   \
    } while (0)

size_t alignUp(size_t x) { return (x + 255u) & ~size_t(255u); }

uint32_t shardSize(uint32_t cellCount, uint32_t world) { return (cellCount + world - 1u) / world; }

void shardRange(uint32_t cellCount, uint32_t world, uint32_t rank, uint32_t& begin, uint32_t& end)
{
    const uint64_t size = shardSize(cellCount, world);
    begin = uint32_t(size * rank < cellCount ? size * rank : cellCount);
    end = uint32_t(uint64_t(begin) + size < cellCount ? uint64_t(begin) + size : cellCount);
}

// ---- device helpers of the symmetric form ----

// sorted[0, n) is grouped by owner = (key >> shift) & (world - 1), ascending; bounds[r] = first index whose owner >= r,
// bounds[world] = n.
__global__ void ownerBoundsKernel(const uint64_t* __restrict__ sorted, uint64_t n, uint32_t shift, uint32_t world,
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

// out[i] = in[rows[i]] (rows of k pairs and their used counts): packs the finished rows of the blocks this rank owns in
// ascending cell order (which is grouped by destination rank), or scatters received rows into the shard.
// The outcome flag of a rank, written from the launch's own argument (no host buffer has to outlive the call).
__global__ void setOutcomeFlagKernel(int32_t* flag, int32_t value) { *flag = value; }

__global__ void __launch_bounds__(256)
moveRowsKernel(const em2_pair* __restrict__ inPairs, const uint32_t* __restrict__ inUsed, const uint32_t* __restrict__ inRow,
               const uint32_t* __restrict__ outRow, uint32_t rowCount, uint32_t k, em2_pair* __restrict__ outPairs,
               uint32_t* __restrict__ outUsed)
{
    const uint32_t i = blockIdx.x;
    if (i >= rowCount) return;
    const size_t from = inRow ? inRow[i] : i, to = outRow ? outRow[i] : i;
    for (uint32_t j = threadIdx.x; j < k; j += blockDim.x) outPairs[to * k + j] = inPairs[from * k + j];
    if (threadIdx.x == 0) outUsed[to] = inUsed[from];
}

struct Layout {
    bool symmetric;
    uint64_t plan[12];
    size_t rowScan, rowScanBytes;           // em2_dev_find_similar_pairs4 workspace (rows form, and the fallback)
    size_t sharded;                          // the plan's workspace
    size_t globalPairs, globalUsed;          // [cellCount][k], [cellCount]
    size_t staging, stagingUsed;             // owned rows packed for the redistribution
    size_t received, receivedUsed;           // rows of the shard as received (grouped by source rank)
    size_t ownedRows, shardRows;             // uint32 index lists
    size_t counts;                           // uint64[world * (world + 2)] + bounds
    size_t total;
    uint32_t ownedRowCount;
};

uint32_t ownedRowCountOf(uint32_t cellCount, uint32_t rank, uint32_t world)
{
    const uint32_t blocks = (cellCount + 63u) / 64u;
    uint32_t n = 0;
    for (uint32_t b = rank; b < blocks; b += world) n += (uint64_t(b) * 64u + 64u <= cellCount) ? 64u : cellCount - b * 64u;
    return n;
}

bool symmetricWanted(uint32_t cellCount, uint32_t world, uint32_t k)
{
    if (k == 0) return false;
    if (world < 2 && !(getenv("EM2_SHARDED_WORLD_ONE") && getenv("EM2_SHARDED_WORLD_ONE")[0] == '1')) return false;       // (tests: RCCL on one rank)
    const char* v = getenv("EM2_SHARDED_SCAN");
    if (v && v[0] == '0') return false;
    const char* m = getenv("EM2_SHARDED_MIN_CELLS");
    const uint64_t minCells = m ? strtoull(m, nullptr, 10) : 100000ull;           // as sharded.py
    return cellCount >= minCells;
}

Layout layoutOf(uint32_t cellCount, uint32_t lshCount, uint32_t k, uint32_t rank, uint32_t world)
{
    Layout l;
    std::memset(&l, 0, sizeof(l));
    uint32_t begin = 0, end = 0;
    shardRange(cellCount, world, rank, begin, end);
    const uint32_t rows = end - begin;
    size_t at = 0;
    l.rowScan = at;
    l.rowScanBytes = em2_dev_find_similar_pairs4_workspace(cellCount, rows, lshCount, k);
    at += alignUp(l.rowScanBytes);
    // (the owner of a deferred candidate's target travels in eight bits of the entry: beyond 255 ranks every rank takes the
    // always-valid rows form, by this same arithmetic)
    if (world <= 255u && symmetricWanted(cellCount, world, k)) {
        em2_dev_fsp4_sharded_plan(cellCount, lshCount, k, rank, world, l.plan, 12);
        l.symmetric = l.plan[0] != 0;
    }
    if (l.symmetric) {
        l.ownedRowCount = ownedRowCountOf(cellCount, rank, world);
        l.sharded = at;       at += alignUp(size_t(l.plan[1]));
        l.globalPairs = at;   at += alignUp(size_t(cellCount) * k * sizeof(em2_pair));
        l.globalUsed = at;    at += alignUp(size_t(cellCount) * 4u);
        l.staging = at;       at += alignUp(size_t(l.ownedRowCount) * k * sizeof(em2_pair));
        l.stagingUsed = at;   at += alignUp(size_t(l.ownedRowCount) * 4u);
        l.received = at;      at += alignUp(size_t(rows) * k * sizeof(em2_pair));
        l.receivedUsed = at;  at += alignUp(size_t(rows) * 4u);
        l.ownedRows = at;     at += alignUp(size_t(l.ownedRowCount) * 4u);
        l.shardRows = at;     at += alignUp(size_t(rows) * 4u);
        l.counts = at;        at += alignUp(size_t(world) * (world + 4u) * 8u);
    }
    l.total = at + 256u;
    return l;
}

struct Timer {
    double* out;
    hipStream_t stream;
    std::chrono::steady_clock::time_point last;
    Timer(double* o, hipStream_t s) : out(o), stream(s), last(std::chrono::steady_clock::now()) {}
    // Closes a stage: only when the caller asked for timings does this synchronise the stream.
    void stage(int index)
    {
        if (!out) return;
        (void)hipStreamSynchronize(stream);
        const auto now = std::chrono::steady_clock::now();
        out[index] += std::chrono::duration<double, std::milli>(now - last).count();
        last = now;
    }
};

int rowsForm(const em2_collectives* c, const uint64_t* dAll, uint32_t cellCount, uint32_t lshCount, uint32_t k, double thr,
             em2_pair* dPairs, uint32_t* dUsed, char* ws, const Layout& l, hipStream_t stream, Timer& timer)
{
    uint32_t begin = 0, end = 0;
    shardRange(cellCount, uint32_t(c->world), uint32_t(c->rank), begin, end);
    if (end > begin) {
        EM2_DIST_OK(em2_dev_find_similar_pairs4(dAll, cellCount, begin, end, lshCount, k, thr, dPairs, dUsed, ws + l.rowScan,
                                                l.rowScanBytes, stream));
        if (k) EM2_DIST_OK(em2_dev_find_similar_pairs4_status(ws + l.rowScan, end - begin, k, stream));
    }
    timer.stage(EM2_DIST_MS_SCAN);
    return EM2_OK;
}

}  // namespace


extern "C" {

size_t em2_dist_find_similar_pairs4_workspace(uint32_t cellCount, uint32_t lshCount, uint32_t k, uint32_t rank, uint32_t world)
{
    if (lshCount == 0 || world == 0 || rank >= world) return 0;
    return layoutOf(cellCount, lshCount, k, rank, world).total;
}

int em2_dist_find_similar_pairs4_form(uint32_t cellCount, uint32_t lshCount, uint32_t k, uint32_t world)
{
    if (lshCount == 0 || world == 0) return 0;
    return layoutOf(cellCount, lshCount, k, 0, world).symmetric ? 2 : 0;
}

int em2_dist_find_similar_pairs4_with(const em2_collectives* c, const uint64_t* d_localSignatures, uint32_t cellCount,
                                      uint32_t lshCount, uint32_t k, double similarityThreshold, uint64_t* d_allSignatures,
                                      em2_pair* d_pairs, uint32_t* d_usedCount, void* d_workspace, size_t workspaceBytes,
                                      void* streamArg, double* stageMs)
{
    if (!c || !c->all_gather || !c->all_reduce_max_i32 || !c->all_to_all_v) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dist_find_similar_pairs4: incomplete collective table");
    if (c->world < 1 || c->rank < 0 || c->rank >= c->world) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dist_find_similar_pairs4: rank / world out of range");
    if (lshCount == 0) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dist_find_similar_pairs4: lshCount must be positive");
    if (cellCount == 0) return EM2_OK;
    if (!d_localSignatures || !d_allSignatures || !d_usedCount || (!d_pairs && k) || !d_workspace) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dist_find_similar_pairs4: null pointer");
    const uint32_t world = uint32_t(c->world), rank = uint32_t(c->rank);
    const Layout l = layoutOf(cellCount, lshCount, k, rank, world);
    char* ws = reinterpret_cast<char*>(alignUp(reinterpret_cast<size_t>(d_workspace)));
    if (workspaceBytes < l.total) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dist_find_similar_pairs4: workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(streamArg);
    if (stageMs) for (int i = 0; i < EM2_DIST_MS_COUNT; i++) stageMs[i] = 0.;
    Timer timer(stageMs, stream);

    const uint32_t words = (lshCount - 1u) / 64u + 1u;
    const uint32_t shard = shardSize(cellCount, world);
    uint32_t begin = 0, end = 0;
    shardRange(cellCount, world, rank, begin, end);
    const uint32_t rows = end - begin;

    // ---- the one exchange of north_star: the signature shards (padded to the common shard size) ----
    if (c->all_gather(c->context, d_localSignatures, d_allSignatures, size_t(shard) * words * 8u, stream) != 0) {
        return fail(EM2_ERROR_RUNTIME, "em2_dist_find_similar_pairs4: all_gather of the signatures failed");
    }
    timer.stage(EM2_DIST_MS_GATHER_SIGNATURES);
    if (!l.symmetric) return rowsForm(c, d_allSignatures, cellCount, lshCount, k, similarityThreshold, d_pairs, d_usedCount, ws, l, stream, timer);

    // ---- symmetric form ----
    // A failure on ONE rank must not leave the others blocked in a collective that rank never enters.  Up to the agreement
    // point a local failure (a phase's status, a HIP error, a hand-off time-out) is therefore only recorded: the rank goes
    // on issuing every collective of the choreography -- the phases it can no longer run are skipped -- and reports
    // overflow = 1, so that ALL ranks take the rows form together, after which this rank returns the recorded error.
    // What is checked after the agreement (capacities of the exchange areas) is evaluated by every rank for every rank
    // from the gathered count matrix, i.e. identically everywhere.  A failure behind the exchange (phase 3, the copies
    // of the redistribution) is recorded as well; the rank keeps entering the collectives of the redistribution -- so its
    // peers receive rows that mean nothing -- and the call therefore ends with ONE more reduction, of the ranks' failure
    // flags: if any rank failed behind the agreement, EVERY rank returns an error (its own, or "another rank failed"),
    // nobody hands back rows that look valid and are not.  Only a collective that itself reports failure ends the call at
    // once: the transport is gone, there is nothing left to keep in step with.
    // (As in expressionmatrix2_amd/sharded.py: guarded() / the agreed overflow flag.)
    em2_pair* globalPairs = reinterpret_cast<em2_pair*>(ws + l.globalPairs);
    uint32_t* globalUsed = reinterpret_cast<uint32_t*>(ws + l.globalUsed);
    char* shardWs = ws + l.sharded;
    int localError = EM2_OK;
    std::string localMessage;
    auto record = [&](int rc) {
        if (rc != EM2_OK && localError == EM2_OK) {
            localError = rc;
            localMessage = em2_last_error();
        }
        return rc == EM2_OK;
    };
    auto recordHip = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && localError == EM2_OK) {
            localError = EM2_ERROR_HIP;
            localMessage = std::string(what) + ": " + hipGetErrorString(e);
        }
        return e == hipSuccess;
    };
    auto phase = [&](int number, uint64_t gatheredCount) {
        if (localError != EM2_OK) return false;
#ifdef EM2_DIAG
        // fault injection (diagnostic build only, tests/test_gpu_dist_entry.py): phase EM2_DIST_FAIL_PHASE fails on rank EM2_DIST_FAIL_RANK
        if (getenv("EM2_DIST_FAIL_PHASE") && getenv("EM2_DIST_FAIL_RANK") && atoi(getenv("EM2_DIST_FAIL_PHASE")) == number &&
            uint32_t(atoi(getenv("EM2_DIST_FAIL_RANK"))) == rank) {
            return record(fail(EM2_ERROR_RUNTIME, "em2_dist_find_similar_pairs4: injected failure of phase " + std::to_string(number)));
        }
#endif
        return record(em2_dev_fsp4_sharded_phase(number, d_allSignatures, cellCount, lshCount, k, similarityThreshold, rank, world, globalPairs,
                                                 globalUsed, shardWs, size_t(l.plan[1]), gatheredCount, stream));
    };
    int32_t* snap = reinterpret_cast<int32_t*>(shardWs + l.plan[2]);
    uint64_t* pool = reinterpret_cast<uint64_t*>(shardWs + l.plan[3]);
    uint64_t* gathered = reinterpret_cast<uint64_t*>(shardWs + l.plan[5]);
    uint64_t* sorted = reinterpret_cast<uint64_t*>(shardWs + l.plan[10]);
    uint64_t* counts = reinterpret_cast<uint64_t*>(ws + l.counts);            // [world][world + 2] gathered, then bounds[world + 1]
    const uint32_t stride = world + 2u;

    phase(0, 0);
    timer.stage(EM2_DIST_MS_SCAN);
    if (c->all_reduce_max_i32(c->context, snap, cellCount, stream) != 0) return fail(EM2_ERROR_RUNTIME, "em2_dist_find_similar_pairs4: all_reduce failed");
    timer.stage(EM2_DIST_MS_ALL_REDUCE);
    phase(1, 0);
    timer.stage(EM2_DIST_MS_SCAN);
    if (c->all_reduce_max_i32(c->context, snap, cellCount, stream) != 0) return fail(EM2_ERROR_RUNTIME, "em2_dist_find_similar_pairs4: all_reduce failed");
    timer.stage(EM2_DIST_MS_ALL_REDUCE);
    phase(2, 0);
    uint64_t used = 0;
    uint32_t overflow = 0;
    if (localError == EM2_OK) record(em2_dev_fsp4_sharded_status(cellCount, k, rank, world, shardWs, stream, &used, &overflow));
    timer.stage(EM2_DIST_MS_SCAN);

    // ---- what every rank sends to every rank, and whether anybody overflowed or failed: one small all_gather, one read-back ----
    const bool routed = (world & (world - 1u)) == 0u;          // (the owner of a target cell is a bit field of the entry key)
    std::vector<uint64_t> mine(stride, 0), all(size_t(world) * stride, 0);
    if (routed && !overflow && localError == EM2_OK) {
        if (used) phase(4, used);
        uint64_t* bounds = counts + size_t(world) * stride;
        std::vector<uint64_t> hostBounds(world + 1u, 0);
        if (localError == EM2_OK) {
            ownerBoundsKernel<<<dim3(1), dim3(256), 0, stream>>>(sorted, used, uint32_t(l.plan[11]), world, bounds);
            recordHip(hipGetLastError(), "ownerBoundsKernel") &&
                recordHip(hipMemcpyAsync(hostBounds.data(), bounds, (world + 1u) * 8u, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync(bounds)") &&
                recordHip(hipStreamSynchronize(stream), "hipStreamSynchronize");
        }
        if (localError == EM2_OK) {
            for (uint32_t r = 0; r < world; r++) mine[r] = hostBounds[r + 1u] - hostBounds[r];
        }
    }
    if (localError != EM2_OK) {
        used = 0;
        overflow = 1;
        for (uint32_t r = 0; r < world; r++) mine[r] = 0;
    }
    mine[world] = used;
    mine[world + 1u] = overflow;
    // (a failed copy of this rank's own row would send stale counts: the row is staged through a pinned-size host vector and
    // the flag travels in it, so a HIP error here is the one place where this rank cannot tell the others -- the transport
    // call below then fails or the others time out; such an error means the device is lost)
    EM2_DIST_HIP(hipMemcpyAsync(counts + size_t(rank) * stride, mine.data(), stride * 8u, hipMemcpyHostToDevice, stream));
    if (c->all_gather(c->context, counts + size_t(rank) * stride, counts, stride * 8u, stream) != 0) return fail(EM2_ERROR_RUNTIME, "em2_dist_find_similar_pairs4: all_gather of the counts failed");
    EM2_DIST_HIP(hipMemcpyAsync(all.data(), counts, all.size() * 8u, hipMemcpyDeviceToHost, stream));
    EM2_DIST_HIP(hipStreamSynchronize(stream));
    // ---- the agreement: every rank evaluates the same matrix the same way ----
    bool fallBack = false;
    uint64_t maxUsed = 0;
    for (uint32_t r = 0; r < world; r++) {
        fallBack = fallBack || all[size_t(r) * stride + world + 1u] != 0;
        if (all[size_t(r) * stride + world] > maxUsed) maxUsed = all[size_t(r) * stride + world];
    }
    if (routed) {
        for (uint32_t receiver = 0; receiver < world; receiver++) {        // what each rank would receive must fit its exchange area
            uint64_t entries = 0;
            for (uint32_t sender = 0; sender < world; sender++) entries += all[size_t(sender) * stride + receiver];
            fallBack = fallBack || entries > l.plan[6];
        }
    } else {
        fallBack = fallBack || uint64_t(world) * maxUsed > l.plan[6];
    }
    timer.stage(EM2_DIST_MS_EXCHANGE);
    if (fallBack) {
        const int rc = rowsForm(c, d_allSignatures, cellCount, lshCount, k, similarityThreshold, d_pairs, d_usedCount, ws, l, stream, timer);
        if (localError != EM2_OK) return fail(localError, localMessage);
        return rc;
    }

    // ---- the deferred candidates travel to the owners of their target cells ----
    uint64_t receivedEntries = 0;
    if (routed) {
        std::vector<uint64_t> sendBytes(world), sendOffsets(world), recvBytes(world), recvOffsets(world);
        uint64_t so = 0, ro = 0;
        for (uint32_t r = 0; r < world; r++) {
            sendBytes[r] = mine[r] * 8u;
            sendOffsets[r] = so;
            so += sendBytes[r];
            recvBytes[r] = all[size_t(r) * stride + rank] * 8u;
            recvOffsets[r] = ro;
            ro += recvBytes[r];
        }
        receivedEntries = ro / 8u;
        if (c->all_to_all_v(c->context, sorted, sendBytes.data(), sendOffsets.data(), gathered, recvBytes.data(), recvOffsets.data(), stream) != 0) {
            return fail(EM2_ERROR_RUNTIME, "em2_dist_find_similar_pairs4: all_to_all of the deferred candidates failed");
        }
    } else if (maxUsed) {
        if (used < maxUsed) recordHip(hipMemsetAsync(pool + used, 0xff, (maxUsed - used) * 8u, stream), "hipMemsetAsync(pool tail)");      // ~0 sorts behind every entry
        if (c->all_gather(c->context, pool, gathered, maxUsed * 8u, stream) != 0) return fail(EM2_ERROR_RUNTIME, "em2_dist_find_similar_pairs4: all_gather of the deferred candidates failed");
        receivedEntries = uint64_t(world) * maxUsed;
    }
    timer.stage(EM2_DIST_MS_EXCHANGE);
    // The outcome word (the first word of the count matrix, which has served) says "failed" from here until this rank
    // clears it at the end: a rank whose device falls into a sticky error behind this point cannot write its flag any more,
    // and must not enter the reduction with a stale zero.
    int32_t* dFailed = reinterpret_cast<int32_t*>(counts);
    recordHip(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(dFailed), 1, 1, stream), "hipMemsetD32Async(outcome flag)");
    phase(3, receivedEntries);          // (a failure from here on is returned after the last collective)
    timer.stage(EM2_DIST_MS_SCAN);

    // ---- finished rows go from the ranks that own their blocks to the ranks of their contiguous ranges ----
    // Rows a rank owns, ascending, are grouped by destination (the ranges ascend with the rank); the rows of a shard as
    // received are grouped by source, each group ascending.  Both lists follow from arithmetic alone.
    std::vector<uint32_t> ownedRows, shardRows;
    std::vector<uint64_t> sendBytes(world, 0), sendOffsets(world, 0), recvBytes(world, 0), recvOffsets(world, 0);
    const uint32_t blocks = (cellCount + 63u) / 64u;
    const size_t rowBytes = size_t(k) * sizeof(em2_pair);
    ownedRows.reserve(l.ownedRowCount);
    for (uint32_t b = rank; b < blocks; b += world) {
        for (uint32_t cell = b * 64u; cell < cellCount && cell < b * 64u + 64u; cell++) {
            ownedRows.push_back(cell);
            sendBytes[cell / shard] += 1;
        }
    }
    shardRows.reserve(rows);
    for (uint32_t source = 0; source < world; source++) {
        for (uint32_t cell = begin; cell < end; cell++) {
            if ((cell / 64u) % world == source) {
                shardRows.push_back(cell - begin);
                recvBytes[source] += 1;
            }
        }
    }
    uint32_t* dOwnedRows = reinterpret_cast<uint32_t*>(ws + l.ownedRows);
    uint32_t* dShardRows = reinterpret_cast<uint32_t*>(ws + l.shardRows);
    em2_pair* staging = reinterpret_cast<em2_pair*>(ws + l.staging);
    uint32_t* stagingUsed = reinterpret_cast<uint32_t*>(ws + l.stagingUsed);
    em2_pair* received = reinterpret_cast<em2_pair*>(ws + l.received);
    uint32_t* receivedUsed = reinterpret_cast<uint32_t*>(ws + l.receivedUsed);
    if (!ownedRows.empty() && localError == EM2_OK) {
        if (recordHip(hipMemcpyAsync(dOwnedRows, ownedRows.data(), ownedRows.size() * 4u, hipMemcpyHostToDevice, stream), "hipMemcpyAsync(owned rows)")) {
            moveRowsKernel<<<dim3(uint32_t(ownedRows.size())), dim3(64), 0, stream>>>(globalPairs, globalUsed, dOwnedRows, nullptr,
                                                                                     uint32_t(ownedRows.size()), k, staging, stagingUsed);
            recordHip(hipGetLastError(), "moveRowsKernel");
        }
    }
    for (int pass = 0; pass < 2; pass++) {          // the pairs, then the used counts
        const size_t unit = pass == 0 ? rowBytes : 4u;
        if (unit == 0) continue;
        std::vector<uint64_t> sb(world), so(world), rb(world), ro(world);
        uint64_t s = 0, r = 0;
        for (uint32_t p = 0; p < world; p++) {
            sb[p] = sendBytes[p] * unit;
            so[p] = s;
            s += sb[p];
            rb[p] = recvBytes[p] * unit;
            ro[p] = r;
            r += rb[p];
        }
        const void* from = pass == 0 ? static_cast<const void*>(staging) : static_cast<const void*>(stagingUsed);
        void* to = pass == 0 ? static_cast<void*>(received) : static_cast<void*>(receivedUsed);
        if (c->all_to_all_v(c->context, from, sb.data(), so.data(), to, rb.data(), ro.data(), stream) != 0) {
            return fail(EM2_ERROR_RUNTIME, "em2_dist_find_similar_pairs4: all_to_all of the finished rows failed");
        }
    }
    if (rows && localError == EM2_OK) {
        if (recordHip(hipMemcpyAsync(dShardRows, shardRows.data(), shardRows.size() * 4u, hipMemcpyHostToDevice, stream), "hipMemcpyAsync(shard rows)")) {
            moveRowsKernel<<<dim3(rows), dim3(64), 0, stream>>>(received, receivedUsed, nullptr, dShardRows, rows, k, d_pairs, d_usedCount);
            recordHip(hipGetLastError(), "moveRowsKernel");
        }
    }
    recordHip(hipStreamSynchronize(stream), "hipStreamSynchronize");          // the host lists above must outlive the copies
    // ---- the outcome is collective: a rank that failed behind the agreement has sent rows that mean nothing ----
    // (nothing returns between the last all_to_all and this reduction.  The flag is written by a kernel from its own argument:
    // no copy out of this frame is left queued behind a return; if the launch fails -- a device in a sticky error -- the word
    // keeps the 1 it was preset to.  Whatever happens, the stream is drained before the call returns: the reduction runs on the
    // caller's workspace.)
    int32_t failed = localError != EM2_OK ? 1 : 0;
    setOutcomeFlagKernel<<<dim3(1), dim3(1), 0, stream>>>(dFailed, failed);
    recordHip(hipGetLastError(), "setOutcomeFlagKernel");
    const bool reduced = c->all_reduce_max_i32(c->context, dFailed, 1, stream) == 0;
    if (reduced && localError == EM2_OK) recordHip(hipMemcpyAsync(&failed, dFailed, 4u, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync(outcome)");
    recordHip(hipStreamSynchronize(stream), "hipStreamSynchronize");
    if (!reduced) return fail(EM2_ERROR_RUNTIME, "em2_dist_find_similar_pairs4: all_reduce of the outcome failed");
    timer.stage(EM2_DIST_MS_REDISTRIBUTE);
    if (localError != EM2_OK) return fail(localError, localMessage);
    if (failed) return fail(EM2_ERROR_RUNTIME, "em2_dist_find_similar_pairs4: another rank failed behind the ranks' agreement (phase 3 or the "
                                               "redistribution of the finished rows); the rows received from it are not valid");
    return EM2_OK;
}


// ---------------------------------------------------------------------------------------------------------
// The RCCL transport.
// ---------------------------------------------------------------------------------------------------------

namespace {

struct Rccl {
    decltype(&ncclAllGather) allGather = nullptr;
    decltype(&ncclAllReduce) allReduce = nullptr;
    decltype(&ncclGroupStart) groupStart = nullptr;
    decltype(&ncclGroupEnd) groupEnd = nullptr;
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclCommCount) commCount = nullptr;
    decltype(&ncclCommUserRank) commUserRank = nullptr;
    decltype(&ncclGetErrorString) errorString = nullptr;
    bool bound = false;
    std::string error;
};

Rccl& rccl()
{
    static Rccl r;
    if (r.bound || !r.error.empty()) return r;
    void* handles[3] = {RTLD_DEFAULT, nullptr, nullptr};
    for (int attempt = 0; attempt < 3; attempt++) {
        void* h = handles[attempt];
        if (attempt == 1) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (attempt == 2) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (attempt > 0 && !h) continue;
        if (!dlsym(h, "ncclAllGather")) continue;
        r.allGather = reinterpret_cast<decltype(r.allGather)>(dlsym(h, "ncclAllGather"));
        r.allReduce = reinterpret_cast<decltype(r.allReduce)>(dlsym(h, "ncclAllReduce"));
        r.groupStart = reinterpret_cast<decltype(r.groupStart)>(dlsym(h, "ncclGroupStart"));
        r.groupEnd = reinterpret_cast<decltype(r.groupEnd)>(dlsym(h, "ncclGroupEnd"));
        r.send = reinterpret_cast<decltype(r.send)>(dlsym(h, "ncclSend"));
        r.recv = reinterpret_cast<decltype(r.recv)>(dlsym(h, "ncclRecv"));
        r.commCount = reinterpret_cast<decltype(r.commCount)>(dlsym(h, "ncclCommCount"));
        r.commUserRank = reinterpret_cast<decltype(r.commUserRank)>(dlsym(h, "ncclCommUserRank"));
        r.errorString = reinterpret_cast<decltype(r.errorString)>(dlsym(h, "ncclGetErrorString"));
        r.bound = r.allGather && r.allReduce && r.groupStart && r.groupEnd && r.send && r.recv && r.commCount && r.commUserRank;
        if (r.bound) return r;
    }
    r.error = "RCCL is not loaded in this process and librccl.so.1 / librccl.so cannot be opened";
    return r;
}

struct RcclContext {
    ncclComm_t comm;
    int world;
};

int rcclAllGather(void* context, const void* send, void* recv, size_t bytesPerRank, void* stream)
{
    RcclContext* x = static_cast<RcclContext*>(context);
    return rccl().allGather(send, recv, bytesPerRank, ncclUint8, x->comm, static_cast<hipStream_t>(stream)) == ncclSuccess ? 0 : 1;
}

int rcclAllReduceMaxI32(void* context, void* buffer, size_t count, void* stream)
{
    RcclContext* x = static_cast<RcclContext*>(context);
    return rccl().allReduce(buffer, buffer, count, ncclInt32, ncclMax, x->comm, static_cast<hipStream_t>(stream)) == ncclSuccess ? 0 : 1;
}

int rcclAllToAllV(void* context, const void* send, const uint64_t* sendBytes, const uint64_t* sendOffsets, void* recv,
                  const uint64_t* recvBytes, const uint64_t* recvOffsets, void* stream)
{
    RcclContext* x = static_cast<RcclContext*>(context);
    Rccl& r = rccl();
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (r.groupStart() != ncclSuccess) return 1;
    bool ok = true;
    for (int p = 0; p < x->world; p++) {
        if (sendBytes[p]) ok = ok && r.send(static_cast<const char*>(send) + sendOffsets[p], sendBytes[p], ncclUint8, p, x->comm, s) == ncclSuccess;
        if (recvBytes[p]) ok = ok && r.recv(static_cast<char*>(recv) + recvOffsets[p], recvBytes[p], ncclUint8, p, x->comm, s) == ncclSuccess;
    }
    return (r.groupEnd() == ncclSuccess && ok) ? 0 : 1;
}

}  // namespace

int em2_dist_find_similar_pairs4(void* ncclCommunicator, const uint64_t* d_localSignatures, uint32_t cellCount, uint32_t lshCount,
                                 uint32_t k, double similarityThreshold, uint64_t* d_allSignatures, em2_pair* d_pairs,
                                 uint32_t* d_usedCount, void* d_workspace, size_t workspaceBytes, void* stream, double* stageMs)
{
    if (!ncclCommunicator) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dist_find_similar_pairs4: null communicator");
    Rccl& r = rccl();
    if (!r.bound) return fail(EM2_ERROR_UNSUPPORTED, "em2_dist_find_similar_pairs4: " + r.error);
    RcclContext context;
    context.comm = static_cast<ncclComm_t>(ncclCommunicator);
    int world = 0, rank = 0;
    if (r.commCount(context.comm, &world) != ncclSuccess || r.commUserRank(context.comm, &rank) != ncclSuccess) {
        return fail(EM2_ERROR_RUNTIME, "em2_dist_find_similar_pairs4: the communicator does not answer ncclCommCount / ncclCommUserRank");
    }
    context.world = world;
    em2_collectives table;
    table.context = &context;
    table.world = world;
    table.rank = rank;
    table.all_gather = rcclAllGather;
    table.all_reduce_max_i32 = rcclAllReduceMaxI32;
    table.all_to_all_v = rcclAllToAllV;
    return em2_dist_find_similar_pairs4_with(&table, d_localSignatures, cellCount, lshCount, k, similarityThreshold, d_allSignatures, d_pairs,
                                             d_usedCount, d_workspace, workspaceBytes, stream, stageMs);
}

}  // extern "C"
