// em2_cluster.hip -- label propagation over the cell graph on the GPU, bit-identical to the serial reference.
// CellGraph::labelPropagationClustering (src/CellGraph.cpp:443-612) with ClusterTable
// (src/CellGraph.hpp:50-121), the first step of ExpressionMatrix::createClusterGraph (src/ExpressionMatrix.cpp:2145).
//
// Reference: per iteration the vertices are visited in one std::shuffle order; a vertex whose label differs from
// the best cluster of its table takes that cluster and PUSHES (+similarity on the new label, -similarity on the old)
// into the table of every neighbour.  A table is an insertion-ordered list of (cluster, float weight) with a
// (bestCluster, bestWeight) pair maintained incrementally, so its state depends on the ORDER of the pushes it
// receives -- but on nothing else: tables of different vertices never interact.
//
// Here the pushes become pulls.  A label change of u is an event at time (iteration, position of u in that
// iteration's order).  When vertex v gets its turn it applies, in time order, the events of its neighbours since its
// previous turn: those of the previous iteration that came after v (phase A: labels of two iterations ago and of
// last iteration differ, and posPrev[u] > posPrev[v]) and those of this iteration that came before v (phase B:
// posCur[u] < posCur[v]).  The sequence of addWeight calls every table sees is exactly the reference's, so are the
// float sums, the tie decisions and the labels.  What is gained is parallelism: v only has to wait for the
// neighbours that precede it in this iteration's order.
//
// The kernel, labelPropagationCachedKernel: the table of the vertex whose turn it is lives in registers (up to 256 entries) or
// LDS (up to 640) for the turn, a neighbour is ONE 32-byte record, the events of a turn are ranked and applied in batches.  It
// reproduces the reference's addWeight sequences exactly.  (Rounds 1-2 searched the tables in global memory and gathered five
// arrays per neighbour: labelPropagationKernel in the git history, 102 ms where this one takes 34.)
//
// Schedule: one launch per iteration; a wave handles one vertex at a time and only ever waits (phase B) for vertices at
// smaller positions, which running waves hold or will draw before anything larger, so the waits cannot deadlock.  How the
// positions reach the waves is the SCHEDULE parameter of the cached kernel (a ticket per compute unit by default; see
// there).  The only data shared inside a launch is the turn word of a vertex, (iteration + 1) << 32 | label, one 64-bit
// agent-scope atomic word, so no fences are needed; everything else a wave touches is private to its vertex until the next
// launch.  The host draws the orders (std::shuffle) in a thread of its own and uploads them on a second stream.
//
// Layout: adjacency CSR in add_edge order (what out_edges() of adjacency_list<listS,listS,undirectedS> walks);
// tables in one arena of (cluster, weight), 2*degree+8 entries per vertex to start with, relocated to the bump-
// allocated tail when full (entries are never removed, as in the reference); two arrays of per-vertex records alternating
// between iterations; for vertices of degree > 64 the candidate events are sorted as (time, adjacency index) keys in LDS
// (beyond 512 of them: in global memory).

#include "em2_device.h"
#include "em2_select_wave.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>

#include <rocprim/rocprim.hpp>

namespace em2 {
namespace {

constexpr uint32_t kNone = 0xffffffffu;

struct TableEntry {
    uint32_t cluster;
    float weight;
};

struct TableMeta {
    uint64_t begin;
    uint32_t size;
    uint32_t capacity;
    uint32_t best;          // ClusterTable::bestClusterId, kNone = the reference's numeric_limits<uint32_t>::max()
    float bestWeight;
};

// Phase A: a = old label, b = new label.  Phase B: a = the neighbour, b = its label before this iteration, and when
// the neighbour had already had its turn at the first look, known = 1 and c = its new label.
struct Candidate {
    uint32_t key;
    uint32_t a;
    uint32_t b;
    float weight;
    uint32_t c;
    uint32_t known;
};

// The LDS form of the kernel reads what it needs of a neighbour with ONE 16-byte gather (+ its turn word from the same 64 bytes)
// instead of five 4-byte gathers from five arrays: a turn without events is bound by exactly that traffic.  Two arrays of records
// alternate: the turns of iteration t read records[t & 1] and write the label fields and their own position into
// records[(t + 1) & 1], whose posCur the positions kernel of iteration t + 1 fills (other bytes of the same record).  A stale
// turn word is that of iteration t - 2 and never equals (t + 1).
struct alignas(32) VertexRecord {
    uint32_t labelPrev;     // the label after the previous iteration
    uint32_t labelPrev2;    // ... and after the one before
    uint32_t posPrev;       // position in the previous iteration's order
    uint32_t posCur;        // position in this iteration's order
    uint64_t turn;          // (iteration + 1) << 32 | label once the vertex has had its turn in this iteration
    uint64_t unused;
};

struct ClusterArgs {
    uint32_t vertexCount;
    uint32_t iteration;
    const uint64_t* offsets;
    const uint32_t* neighbour;
    const float* weight;
    const uint32_t* order;
    const uint32_t* posCur;
    const uint32_t* posPrev;
    const uint32_t* labelPrev;
    const uint32_t* labelPrev2;
    uint32_t* labelCur;
    uint64_t* state;
    const VertexRecord* records;    // LDS form only: this iteration's records ...
    VertexRecord* recordsNext;      // ... and the next one's
    TableMeta* meta;
    TableEntry* arena;
    unsigned long long* arenaTop;
    uint64_t arenaCapacity;
    uint32_t* control;          // [0] ticket, [1] label changes, [2] error (1 wait timed out, 2 arena exhausted)
    uint32_t ticketBatch;       // 0: wave w of W takes positions w, w+W, ..; else positions drawn from the ticket, this many at a time
    uint32_t poolAreas;         // LDS form: areas in the block's pool (one per wave, or fewer: see the unit schedule)
    Candidate* scratchA;
    Candidate* scratchB;
#ifdef EM2_DIAG
    unsigned long long* diag;   // kDiagWords sums over the waves of one launch (EM2_TIMING=1 prints them)
#endif
    uint64_t* sortKeys;         // 4 x adjacency slots: (event time << 32 | list index) of either list, padded to a power of two
    uint64_t slots;
};

// Diagnostic build: where a wave's cycles go.  Slots: 0 draw + header loads, 1 gather (degree <= 64), 2 phase A events,
// 3 phase B events, 4 waits for a neighbour's turn, 5 hub gather, 6 hub sort, 7 hub events, 8 the turn's stores, 9 whole wave;
// counts: 10 events A, 11 events B, 12 waits, 13 hub turns, 14 findBest rescans, 15 table relocations.
constexpr uint32_t kDiagWords = 32;       // (16: ordering a turn's events, 17: applying them)
#ifdef EM2_DIAG
#define LP_CLOCK(slot)                                          \
    do {                                                        \
        const uint64_t lpNow = __builtin_amdgcn_s_memtime();    \
        diagAcc[slot] += lpNow - diagLast;                      \
        diagLast = lpNow;                                       \
    } while (0)
#define LP_COUNT(slot) (++diagAcc[slot])
#define LP_DIAG_PARAM , uint64_t* diagAcc, uint64_t& diagLast
#define LP_DIAG_PASS , diagAcc, diagLast
#else
#define LP_CLOCK(slot) ((void)0)
#define LP_COUNT(slot) ((void)0)
#define LP_DIAG_PARAM
#define LP_DIAG_PASS
#endif

__device__ __forceinline__ uint32_t uniform(uint32_t x) { return uint32_t(__builtin_amdgcn_readfirstlane(int(x))); }
__device__ __forceinline__ float uniform(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }
__device__ __forceinline__ uint64_t uniform(uint64_t x) { return uint64_t(uniform(uint32_t(x))) | (uint64_t(uniform(uint32_t(x >> 32))) << 32); }

__device__ __forceinline__ uint32_t waveMin(uint32_t x)
{
    for (int offset = 32; offset; offset >>= 1) x = min(x, uint32_t(__shfl_xor(int(x), offset)));
    return x;
}

// ClusterTable::findBestCluster (CellGraph.hpp:104-114): the first entry holding the largest weight, if above -1.
__device__ __forceinline__ void findBest(TableMeta& t, const TableEntry* arena, uint32_t lane)
{
    float bestWeight = -1.f;
    uint32_t bestIndex = kNone;
    for (uint32_t base = 0; base < t.size; base += 64u) {
        const uint32_t i = base + lane;
        if (i < t.size) {
            const float w = arena[t.begin + i].weight;
            if (w > bestWeight) {
                bestWeight = w;
                bestIndex = i;
            }
        }
    }
    for (int offset = 32; offset; offset >>= 1) {
        const float otherWeight = __shfl_xor(bestWeight, offset);
        const uint32_t otherIndex = uint32_t(__shfl_xor(int(bestIndex), offset));
        if (otherWeight > bestWeight || (otherWeight == bestWeight && otherIndex < bestIndex)) {
            bestWeight = otherWeight;
            bestIndex = otherIndex;
        }
    }
    bestIndex = uniform(bestIndex);
    if (bestIndex == kNone) {
        t.best = kNone;
        t.bestWeight = -1.f;
    } else {
        t.best = uniform(arena[t.begin + bestIndex].cluster);
        t.bestWeight = uniform(bestWeight);
    }
}

// ClusterTable::addWeight (CellGraph.hpp:70-99).  Returns false when the arena is exhausted.
__device__ __forceinline__ bool addWeight(TableMeta& t, const ClusterArgs& args, uint32_t cluster, float weight, uint32_t lane LP_DIAG_PARAM)
{
    TableEntry* arena = args.arena;
    for (uint32_t base = 0; base < t.size; base += 64u) {
        const uint32_t i = base + lane;
        bool hit = false;
        float w = 0.f;
        if (i < t.size) {
            const TableEntry e = arena[t.begin + i];
            hit = e.cluster == cluster;
            w = e.weight;
        }
        const uint64_t mask = __builtin_amdgcn_ballot_w64(hit);
        if (mask == 0ull) continue;
        // The first entry of that cluster, like the reference's linear search (parallel edges can leave an initial
        // table with the same cluster twice).
        const int owner = __ffsll((unsigned long long)mask) - 1;
        if (int(lane) == owner) {
            w += weight;
            arena[t.begin + i].weight = w;
        }
        const float updated = __shfl(w, owner);
        if (cluster == t.best) {
            if (weight < 0.f) {
                LP_COUNT(14);
                findBest(t, arena, lane);
            }
            else t.bestWeight = updated;
        } else if (updated > t.bestWeight) {
            t.best = cluster;
            t.bestWeight = updated;
        }
        return true;
    }
    if (t.size == t.capacity) {
        const uint32_t capacity = t.capacity * 2u + 8u;
        unsigned long long at = 0;
        if (lane == 0u) at = atomicAdd(args.arenaTop, (unsigned long long)capacity);
        at = uniform(uint64_t(at));
        if (at + capacity > args.arenaCapacity) return false;
        LP_COUNT(15);
        for (uint32_t i = lane; i < t.size; i += 64u) arena[at + i] = arena[t.begin + i];
        t.begin = at;
        t.capacity = capacity;
    }
    if (lane == 0u) arena[t.begin + t.size] = TableEntry{cluster, weight};
    ++t.size;
    if (weight > t.bestWeight) {
        t.best = cluster;
        t.bestWeight = weight;
    }
    return true;
}

// One label change of a neighbour as the table of the current vertex sees it (CellGraph.cpp:529-530).
__device__ __forceinline__ bool applyEvent(TableMeta& t, const ClusterArgs& args, uint32_t oldLabel, uint32_t newLabel,
                                           float weight, uint32_t lane LP_DIAG_PARAM)
{
    return addWeight(t, args, newLabel, weight, lane LP_DIAG_PASS) && addWeight(t, args, oldLabel, -weight, lane LP_DIAG_PASS);
}

// Waits until the vertex has had its turn in this iteration; returns its label, or sets failed.
__device__ __forceinline__ uint32_t labelAfterTurnAt(const ClusterArgs& args, const uint64_t* word, bool& failed LP_DIAG_PARAM)
{
    const uint32_t want = args.iteration + 1u;
    // (every value that steers the loop goes through readfirstlane: the compiler must see the polling as uniform control flow,
    // or everything live across it -- the table's size, its mode, the error -- ends up in vector registers behind exec masks)
    uint64_t s = uniform(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (uint32_t(s >> 32) != want) {
#ifdef EM2_DIAG
        const uint64_t lpEntered = __builtin_amdgcn_s_memtime();
        LP_COUNT(12);
#endif
        // A waiting wave must cost the working waves of its CU next to nothing: every instruction of the poll, and above all
        // its branches, competes with theirs (measured: with a poll every ~130 cycles + load, 70 % of a launch's branch
        // instructions were polls, and the working waves ran at a third of their speed).  So the polls are spaced by ~2000
        // cycles -- a wait lasts 10^5 .. 10^6 -- and the error word and the clock are looked at every 64th poll only.
        const uint64_t start = __builtin_amdgcn_s_memrealtime();             // 100 MHz
        for (uint32_t polls = 1;; ++polls) {
            __builtin_amdgcn_s_sleep(31);
            s = uniform(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (uint32_t(s >> 32) == want) break;
            if ((polls & 63u) == 0u &&
                (uniform(__hip_atomic_load(args.control + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0u ||
                 __builtin_amdgcn_s_memrealtime() - start > 400000000ull)) {
                failed = true;
                break;
            }
        }
#ifdef EM2_DIAG
        const uint64_t lpLeft = __builtin_amdgcn_s_memtime();      // the wait leaves the surrounding phase's account
        diagAcc[4] += lpLeft - lpEntered;
        diagLast += lpLeft - lpEntered;
#endif
    }
    return uniform(uint32_t(s));
}

__device__ __forceinline__ uint32_t labelAfterTurn(const ClusterArgs& args, uint32_t vertex, bool& failed LP_DIAG_PARAM)
{
    return labelAfterTurnAt(args, args.state + vertex, failed LP_DIAG_PASS);
}

// Bitonic sort of n (a power of two) 64-bit keys in global memory by one wave: the candidate events of a vertex of
// large degree, by event time.  (The repeated minimum search this replaces was quadratic: a hub of degree 3000 took
// half a second per turn.)  A phase's pairs are disjoint, and a wave's loads follow its stores in order.
__device__ void sortKeysByWave(uint64_t* keys, uint32_t n, uint32_t lane)
{
    for (uint32_t k = 2u; k <= n; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0u; j >>= 1) {
            for (uint32_t i = lane; i < n; i += 64u) {
                const uint32_t partner = i ^ j;
                if (partner > i) {
                    const uint64_t x = keys[i], y = keys[partner];
                    const bool ascending = (i & k) == 0u;
                    if ((x > y) == ascending) {
                        keys[i] = y;
                        keys[partner] = x;
                    }
                }
            }
        }
    }
}

// The product's form of the turn (round 3).
//
// Measured with the diagnostic build (LP_CLOCK) at 1M vertices / 15M edges, on the older kernel: in the iterations with many
// label changes a wave spent 36 % of its cycles applying events -- two addWeight calls each, every one a dependent global
// load of the table, a ballot and a store: 3000 cycles -- 8 % in the 4 % of turns whose vertex has more than 64 neighbours
// (tables of hundreds of entries scanned from global memory 64 at a time, candidates staged and sorted in global memory), and
// 40-65 % WAITING.  Here a table is read once per turn, on the first event, into registers or (two entries per lane, 16
// bytes) into LDS, searched and updated there, and written back once; a table that outgrows its allocation only changes its
// address (no copy: the content is on chip); the candidate events of a vertex of large degree are sorted as (time, adjacency
// index) keys in LDS and their data gathered again 64 at a time.  Tables above kCacheEntries and candidate lists above
// kHubKeys take the global-memory forms above.  The waits were the schedule's (see SCHEDULE below).
constexpr uint32_t kCacheEntries = 640;
constexpr uint32_t kHubKeys = 512;

struct alignas(16) WaveArea {
    uint2 table[kCacheEntries];         // (cluster, weight bits)
    uint64_t keys[kHubKeys];
};

__device__ __forceinline__ uint32_t laneValue(uint32_t x, int owner) { return uint32_t(__builtin_amdgcn_readlane(int(x), owner)); }
__device__ __forceinline__ float laneValue(float x, int owner) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), owner)); }

// Most tables have a few dozen entries: up to kRegisterSlots * 64 of them are held in REGISTERS for the turn (entry i in lane
// i & 63, slot i >> 6), where an addWeight is a compare, a ballot and a predicated add -- no LDS round trip; a table that is or
// grows larger goes through the LDS area, and beyond that through global memory.  (The slots are named members and the code
// per slot is spelled out by EM2_TABLE_SLOTS: an array member sent the whole structure to scratch memory.)
constexpr uint32_t kRegisterSlots = 4;
#define EM2_TABLE_SLOTS(F) F(0, cluster0, weight0) F(1, cluster1, weight1) F(2, cluster2, weight2) F(3, cluster3, weight3)

struct CachedTable {
    TableMeta t;
    uint32_t mode;          // 0 untouched in global memory, 3 in registers, 1 in LDS, 2 worked on in global memory
    bool dirty;
    uint32_t cluster0, cluster1, cluster2, cluster3;        // mode 3: kNone where the table has no entry
    float weight0, weight1, weight2, weight3;
};

// findBestCluster over the registers: the first entry holding the largest weight, if above -1.
__device__ __forceinline__ void registerFindBest(CachedTable& c, uint32_t lane)
{
    float bestWeight = -1.f;
    uint32_t bestIndex = kNone;
#define EM2_SLOT(SLOT, CLUSTER, WEIGHT)                                   \
    if (SLOT * 64u + lane < c.t.size && c.WEIGHT > bestWeight) {          \
        bestWeight = c.WEIGHT;                                            \
        bestIndex = SLOT * 64u + lane;                                    \
    }
    EM2_TABLE_SLOTS(EM2_SLOT)
#undef EM2_SLOT
    for (int offset = 32; offset; offset >>= 1) {
        const float otherWeight = __shfl_xor(bestWeight, offset);
        const uint32_t otherIndex = uint32_t(__shfl_xor(int(bestIndex), offset));
        if (otherWeight > bestWeight || (otherWeight == bestWeight && otherIndex < bestIndex)) {
            bestWeight = otherWeight;
            bestIndex = otherIndex;
        }
    }
    bestIndex = uniform(bestIndex);
    if (bestIndex == kNone) {
        c.t.best = kNone;
        c.t.bestWeight = -1.f;
        return;
    }
    uint32_t best = 0;
#define EM2_SLOT(SLOT, CLUSTER, WEIGHT) \
    if ((bestIndex >> 6) == SLOT) best = laneValue(c.CLUSTER, int(bestIndex & 63u));
    EM2_TABLE_SLOTS(EM2_SLOT)
#undef EM2_SLOT
    c.t.best = best;
    c.t.bestWeight = uniform(bestWeight);
}

__device__ __forceinline__ void registerFlush(const CachedTable& c, const ClusterArgs& args, uint32_t lane)
{
    uint2* out = reinterpret_cast<uint2*>(args.arena + c.t.begin);
#define EM2_SLOT(SLOT, CLUSTER, WEIGHT) \
    if (SLOT * 64u + lane < c.t.size) out[SLOT * 64u + lane] = make_uint2(c.CLUSTER, __float_as_uint(c.WEIGHT));
    EM2_TABLE_SLOTS(EM2_SLOT)
#undef EM2_SLOT
}


__device__ __forceinline__ void cachedFindBest(CachedTable& c, const uint2* table, uint32_t lane)
{
    float bestWeight = -1.f;
    uint32_t bestIndex = kNone;
    for (uint32_t i = lane; i < c.t.size; i += 64u) {
        const float w = __uint_as_float(table[i].y);
        if (w > bestWeight) {
            bestWeight = w;
            bestIndex = i;
        }
    }
    for (int offset = 32; offset; offset >>= 1) {
        const float otherWeight = __shfl_xor(bestWeight, offset);
        const uint32_t otherIndex = uint32_t(__shfl_xor(int(bestIndex), offset));
        if (otherWeight > bestWeight || (otherWeight == bestWeight && otherIndex < bestIndex)) {
            bestWeight = otherWeight;
            bestIndex = otherIndex;
        }
    }
    bestIndex = uniform(bestIndex);
    if (bestIndex == kNone) {
        c.t.best = kNone;
        c.t.bestWeight = -1.f;
    } else {
        c.t.best = uniform(table[bestIndex].x);
        c.t.bestWeight = uniform(bestWeight);
    }
}

// Writes the LDS copy back (the whole table: its entries are contiguous, a turn with events touches most of them).
__device__ __forceinline__ void cachedFlush(const CachedTable& c, const ClusterArgs& args, const uint2* table, uint32_t lane)
{
    uint2* out = reinterpret_cast<uint2*>(args.arena + c.t.begin);
    for (uint32_t i = lane; i < c.t.size; i += 64u) out[i] = table[i];
}

// The table of the current turn leaves global memory: into registers, the LDS area, or (too large) stays where it is.
__device__ __forceinline__ void openTable(CachedTable& c, const ClusterArgs& args, uint2* table, uint32_t lane)
{
    if (c.mode != 0u) return;
    if (c.t.size <= kRegisterSlots * 64u) {
        const uint2* in = reinterpret_cast<const uint2*>(args.arena + c.t.begin);
#define EM2_SLOT(SLOT, CLUSTER, WEIGHT)                                                   \
    {                                                                                     \
        uint2 e = make_uint2(kNone, 0u);                                                  \
        if (SLOT * 64u < c.t.size && SLOT * 64u + lane < c.t.size) e = in[SLOT * 64u + lane]; \
        c.CLUSTER = e.x;                                                                  \
        c.WEIGHT = __uint_as_float(e.y);                                                  \
    }
        EM2_TABLE_SLOTS(EM2_SLOT)
#undef EM2_SLOT
        c.mode = 3u;
    } else if (c.t.size <= kCacheEntries && table != nullptr) {
        // begin and capacity are even (2 * degree + 8, 2 * capacity + 8), the arena 16-byte aligned: two entries per lane
        const uint4* in = reinterpret_cast<const uint4*>(args.arena + c.t.begin);
        uint4* out = reinterpret_cast<uint4*>(table);
        for (uint32_t j = lane; 2u * j < c.t.size; j += 64u) out[j] = in[j];
        c.mode = 1u;
        waveSync();
    } else {
        c.mode = 2u;
    }
}

// ClusterTable::addWeight (CellGraph.hpp:70-99) on the table of the current turn.  Returns false when the arena is exhausted.
__device__ __forceinline__ bool cachedAddWeight(CachedTable& c, const ClusterArgs& args, uint2* table, uint32_t cluster, float weight,
                                                uint32_t lane LP_DIAG_PARAM)
{
    // (wave-uniform by construction; said again so that the branches below are scalar ones)
    c.mode = uniform(c.mode);
    c.t.size = uniform(c.t.size);
    c.t.capacity = uniform(c.t.capacity);
    c.t.begin = uniform(c.t.begin);
    c.t.best = uniform(c.t.best);
    c.t.bestWeight = uniform(c.t.bestWeight);
    cluster = uniform(cluster);
    weight = uniform(weight);
    openTable(c, args, table, lane);
    if (c.mode == 3u) {
        c.dirty = true;
        // the first entry of that cluster, like the reference's linear search
#define EM2_SLOT(SLOT, CLUSTER, WEIGHT)                                                   \
    if (SLOT * 64u < c.t.size) {                                                          \
        const uint64_t mask = __builtin_amdgcn_ballot_w64(c.CLUSTER == cluster);          \
        if (mask != 0ull) {                                                               \
            const int owner = __ffsll((unsigned long long)mask) - 1;                      \
            if (int(lane) == owner) c.WEIGHT += weight;                                   \
            const float updated = laneValue(c.WEIGHT, owner);                             \
            if (cluster == c.t.best) {                                                    \
                if (weight < 0.f) {                                                       \
                    LP_COUNT(14);                                                         \
                    registerFindBest(c, lane);                                            \
                } else {                                                                  \
                    c.t.bestWeight = updated;                                             \
                }                                                                         \
            } else if (updated > c.t.bestWeight) {                                        \
                c.t.best = cluster;                                                       \
                c.t.bestWeight = updated;                                                 \
            }                                                                             \
            return true;                                                                  \
        }                                                                                 \
    }
        EM2_TABLE_SLOTS(EM2_SLOT)
#undef EM2_SLOT
        if (c.t.size < kRegisterSlots * 64u) {
            if (c.t.size == c.t.capacity) {
                // the content is in registers: a table that outgrows its allocation only changes the address it is written back to
                const uint32_t capacity = c.t.capacity * 2u + 8u;
                unsigned long long at = 0;
                if (lane == 0u) at = atomicAdd(args.arenaTop, (unsigned long long)capacity);
                at = uniform(uint64_t(at));
                if (at + capacity > args.arenaCapacity) return false;
                LP_COUNT(15);
                c.t.begin = at;
                c.t.capacity = capacity;
            }
#define EM2_SLOT(SLOT, CLUSTER, WEIGHT)                                \
    if ((c.t.size >> 6) == SLOT && lane == (c.t.size & 63u)) {        \
        c.CLUSTER = cluster;                                           \
        c.WEIGHT = weight;                                             \
    }
            EM2_TABLE_SLOTS(EM2_SLOT)
#undef EM2_SLOT
            ++c.t.size;
            if (weight > c.t.bestWeight) {
                c.t.best = cluster;
                c.t.bestWeight = weight;
            }
            return true;
        }
        // the registers are full: the table moves to the LDS area for the rest of the turn -- or, for a turn that holds no
        // area, back to global memory
        if (table != nullptr) {
#define EM2_SLOT(SLOT, CLUSTER, WEIGHT) table[SLOT * 64u + lane] = make_uint2(c.CLUSTER, __float_as_uint(c.WEIGHT));
            EM2_TABLE_SLOTS(EM2_SLOT)
#undef EM2_SLOT
            c.mode = 1u;
            waveSync();
        } else {
            registerFlush(c, args, lane);
            waveSyncGlobal();
            c.mode = 2u;
        }
    }
    if (c.mode == 1u) {
        c.dirty = true;
        const uint32_t pairs = (c.t.size + 1u) >> 1;
        for (uint32_t first = 0; first < pairs; first += 64u) {
            const uint32_t j = first + lane;
            bool hit0 = false, hit1 = false;
            uint2 e0 = make_uint2(0u, 0u), e1 = make_uint2(0u, 0u);
            if (j < pairs) {
                e0 = table[2u * j];
                e1 = table[2u * j + 1u];
                hit0 = e0.x == cluster;
                hit1 = 2u * j + 1u < c.t.size && e1.x == cluster;
            }
            const uint64_t mask = __builtin_amdgcn_ballot_w64(hit0 || hit1);
            if (mask == 0ull) continue;
            // the first entry of that cluster, like the reference's linear search
            const int owner = __ffsll((unsigned long long)mask) - 1;
            float w = __uint_as_float(hit0 ? e0.y : e1.y);
            if (int(lane) == owner) {
                w += weight;
                table[2u * j + (hit0 ? 0u : 1u)].y = __float_as_uint(w);
            }
            const float updated = laneValue(w, owner);
            if (cluster == c.t.best) {
                if (weight < 0.f) {
                    LP_COUNT(14);
                    waveSync();
                    cachedFindBest(c, table, lane);
                } else {
                    c.t.bestWeight = updated;
                }
            } else if (updated > c.t.bestWeight) {
                c.t.best = cluster;
                c.t.bestWeight = updated;
            }
            return true;
        }
        if (c.t.size < kCacheEntries) {
            if (c.t.size == c.t.capacity) {
                // the content is in LDS: a table that outgrows its allocation only changes the address it is written back to
                const uint32_t capacity = c.t.capacity * 2u + 8u;
                unsigned long long at = 0;
                if (lane == 0u) at = atomicAdd(args.arenaTop, (unsigned long long)capacity);
                at = uniform(uint64_t(at));
                if (at + capacity > args.arenaCapacity) return false;
                LP_COUNT(15);
                c.t.begin = at;
                c.t.capacity = capacity;
            }
            if (lane == 0u) table[c.t.size] = make_uint2(cluster, __float_as_uint(weight));
            ++c.t.size;
            waveSync();
            if (weight > c.t.bestWeight) {
                c.t.best = cluster;
                c.t.bestWeight = weight;
            }
            return true;
        }
        // no room in LDS for another entry: back to global memory for the rest of the turn
        waveSync();
        cachedFlush(c, args, table, lane);
        waveSyncGlobal();
        c.mode = 2u;
    }
    return addWeight(c.t, args, cluster, weight, lane LP_DIAG_PASS);
}

// n label changes of neighbours (CellGraph.cpp:529-530: +weight on the new label, then -weight on the old one), held by the
// lanes 0 .. n-1 in the order of their times, applied to the table of the current turn.  0, or the error (2: arena exhausted).
//
// The loop is what a turn with events spends its time in, and it is a chain of dependent scalar and vector instructions
// executed by one wave: what counts is the NUMBER of instructions and branches per addWeight (measured: 2200 cycles per event
// in the form that picked every event with a wave-wide minimum and went through the general addWeight).  So for a table in
// registers the common case -- the cluster is there, in one of the first 64 entries -- is spelled out here: a compare, the
// first set bit, a predicated add, one readlane, and the update of (best, bestWeight).
__device__ __forceinline__ uint32_t applyBatch(CachedTable& c, const ClusterArgs& args, uint2* table, uint32_t n, uint32_t oldLabels,
                                               uint32_t newLabels, float weights, uint32_t lane LP_DIAG_PARAM)
{
    openTable(c, args, table, lane);
#ifdef EM2_DIAG
    if (c.mode == 3u) {
        // (what the table's load costs: wait for it here instead of at the first compare)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint64_t lpNow = __builtin_amdgcn_s_memtime();
        diagAcc[18] += lpNow - diagLast;
    }
#endif
    for (uint32_t j = 0; j < n; ++j) {
        const uint32_t oldLabel = laneValue(oldLabels, int(j)), newLabel = laneValue(newLabels, int(j));
        const float weight = laneValue(weights, int(j));
        if (newLabel == oldLabel) continue;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const uint32_t cluster = half ? oldLabel : newLabel;
            const float delta = half ? -weight : weight;
            if (c.mode == 3u) {
#ifdef EM2_DIAG
                const uint64_t lpOp = __builtin_amdgcn_s_memtime();
#endif
                const uint64_t mask = __builtin_amdgcn_ballot_w64(c.cluster0 == cluster);
                if (mask != 0ull) {
                    const int owner = __ffsll((unsigned long long)mask) - 1;
                    const float sum = c.weight0 + delta;
                    c.weight0 = int(lane) == owner ? sum : c.weight0;
                    const float updated = laneValue(sum, owner);
                    c.dirty = true;
                    const bool isBest = cluster == c.t.best;
                    if (isBest && delta < 0.f) {
                        LP_COUNT(14);
                        registerFindBest(c, lane);
                    } else if (isBest || updated > c.t.bestWeight) {
                        c.t.best = cluster;
                        c.t.bestWeight = updated;
                    }
#ifdef EM2_DIAG
                    diagAcc[20] += __builtin_amdgcn_s_memtime() - lpOp;
                    ++diagAcc[21];
#endif
                    continue;
                }
            }
            LP_COUNT(19);
#ifdef EM2_DIAG
            const uint64_t lpSlow = __builtin_amdgcn_s_memtime();
            const uint32_t lpMode = c.mode;
#endif
            if (!cachedAddWeight(c, args, table, cluster, delta, lane LP_DIAG_PASS)) return 2u;
#ifdef EM2_DIAG
            {
                const uint64_t spent = __builtin_amdgcn_s_memtime() - lpSlow;
                if (lpMode == 3u) { diagAcc[24] += spent; ++diagAcc[25]; }
                else if (lpMode == 1u) { diagAcc[26] += spent; ++diagAcc[27]; }
                else { diagAcc[28] += spent; ++diagAcc[29]; }
            }
#endif
        }
    }
    return 0u;
}

// Bitonic sort of n (a power of two) keys by one wave; KEYS is a pointer into LDS or into global memory (two instantiations).
template <bool GLOBAL>
__device__ __forceinline__ void sortKeysOfTurn(uint64_t* keys, uint32_t n, uint32_t lane)
{
    for (uint32_t k = 2u; k <= n; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0u; j >>= 1) {
            for (uint32_t i = lane; i < n; i += 64u) {
                const uint32_t partner = i ^ j;
                if (partner > i) {
                    const uint64_t x = keys[i], y = keys[partner];
                    const bool ascending = (i & k) == 0u;
                    if ((x > y) == ascending) {
                        keys[i] = y;
                        keys[partner] = x;
                    }
                }
            }
            waveSyncFor<GLOBAL>();
        }
    }
}

// The candidates' keys of one phase of a vertex with more than 64 neighbours, padded and sorted.
template <bool GLOBAL>
__device__ __forceinline__ void sortHubKeys(uint64_t* keys, uint32_t count, uint32_t lane)
{
    uint32_t padded = 1u;
    while (padded < count) padded <<= 1;
    for (uint32_t i = count + lane; i < padded; i += 64u) keys[i] = ~0ull;
    waveSyncFor<GLOBAL>();
    sortKeysOfTurn<GLOBAL>(keys, padded, lane);
}

// SCHEDULE: how the positions of an iteration's order reach the waves.
//   kScheduleTicket   positions drawn from one global ticket, ticketBatch at a time (no residency requirement; 45 ns per draw)
//   kScheduleStrided  wave w of W takes positions w, w + W, ... and, knowing them in advance, keeps the loads of three turns in
//                     flight.  Free, and the fastest form of an iteration WITHOUT label changes -- but every position a wave
//                     holds ahead of time is a hostage of its current turn: while that turn waits for a neighbour (or is a
//                     hub's), nobody can take those positions over, their dependents wait in turn, and the waits feed each
//                     other: 40-60 % of all wave cycles in the iterations with many changes, whatever the turns themselves
//                     cost (three rewrites of the event arithmetic changed nothing).
//   kScheduleUnit     one block of 16 waves per compute unit; unit u of U owns positions u, u + U, ... and a wave draws ONE
//                     position from the unit's ticket in LDS when it is free to work on it, loading nothing ahead: a turn that
//                     waits holds up nothing but itself (clustering of 1M cells: 53.6 -> 36.6 ms on the same box).
constexpr int kScheduleTicket = 0, kScheduleStrided = 1, kScheduleUnit = 2;

template <int SCHEDULE>
__global__ void __launch_bounds__(SCHEDULE == kScheduleUnit ? 1024 : 256, SCHEDULE == kScheduleUnit ? 8 : 1)  // (HIP: maximum threads per block, minimum WAVES per SIMD)
labelPropagationCachedKernel(ClusterArgs args)
{
    constexpr bool STRIDED = SCHEDULE == kScheduleStrided;          // (the loads of three turns in flight)
    // LDS: the pool of areas (tables beyond the registers, keys of large neighbourhoods: 4 % of the turns need one), then the
    // unit's ticket and the mask of free areas.  With one area per wave (256-thread blocks) a turn always gets one; a unit of
    // 16 waves shares fewer, so that two units fit a compute unit: a turn that finds none takes the global-memory forms.
    extern __shared__ __attribute__((aligned(16))) unsigned char labelLds[];
    WaveArea* areas = reinterpret_cast<WaveArea*>(labelLds);
    uint32_t* unitTicket = reinterpret_cast<uint32_t*>(labelLds + args.poolAreas * sizeof(WaveArea));
    uint32_t* freeAreas = unitTicket + 1;
    if (threadIdx.x == 0u) {
        *unitTicket = 0u;
        *freeAreas = args.poolAreas >= 32u ? 0xffffffffu : (1u << args.poolAreas) - 1u;
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t want = args.iteration + 1u;
    const uint32_t waves = gridDim.x * (blockDim.x / 64u);
    uint32_t next = blockIdx.x * (blockDim.x / 64u) + threadIdx.x / 64u, end = 0, changes = 0;
#ifdef EM2_DIAG
    uint64_t diagAcc[kDiagWords] = {};
    const uint64_t diagStart = __builtin_amdgcn_s_memtime();
    uint64_t diagLast = diagStart;
#endif
    // What a turn reads before it can look at its neighbours -- its vertex, the vertex's header (adjacency range, table, label,
    // previous position), the neighbour list -- is a chain of three dependent loads that depends on nothing the launch writes
    // (a vertex's table header is only written by its own turn).  With the strided schedule (STRIDED) a wave knows its positions
    // in advance, so the chain is loaded ahead: at the start of turn k the neighbours of turn k+1, the header of turn k+2 and
    // the vertex of turn k+3 are requested -- unconditionally, with clamped indices and per-lane addresses, so that no branch
    // and no readfirstlane, hence no wait, lies between the requests -- and the stages move up by one at the END of the turn,
    // before its stores are issued (the moves wait for loads that arrived long ago; a wait at the start of the next turn
    // would also wait for those stores).  With the ticket schedule the chain is loaded at the start of the turn.
    struct Header {         // wave-uniform
        uint64_t base;
        uint32_t degree;
        TableMeta t;
        uint32_t label, posPrev;
    };
    struct RawHeader {      // as loaded: the same value in every lane
        uint64_t base, end;
        TableMeta t;
        uint32_t label, posPrev;
    };
    const bool later = args.iteration > 0u;
    const uint32_t lastPosition = args.vertexCount - 1u;
    const uint64_t lastSlot = args.slots ? args.slots - 1u : 0u;
    auto loadVertex = [&](uint32_t position) -> uint32_t { return args.order[min(position, lastPosition)]; };
    auto loadHeader = [&](uint32_t vertex) -> RawHeader {
        RawHeader h;
        h.base = args.offsets[vertex];
        h.end = args.offsets[vertex + 1];
        h.t = args.meta[vertex];
        const uint4 own = *reinterpret_cast<const uint4*>(args.records + vertex);
        h.label = own.x;
        h.posPrev = own.z;                              // (of iteration 0: not used)
        return h;
    };
    auto pickUp = [&](const RawHeader& raw) -> Header {
        Header h;
        h.base = uniform(raw.base);
        h.degree = uniform(uint32_t(raw.end - raw.base));
        h.t.begin = uniform(raw.t.begin);
        h.t.size = uniform(raw.t.size);
        h.t.capacity = uniform(raw.t.capacity);
        h.t.best = uniform(raw.t.best);
        h.t.bestWeight = uniform(raw.t.bestWeight);
        h.label = uniform(raw.label);
        h.posPrev = uniform(raw.posPrev);
        return h;
    };
    // (lanes beyond the degree read some valid slot and never use it)
    auto loadNeighbour = [&](uint64_t base, uint32_t& u, float& w) {
        const uint64_t slot = min(base + lane, lastSlot);
        u = args.neighbour[slot];
        w = args.weight[slot];
    };
    // the next position of this wave under the strided schedule (beyond the order's end: clamped by the loads, ends the loop)
    auto drawPosition = [&]() -> uint32_t {
        const uint32_t position = next;
        next = next < args.vertexCount ? next + waves : next;
        return min(position, args.vertexCount);
    };
    uint32_t rawVertexA = 0, rawVertexB = 0, rawVertexC = 0, uA = 0, positionA = 0, positionB = 0, positionC = 0;
    float wA = 0.f;
    RawHeader rawHeaderA = {}, rawHeaderB = {};
    if (STRIDED) {
        positionA = drawPosition();
        positionB = drawPosition();
        positionC = drawPosition();
        rawVertexA = loadVertex(positionA);
        rawVertexB = loadVertex(positionB);
        rawVertexC = loadVertex(positionC);
        rawHeaderA = loadHeader(rawVertexA);
        rawHeaderB = loadHeader(rawVertexB);
        loadNeighbour(rawHeaderA.base, uA, wA);
    }
    for (;;) {
        uint32_t p;                                   // (the schedule: see labelPropagationKernel)
        uint32_t v, u = 0, rawVertexD = 0, uB = 0, positionD = 0;
        float w = 0.f, wB = 0.f;
        Header header;
        RawHeader rawHeaderC = {};
        if (STRIDED) {
            if (positionA >= args.vertexCount) break;
            p = positionA;
            positionD = drawPosition();
            v = uniform(rawVertexA);
            header = pickUp(rawHeaderA);
            u = uA;
            w = wA;
            loadNeighbour(rawHeaderB.base, uB, wB);
            rawHeaderC = loadHeader(rawVertexC);
            rawVertexD = loadVertex(positionD);
        } else if (SCHEDULE == kScheduleUnit) {
            uint32_t t = 0;
            if (lane == 0u) t = atomicAdd(unitTicket, 1u);
            t = uniform(t);
            const uint64_t position = uint64_t(blockIdx.x) + uint64_t(t) * gridDim.x;
            if (position >= args.vertexCount) break;
            p = uint32_t(position);
            v = uniform(args.order[p]);
            header = pickUp(loadHeader(v));
            loadNeighbour(header.base, u, w);
        } else {
            if (next >= end) {
                uint32_t first = kNone;
                if (lane == 0u && __hip_atomic_load(args.control + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                    first = __hip_atomic_fetch_add(args.control, args.ticketBatch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                first = uniform(first);
                if (first >= args.vertexCount) break;
                next = first;
                end = min(first + args.ticketBatch, args.vertexCount);
            }
            p = next++;
            v = uniform(args.order[p]);
            header = pickUp(loadHeader(v));
            loadNeighbour(header.base, u, w);
        }
        const uint64_t base = header.base;
        const uint32_t degree = header.degree;
        CachedTable c;
        c.t = header.t;
        c.mode = 0u;
        c.dirty = false;
        c.cluster0 = c.cluster1 = c.cluster2 = c.cluster3 = kNone;
        c.weight0 = c.weight1 = c.weight2 = c.weight3 = 0.f;
        // an area of the pool for the turns that need one: a large neighbourhood, or a table beyond the registers (a table that
        // grows beyond them during a turn without an area goes back to global memory)
        uint32_t areaIndex = kNone;
        if (degree > 64u || c.t.size > kRegisterSlots * 64u) {
            if (lane == 0u) {
                uint32_t seen = __hip_atomic_load(freeAreas, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                while (seen != 0u) {
                    const uint32_t bit = seen & (0u - seen);
                    if (__hip_atomic_compare_exchange_strong(freeAreas, &seen, seen & ~bit, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_WORKGROUP)) {
                        areaIndex = uint32_t(__ffs(int(bit)) - 1);
                        break;
                    }
                }
            }
            areaIndex = uniform(areaIndex);
        }
        WaveArea* const area = areaIndex != kNone ? areas + areaIndex : nullptr;
        uint2* const table = area ? area->table : nullptr;
        uint32_t label = header.label;
        const uint32_t posPrevV = later ? header.posPrev : 0u;
        uint32_t error = 0;
        LP_CLOCK(0);

        if (degree <= 64u) {
            // ---- one neighbour per lane; both candidate lists stay in registers ----
            uint32_t labelU = 0, beforeU = 0, keyA = kNone, keyB = kNone;
            if (lane < degree) {
                const uint4 r = *reinterpret_cast<const uint4*>(args.records + u);
                labelU = r.x;
                beforeU = r.y;
                if (later && labelU != beforeU && r.z > posPrevV) keyA = r.z;
                if (r.w < p) keyB = r.w;
            }
            // First look at the earlier neighbours, all at once: most have had their turn and kept their label.
            uint32_t afterU = 0;
            bool known = false;
            if (keyB != kNone) {
                const uint64_t s = __hip_atomic_load(&args.records[u].turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                known = uint32_t(s >> 32) == want;
                afterU = uint32_t(s);
                if (known && afterU == labelU) keyB = kNone;
            }
            LP_CLOCK(1);
            // The earlier neighbours that have not had their turn yet: wait for them now, one after the other (in any order:
            // they all hold smaller positions), so that every event of the turn is known before the first is applied.
            uint64_t unknown = __builtin_amdgcn_ballot_w64(keyB != kNone && !known);
            while (unknown != 0ull) {
                const int owner = __ffsll((unsigned long long)unknown) - 1;
                unknown &= unknown - 1ull;
                bool failed = false;
                const uint32_t newLabel = labelAfterTurnAt(args, &args.records[laneValue(u, owner)].turn, failed LP_DIAG_PASS);
                if (failed) {
                    error = 1;
                    break;
                }
                if (int(lane) == owner) afterU = newLabel;
            }
            if (afterU == labelU) keyB = kNone;
            LP_CLOCK(3);
            // phase 0: the events of the previous iteration that came after this vertex; phase 1: those of this iteration
            // before it -- each in the order of their times (equal times, i.e. parallel edges, in the order of the adjacency):
            // every candidate counts the candidates before it, and the events move to the lanes of those ranks
#pragma unroll 1
            for (int phase = 0; phase < 2 && !error; ++phase) {
                const uint32_t key = phase ? keyB : keyA;
                const uint64_t candidates = __builtin_amdgcn_ballot_w64(key != kNone);
                if (candidates == 0ull) continue;
                const uint32_t n = uint32_t(__builtin_popcountll(candidates));
                const uint64_t mine = (uint64_t(key) << 6) | lane;
                uint32_t rank = 0;
                for (uint64_t rest = candidates; rest != 0ull; rest &= rest - 1ull) {
                    const int other = __ffsll((unsigned long long)rest) - 1;
                    const uint64_t theirs = (uint64_t(laneValue(key, other)) << 6) | uint32_t(other);
                    rank += theirs < mine ? 1u : 0u;
                }
                // (ds_permute: a lane that pushes names the lane that receives, and only lanes that take part receive; so the lanes
                // without a candidate take part too and push to the lanes behind the events)
                const int to = int((key != kNone ? rank : n + lanesBelow(~candidates)) << 2);
                const uint32_t oldLabels = uint32_t(__builtin_amdgcn_ds_permute(to, int(phase ? labelU : beforeU)));
                const uint32_t newLabels = uint32_t(__builtin_amdgcn_ds_permute(to, int(phase ? afterU : labelU)));
                const float weights = __int_as_float(__builtin_amdgcn_ds_permute(to, __float_as_int(w)));
#ifdef EM2_DIAG
                diagAcc[phase ? 11 : 10] += n;
#endif
                LP_CLOCK(16);
                error = applyBatch(c, args, table, n, oldLabels, newLabels, weights, lane LP_DIAG_PASS);
                LP_CLOCK(17);
            }
        } else {
            LP_COUNT(13);
#pragma unroll 1
            for (int phase = later ? 0 : 1; phase < 2 && !error; ++phase) {
                // the candidates of this phase: (event time << 32 | adjacency index); equal times (parallel edges) keep the
                // order of the adjacency.  In LDS while they fit, else in the vertex's slots of the global key area.
                uint64_t* global = args.sortKeys + (phase ? 2u * args.slots : 0u) + 2u * base;
                uint32_t count = 0;
                bool inLds = area != nullptr;
                for (uint32_t first = 0; first < degree; first += 64u) {
                    const uint32_t i = first + lane;
                    bool candidate = false;
                    uint32_t key = 0;
                    if (i < degree) {
                        const uint32_t u = args.neighbour[base + i];
                        const uint4 r = *reinterpret_cast<const uint4*>(args.records + u);
                        if (phase == 0) {
                            key = r.z;
                            candidate = key > posPrevV && r.x != r.y;
                        } else {
                            key = r.w;
                            candidate = key < p;
                            if (candidate) {
                                const uint64_t s = __hip_atomic_load(&args.records[u].turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (uint32_t(s >> 32) == want && uint32_t(s) == r.x) candidate = false;
                            }
                        }
                    }
                    const uint64_t mask = __builtin_amdgcn_ballot_w64(candidate);
                    const uint32_t more = uint32_t(__builtin_popcountll(mask));
                    if (inLds && count + more > kHubKeys) {
                        // the list leaves LDS: what was collected so far moves to the global area
                        waveSync();
                        for (uint32_t k = lane; k < count; k += 64u) global[k] = area->keys[k];
                        inLds = false;
                    }
                    if (candidate) {
                        const uint64_t entry = (uint64_t(key) << 32) | i;
                        if (inLds) area->keys[count + lanesBelow(mask)] = entry;
                        else global[count + lanesBelow(mask)] = entry;
                    }
                    count += more;
                }
                LP_CLOCK(5);
                if (count == 0u) continue;
                if (inLds) {
                    waveSync();
                    sortHubKeys<false>(area->keys, count, lane);
                } else {
                    waveSyncGlobal();
                    sortHubKeys<true>(global, count, lane);
                }
                LP_CLOCK(6);
                // applied 64 at a time: the lane that holds a key gathers the neighbour's data again (the lines are still in L2)
                for (uint32_t first = 0; first < count && !error; first += 64u) {
                    const uint32_t n = min(64u, count - first);
                    uint32_t u = 0, before = 0, after = 0;
                    float w = 0.f;
                    bool known = true;
                    if (lane < n) {
                        uint64_t entry;
                        if (inLds) entry = area->keys[first + lane];
                        else entry = global[first + lane];
                        const uint32_t index = uint32_t(entry);
                        u = args.neighbour[base + index];
                        w = args.weight[base + index];
                        const uint4 r = *reinterpret_cast<const uint4*>(args.records + u);
                        if (phase == 0) {
                            before = r.y;
                            after = r.x;
                        } else {
                            before = r.x;
                            const uint64_t s = __hip_atomic_load(&args.records[u].turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            known = uint32_t(s >> 32) == want;
                            after = uint32_t(s);
                        }
                    }
                    // the neighbours that have not had their turn yet: wait for them (any order), then all events are known
                    uint64_t unknown = __builtin_amdgcn_ballot_w64(!known);
                    while (unknown != 0ull) {
                        const int owner = __ffsll((unsigned long long)unknown) - 1;
                        unknown &= unknown - 1ull;
                        bool failed = false;
                        const uint32_t newLabel = labelAfterTurnAt(args, &args.records[laneValue(u, owner)].turn, failed LP_DIAG_PASS);
                        if (failed) {
                            error = 1;
                            break;
                        }
                        if (int(lane) == owner) after = newLabel;
                    }
                    if (error) break;
#ifdef EM2_DIAG
                    diagAcc[phase ? 11 : 10] += n;
#endif
                    error = applyBatch(c, args, table, n, before, after, w, lane LP_DIAG_PASS);
                }
                LP_CLOCK(7);
            }
        }

        if (error) {
            if (lane == 0u) {
                // Keep the first cause: a timeout that follows an exhausted arena is only its consequence.
                uint32_t expected = 0;
                __hip_atomic_compare_exchange_strong(args.control + 2, &expected, error, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }

        // ---- the turn proper (CellGraph.cpp:507-524) ----
        const uint32_t labelBefore = label;
        const bool change = c.t.size != 0u && label != c.t.best;
        if (change) {
            label = c.t.best;
            ++changes;
        }
        if (STRIDED) {
            // the turns ahead move up by one (their loads were requested at the start of this turn)
            rawVertexA = rawVertexB;
            rawHeaderA = rawHeaderB;
            uA = uB;
            wA = wB;
            rawVertexB = rawVertexC;
            rawHeaderB = rawHeaderC;
            rawVertexC = rawVertexD;
            positionA = positionB;
            positionB = positionC;
            positionC = positionD;
        }
        if (lane == 0u) {
            // the next iteration's record of this vertex: its labels and this position (posCur is the positions kernel's)
            uint32_t* nextRecord = reinterpret_cast<uint32_t*>(args.recordsNext + v);
            nextRecord[0] = label;
            nextRecord[1] = labelBefore;
            nextRecord[2] = p;
            args.meta[v] = c.t;
            __hip_atomic_store(const_cast<uint64_t*>(&args.records[v].turn), (uint64_t(want) << 32) | label, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        if (c.mode == 3u && c.dirty) {
            registerFlush(c, args, lane);
        } else if (c.mode == 1u && c.dirty) {
            waveSync();
            cachedFlush(c, args, table, lane);
        }
        if (areaIndex != kNone) {
            waveSync();                                       // (the flush has read the area)
            if (lane == 0u) __hip_atomic_fetch_or(freeAreas, 1u << areaIndex, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        LP_CLOCK(8);
    }
#ifdef EM2_DIAG
    diagAcc[9] = __builtin_amdgcn_s_memtime() - diagStart;
    if (lane == 0u && args.diag) {
        for (uint32_t i = 0; i < kDiagWords; ++i) atomicAdd(args.diag + i, (unsigned long long)diagAcc[i]);
    }
#endif
    if (lane == 0u && changes) __hip_atomic_fetch_add(args.control + 1, changes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// CellGraph.cpp:459-476: every vertex starts in the cluster named by its own cell id; its table lists the clusters of
// its neighbours in out-edge order (addWeightQuick) and then finds its best cluster.
__global__ void __launch_bounds__(256)
initialTablesKernel(uint32_t vertexCount, const uint64_t* __restrict__ offsets, const uint32_t* __restrict__ neighbour,
                    const float* __restrict__ weight, const uint32_t* __restrict__ vertexCellIds, TableMeta* __restrict__ meta,
                    TableEntry* __restrict__ arena, uint32_t* __restrict__ label0, uint64_t* __restrict__ state)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wavesPerGrid = gridDim.x * (blockDim.x / 64u);
    for (uint32_t v = blockIdx.x * (blockDim.x / 64u) + threadIdx.x / 64u; v < vertexCount; v += wavesPerGrid) {
        const uint64_t base = offsets[v];
        const uint32_t degree = uint32_t(offsets[v + 1] - base);
        TableMeta t;
        t.begin = 2ull * base + 8ull * v;
        t.size = degree;
        t.capacity = 2u * degree + 8u;
        for (uint32_t i = lane; i < degree; i += 64u) {
            arena[t.begin + i] = TableEntry{vertexCellIds[neighbour[base + i]], weight[base + i]};
        }
        findBest(t, arena, lane);
        if (lane == 0u) {
            meta[v] = t;
            label0[v] = vertexCellIds[v];
            state[v] = 0ull;
        }
    }
}

__global__ void __launch_bounds__(256)
positionsKernel(const uint32_t* __restrict__ order, uint32_t count, uint32_t* __restrict__ position)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < count) position[order[p]] = p;
}

__global__ void __launch_bounds__(256)
recordPositionsKernel(const uint32_t* __restrict__ order, uint32_t count, VertexRecord* __restrict__ records)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < count) records[order[p]].posCur = p;
}

__global__ void __launch_bounds__(256)
initialRecordsKernel(const uint32_t* __restrict__ vertexCellIds, uint32_t count, VertexRecord* __restrict__ first, VertexRecord* __restrict__ second)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= count) return;
    // posCur is the positions kernels' field, written on another stream (that of the second iteration possibly before this
    // kernel has run): leave it alone in both records
    const uint32_t cell = vertexCellIds[v];
    uint32_t* a = reinterpret_cast<uint32_t*>(first + v);
    uint32_t* b = reinterpret_cast<uint32_t*>(second + v);
    a[0] = cell;
    a[1] = cell;
    a[2] = 0u;
    a[4] = 0u;
    a[5] = 0u;
    b[0] = 0u;
    b[1] = 0u;
    b[2] = 0u;
    b[4] = 0u;
    b[5] = 0u;
}

__global__ void __launch_bounds__(256)
recordLabelsKernel(const VertexRecord* __restrict__ records, uint32_t count, uint32_t* __restrict__ labels)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < count) labels[v] = records[v].labelPrev;
}

// The two ends of edge e are records 2e and 2e+1: key = the vertex whose list the record joins.
__global__ void __launch_bounds__(256)
edgeEndsKernel(const uint32_t* __restrict__ edge0, const uint32_t* __restrict__ edge1, uint32_t slots,
               uint32_t* __restrict__ keys, uint32_t* __restrict__ ends, uint32_t* __restrict__ degree)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= slots) return;
    const uint32_t vertex = (r & 1u) ? edge1[r >> 1] : edge0[r >> 1];
    keys[r] = vertex;
    ends[r] = r;
    atomicAdd(degree + vertex, 1u);
}

__global__ void __launch_bounds__(256)
adjacencyKernel(const uint32_t* __restrict__ endsSorted, uint32_t slots, const uint32_t* __restrict__ edge0,
                const uint32_t* __restrict__ edge1, const float* __restrict__ similarity, uint32_t* __restrict__ neighbour,
                float* __restrict__ weight)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= slots) return;
    const uint32_t r = endsSorted[j];
    neighbour[j] = (r & 1u) ? edge0[r >> 1] : edge1[r >> 1];
    weight[j] = similarity[r >> 1];
}

// A buffer of its own, or a 256-byte aligned piece of a Pool: releasing device memory is what costs (a dozen hipFree calls on
// 1.5 GB took 3 ms of a 58 ms call), so the buffers of one phase share one allocation.
struct Pool {
    char* base = nullptr;
    size_t size = 0, used = 0;
    ~Pool() { if (base) (void)hipFree(base); }
    hipError_t reserve(size_t bytes)
    {
        size = bytes;
        return hipMalloc(reinterpret_cast<void**>(&base), bytes ? bytes : 16);
    }
    static size_t rounded(size_t bytes) { return (std::max<size_t>(bytes, 16) + 255u) & ~size_t(255); }
};

struct Buffer {
    void* p = nullptr;
    bool owned = false;
    ~Buffer() { release(); }
    void release()
    {
        if (p && owned) (void)hipFree(p);
        p = nullptr;
        owned = false;
    }
    hipError_t allocate(size_t bytes)
    {
        owned = true;
        return hipMalloc(&p, bytes ? bytes : 16);
    }
    hipError_t allocate(Pool& pool, size_t bytes)
    {
        const size_t need = Pool::rounded(bytes);
        if (pool.base && pool.used + need <= pool.size) {
            p = pool.base + pool.used;
            pool.used += need;
            owned = false;
            return hipSuccess;
        }
        return allocate(bytes);
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// EM2_TIMING=1: wall time of the stages of one call, on stderr.
struct StageClock {
    bool on = getenv("EM2_TIMING") && getenv("EM2_TIMING")[0] == '1';
    std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    double lap()
    {
        const auto now = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(now - last).count();
        last = now;
        return ms;
    }
    void stage(const char* name)
    {
        if (on) fprintf(stderr, "[em2 timing]   label propagation: %s %.1f ms\n", name, lap());
    }
};

// The visiting orders of the iterations (CellGraph.cpp:480-489: one std::mt19937 seeded once, one std::shuffle of the vertices
// in ascending cell id per iteration -- libstdc++'s, which is what the reference links), drawn by a thread of its own ahead of
// the GPU: an order costs 3.4 ms of host time at a million vertices, more than an iteration without label changes costs the
// GPU, and depends on nothing but the generator.  The orders land in pinned memory so that their upload overlaps the kernels.
class OrderProducer {
public:
    OrderProducer(const uint32_t* input, uint32_t count, uint64_t seed, uint64_t iterations)
        : input_(input), count_(count), iterations_(iterations), generator_(seed)
    {
        // one pinned block for all slots, kept by the process between calls (pinning and unpinning 32 MB costs milliseconds)
        const size_t bytes = std::max<size_t>(size_t(kSlots) * count * sizeof(uint32_t), 64);
        block_ = pinnedPool().take(bytes);
        if (block_) {
            for (uint32_t i = 0; i < kSlots; ++i) slots_[i] = static_cast<uint32_t*>(block_) + size_t(i) * count;
        } else {
            pageable_.resize(size_t(kSlots) * count);
            for (uint32_t i = 0; i < kSlots; ++i) slots_[i] = pageable_.data() + size_t(i) * count;
        }
        try {
            const char* wanted = getenv("EM2_LABEL_ORDER_THREAD");          // (0: the fallback below, for its test)
            if (wanted && wanted[0] == '0') throw 0;
            thread_ = std::thread([this] { run(); });
            threaded_ = true;
        } catch (...) {
            threaded_ = false;          // no thread to be had: the orders are drawn by the caller, when it asks for them
        }
    }
    ~OrderProducer()
    {
        {
            std::lock_guard<std::mutex> guard(mutex_);
            stop_ = true;
        }
        stopRequested_.store(true, std::memory_order_relaxed);
        changed_.notify_all();
        if (threaded_) thread_.join();
        if (block_) pinnedPool().give(block_, blockBytes());
    }
    OrderProducer(const OrderProducer&) = delete;
    OrderProducer& operator=(const OrderProducer&) = delete;

    // The order of that iteration; waits for it to be drawn.
    const uint32_t* order(uint64_t iteration)
    {
        if (!threaded_) {
            while (produced_ <= iteration) draw(produced_++);
            return slots_[iteration % kSlots];
        }
        std::unique_lock<std::mutex> lock(mutex_);
        changed_.wait(lock, [&] { return produced_ > iteration; });
        return slots_[iteration % kSlots];
    }
    bool ready(uint64_t iteration)
    {
        if (!threaded_) return false;
        std::lock_guard<std::mutex> guard(mutex_);
        return produced_ > iteration;
    }
    // The orders of the iterations below this one are no longer read (their uploads have completed).
    void release(uint64_t iteration)
    {
        {
            std::lock_guard<std::mutex> guard(mutex_);
            released_ = std::max(released_, iteration);
        }
        changed_.notify_all();
    }

private:
    static constexpr uint32_t kSlots = 8;
    size_t blockBytes() const { return std::max<size_t>(size_t(kSlots) * count_ * sizeof(uint32_t), 64); }
    // At most one pinned block is kept; a call that finds it taken (another thread) or too small pins its own.
    struct PinnedPool {
        std::mutex mutex;
        void* kept = nullptr;
        size_t keptBytes = 0;
        void* take(size_t bytes)
        {
            {
                std::lock_guard<std::mutex> guard(mutex);
                if (kept && keptBytes >= bytes) {
                    void* p = kept;
                    kept = nullptr;
                    return p;
                }
            }
            void* p = nullptr;
            if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                return nullptr;
            }
            return p;
        }
        void give(void* p, size_t bytes)
        {
            void* release = p;
            {
                std::lock_guard<std::mutex> guard(mutex);
                if (!kept || keptBytes < bytes) {
                    release = kept;
                    kept = p;
                    keptBytes = bytes;
                }
            }
            if (release) (void)hipHostFree(release);
        }
    };
    static PinnedPool& pinnedPool()
    {
        static PinnedPool pool;
        return pool;
    }
    // The order of iteration i into its slot; false when the call ended meanwhile.
    bool draw(uint64_t i)
    {
        uint32_t* slot = slots_[i % kSlots];
        std::copy(input_, input_ + count_, slot);
        if (count_ < 65536u) {
            std::shuffle(slot, slot + count_, generator_);
        } else {
            // std::shuffle itself, spelled out so that a call that has ended does not wait milliseconds for an order nobody
            // will read: with a 32-bit generator and 65536 elements or more (range of the generator / count < count)
            // libstdc++'s shuffle is exactly this loop over its own uniform_int_distribution<unsigned long> (bits/stl_algo.h;
            // the paired draws only exist below that size).  The GPU tests on both sides of the limit hold it to the oracle,
            // which calls std::shuffle.
            typedef std::uniform_int_distribution<unsigned long> Distribution;
            Distribution distribution;
            for (unsigned long at = 1; at < count_; ++at) {
                std::swap(slot[at], slot[distribution(generator_, Distribution::param_type(0, at))]);
                if ((at & 0x3fffu) == 0u && stopRequested_.load(std::memory_order_relaxed)) return false;
            }
        }
        return true;
    }
    void run()
    {
        for (uint64_t i = 0; i < iterations_; ++i) {
            {
                std::unique_lock<std::mutex> lock(mutex_);
                changed_.wait(lock, [&] { return stop_ || i < released_ + kSlots; });
                if (stop_) return;
            }
            if (!draw(i)) return;
            {
                std::lock_guard<std::mutex> guard(mutex_);
                produced_ = i + 1;
            }
            changed_.notify_all();
        }
    }
    const uint32_t* input_;
    uint32_t count_;
    uint64_t iterations_;
    std::mt19937 generator_;
    uint32_t* slots_[kSlots] = {};
    void* block_ = nullptr;
    std::vector<uint32_t> pageable_;
    std::mutex mutex_;
    std::condition_variable changed_;
    uint64_t produced_ = 0, released_ = 0;
    bool stop_ = false;
    std::atomic<bool> stopRequested_{false};
    bool threaded_ = false;
    std::thread thread_;
};

struct StreamHolder {
    hipStream_t stream = nullptr;
    hipEvent_t ready[2] = {nullptr, nullptr};
    ~StreamHolder()
    {
        for (hipEvent_t e : ready) {
            if (e) (void)hipEventDestroy(e);
        }
        if (stream) {
            (void)hipStreamSynchronize(stream);
            (void)hipStreamDestroy(stream);
        }
    }
};

#define EM2_TRY(call)                        \
    do {                                     \
        hipError_t em2Err_ = (call);         \
        if (em2Err_ != hipSuccess) return em2Err_; \
    } while (0)

}  // namespace

// Host buffers in, raw labels (cell ids of the cluster seeds) out; the caller renumbers them.  shuffleInput is the
// vector the reference shuffles every iteration: the vertices in ascending cell id (CellGraph.cpp:484-489).  Edges
// must not be self loops (a vertex never pulls from itself).  *error: 0 ok, 1 a wait timed out, 2 table arena
// exhausted.  Synchronises the stream.
static hipError_t runLabelPropagationBody(const uint32_t* vertexCellIds, uint32_t vertexCount, const uint32_t* edgeVertex0,
                                          const uint32_t* edgeVertex1, const float* edgeSimilarity, uint64_t edgeCount,
                                          const uint32_t* shuffleInput, uint64_t seed, uint64_t stableIterationCountThreshold,
                                          uint64_t maxIterationCount, uint32_t* labels, uint64_t* iterationCount, uint32_t* error,
                                          hipStream_t stream)
{
    *iterationCount = 0;
    *error = 0;
    if (vertexCount == 0) return hipSuccess;
    StageClock clock;
    const dim3 block(256);
    // (declared before every device buffer: the thread is joined, the copy stream drained, after the buffers are released --
    // both only touch memory of their own by then)
    std::unique_ptr<OrderProducer> orders(new OrderProducer(shuffleInput, vertexCount, seed, maxIterationCount));
    StreamHolder copies;
    EM2_TRY(hipStreamCreateWithFlags(&copies.stream, hipStreamNonBlocking));
    for (hipEvent_t& e : copies.ready) EM2_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    clock.stage("pinned order buffers, copy stream");

    // ---- out_edges() of every vertex in add_edge order: a stable sort of the 2E edge ends by vertex ----
    const uint64_t slots = 2 * edgeCount;
    if (slots >= (1ull << 32)) return hipErrorInvalidValue;
    Buffer dOffsets, dNeighbour, dWeight, dCells, dDegree;
    Pool graphPool;
    EM2_TRY(graphPool.reserve(Pool::rounded((size_t(vertexCount) + 1) * sizeof(uint64_t)) + 2 * Pool::rounded(slots * sizeof(uint32_t)) +
                              Pool::rounded(size_t(vertexCount) * sizeof(uint32_t)) + Pool::rounded((size_t(vertexCount) + 1) * sizeof(uint32_t))));
    EM2_TRY(dOffsets.allocate(graphPool, (size_t(vertexCount) + 1) * sizeof(uint64_t)));
    EM2_TRY(dNeighbour.allocate(graphPool, slots * sizeof(uint32_t)));
    EM2_TRY(dWeight.allocate(graphPool, slots * sizeof(float)));
    EM2_TRY(dCells.allocate(graphPool, size_t(vertexCount) * sizeof(uint32_t)));
    EM2_TRY(dDegree.allocate(graphPool, (size_t(vertexCount) + 1) * sizeof(uint32_t)));
    EM2_TRY(hipMemcpyAsync(dCells.p, vertexCellIds, size_t(vertexCount) * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    EM2_TRY(hipMemsetAsync(dDegree.p, 0, (size_t(vertexCount) + 1) * sizeof(uint32_t), stream));
    uint32_t maxDegree = 0;
    if (slots) {
        Buffer dEdge0, dEdge1, dSimilarity, dKeys, dKeysSorted, dEnds, dEndsSorted, dTemp, dMax;
        uint32_t keyBits = 1;
        while (keyBits < 32u && (1ull << keyBits) < vertexCount) ++keyBits;
        size_t sortBytes = 0, scanBytes = 0, maxBytes = 0;
        {
            uint32_t* const noWords = nullptr;          // (the queries only look at types and sizes)
            uint64_t* const noOffsets = nullptr;
            EM2_TRY(rocprim::radix_sort_pairs(nullptr, sortBytes, noWords, noWords, noWords, noWords, size_t(slots), 0u, keyBits, stream));
            EM2_TRY(rocprim::exclusive_scan(nullptr, scanBytes, noWords, noOffsets, uint64_t(0), size_t(vertexCount) + 1,
                                            rocprim::plus<uint64_t>(), stream));
            EM2_TRY(rocprim::reduce(nullptr, maxBytes, noWords, noWords, 0u, size_t(vertexCount), rocprim::maximum<uint32_t>(), stream));
        }
        const size_t tempBytes = std::max(sortBytes, std::max(scanBytes, maxBytes));
        Pool edgePool;
        EM2_TRY(edgePool.reserve(3 * Pool::rounded(edgeCount * sizeof(uint32_t)) + 4 * Pool::rounded(slots * sizeof(uint32_t)) +
                                 Pool::rounded(sizeof(uint32_t)) + Pool::rounded(tempBytes)));
        EM2_TRY(dEdge0.allocate(edgePool, edgeCount * sizeof(uint32_t)));
        EM2_TRY(dEdge1.allocate(edgePool, edgeCount * sizeof(uint32_t)));
        EM2_TRY(dSimilarity.allocate(edgePool, edgeCount * sizeof(float)));
        EM2_TRY(dKeys.allocate(edgePool, slots * sizeof(uint32_t)));
        EM2_TRY(dKeysSorted.allocate(edgePool, slots * sizeof(uint32_t)));
        EM2_TRY(dEnds.allocate(edgePool, slots * sizeof(uint32_t)));
        EM2_TRY(dEndsSorted.allocate(edgePool, slots * sizeof(uint32_t)));
        EM2_TRY(dMax.allocate(edgePool, sizeof(uint32_t)));
        EM2_TRY(dTemp.allocate(edgePool, tempBytes));
        EM2_TRY(hipMemcpyAsync(dEdge0.p, edgeVertex0, edgeCount * sizeof(uint32_t), hipMemcpyDefault, stream));       // (host or device arrays)
        EM2_TRY(hipMemcpyAsync(dEdge1.p, edgeVertex1, edgeCount * sizeof(uint32_t), hipMemcpyDefault, stream));       // (host or device arrays)
        EM2_TRY(hipMemcpyAsync(dSimilarity.p, edgeSimilarity, edgeCount * sizeof(float), hipMemcpyDefault, stream));       // (host or device arrays)
        const dim3 endGrid(uint32_t((slots + 255u) / 256u));
        edgeEndsKernel<<<endGrid, block, 0, stream>>>(dEdge0.as<uint32_t>(), dEdge1.as<uint32_t>(), uint32_t(slots),
                                                      dKeys.as<uint32_t>(), dEnds.as<uint32_t>(), dDegree.as<uint32_t>());
        EM2_TRY(hipGetLastError());
        EM2_TRY(rocprim::radix_sort_pairs(dTemp.p, sortBytes, dKeys.as<uint32_t>(), dKeysSorted.as<uint32_t>(), dEnds.as<uint32_t>(),
                                          dEndsSorted.as<uint32_t>(), size_t(slots), 0u, keyBits, stream));
        EM2_TRY(rocprim::exclusive_scan(dTemp.p, scanBytes, dDegree.as<uint32_t>(), dOffsets.as<uint64_t>(), uint64_t(0),
                                        size_t(vertexCount) + 1, rocprim::plus<uint64_t>(), stream));
        EM2_TRY(rocprim::reduce(dTemp.p, maxBytes, dDegree.as<uint32_t>(), dMax.as<uint32_t>(), 0u, size_t(vertexCount),
                                rocprim::maximum<uint32_t>(), stream));
        adjacencyKernel<<<endGrid, block, 0, stream>>>(dEndsSorted.as<uint32_t>(), uint32_t(slots), dEdge0.as<uint32_t>(),
                                                       dEdge1.as<uint32_t>(), dSimilarity.as<float>(), dNeighbour.as<uint32_t>(),
                                                       dWeight.as<float>());
        EM2_TRY(hipGetLastError());
        EM2_TRY(hipMemcpyAsync(&maxDegree, dMax.p, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        EM2_TRY(hipStreamSynchronize(stream));
    } else {
        EM2_TRY(hipMemsetAsync(dOffsets.p, 0, (size_t(vertexCount) + 1) * sizeof(uint64_t), stream));
    }
    clock.stage("adjacency");

    const uint64_t initialEntries = 2 * slots + 8ull * vertexCount;
    // Tables only grow (the reference never removes an entry either) and a table that fills up moves to the tail, so
    // the tail is sized generously and, should a graph whose labels keep churning outgrow it anyway, the whole run is
    // repeated with four times the tail (the algorithm is deterministic).  EM2_LABEL_ARENA_TAIL (entries) is a test knob.
    uint64_t arenaTail = std::max<uint64_t>(initialEntries, 1ull << 20);
    if (const char* v = getenv("EM2_LABEL_ARENA_TAIL")) {
        if (atoll(v) >= 0) arenaTail = uint64_t(atoll(v));
    }
    uint64_t arenaCapacity = initialEntries + arenaTail;
    Buffer dLabels, dState, dPositions, dOrder, dMeta, dArena, dControl, dScratch, dRecords, dSortKeys;
#ifdef EM2_DIAG
    Buffer dDiag;
#endif
    const bool hubs = maxDegree > 64u;
    Pool statePool;
    EM2_TRY(statePool.reserve(Pool::rounded(4 * size_t(vertexCount) * sizeof(uint32_t)) + Pool::rounded(size_t(vertexCount) * sizeof(uint64_t)) +
                              Pool::rounded(2 * size_t(vertexCount) * sizeof(VertexRecord)) +
                              Pool::rounded(3 * size_t(vertexCount) * sizeof(uint32_t)) + Pool::rounded(2 * size_t(vertexCount) * sizeof(uint32_t)) +
                              Pool::rounded(size_t(vertexCount) * sizeof(TableMeta)) + Pool::rounded(64) + Pool::rounded(kDiagWords * 8) +
                              Pool::rounded(hubs ? 4 * slots * sizeof(uint64_t) : 0)));
    EM2_TRY(dLabels.allocate(statePool, 4 * size_t(vertexCount) * sizeof(uint32_t)));
    EM2_TRY(dState.allocate(statePool, size_t(vertexCount) * sizeof(uint64_t)));
    EM2_TRY(dRecords.allocate(statePool, 2 * size_t(vertexCount) * sizeof(VertexRecord)));
    EM2_TRY(dPositions.allocate(statePool, 3 * size_t(vertexCount) * sizeof(uint32_t)));
    EM2_TRY(dOrder.allocate(statePool, 2 * size_t(vertexCount) * sizeof(uint32_t)));
    EM2_TRY(dMeta.allocate(statePool, size_t(vertexCount) * sizeof(TableMeta)));
    EM2_TRY(dControl.allocate(statePool, 4 * sizeof(uint32_t) + sizeof(unsigned long long)));
#ifdef EM2_DIAG
    EM2_TRY(dDiag.allocate(statePool, kDiagWords * sizeof(unsigned long long)));
#endif
    if (hubs) {
        EM2_TRY(dSortKeys.allocate(statePool, 4 * slots * sizeof(uint64_t)));
    }
    // (the tables' arena stays an allocation of its own: a run whose tables outgrow it replaces it)
    EM2_TRY(dArena.allocate(arenaCapacity * sizeof(TableEntry)));

    int device = 0, computeUnits = 0;
    EM2_TRY(hipGetDevice(&device));
    EM2_TRY(hipDeviceGetAttribute(&computeUnits, hipDeviceAttributeMultiprocessorCount, device));
    // The strided schedule needs every wave of the grid resident: at most 4 blocks (16 waves) per CU, where the kernel's
    // registers allow 7.  EM2_LABEL_TICKET_BATCH=n (n > 0) selects the ticket schedule instead, which does not care
    // (a GPU shared with another process); a strided run that times out is repeated that way, from the start.
    const char* batchText = getenv("EM2_LABEL_TICKET_BATCH");
    uint32_t ticketBatch = batchText && atoi(batchText) > 0 ? uint32_t(atoi(batchText)) : 0u;
    // a ticket per compute unit (kScheduleUnit) unless the global ticket is asked for or a unit's block does not fit the device
    bool unitSchedule = ticketBatch == 0u;
    // 256-thread blocks: an area per wave.  Units of 16 waves: a pool of kUnitAreas, so that two units fit a compute unit.
    constexpr uint32_t kUnitAreas = 8;
    const size_t areaBytes4 = 4 * sizeof(WaveArea) + 64, areaBytes16 = kUnitAreas * sizeof(WaveArea) + 64;
    int unitsPerCu = 0;
    if (unitSchedule) {
        // (a device that cannot hold a 16-wave block with its LDS areas takes the global ticket)
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(labelPropagationCachedKernel<kScheduleUnit>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, int(areaBytes16)) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&unitsPerCu, labelPropagationCachedKernel<kScheduleUnit>, 1024, areaBytes16) != hipSuccess ||
            unitsPerCu < 1) {
            (void)hipGetLastError();
            unitSchedule = false;
        }
        unitsPerCu = std::min(unitsPerCu, 2);
    }
    if (!unitSchedule && ticketBatch == 0u) ticketBatch = 4u;          // (no room for a unit's block: the global ticket)
    int smallBlocksPerUnit = 0;
    EM2_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&smallBlocksPerUnit, labelPropagationCachedKernel<kScheduleTicket>, 256, areaBytes4));
    smallBlocksPerUnit = std::max(1, std::min(smallBlocksPerUnit, 4));
    // 256-thread blocks (strided or ticket schedule, the older kernel, the first tables) and one 1024-thread block per unit
    const dim3 grid(std::min<uint32_t>((vertexCount + 3u) / 4u, uint32_t(computeUnits) * uint32_t(smallBlocksPerUnit)));
    if (clock.on && unitSchedule) fprintf(stderr, "[em2 timing]   label propagation: %d unit(s) of 16 waves per compute unit\n", unitsPerCu);
    const dim3 unitGrid(std::min<uint32_t>((vertexCount + 15u) / 16u, uint32_t(computeUnits) * uint32_t(std::max(1, unitsPerCu))));
    uint32_t* label[4];
    for (int i = 0; i < 4; i++) label[i] = dLabels.as<uint32_t>() + size_t(i) * vertexCount;
    // three position arrays and two order arrays: those of iteration t + 1 are filled on the copy stream while the kernel of
    // iteration t still reads those of t and t - 1
    uint32_t* position[3] = {dPositions.as<uint32_t>(), dPositions.as<uint32_t>() + vertexCount, dPositions.as<uint32_t>() + 2 * size_t(vertexCount)};
    uint32_t* orderOf[2] = {dOrder.as<uint32_t>(), dOrder.as<uint32_t>() + vertexCount};
    VertexRecord* recordsOf[2] = {dRecords.as<VertexRecord>(), dRecords.as<VertexRecord>() + vertexCount};
    uint32_t* control = dControl.as<uint32_t>();
    unsigned long long* arenaTop = reinterpret_cast<unsigned long long*>(control + 4);
    clock.stage("allocate");
    // Order and positions of an iteration, on the copy stream, from the producer's pinned memory.
    auto upload = [&](uint64_t iteration) -> hipError_t {
        const uint32_t* order = orders->order(iteration);
        EM2_TRY(hipMemcpyAsync(orderOf[iteration & 1u], order, size_t(vertexCount) * sizeof(uint32_t), hipMemcpyHostToDevice, copies.stream));
        recordPositionsKernel<<<dim3((vertexCount + 255u) / 256u), block, 0, copies.stream>>>(orderOf[iteration & 1u], vertexCount,
                                                                                             recordsOf[iteration & 1u]);
        EM2_TRY(hipGetLastError());
        return hipEventRecord(copies.ready[iteration & 1u], copies.stream);
    };

    // (test knob of the schedule, read once: a smaller pool of LDS areas -- turns that find none)
    uint32_t poolAreas = unitSchedule ? kUnitAreas : 4u;
    if (const char* text = getenv("EM2_LABEL_POOL_AREAS")) {
        if (atoi(text) >= 0 && uint32_t(atoi(text)) < poolAreas) poolAreas = uint32_t(atoi(text));
    }
    constexpr uint32_t unitWaves = 16u;

    int arenaGrowths = 0;
    bool triedTicket = false;           // (its own flag: an arena growth before the time-out must not forfeit the ticket repeat)
    for (int attempt = 0;; ++attempt) {
        initialTablesKernel<<<grid, block, 0, stream>>>(vertexCount, dOffsets.as<uint64_t>(), dNeighbour.as<uint32_t>(),
                                                        dWeight.as<float>(), dCells.as<uint32_t>(), dMeta.as<TableMeta>(),
                                                        dArena.as<TableEntry>(), label[0], dState.as<uint64_t>());
        EM2_TRY(hipGetLastError());
        initialRecordsKernel<<<dim3((vertexCount + 255u) / 256u), block, 0, stream>>>(dCells.as<uint32_t>(), vertexCount, recordsOf[0],
                                                                                     recordsOf[1]);
        EM2_TRY(hipGetLastError());
        const uint32_t zero[4] = {0, 0, 0, 0};
        EM2_TRY(hipMemcpyAsync(control, zero, sizeof(zero), hipMemcpyHostToDevice, stream));
        const unsigned long long top = initialEntries;
        EM2_TRY(hipMemcpyAsync(arenaTop, &top, sizeof(top), hipMemcpyHostToDevice, stream));

        // :480-549.  The orders come from the producer thread; order and positions of iteration t + 1 are uploaded while the
        // kernel of iteration t runs.
        if (attempt > 0) {
            EM2_TRY(hipStreamSynchronize(copies.stream));
            orders.reset();
            orders.reset(new OrderProducer(shuffleInput, vertexCount, seed, maxIterationCount));
        }
        uint64_t stable = 0, iterations = 0;
        uint32_t failure = 0;
        if (maxIterationCount > 0) EM2_TRY(upload(0));
        clock.stage("first tables, first order");
        while (iterations < maxIterationCount) {
            const uint32_t t = uint32_t(iterations);
            EM2_TRY(hipStreamWaitEvent(stream, copies.ready[t & 1u], 0));
            EM2_TRY(hipMemsetAsync(control, 0, 2 * sizeof(uint32_t), stream));
            ClusterArgs args;
            args.vertexCount = vertexCount;
            args.iteration = t;
            args.offsets = dOffsets.as<uint64_t>();
            args.neighbour = dNeighbour.as<uint32_t>();
            args.weight = dWeight.as<float>();
            args.order = orderOf[t & 1u];
            args.posCur = position[t % 3u];
            args.posPrev = position[(t + 2u) % 3u];
            args.labelPrev = label[t & 3u];
            args.labelPrev2 = label[(t + 3u) & 3u];
            args.labelCur = label[(t + 1u) & 3u];
            args.state = dState.as<uint64_t>();
            args.records = recordsOf[t & 1u];
            args.recordsNext = recordsOf[(t + 1u) & 1u];
            args.meta = dMeta.as<TableMeta>();
            args.arena = dArena.as<TableEntry>();
            args.arenaTop = arenaTop;
            args.arenaCapacity = arenaCapacity;
            args.control = control;
            args.ticketBatch = ticketBatch;
            args.poolAreas = unitSchedule ? std::min(poolAreas, kUnitAreas) : std::min(poolAreas, 4u);
            args.scratchA = dScratch.as<Candidate>();
            args.scratchB = dScratch.as<Candidate>() + slots;
            args.sortKeys = dSortKeys.as<uint64_t>();
            args.slots = slots;
#ifdef EM2_DIAG
            args.diag = clock.on ? dDiag.as<unsigned long long>() : nullptr;
            EM2_TRY(hipMemsetAsync(dDiag.p, 0, kDiagWords * sizeof(unsigned long long), stream));
#endif
            if (unitSchedule) {
                labelPropagationCachedKernel<kScheduleUnit><<<unitGrid, dim3(64u * unitWaves), areaBytes16, stream>>>(args);
            }
            else labelPropagationCachedKernel<kScheduleTicket><<<grid, block, areaBytes4, stream>>>(args);
            EM2_TRY(hipGetLastError());
            ++iterations;
            // the next order goes up while this kernel runs -- if it has been drawn already: this iteration may be the last, and
            // then nobody should wait milliseconds for an order that is not needed
            bool uploaded = false;
            if (iterations < maxIterationCount && orders->ready(iterations)) {
                EM2_TRY(upload(iterations));
                uploaded = true;
            }
            uint32_t result[3] = {0, 0, 0};
            EM2_TRY(hipMemcpyAsync(result, control, sizeof(result), hipMemcpyDeviceToHost, stream));
            EM2_TRY(hipStreamSynchronize(stream));
            orders->release(iterations);          // the kernel that waited for this order's upload has completed
            if (clock.on) fprintf(stderr, "[em2 timing]   label propagation: iteration %u, %u changes, %.1f ms\n", t, result[1], clock.lap());
#ifdef EM2_DIAG
            if (clock.on) {
                unsigned long long d[kDiagWords];
                EM2_TRY(hipMemcpy(d, dDiag.p, sizeof(d), hipMemcpyDeviceToHost));
                const double whole = double(d[9]) > 0 ? double(d[9]) : 1.0;
                fprintf(stderr, "[em2 timing]     wave cycles %%: header %.1f gather %.1f A %.1f B %.1f wait %.1f hub gather %.1f hub sort %.1f hub events %.1f "
                        "store %.1f | events A %llu B %llu, waits %llu, hub turns %llu, rescans %llu, relocations %llu, mean wave %.2f Mticks\n",
                        100 * d[0] / whole, 100 * d[1] / whole, 100 * d[2] / whole, 100 * d[3] / whole, 100 * d[4] / whole, 100 * d[5] / whole,
                        100 * d[6] / whole, 100 * d[7] / whole, 100 * d[8] / whole, d[10], d[11], d[12], d[13], d[14], d[15],
                        whole / (unitSchedule ? double(unitGrid.x) * 16.0 : double(grid.x) * 4.0) / 1e6);
                fprintf(stderr, "[em2 timing]     LDS form, degree <= 64: ordering the events %.1f %%, applying them %.1f %% of the wave cycles (B = the first polls)\n",
                        100 * d[16] / whole, 100 * d[17] / whole);
                fprintf(stderr, "[em2 timing]     ... of which waiting for the table's load %.1f %%; addWeight calls that left the short path %llu\n",
                        100 * d[18] / whole, d[19]);
                fprintf(stderr, "[em2 timing]     general addWeight from the batch: registers %llu x %.0f cycles, LDS %llu x %.0f, global %llu x %.0f\n", d[25],
                        d[25] ? double(d[24]) / double(d[25]) : 0.0, d[27], d[27] ? double(d[26]) / double(d[27]) : 0.0, d[29],
                        d[29] ? double(d[28]) / double(d[29]) : 0.0);
                fprintf(stderr, "[em2 timing]     short-path addWeight: %llu calls, %.0f cycles each between two s_memtime\n", d[21], d[21] ? double(d[20]) / double(d[21]) : 0.0);
                clock.lap();
            }
#endif
            if (result[2] != 0) {
                failure = result[2];
                break;
            }
            stable = result[1] ? 0 : stable + 1;
            if (stable == stableIterationCountThreshold) break;
            if (!uploaded && iterations < maxIterationCount) EM2_TRY(upload(iterations));
        }
        if (failure == 1 && ticketBatch == 0 && !triedTicket) {
            triedTicket = true;
            if (clock.on) fprintf(stderr, "[em2 timing]   label propagation: the strided schedule timed out, repeating with the ticket\n");
            ticketBatch = 4;
            unitSchedule = false;
            continue;
        }
        if (failure == 2 && arenaGrowths < 6) {
            ++arenaGrowths;
            arenaTail = std::max<uint64_t>(arenaTail, 1ull << 16) * 4u;
            arenaCapacity = initialEntries + arenaTail;
            if (clock.on) fprintf(stderr, "[em2 timing]   label propagation: tables outgrew the arena, repeating with %llu entries\n",
                                  (unsigned long long)arenaCapacity);
            dArena.release();
            const hipError_t grown = dArena.allocate(arenaCapacity * sizeof(TableEntry));
            if (grown != hipSuccess) {
                *iterationCount = iterations;
                *error = failure;           // no memory for a larger arena: report the exhaustion
                (void)hipGetLastError();
                return hipSuccess;
            }
            continue;
        }
        *iterationCount = iterations;
        if (failure) {
            *error = failure;
            return hipSuccess;
        }
        recordLabelsKernel<<<dim3((vertexCount + 255u) / 256u), block, 0, stream>>>(recordsOf[iterations & 1u], vertexCount, label[0]);
        EM2_TRY(hipGetLastError());
        EM2_TRY(hipMemcpyAsync(labels, label[0], size_t(vertexCount) * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        EM2_TRY(hipStreamSynchronize(stream));
        clock.stage("labels to the host");
        return hipSuccess;
    }
}

hipError_t runLabelPropagation(const uint32_t* vertexCellIds, uint32_t vertexCount, const uint32_t* edgeVertex0,
                               const uint32_t* edgeVertex1, const float* edgeSimilarity, uint64_t edgeCount,
                               const uint32_t* shuffleInput, uint64_t seed, uint64_t stableIterationCountThreshold,
                               uint64_t maxIterationCount, uint32_t* labels, uint64_t* iterationCount, uint32_t* error,
                               hipStream_t stream)
{
    StageClock whole;
    const hipError_t status = runLabelPropagationBody(vertexCellIds, vertexCount, edgeVertex0, edgeVertex1, edgeSimilarity, edgeCount,
                                                      shuffleInput, seed, stableIterationCountThreshold, maxIterationCount, labels,
                                                      iterationCount, error, stream);
    whole.stage("the whole call, buffers released");
    return status;
}

}  // namespace em2
