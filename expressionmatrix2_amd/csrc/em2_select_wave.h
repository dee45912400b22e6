// em2_select_wave.h -- the exact keepBest selection of em2_select.h executed by a whole wave (device only).
//
// std::nth_element's work is its Hoare partitions.  Which elements one partition swaps, and where it cuts, is a
// function of the ORIGINAL positions of the elements at which the two scans stop:
//     L = positions x in [lo,hi), ascending,  with !(a[x] < pivot)   (the upward scan stops there)
//     R = positions y in [lo,hi), descending, with !(pivot < a[y])   (the downward scan stops there)
// The sequential loop swaps the pairs (L[t], R[t]) for t = 0,1,.. while L[t] < R[t] -- all disjoint -- and returns
// L[T] at the first t = T where that fails, or R[T-1] when no untouched left stopper lies below it (the element
// swapped in there stops the scan).  So a wave builds L and R with ballots + prefix popcounts, performs all
// swaps at once and reads the cut: a few dozen instructions per partition instead of a dependent LDS access per
// element from a single lane.  Median-of-3, the <=3-element insertion sort and the depth-limit heap fallback stay
// on lane 0 (em2_select.h).  tests/native/em2_host_checks.cpp holds a lane-by-lane host model of exactly this
// formulation, checked against std::nth_element / std::__introselect (tests/test_select_emulation.py).
#ifndef EM2_SELECT_WAVE_H
#define EM2_SELECT_WAVE_H

#include "em2_select.h"

namespace em2 {

__device__ __forceinline__ void waveSync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// The same for arrays in global memory: what one lane stored must be what another lane of the wave loads next, and a
// lane's loads may not come from a line its L1 fetched before that store -- the agent-scope fence writes back and
// invalidates.
__device__ __forceinline__ void waveSyncGlobal()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
    __builtin_amdgcn_wave_barrier();
}

// Sorts lds[0, n) ascending by (key, cell) -- the order of the reference's final sort by similarity, ties by cell id
// (src/ExpressionMatrixLsh.cpp:330-340 via keepBest) -- with a bitonic network over the next power of two (< 2n <= 2k
// slots, which the wave's area has), sentinels behind the entries.  (It was a rank count, n^2 comparisons: 2.8 of the
// 9.5 ms of the inbox replay kernel at 1M cells, k = 100.)
// Up to 128 entries (the k = 100 lists of the benchmark configurations): the same bitonic network in registers, two entries per
// lane (slots lane and lane + 64) as 64-bit (key, cell) values, partners fetched across lanes with shuffles -- no LDS round trips
// and no synchronisation between the 28 stages (the LDS form spends ~13k cycles on 128 slots, a seventh of findSimilarPairs5's
// selection kernel and a third of the inbox replay's finish).
__device__ __attribute__((noinline)) void sortListWaveRegisters(Entry* lds, uint32_t n, uint32_t lane)
{
    uint64_t v0 = ~0ull, v1 = ~0ull;          // sentinels sort behind every entry
    if (lane < n) {
        const Entry e = lds[lane];
        v0 = (uint64_t(e.key) << 32) | e.cell;
    }
    if (lane + 64u < n) {
        const Entry e = lds[lane + 64u];
        v1 = (uint64_t(e.key) << 32) | e.cell;
    }
    auto exchange = [&](uint64_t x, uint32_t stride, bool keepMin) {
        const uint32_t lo = uint32_t(__shfl_xor(int(uint32_t(x)), int(stride), 64));
        const uint32_t hi = uint32_t(__shfl_xor(int(uint32_t(x >> 32)), int(stride), 64));
        const uint64_t y = (uint64_t(hi) << 32) | lo;
        const bool takeOther = keepMin ? y < x : y > x;
        return takeOther ? y : x;
    };
#pragma unroll
    for (uint32_t size = 2; size <= 128u; size <<= 1) {
#pragma unroll
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride == 64u) {
                // (size 128: every pair (i, i + 64) ascends)
                const uint64_t a = v0 < v1 ? v0 : v1, b = v0 < v1 ? v1 : v0;
                v0 = a;
                v1 = b;
            } else {
                const bool lower = (lane & stride) == 0u;
                const bool ascending0 = (lane & size) == 0u;                  // slot `lane`
                const bool ascending1 = ((lane + 64u) & size) == 0u;          // slot `lane + 64`
                v0 = exchange(v0, stride, lower == ascending0);
                v1 = exchange(v1, stride, lower == ascending1);
            }
        }
    }
    waveSync();          // every lane has read its entries
    if (lane < n) {
        Entry e;
        e.cell = uint32_t(v0);
        e.key = uint32_t(v0 >> 32);
        lds[lane] = e;
    }
    if (lane + 64u < n) {
        Entry e;
        e.cell = uint32_t(v1);
        e.key = uint32_t(v1 >> 32);
        lds[lane + 64u] = e;
    }
    waveSync();
}

__device__ __forceinline__ void sortListWave(Entry* lds, uint32_t n, uint32_t lane)
{
    if (n <= 128u) {
        sortListWaveRegisters(lds, n, lane);
        return;
    }
    uint32_t padded = 1;
    while (padded < n) padded <<= 1;
    for (uint32_t i = n + lane; i < padded; i += 64u) {
        Entry sentinel;
        sentinel.cell = 0xffffffffu;
        sentinel.key = 0xffffffffu;
        lds[i] = sentinel;
    }
    waveSync();
    for (uint32_t size = 2; size <= padded; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = lane; t < (padded >> 1); t += 64u) {
                const uint32_t a = ((t & ~(stride - 1u)) << 1) | (t & (stride - 1u));
                const uint32_t b = a | stride;
                const Entry ea = lds[a], eb = lds[b];
                const bool bFirst = (eb.key < ea.key) || (eb.key == ea.key && eb.cell < ea.cell);
                const bool ascending = (a & size) == 0u;
                if (bFirst == ascending) {
                    lds[a] = eb;
                    lds[b] = ea;
                }
            }
            waveSync();
        }
    }
}

template <bool GLOBAL>
__device__ __forceinline__ void waveSyncFor()
{
    if (GLOBAL) waveSyncGlobal();
    else waveSync();
}

// __move_median_to_first for the wave forms in LDS (one lane calls it): the four elements are loaded at once -- one LDS round trip
// instead of a chain of dependent ones (a 200-entry selection of the scan spends most of its time between partitions) --, the
// decision tree is medianToFirst's, and swapping a[result] with the median is two stores.  result, ia, ib, ic are distinct
// (introselect calls it on ranges of more than three elements).
template <class E> __device__ __forceinline__ uint32_t medianToFirstLoaded(E* a, int result, int ia, int ib, int ic)
{
    const E er = a[result], ea = a[ia], eb = a[ib], ec = a[ic];
    int m;
    if (ea.key < eb.key) {
        if (eb.key < ec.key) m = ib;
        else if (ea.key < ec.key) m = ic;
        else m = ia;
    }
    else if (ea.key < ec.key) m = ia;
    else if (eb.key < ec.key) m = ic;
    else m = ib;
    const E em = m == ia ? ea : (m == ib ? eb : ec);
    a[result] = em;
    a[m] = er;
    return uint32_t(em.key);            // (the partition's pivot: its caller hands it on instead of reading a[result] back)
}

__device__ __forceinline__ uint32_t lanesBelow(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
}

// __unguarded_partition(a+lo, a+hi, pivot = a[lo-1]); returns the cut (wave-uniform).  Index: uint16_t positions for lists
// staged in LDS, uint32_t with GLOBAL for lists of any length left in global memory.
// CHUNKS > 1 (LDS form only): the pipelined form for long lists -- findSimilarPairs5's, thousands of entries, one wave per SIMD.
// The scan's 2k-entry selections keep CHUNKS = 1: their ranges fit one or two chunks, and the unrolled bodies cost the
// scan kernels more than they saved (same box, alternating: +0.3 ms kernel, +0.7 ms scan at 1M cells).
template <class Index, bool GLOBAL, class E = Entry, int CHUNKS = 1>
__device__ inline int partitionWaveT(E* a, int lo, int hi, Index* Lpos, Index* Rpos, uint32_t lane, bool havePivot = false, uint32_t pivotKey = 0u)
{
    // (a 2k-entry selection of the scan is a chain of LDS round trips under the neighbour wave's matrix instructions: every one
    // that can be had from a register is taken from there -- the pivot from the lane that placed it, the next chunk's keys a turn
    // ahead, the cut from the lanes that hold the crossing pair)
    const uint32_t pk = havePivot ? pivotKey : uint32_t(a[lo - 1].key);
    int nL = 0, nR = 0;
    // (LDS form: the keys of four chunks are loaded before the first of them is used -- the scan writes only the position
    // arrays, and a wave that waited for every chunk's load spent its time in LDS latency, one wave per SIMD being the rule)
    constexpr int kChunks = GLOBAL ? 1 : CHUNKS;
    uint32_t ahead = (!GLOBAL && CHUNKS == 1 && lo + int(lane) < hi) ? uint32_t(a[lo + int(lane)].key) : 0u;
    for (int base = lo; base < hi; base += 64 * kChunks) {
        uint32_t key[kChunks];
        bool valid[kChunks];
#pragma unroll
        for (int j = 0; j < kChunks; ++j) {
            const int x = base + 64 * j + int(lane);
            valid[j] = x < hi;
            if (!GLOBAL && CHUNKS == 1) {
                key[j] = ahead;
                ahead = x + 64 < hi ? uint32_t(a[x + 64].key) : 0u;
            } else {
                key[j] = valid[j] ? a[x].key : 0u;
            }
        }
#pragma unroll
        for (int j = 0; j < kChunks; ++j) {
            if (j > 0 && base + 64 * j >= hi) break;          // (uniform: the short ranges of a 2k-entry list end in the first chunk)
            const int x = base + 64 * j + int(lane);
            const bool isL = valid[j] && key[j] >= pk;
            const bool isR = valid[j] && key[j] <= pk;
            const uint64_t mL = __builtin_amdgcn_ballot_w64(isL);
            const uint64_t mR = __builtin_amdgcn_ballot_w64(isR);
            if (isL) Lpos[nL + int(lanesBelow(mL))] = Index(x);
            if (isR) Rpos[nR + int(lanesBelow(mR))] = Index(x);
            nL += __builtin_popcountll(mL);
            nR += __builtin_popcountll(mR);
        }
    }
    waveSyncFor<GLOBAL>();
    int T = 0;
    if (!GLOBAL && CHUNKS > 1) {
        // The pairs (L[t], R[t]) are disjoint for all t below T (header), so a lane's loads and stores of its own pairs need no
        // order against any other lane's: two chunks of pairs per turn, all four loads before the stores.
        for (int base = 0; base < nL; base += 128) {
            int x[2] = {0, 0}, y[2] = {0, 0};
            bool c[2] = {false, false}, valid[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int t = base + 64 * j + int(lane);
                valid[j] = t < nL;
                if (valid[j]) {
                    x[j] = int(Lpos[t]);
                    y[j] = t < nR ? int(Rpos[nR - 1 - t]) : lo - 1;
                    c[j] = x[j] < y[j];
                }
            }
            E ex[2] = {E(), E()}, ey[2] = {E(), E()};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (c[j]) {
                    ex[j] = a[x[j]];
                    ey[j] = a[y[j]];
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (c[j]) {
                    a[x[j]] = ey[j];
                    a[y[j]] = ex[j];
                }
            }
            const uint64_t m0 = __builtin_amdgcn_ballot_w64(c[0]);
            T += __builtin_popcountll(m0);
            if (m0 != __builtin_amdgcn_ballot_w64(valid[0])) break;
            if (base + 64 >= nL) break;
            const uint64_t m1 = __builtin_amdgcn_ballot_w64(c[1]);
            T += __builtin_popcountll(m1);
            if (m1 != __builtin_amdgcn_ballot_w64(valid[1])) break;
        }
    } else {
        for (int base = 0; base < nL; base += 64) {
            const int t = base + int(lane);
            const bool valid = t < nL;
            int x = 0, y = 0;
            bool c = false;
            if (valid) {
                x = int(Lpos[t]);
                y = t < nR ? int(Rpos[nR - 1 - t]) : lo - 1;
                c = x < y;
            }
            E ex = E(), ey = E();
            if (c) {
                ex = a[x];
                ey = a[y];
            }
            waveSyncFor<GLOBAL>();
            if (c) {
                a[x] = ey;
                a[y] = ex;
            }
            const uint64_t mc = __builtin_amdgcn_ballot_w64(c);
            const uint64_t mv = __builtin_amdgcn_ballot_w64(valid);
            T += __builtin_popcountll(mc);
            if (mc != mv) {
                // The pairs stop crossing inside this chunk, at the lane f = T - base (c is a prefix of the valid lanes: L
                // ascends, R descends).  Lane f holds L[T]; R[nR - T] is what lane f - 1 holds as its y -- or, for f == 0, the
                // entry behind R[nR - 1 - T] = lane 0's y, which only the position array has.  Both from registers when f > 0.
                const int f = T - base;
                if (!GLOBAL && f > 0 && T < nR) {
                    const int lT = __builtin_amdgcn_readlane(x, f);
                    const int rT = __builtin_amdgcn_readlane(y, f - 1);
                    waveSyncFor<GLOBAL>();
                    return lT < rT ? lT : rT;           // (T > 0 and T < nL here: the rule below with both operands at hand)
                }
                break;
            }
        }
    }
    waveSyncFor<GLOBAL>();
    int cut;
    if (T < nL && (T == 0 || int(Lpos[T]) < int(Rpos[nR - T]))) cut = int(Lpos[T]);
    else cut = int(Rpos[nR - T]);
    return __builtin_amdgcn_readfirstlane(cut);
}

__device__ inline int partitionWave(Entry* a, int lo, int hi, uint16_t* Lpos, uint16_t* Rpos, uint32_t lane)
{
    return partitionWaveT<uint16_t, false, Entry, 1>(a, lo, hi, Lpos, Rpos, lane);
}

// std::nth_element(a, a+nth, a+n, cmp) by one wave; a, Lpos, Rpos in LDS (Lpos/Rpos: n uint16 each, n <= 65535), or all
// three in global memory (GLOBAL, Index = uint32_t).
template <class Index, bool GLOBAL, class E = Entry, int CHUNKS = 1>
__device__ inline void nthElementWaveT(E* a, Index* Lpos, Index* Rpos, int nth, int n, uint32_t lane)
{
    if (n == 0 || nth == n) return;
    int first = 0, last = n;
    int depthLimit = 2 * floorLog2(uint32_t(n));
    while (last - first > 3) {
        if (depthLimit == 0) {
            if (lane == 0u) {
                heapSelect(a, first, nth + 1, last);
                entrySwap(a, first, nth);
            }
            waveSyncFor<GLOBAL>();
            return;
        }
        --depthLimit;
        uint32_t pivotKey = 0u;
        if (lane == 0u) {
            if (!GLOBAL) pivotKey = medianToFirstLoaded(a, first, first + 1, first + (last - first) / 2, last - 1);
            else medianToFirst(a, first, first + 1, first + (last - first) / 2, last - 1);
        }
        if (!GLOBAL) pivotKey = uint32_t(__builtin_amdgcn_readfirstlane(int(pivotKey)));
        waveSyncFor<GLOBAL>();
        const int cut = partitionWaveT<Index, GLOBAL, E, CHUNKS>(a, first + 1, last, Lpos, Rpos, lane, !GLOBAL, pivotKey);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    if (lane == 0u) insertionSort(a, first, last);
    waveSyncFor<GLOBAL>();
}

__device__ inline void nthElementWave(Entry* a, uint16_t* Lpos, uint16_t* Rpos, int nth, int n, uint32_t lane)
{
    nthElementWaveT<uint16_t, false>(a, Lpos, Rpos, nth, n, lane);
}

}  // namespace em2

#endif
