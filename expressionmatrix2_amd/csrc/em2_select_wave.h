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

__device__ __forceinline__ uint32_t lanesBelow(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
}

// __unguarded_partition(a+lo, a+hi, pivot = a[lo-1]); returns the cut (wave-uniform).
__device__ inline int partitionWave(Entry* a, int lo, int hi, uint16_t* Lpos, uint16_t* Rpos, uint32_t lane)
{
    const uint32_t pk = a[lo - 1].key;
    int nL = 0, nR = 0;
    for (int base = lo; base < hi; base += 64) {
        const int x = base + int(lane);
        const bool valid = x < hi;
        const uint32_t key = valid ? a[x].key : 0u;
        const bool isL = valid && key >= pk;
        const bool isR = valid && key <= pk;
        const uint64_t mL = __builtin_amdgcn_ballot_w64(isL);
        const uint64_t mR = __builtin_amdgcn_ballot_w64(isR);
        if (isL) Lpos[nL + int(lanesBelow(mL))] = uint16_t(x);
        if (isR) Rpos[nR + int(lanesBelow(mR))] = uint16_t(x);
        nL += __builtin_popcountll(mL);
        nR += __builtin_popcountll(mR);
    }
    waveSync();
    int T = 0;
    for (int base = 0; base < nL; base += 64) {
        const int t = base + int(lane);
        const bool valid = t < nL;
        int x = 0, y = 0;
        bool c = false;
        if (valid) {
            x = Lpos[t];
            y = t < nR ? int(Rpos[nR - 1 - t]) : lo - 1;
            c = x < y;
        }
        Entry ex, ey;
        ex.cell = ex.key = ey.cell = ey.key = 0u;
        if (c) {
            ex = a[x];
            ey = a[y];
        }
        waveSync();
        if (c) {
            a[x] = ey;
            a[y] = ex;
        }
        const uint64_t mc = __builtin_amdgcn_ballot_w64(c);
        T += __builtin_popcountll(mc);
        if (mc != __builtin_amdgcn_ballot_w64(valid)) break;
    }
    waveSync();
    int cut;
    if (T < nL && (T == 0 || int(Lpos[T]) < int(Rpos[nR - T]))) cut = Lpos[T];
    else cut = Rpos[nR - T];
    return __builtin_amdgcn_readfirstlane(cut);
}

// std::nth_element(a, a+nth, a+n, cmp) by one wave; a, Lpos, Rpos in LDS (Lpos/Rpos: n uint16 each, n <= 65535).
__device__ inline void nthElementWave(Entry* a, uint16_t* Lpos, uint16_t* Rpos, int nth, int n, uint32_t lane)
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
            waveSync();
            return;
        }
        --depthLimit;
        if (lane == 0u) medianToFirst(a, first, first + 1, first + (last - first) / 2, last - 1);
        waveSync();
        const int cut = partitionWave(a, first + 1, last, Lpos, Rpos, lane);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    if (lane == 0u) insertionSort(a, first, last);
    waveSync();
}

}  // namespace em2

#endif
