// em2_select.h -- exact emulation of the selection the reference applies to a cell's candidate list.
//
// The reference keeps the best k candidates of a cell with
//     keepBest(v, k, OrderPairsBySecondGreater<pair<CellId,float>>())          (src/heap.hpp:116-126)
// i.e. std::nth_element(v.begin(), v.begin()+k, v.end(), cmp); v.resize(k);
// and findSimilarPairs4 then uses v.back().second as the cell's new cut-off
// (src/ExpressionMatrixLsh.cpp:247-250).  Similarities are a function of an integer mismatch count, so ties
// are the rule, and WHICH tied elements survive and WHICH element ends at position k-1 is decided by the
// internals of libstdc++'s introselect.  Bit-exact SimilarPairs therefore need the same sequence of
// element moves.  This header restates that algorithm (GCC libstdc++ bits/stl_algo.h __introselect,
// __unguarded_partition_pivot, __move_median_to_first, __heap_select, __insertion_sort and
// bits/stl_heap.h __adjust_heap/__push_heap/__make_heap/__pop_heap) for one fixed element type and
// comparator, in a form that compiles for the host and for gfx950.
//
// Element: {cell, key}.  key is the rank of the float similarity (smaller key == larger similarity, equal
// key == bit-identical float), so the reference comparator  x.second > y.second  is  x.key < y.key.

#ifndef EM2_SELECT_H
#define EM2_SELECT_H

#include <stdint.h>

#if defined(__HIPCC__)
#define EM2_HD __host__ __device__ __forceinline__
#else
#define EM2_HD inline
#endif

namespace em2 {

struct Entry {
    uint32_t cell;
    uint32_t key;
};

// The functions below are templates over the element type E: anything with a `key` member that orders it (Entry here and in
// the scan kernels; the 4-byte {key, index} element findSimilarPairs5 stages in LDS).
template <class E> EM2_HD bool entryBefore(const E& x, const E& y) { return x.key < y.key; }

template <class E> EM2_HD void entrySwap(E* a, int i, int j)
{
    const E t = a[i];
    a[i] = a[j];
    a[j] = t;
}

// floor(log2(n)) for n >= 1   (std::__lg)
EM2_HD int floorLog2(uint32_t n)
{
    int r = 0;
    while (n >>= 1) ++r;
    return r;
}

// Swap the median of a[ia], a[ib], a[ic] into a[result].
template <class E> EM2_HD void medianToFirst(E* a, int result, int ia, int ib, int ic)
{
    if (entryBefore(a[ia], a[ib])) {
        if (entryBefore(a[ib], a[ic])) entrySwap(a, result, ib);
        else if (entryBefore(a[ia], a[ic])) entrySwap(a, result, ic);
        else entrySwap(a, result, ia);
    }
    else if (entryBefore(a[ia], a[ic])) entrySwap(a, result, ia);
    else if (entryBefore(a[ib], a[ic])) entrySwap(a, result, ic);
    else entrySwap(a, result, ib);
}

// Hoare partition of [first,last) around the value at pivot (which lies outside the range).
template <class E> EM2_HD int unguardedPartition(E* a, int first, int last, int pivot)
{
    const E p = a[pivot];
    for (;;) {
        while (entryBefore(a[first], p)) ++first;
        --last;
        while (entryBefore(p, a[last])) --last;
        if (!(first < last)) return first;
        entrySwap(a, first, last);
        ++first;
    }
}

template <class E> EM2_HD int unguardedPartitionPivot(E* a, int first, int last)
{
    const int mid = first + (last - first) / 2;
    medianToFirst(a, first, first + 1, mid, last - 1);
    return unguardedPartition(a, first + 1, last, first);
}

// Sift `value` up from hole towards top in the heap a[base ...).
template <class E> EM2_HD void pushHeap(E* a, int base, int hole, int top, E value)
{
    int parent = (hole - 1) / 2;
    while (hole > top && entryBefore(a[base + parent], value)) {
        a[base + hole] = a[base + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a[base + hole] = value;
}

template <class E> EM2_HD void adjustHeap(E* a, int base, int hole, int len, E value)
{
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (entryBefore(a[base + child], a[base + child - 1])) child--;
        a[base + hole] = a[base + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a[base + hole] = a[base + child - 1];
        hole = child - 1;
    }
    pushHeap(a, base, hole, top, value);
}

// Heap of the best (middle-first) elements of [first,last), worst of them at a[first].
template <class E> EM2_HD void heapSelect(E* a, int first, int middle, int last)
{
    const int len = middle - first;
    if (len >= 2) {
        int parent = (len - 2) / 2;
        for (;;) {
            const E value = a[first + parent];
            adjustHeap(a, first, parent, len, value);
            if (parent == 0) break;
            parent--;
        }
    }
    for (int i = middle; i < last; ++i) {
        if (entryBefore(a[i], a[first])) {
            const E value = a[i];
            a[i] = a[first];
            adjustHeap(a, first, 0, len, value);
        }
    }
}

template <class E> EM2_HD void insertionSort(E* a, int first, int last)
{
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        const E value = a[i];
        if (entryBefore(value, a[first])) {
            for (int j = i; j > first; --j) a[j] = a[j - 1];
            a[first] = value;
        } else {
            int j = i;
            while (entryBefore(value, a[j - 1])) {
                a[j] = a[j - 1];
                --j;
            }
            a[j] = value;
        }
    }
}

// std::nth_element(a+0, a+nth, a+n, cmp).  depthLimit < 0 selects the library's own 2*floor(log2(n)).
template <class E> EM2_HD void nthElement(E* a, int nth, int n, int depthLimit = -1)
{
    if (n == 0 || nth == n) return;
    int first = 0, last = n;
    if (depthLimit < 0) depthLimit = 2 * floorLog2(uint32_t(n));
    while (last - first > 3) {
        if (depthLimit == 0) {
            heapSelect(a, first, nth + 1, last);
            entrySwap(a, first, nth);
            return;
        }
        --depthLimit;
        const int cut = unguardedPartitionPivot(a, first, last);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    insertionSort(a, first, last);
}

// keepBest (src/heap.hpp:116-126): returns the new element count.
template <class E> EM2_HD int keepBest(E* a, int n, int k)
{
    if (n > k) {
        nthElement(a, k, n);
        return k;
    }
    return n;
}

}  // namespace em2

#endif
