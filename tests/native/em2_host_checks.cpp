// em2_host_checks.cpp -- TEST-ONLY host build of the product's host/device-shared headers.
//
// Compiles expressionmatrix2_amd/csrc/em2_select.h and em2_tables.cpp for the host so that the exact
// selection emulation and the integer acceptance tables can be checked on a machine without a GPU:
//   * em2t_nth_element        the product's introselect restatement on (cell,key) entries
//   * em2t_std_introselect    libstdc++'s own std::__introselect on the same data with a caller-chosen
//                             depth limit (forces the heap-select fallback that random data never reaches)
//   * em2t_fsp4_rows          the per-row state machine of em2_scan.hip replayed on the host with the same
//                             tables and the same selection code, to be compared with the oracle
// Nothing here is shipped or used by the product path.

#include "../../expressionmatrix2_amd/csrc/em2_select.h"
#include "../../expressionmatrix2_amd/csrc/em2_tables.h"

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

extern "C" {

void em2t_nth_element(uint32_t* cell, uint32_t* key, uint32_t n, uint32_t nth, int depthLimit)
{
    std::vector<em2::Entry> a(n);
    for (uint32_t i = 0; i < n; i++) { a[i].cell = cell[i]; a[i].key = key[i]; }
    em2::nthElement(a.data(), int(nth), int(n), depthLimit);
    for (uint32_t i = 0; i < n; i++) { cell[i] = a[i].cell; key[i] = a[i].key; }
}

struct KeyLess {
    bool operator()(const std::pair<uint32_t, uint32_t>& x, const std::pair<uint32_t, uint32_t>& y) const
    {
        return x.second < y.second;
    }
};

void em2t_std_introselect(uint32_t* cell, uint32_t* key, uint32_t n, uint32_t nth, int depthLimit)
{
    std::vector< std::pair<uint32_t, uint32_t> > a(n);
    for (uint32_t i = 0; i < n; i++) a[i] = std::make_pair(cell[i], key[i]);
    if (n != 0 && nth != n) {
        if (depthLimit < 0) {
            std::nth_element(a.begin(), a.begin() + nth, a.end(), KeyLess());
        } else {
            std::__introselect(a.begin(), a.begin() + nth, a.end(), long(depthLimit),
                               __gnu_cxx::__ops::__iter_comp_iter(KeyLess()));
        }
    }
    for (uint32_t i = 0; i < n; i++) { cell[i] = a[i].first; key[i] = a[i].second; }
}

// Tables: returns keyCount; arrays sized lshCount+1 by the caller.
uint32_t em2t_tables(uint32_t lshCount, double threshold, double* similarity, uint32_t* keyOfMismatch,
                     float* keySimilarity, int32_t* acceptMaxByKey, int32_t* mGlobal, int32_t* mMaxInitial)
{
    em2::SimilarityTables t;
    const char* error = nullptr;
    if (!em2::buildSimilarityTables(lshCount, threshold, t, &error)) return 0;
    std::memcpy(similarity, t.similarity.data(), t.similarity.size() * sizeof(double));
    std::memcpy(keyOfMismatch, t.keyOfMismatch.data(), t.keyOfMismatch.size() * sizeof(uint32_t));
    std::memcpy(keySimilarity, t.keySimilarity.data(), t.keySimilarity.size() * sizeof(float));
    std::memcpy(acceptMaxByKey, t.acceptMaxByKey.data(), t.acceptMaxByKey.size() * sizeof(int32_t));
    *mGlobal = t.mGlobal;
    *mMaxInitial = t.mMaxInitial;
    return uint32_t(t.keySimilarity.size());
}

// Host replay of fsp4ScanKernel's per-row logic (em2_scan.hip) for rows [rowBegin,rowEnd).
int em2t_fsp4_rows(const uint64_t* signatures, uint32_t cellCount, uint32_t lshCount, uint32_t k,
                   double threshold, uint32_t rowBegin, uint32_t rowEnd,
                   uint32_t* outCell, float* outSim, uint32_t* outUsed)
{
    em2::SimilarityTables t;
    const char* error = nullptr;
    if (!em2::buildSimilarityTables(lshCount, threshold, t, &error)) return 1;
    const size_t W = (size_t(lshCount) - 1) / 64 + 1;
    std::vector<em2::Entry> list(2 * size_t(k));
    for (uint32_t row = rowBegin; row < rowEnd; row++) {
        int32_t mMax = t.mMaxInitial;
        uint32_t count = 0;
        const uint64_t* r = signatures + size_t(row) * W;
        for (uint32_t col = 0; col < cellCount && k > 0; col++) {
            const uint64_t* c = signatures + size_t(col) * W;
            int32_t m = 0;
            for (size_t w = 0; w < W; w++) m += __builtin_popcountll(r[w] ^ c[w]);
            if (m <= mMax && col != row) {
                list[count].cell = col;
                list[count].key = t.keyOfMismatch[size_t(m)];
                ++count;
                if (count == 2 * k) {
                    em2::nthElement(list.data(), int(k), int(count));
                    count = k;
                    mMax = t.acceptMaxByKey[list[k - 1].key];
                }
            }
        }
        if (count > k) {
            em2::nthElement(list.data(), int(k), int(count));
            count = k;
        }
        std::sort(list.begin(), list.begin() + count, [](const em2::Entry& x, const em2::Entry& y) {
            return x.key < y.key || (x.key == y.key && x.cell < y.cell);
        });
        const size_t base = size_t(row - rowBegin) * k;
        for (uint32_t j = 0; j < k; j++) {
            outCell[base + j] = j < count ? list[j].cell : 0u;
            outSim[base + j] = j < count ? t.keySimilarity[list[j].key] : 0.0f;
        }
        outUsed[row - rowBegin] = count;
    }
    return 0;
}

}  // extern "C"


// ---------------------------------------------------------------------------------------------------------
// Host MODEL of the wave-parallel selection used on the device (csrc/em2_select_wave.h): the same algorithm with
// the 64 lanes emulated by loops.  It validates the formulation (which elements a Hoare partition swaps and where
// it cuts can be computed from the ORIGINAL positions of the "stopper" elements) against std::nth_element.
// ---------------------------------------------------------------------------------------------------------
namespace wave_model {

using em2::Entry;

// Exact result of __unguarded_partition(a+lo, a+hi, pivot = a[lo-1]) computed "in parallel".
int partition(std::vector<Entry>& a, int lo, int hi, std::vector<uint16_t>& Lpos, std::vector<uint16_t>& Rpos)
{
    const uint32_t pk = a[lo - 1].key;
    int nL = 0, nR = 0;
    for (int base = lo; base < hi; base += 64) {                 // pass 1: stoppers in ascending position order
        for (int lane = 0; lane < 64; lane++) {
            const int x = base + lane;
            if (x >= hi) break;
            if (a[x].key >= pk) Lpos[nL++] = uint16_t(x);        // left scan stops here:  !(a[x] < pivot)
            if (a[x].key <= pk) Rpos[nR++] = uint16_t(x);        // right scan stops here: !(pivot < a[x])
        }
    }
    int T = 0;
    for (int base = 0; base < nL; base += 64) {                  // pass 2: swap pair t = (L[t], R[nR-1-t]) while L < R
        bool allTrue = true;
        Entry ex[64], ey[64];
        int xs[64], ys[64];
        bool cond[64];
        for (int lane = 0; lane < 64; lane++) {
            const int t = base + lane;
            cond[lane] = false;
            if (t >= nL) continue;
            xs[lane] = Lpos[t];
            ys[lane] = t < nR ? int(Rpos[nR - 1 - t]) : lo - 1;
            cond[lane] = xs[lane] < ys[lane];
            if (cond[lane]) { ex[lane] = a[xs[lane]]; ey[lane] = a[ys[lane]]; }
            else allTrue = false;
        }
        for (int lane = 0; lane < 64; lane++) {
            if (cond[lane]) { a[xs[lane]] = ey[lane]; a[ys[lane]] = ex[lane]; ++T; }
        }
        if (!allTrue) break;
    }
    if (T < nL && (T == 0 || int(Lpos[T]) < int(Rpos[nR - T]))) return Lpos[T];
    return Rpos[nR - T];                                          // = Rdesc[T-1]
}

void nthElement(std::vector<Entry>& a, int nth, int depthLimit)
{
    const int n = int(a.size());
    if (n == 0 || nth == n) return;
    std::vector<uint16_t> Lpos(n), Rpos(n);
    int first = 0, last = n;
    if (depthLimit < 0) depthLimit = 2 * em2::floorLog2(uint32_t(n));
    while (last - first > 3) {
        if (depthLimit == 0) {
            em2::heapSelect(a.data(), first, nth + 1, last);
            em2::entrySwap(a.data(), first, nth);
            return;
        }
        --depthLimit;
        const int mid = first + (last - first) / 2;
        em2::medianToFirst(a.data(), first, first + 1, mid, last - 1);
        const int cut = partition(a, first + 1, last, Lpos, Rpos);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    em2::insertionSort(a.data(), first, last);
}

}  // namespace wave_model

extern "C" void em2t_wave_model_nth_element(uint32_t* cell, uint32_t* key, uint32_t n, uint32_t nth, int depthLimit)
{
    std::vector<em2::Entry> a(n);
    for (uint32_t i = 0; i < n; i++) { a[i].cell = cell[i]; a[i].key = key[i]; }
    wave_model::nthElement(a, int(nth), depthLimit);
    for (uint32_t i = 0; i < n; i++) { cell[i] = a[i].cell; key[i] = a[i].key; }
}
