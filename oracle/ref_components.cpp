// ref_components.cpp -- driver that exposes, through a C ABI, the parts of the reference's own
// implementation of the LSH path that compile in this image WITHOUT any stand-in header:
//
//   /root/reference/src/heap.hpp        keepBest (heap.hpp:116-126)
//   /root/reference/src/orderPairs.hpp  OrderPairsBySecondGreater (:56-62),
//                                       OrderPairsBySecondGreaterThenByFirstLess (:44-52)
//   /root/reference/src/MurmurHash2.cpp MurmurHash64A (:96-137)
//
// They are included / compiled where they lie (see oracle/Makefile: -I /root/reference/src);
// nothing from the reference is copied into this repository.  The result, oracle/_ref/libem2ref.so,
// exists only in containers that have /root/reference; it is git-ignored, travels with gpurun, and is
// used by tests/ to validate oracle/em2_oracle.cpp.  Every other translation unit on the path
// (Lsh.cpp, ExpressionMatrixLsh.cpp, SimilarPairs.cpp, BitSet.hpp, multipleSetUnion.hpp ...) includes
// Boost headers, which this image does not have: they are unbuildable here ("Oracle").
//
// TEST INFRASTRUCTURE ONLY -- same rule as em2_oracle.cpp.

#include "heap.hpp"
#include "orderPairs.hpp"
#include "MurmurHash2.hpp"

#include <algorithm>
#include <cstdint>
#include <functional>
#include <utility>
#include <vector>

using namespace ChanZuckerberg::ExpressionMatrix2;

typedef std::pair<uint32_t, float> Pair;

extern "C" {

// The reference's keepBest with the comparator findSimilarPairs4/5 pass to it
// (ExpressionMatrixLsh.cpp:247,254,267,457).
uint32_t em2ref_keep_best(uint32_t* cell, float* sim, uint32_t n, uint32_t k)
{
    std::vector<Pair> v(n);
    for (uint32_t i = 0; i < n; i++) v[i] = std::make_pair(cell[i], sim[i]);
    keepBest(v, size_t(k), OrderPairsBySecondGreater<Pair>());
    for (size_t i = 0; i < v.size(); i++) { cell[i] = v[i].first; sim[i] = v[i].second; }
    return uint32_t(v.size());
}

// keepBest on ints with std::greater<int>, the call made by testKeepBest (heap.cpp:32-38).
uint32_t em2ref_keep_best_int_greater(int* values, uint32_t n, uint32_t k)
{
    std::vector<int> v(values, values + n);
    keepBest(v, size_t(k), std::greater<int>());
    std::copy(v.begin(), v.end(), values);
    return uint32_t(v.size());
}

// The sort SimilarPairs::sort applies to one cell's pairs (SimilarPairs.cpp:399-405).
void em2ref_sort_pairs(uint32_t* cell, float* sim, uint32_t n)
{
    std::vector<Pair> v(n);
    for (uint32_t i = 0; i < n; i++) v[i] = std::make_pair(cell[i], sim[i]);
    std::sort(v.begin(), v.end(), OrderPairsBySecondGreaterThenByFirstLess<Pair>());
    for (size_t i = 0; i < v.size(); i++) { cell[i] = v[i].first; sim[i] = v[i].second; }
}

uint64_t em2ref_murmur_hash_64a(const void* key, int len, uint64_t seed)
{
    return MurmurHash64A(key, len, seed);
}

}  // extern "C"
