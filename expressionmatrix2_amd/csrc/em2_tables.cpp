#include "em2_tables.h"

#include <cmath>

namespace em2 {

void computeSimilarityTable(uint32_t lshCount, double* table)
{
    // boost::math::double_constants::pi (Lsh.cpp:241) is the double closest to pi.
    const double pi = 3.141592653589793238462643383279502884;
    for (uint32_t m = 0; m <= lshCount; m++) {
        const double angle = double(m) * pi / double(lshCount);
        table[m] = std::cos(angle);
    }
}

namespace {

// max m with table[m] > x, or -1.  table is non-increasing.
int32_t acceptLimit(const std::vector<double>& table, double x)
{
    int32_t lo = -1, hi = int32_t(table.size());     // table[lo] > x (or lo == -1), !(table[hi] > x) (or hi == size)
    while (hi - lo > 1) {
        const int32_t mid = lo + (hi - lo) / 2;
        if (table[size_t(mid)] > x) lo = mid;
        else hi = mid;
    }
    return lo;
}

}  // namespace

bool buildSimilarityTables(uint32_t lshCount, double similarityThreshold, SimilarityTables& out, const char** error)
{
    out.lshCount = lshCount;
    out.similarity.resize(size_t(lshCount) + 1);
    computeSimilarityTable(lshCount, out.similarity.data());
    for (size_t m = 1; m <= lshCount; m++) {
        if (out.similarity[m] > out.similarity[m - 1]) {
            if (error) *error = "similarity table is not monotone non-increasing";
            return false;
        }
    }

    out.keyOfMismatch.resize(size_t(lshCount) + 1);
    out.keySimilarity.clear();
    for (size_t m = 0; m <= lshCount; m++) {
        const float f = float(out.similarity[m]);
        if (m == 0 || f != out.keySimilarity.back()) out.keySimilarity.push_back(f);
        out.keyOfMismatch[m] = uint32_t(out.keySimilarity.size() - 1);
    }

    out.mGlobal = acceptLimit(out.similarity, similarityThreshold);
    const float initialCellThreshold = float(similarityThreshold);
    const int32_t initial = acceptLimit(out.similarity, double(initialCellThreshold));
    out.mMaxInitial = initial < out.mGlobal ? initial : out.mGlobal;

    out.acceptMaxByKey.resize(out.keySimilarity.size());
    for (size_t q = 0; q < out.keySimilarity.size(); q++) {
        const int32_t a = acceptLimit(out.similarity, double(out.keySimilarity[q]));
        out.acceptMaxByKey[q] = a < out.mGlobal ? a : out.mGlobal;
    }
    return true;
}

}  // namespace em2
