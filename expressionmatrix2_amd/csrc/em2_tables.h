// em2_tables.h -- host-side integer tables that carry the reference's floating-point acceptance rules
// onto the device without any device-side floating point.
//
// Reference rules being encoded (all in src/ExpressionMatrixLsh.cpp and src/Lsh.cpp):
//   similarityTable[m] = std::cos(double(m) * pi / double(lshCount))            Lsh.cpp:229-249
//   findSimilarPairs4 accepts a candidate with mismatch count m for a cell iff
//        similarityTable[m] > similarityThreshold                (double > double)   :244
//     && similarityTable[m] > cellThreshold[cell]                (double > float)    :245,252
//   where cellThreshold starts as float(similarityThreshold) (:207) and, after each keepBest, becomes the
//   float similarity stored in tmp.back() (:247-250).
//   Stored similarities are float(similarityTable[m]) (pair<CellId,float>, :246).
//   findSimilarPairs5 accepts iff similarityTable[m] > similarityThreshold (:441).
//
// similarityTable is non-increasing in m, so every rule above is "m <= some bound".  We precompute
//   key[m]            rank of float(similarityTable[m]) among the distinct float values (0 = largest);
//                     the reference comparator  x.second > y.second  is  key_x < key_y;
//   keySimilarity[q]  the float value of rank q (what is written to SimilarPairs);
//   mGlobal           max m with similarityTable[m] > similarityThreshold, or -1;
//   mMaxInitial       max m accepted while cellThreshold == float(similarityThreshold);
//   acceptMaxByKey[q] max m accepted once cellThreshold == keySimilarity[q].

#ifndef EM2_TABLES_H
#define EM2_TABLES_H

#include <stdint.h>
#include <vector>

namespace em2 {

struct SimilarityTables {
    uint32_t lshCount;
    std::vector<double> similarity;         // [lshCount+1]
    std::vector<uint32_t> keyOfMismatch;    // [lshCount+1]
    std::vector<float> keySimilarity;       // [keyCount]
    std::vector<int32_t> acceptMaxByKey;    // [keyCount]
    int32_t mGlobal;
    int32_t mMaxInitial;
};

// Fills `out`; returns false (and sets *error) if the libm table is not monotone, which would break the
// integer formulation (never observed; checked rather than assumed).
bool buildSimilarityTables(uint32_t lshCount, double similarityThreshold, SimilarityTables& out, const char** error);

// Lsh::computeSimilarityTable (Lsh.cpp:229-249).
void computeSimilarityTable(uint32_t lshCount, double* table);

}  // namespace em2

#endif
