// em2_host.h -- host side of the boundary: the reference's memory-mapped file formats and the
// ExpressionMatrix-level drivers of the LSH path (name lookup, subset construction, result files).
//
// File formats (all little-endian, written by mmap in the reference):
//   MemoryMapped::Vector<T>   src/MemoryMappedVector.hpp:141-197   256-byte header {headerSize, objectSize,
//                             objectCount, pageCount, fileSize, capacity, magic 0xa3756fd4b5d8bcc1, 25 zero words}
//                             + objectCount objects; file size rounded up to 4096.
//   MemoryMapped::Object<T>   src/MemoryMappedObject.hpp:88-140    same header with magic 0xb7756f4515d8bc94,
//                             objectCount = capacity = 1.
//   VectorOfVectors<T,Int>    src/MemoryMappedVectorOfVectors.hpp:29-34  <name>.toc (Vector<Int>, n+1 offsets)
//                             + <name>.data (Vector<T>).
#ifndef EM2_HOST_H
#define EM2_HOST_H

#include <stdint.h>
#include <map>
#include <functional>
#include <string>
#include <vector>

#include "../../include/em2_lsh.h"

namespace em2 {
namespace host {

// Thrown for every condition the reference reports with std::runtime_error; .what() is the reference's text
// where the reference has one.
struct Error {
    int code;
    std::string message;
};

// A read-only or writable mapping of one MemoryMapped file.
class MappedFile {
public:
    MappedFile() : base_(nullptr), size_(0) {}
    ~MappedFile() { close(); }
    MappedFile(const MappedFile&) = delete;
    MappedFile& operator=(const MappedFile&) = delete;

    void openExisting(const std::string& path, bool isObject, size_t objectSize);
    void createNew(const std::string& path, bool isObject, size_t objectSize, size_t objectCount);
    void close();

    bool isOpen() const { return base_ != nullptr; }
    size_t objectCount() const;
    const void* data() const { return static_cast<const char*>(base_) + 256; }
    void* data() { return static_cast<char*>(base_) + 256; }

private:
    void* base_;
    size_t size_;
};

void removeFile(const std::string& path);
bool fileExists(const std::string& path);

// GeneSet (src/GeneSet.hpp:62-77): sorted global ids + table global id -> local id (0xffffffff = absent).
struct GeneSet {
    MappedFile globalIds;
    MappedFile localIds;
    uint32_t size() const { return uint32_t(globalIds.objectCount()); }
    const uint32_t* genes() const { return static_cast<const uint32_t*>(globalIds.data()); }
    uint32_t localId(uint32_t globalId) const
    {
        return globalId < localIds.objectCount() ? static_cast<const uint32_t*>(localIds.data())[globalId] : 0xffffffffu;
    }
};

// The part of ExpressionMatrix (src/ExpressionMatrix.hpp:80-1247) the LSH path touches.
class Matrix {
public:
    explicit Matrix(const std::string& directoryName);       // accessExisting, ExpressionMatrix.cpp:109-160
    ~Matrix();

    const std::string& directory() const { return directoryName_; }
    uint32_t cellCount() const { return uint32_t(toc_.objectCount() - 1); }

    // ExpressionMatrixSubset (src/ExpressionMatrixSubset.cpp:9-42): CSR restricted to the gene set and the
    // cell set, in local ids.  Throws the reference's "Gene set X does not exist." etc.
    void subset(const std::string& geneSetName, const std::string& cellSetName, std::vector<uint64_t>& toc,
                std::vector<em2_count>& data, uint32_t& geneCount, uint32_t& cellCount) const;

    // The lookups and checks of subset() without building it; runLshPath builds it on the device.
    void lookupSubset(const std::string& geneSetName, const std::string& cellSetName, const GeneSet*& genes,
                      const uint32_t*& cellIds, uint32_t& cellCount) const;
    // pairsFor(cellCount) returns where the pairs are to be written (used != nullptr asks for pairs at all).
    void runLshPath(const char* what, const std::string& geneSetName, const std::string& cellSetName, size_t lshCount,
                    unsigned int seed, uint32_t& cellCount, std::vector<uint64_t>* signatures, size_t k,
                    double similarityThreshold, const std::function<em2_pair*(uint32_t)>& pairsFor,
                    std::vector<uint32_t>* used) const;

    void findSimilarPairs4(const std::string& geneSetName, const std::string& cellSetName,
                           const std::string& similarPairsName, size_t k, double similarityThreshold,
                           size_t lshCount, unsigned int seed) const;
    void computeLshSignatures(const std::string& geneSetName, const std::string& cellSetName,
                              const std::string& lshName, size_t lshCount, unsigned int seed) const;
    void findSimilarPairs5(const std::string& geneSetName, const std::string& cellSetName,
                           const std::string& lshName, const std::string& similarPairsName, size_t k,
                           double similarityThreshold, size_t lshSliceLength, size_t bucketOverflow) const;
    void findSimilarPairs7(const std::string& geneSetName, const std::string& cellSetName, const std::string& lshName,
                           const std::string& similarPairsName, size_t k, double similarityThreshold,
                           const std::vector<int32_t>& lshSliceLengths, uint32_t maxCheck, size_t log2BucketCount) const;
    void removeSimilarPairs(const std::string& similarPairsName) const;
    // ExpressionMatrix::analyzeLsh (src/ExpressionMatrixLsh.cpp:1244-1367): writes Lsh-analysis.csv and
    // LSH-analysis-statistics.csv into outputDirectory (the reference: the working directory, "").
    void analyzeLsh(const std::string& geneSetName, const std::string& cellSetName, size_t lshCount, unsigned int seed,
                    double csvDownsample, const std::string& outputDirectory) const;

    const GeneSet& geneSet(const std::string& name) const;                 // throws "Gene set X does not exist."
    const MappedFile& cellSet(const std::string& name) const;              // throws "Cell set X does not exist."

private:
    std::string directoryName_;
    MappedFile toc_;         // CellExpressionCounts.toc  (uint64)
    MappedFile data_;        // CellExpressionCounts.data (em2_count)
    std::map<std::string, GeneSet*> geneSets_;
    std::map<std::string, MappedFile*> cellSets_;
};

// SimilarPairs files (src/SimilarPairs.cpp:11-42 create, :369-379 copy): -Info, -Pairs, -CellInfo.
// SimilarPairs(directory, name, geneSetName, cellSetName, k) for writing (src/SimilarPairs.cpp:11-42): creates the three
// files; the pairs are written in place (k per cell, the caller's order is final), finish() fills CellInfo.
class SimilarPairsWriter {
public:
    SimilarPairsWriter(const std::string& directoryName, const std::string& similarPairsName, const std::string& geneSetName,
                       const std::string& cellSetName, size_t k, uint32_t cellCount);
    ~SimilarPairsWriter();                    // not finished: the temporary files go away, an existing object stays
    em2_pair* pairs();
    void finish(const uint32_t* usedCount);   // fills CellInfo, then renames the three files into place
private:
    void removeStale() const;
    MappedFile infoFile_, pairsFile_, cellInfoFile_;
    uint32_t cellCount_;
    std::string base_;
    bool finished_;
};

void writeSimilarPairs(const std::string& directoryName, const std::string& similarPairsName,
                       const std::string& geneSetName, const std::string& cellSetName, size_t k,
                       uint32_t cellCount, const em2_pair* pairs, const uint32_t* usedCount);

struct SimilarPairsInfo {
    uint64_t k;
    std::string geneSetName;
    uint64_t geneSetHash;
    std::string cellSetName;
    uint64_t cellSetHash;
    uint64_t cellCount;
};
// Existing-object constructor of SimilarPairs (src/SimilarPairs.cpp:47-83) including its consistency checks.
void readSimilarPairs(const std::string& directoryName, const std::string& similarPairsName,
                      SimilarPairsInfo& info, std::vector<em2_pair>* pairs, std::vector<uint32_t>* usedCount);

// Lsh files (src/Lsh.hpp:136-141, src/Lsh.cpp:26-28,148): <prefix>-Info, <prefix>-Signatures.
void writeLsh(const std::string& prefix, uint64_t cellCount, uint64_t lshCount, const uint64_t* signatures);
void readLshInfo(const std::string& prefix, uint64_t& cellCount, uint64_t& lshCount);
void readLsh(const std::string& prefix, uint64_t& cellCount, uint64_t& lshCount, std::vector<uint64_t>& signatures);

// Test / bench tooling (NOT a reference API): a directory holding exactly the files the LSH path reads.
void createDirectoryFromCsr(const std::string& directoryName, uint32_t geneCount, uint32_t cellCount,
                            const uint64_t* toc, const em2_count* data);
void addGeneSet(const std::string& directoryName, const std::string& name, const uint32_t* sortedGlobalIds,
                uint32_t count, uint32_t totalGeneCount);
void addCellSet(const std::string& directoryName, const std::string& name, const uint32_t* sortedCellIds,
                uint32_t count);

}  // namespace host
}  // namespace em2

#endif
