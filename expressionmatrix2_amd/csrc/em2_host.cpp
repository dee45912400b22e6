#include "em2_host.h"

// internal to the library (em2_capi.hip)
extern "C" void em2_internal_set_last_error(const char* message);
extern "C" int em2_internal_subset_find_similar_pairs4(const uint64_t* globalToc, const em2_count* globalData, uint32_t globalCellCount,
                                                       const uint32_t* cellIds, uint32_t cellCount, const uint32_t* geneLocalIds,
                                                       uint32_t globalGeneCount, uint32_t geneCount,
                                                       const double* (*vectorsWhenNeeded)(void*), void* vectorsContext, uint32_t lshCount,
                                                       uint64_t* signatures, uint32_t k, double similarityThreshold, em2_pair* pairs,
                                                       uint32_t* usedCount);

#include <chrono>
#include <functional>
#include <future>
#include <memory>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <cerrno>
#include <cfloat>
#include <cstring>
#include <dirent.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

namespace em2 {
namespace host {

// EM2_TIMING=1: wall time of the stages of the matrix-level calls on stderr (measurements only).
class StageTimer {
public:
    explicit StageTimer(const char* what) : what_(what), on_(getenv("EM2_TIMING") && getenv("EM2_TIMING")[0] == '1'),
                                            last_(std::chrono::steady_clock::now()) {}
    void stage(const char* name)
    {
        if (!on_) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[em2 timing] %s: %s %.1f ms\n", what_, name, std::chrono::duration<double, std::milli>(now - last_).count());
        last_ = now;
    }
private:
    const char* what_;
    bool on_;
    std::chrono::steady_clock::time_point last_;
};

namespace {

const uint64_t kVectorMagic = 0xa3756fd4b5d8bcc1ULL;     // src/MemoryMappedVector.hpp:167
const uint64_t kObjectMagic = 0xb7756f4515d8bc94ULL;     // src/MemoryMappedObject.hpp:113
const size_t kPageSize = 4096;                            // src/MemoryMappedVector.hpp:134
const uint32_t kInvalidId = 0xffffffffu;                  // src/Ids.hpp

struct FileHeader {                                       // src/MemoryMappedVector.hpp:141-172
    uint64_t headerSize;
    uint64_t objectSize;
    uint64_t objectCount;
    uint64_t pageCount;
    uint64_t fileSize;
    uint64_t capacity;
    uint64_t magicNumber;
    uint64_t padding[25];
};
static_assert(sizeof(FileHeader) == 256, "header is 256 bytes");

[[noreturn]] void fail(int code, const std::string& message)
{
    Error e;
    e.code = code;
    e.message = message;
    throw e;
}

// SimilarPairs::Info (src/SimilarPairs.hpp:188-198) with StaticString255 = {uint8 n; char s[255]}
// (src/ShortStaticString.hpp:27-42).
struct StaticString255 {
    uint8_t n;
    char s[255];
};
struct SimilarPairsInfoRecord {
    uint64_t k;
    StaticString255 geneSetName;
    uint64_t geneSetHash;
    StaticString255 cellSetName;
    uint64_t cellSetHash;
};
static_assert(sizeof(SimilarPairsInfoRecord) == 536, "SimilarPairs::Info layout");

struct CellInfoRecord {                                   // src/SimilarPairs.hpp:165-179
    uint32_t usedCount;
    uint32_t lowestSimilarityIndex;
    float lowestSimilarity;
};

struct LshInfoRecord {                                    // src/Lsh.hpp:136-141
    uint64_t cellCount;
    uint64_t lshCount;
};

void setStaticString(StaticString255& dst, const std::string& src)
{
    if (src.size() > 255) fail(EM2_ERROR_RUNTIME, "ShortStaticString capacity exceeded.");
    dst.n = uint8_t(src.size());
    std::memcpy(dst.s, src.data(), src.size());
}

std::string getStaticString(const StaticString255& src) { return std::string(src.s, src.s + src.n); }

uint64_t hashOf(const MappedFile& vectorFile, size_t objectSize)
{
    // MemoryMapped::Vector::hash (src/MemoryMappedVector.hpp:715-723)
    const uint64_t bytes = vectorFile.objectCount() * objectSize;
    if (bytes > 0x7fffffffULL) fail(EM2_ERROR_RUNTIME, "hash: vector too long");
    return em2_murmur_hash_64a(vectorFile.data(), int(bytes), 231);
}

bool isSorted(const uint32_t* p, size_t n) { return std::is_sorted(p, p + n); }

}  // namespace


// ---------------------------------------------------------------------------------------------------------
// MappedFile
// ---------------------------------------------------------------------------------------------------------

void MappedFile::openExisting(const std::string& path, bool isObject, size_t objectSize)
{
    close();
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd == -1) {
        fail(EM2_ERROR_IO, "Error accessing " + path + ": Error " + std::to_string(errno) + " opening " + path + ": " +
                               std::string(strerror(errno)));
    }
    struct stat st;
    if (::fstat(fd, &st) == -1) {
        ::close(fd);
        fail(EM2_ERROR_IO, "Error accessing " + path + ": Error during fstat.");
    }
    if (size_t(st.st_size) < sizeof(FileHeader)) {
        ::close(fd);
        fail(EM2_ERROR_IO, "Error accessing " + path + ": file is shorter than its header.");
    }
    void* p = ::mmap(nullptr, size_t(st.st_size), PROT_READ, MAP_SHARED, fd, 0);
    ::close(fd);
    if (p == MAP_FAILED) fail(EM2_ERROR_IO, "Error accessing " + path + ": Error during mmap.");
    base_ = p;
    size_ = size_t(st.st_size);
    const FileHeader* h = static_cast<const FileHeader*>(base_);
    // The checks of accessExisting (src/MemoryMappedVector.hpp:497-499).
    const bool ok = h->magicNumber == (isObject ? kObjectMagic : kVectorMagic) && h->fileSize == size_ &&
                    h->objectSize == objectSize && h->headerSize == sizeof(FileHeader) &&
                    sizeof(FileHeader) + h->objectCount * h->objectSize <= size_;
    if (!ok) {
        close();
        fail(EM2_ERROR_IO, "Error accessing " + path + ": header is not consistent with the file.");
    }
}

void MappedFile::createNew(const std::string& path, bool isObject, size_t objectSize, size_t objectCount)
{
    close();
    FileHeader h;
    std::memset(&h, 0, sizeof(h));
    h.headerSize = sizeof(FileHeader);
    h.objectSize = objectSize;
    h.objectCount = objectCount;
    h.pageCount = (sizeof(FileHeader) + objectSize * objectCount - 1) / kPageSize + 1;     // computePageCount
    h.fileSize = h.pageCount * kPageSize;
    h.capacity = isObject ? 1 : (h.fileSize - sizeof(FileHeader)) / objectSize;
    h.magicNumber = isObject ? kObjectMagic : kVectorMagic;
    // O_TRUNC: an existing object of the same name is silently replaced (src/MemoryMappedVector.hpp:351-354).
    const int fd = ::open(path.c_str(), O_CREAT | O_TRUNC | O_RDWR, S_IRUSR | S_IWUSR | S_IRGRP | S_IROTH);
    if (fd == -1) fail(EM2_ERROR_IO, "Error creating " + path);
    if (::ftruncate(fd, off_t(h.fileSize)) == -1) {
        ::close(fd);
        fail(EM2_ERROR_IO, "Error creating " + path);
    }
    void* p = ::mmap(nullptr, h.fileSize, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    ::close(fd);
    if (p == MAP_FAILED) fail(EM2_ERROR_IO, "Error creating " + path);
    base_ = p;
    size_ = h.fileSize;
    std::memcpy(base_, &h, sizeof(h));        // data area is already zero = value-initialised objects
}

void MappedFile::close()
{
    if (base_) {
        ::msync(base_, size_, MS_SYNC);       // syncToDisk on close (src/MemoryMappedVector.hpp:536-570)
        ::munmap(base_, size_);
        base_ = nullptr;
        size_ = 0;
    }
}

size_t MappedFile::objectCount() const
{
    return base_ ? size_t(static_cast<const FileHeader*>(base_)->objectCount) : 0;
}

void removeFile(const std::string& path) { ::unlink(path.c_str()); }

bool fileExists(const std::string& path)
{
    struct stat st;
    return ::stat(path.c_str(), &st) == 0;
}


// ---------------------------------------------------------------------------------------------------------
// Matrix
// ---------------------------------------------------------------------------------------------------------

namespace {
template <class Map> void deleteAll(Map& m)
{
    for (auto& p : m) delete p.second;
    m.clear();
}
}  // namespace

Matrix::Matrix(const std::string& directoryName) : directoryName_(directoryName)
{
  try {
    DIR* dir = ::opendir(directoryName.c_str());
    if (!dir) fail(EM2_ERROR_IO, "Directory " + directoryName + " does not exist or cannot be read.");
    std::vector<std::string> names;
    while (struct dirent* e = ::readdir(dir)) names.push_back(e->d_name);
    ::closedir(dir);

    toc_.openExisting(directoryName + "/CellExpressionCounts.toc", false, sizeof(uint64_t));       // ExpressionMatrix.cpp:129
    data_.openExisting(directoryName + "/CellExpressionCounts.data", false, sizeof(em2_count));
    if (toc_.objectCount() == 0) fail(EM2_ERROR_IO, "CellExpressionCounts.toc is empty.");

    const std::string cellSetPrefix = "CellSet-";                                                 // CellSets.cpp:27-49
    const std::string geneSetPrefix = "GeneSet-";                                                 // ExpressionMatrix.cpp:138-148
    const std::string geneSetSuffix = "-GlobalIds";
    for (const std::string& n : names) {
        if (n.compare(0, cellSetPrefix.size(), cellSetPrefix) == 0) {
            MappedFile* f = new MappedFile;
            cellSets_[n.substr(cellSetPrefix.size())] = f;
            f->openExisting(directoryName + "/" + n, false, sizeof(uint32_t));
        } else if (n.size() > geneSetPrefix.size() + geneSetSuffix.size() &&
                   n.compare(0, geneSetPrefix.size(), geneSetPrefix) == 0 &&
                   n.compare(n.size() - geneSetSuffix.size(), geneSetSuffix.size(), geneSetSuffix) == 0) {
            const std::string name = n.substr(geneSetPrefix.size(), n.size() - geneSetPrefix.size() - geneSetSuffix.size());
            GeneSet* g = new GeneSet;
            geneSets_[name] = g;
            g->globalIds.openExisting(directoryName + "/GeneSet-" + name + "-GlobalIds", false, sizeof(uint32_t));
            g->localIds.openExisting(directoryName + "/GeneSet-" + name + "-LocalIds", false, sizeof(uint32_t));
            if (!isSorted(g->genes(), g->size())) {
                fail(EM2_ERROR_RUNTIME, "Gene set " + directoryName + "/GeneSet-" + name + " is not sorted and accessed read-only.");
            }
        }
    }
    if (cellSets_.find("AllCells") == cellSets_.end()) fail(EM2_ERROR_RUNTIME, "Cell set \"AllCells\" is missing.");
    if (geneSets_.find("AllGenes") == geneSets_.end()) fail(EM2_ERROR_RUNTIME, "Gene set \"AllGenes\" is missing.");
  } catch (...) {
    deleteAll(geneSets_);
    deleteAll(cellSets_);
    throw;
  }
}

Matrix::~Matrix()
{
    deleteAll(geneSets_);
    deleteAll(cellSets_);
}

const GeneSet& Matrix::geneSet(const std::string& name) const
{
    const auto it = geneSets_.find(name);
    if (it == geneSets_.end()) fail(EM2_ERROR_RUNTIME, "Gene set " + name + " does not exist.");     // ExpressionMatrixLsh.cpp:170
    return *it->second;
}

const MappedFile& Matrix::cellSet(const std::string& name) const
{
    const auto it = cellSets_.find(name);
    if (it == cellSets_.end()) fail(EM2_ERROR_RUNTIME, "Cell set " + name + " does not exist.");     // ExpressionMatrixLsh.cpp:180
    return *it->second;
}

// Lookup + emptiness checks in the reference's order (ExpressionMatrixLsh.cpp:168-187).
void Matrix::lookupSubset(const std::string& geneSetName, const std::string& cellSetName, const GeneSet*& genes,
                          const uint32_t*& cellIds, uint32_t& cellCount) const
{
    const GeneSet& g = geneSet(geneSetName);
    if (g.size() == 0) fail(EM2_ERROR_RUNTIME, "Gene set " + geneSetName + " is empty.");
    const MappedFile& cells = cellSet(cellSetName);
    if (cells.objectCount() == 0) fail(EM2_ERROR_RUNTIME, "Cell set " + cellSetName + " is empty.");
    cellIds = static_cast<const uint32_t*>(cells.data());
    if (!isSorted(cellIds, cells.objectCount())) fail(EM2_ERROR_RUNTIME, "Cell set " + cellSetName + " is not sorted.");
    cellCount = uint32_t(cells.objectCount());
    const uint64_t globalCells = toc_.objectCount() - 1;
    for (uint32_t local = 0; local < cellCount; local++) {
        if (cellIds[local] >= globalCells) fail(EM2_ERROR_RUNTIME, "Cell set " + cellSetName + " refers to a cell that does not exist.");
    }
    genes = &g;
}

void Matrix::subset(const std::string& geneSetName, const std::string& cellSetName, std::vector<uint64_t>& toc,
                    std::vector<em2_count>& data, uint32_t& geneCount, uint32_t& cellCount) const
{
    const GeneSet* genesPointer = nullptr;
    const uint32_t* cellIds = nullptr;
    lookupSubset(geneSetName, cellSetName, genesPointer, cellIds, cellCount);
    const GeneSet& genes = *genesPointer;
    geneCount = genes.size();
    const uint64_t* globalToc = static_cast<const uint64_t*>(toc_.data());
    const em2_count* globalData = static_cast<const em2_count*>(data_.data());
    toc.assign(size_t(cellCount) + 1, 0);
    data.clear();
    for (uint32_t local = 0; local < cellCount; local++) {                    // ExpressionMatrixSubset.cpp:24-39
        const uint32_t global = cellIds[local];
        for (uint64_t j = globalToc[global]; j < globalToc[global + 1]; j++) {
            const uint32_t localGene = genes.localId(globalData[j].gene);
            if (localGene == kInvalidId) continue;
            em2_count c;
            c.gene = localGene;
            c.count = globalData[j].count;
            data.push_back(c);
        }
        toc[local + 1] = data.size();
    }
}

// Subset + hyperplanes + signatures (+ pairs) with the restricted CSR built on the device
// (em2_subset_find_similar_pairs4): the work of ExpressionMatrixSubset + Lsh (+ the pair loop).
void Matrix::runLshPath(const char* what, const std::string& geneSetName, const std::string& cellSetName, size_t lshCount,
                        unsigned int seed, uint32_t& cellCount, std::vector<uint64_t>* signatures, size_t k,
                        double similarityThreshold, const std::function<em2_pair*(uint32_t)>& pairsFor,
                        std::vector<uint32_t>* used) const
{
    StageTimer timer(what);
    const GeneSet* genes = nullptr;
    const uint32_t* cellIds = nullptr;
    lookupSubset(geneSetName, cellSetName, genes, cellIds, cellCount);
    if (lshCount == 0 || lshCount > 0xffffffffULL || k > 0xffffffffULL) fail(EM2_ERROR_INVALID_ARGUMENT, std::string(what) + ": lshCount or k out of range");
    const uint32_t geneCount = genes->size();
    timer.stage("lookup");
    // Lsh::generateLshVectors (src/Lsh.cpp:68-113) on a thread of its own: the device call asks for the hyperplanes when it has
    // uploaded the expression matrix and taken its subset (em2_internal_subset_find_similar_pairs4)
    // (members are destroyed in reverse order: the future -- whose destructor joins the generator thread -- goes first, so the
    // thread can never write `error` or `values` after their destruction when this frame unwinds)
    struct Hyperplanes {
        std::vector<double> values;
        std::string error;
        std::future<int> drawn;
    } hyperplanes;
    hyperplanes.values.resize(size_t(geneCount) * lshCount);
    hyperplanes.drawn = std::async(std::launch::async, [&hyperplanes, geneCount, lshCount, seed]() {
        const int rc = em2_lsh_generate_vectors(geneCount, uint32_t(lshCount), seed, hyperplanes.values.data());
        if (rc != EM2_OK) hyperplanes.error = em2_last_error();          // (the error text is thread-local: carried over)
        return rc;
    });
    const size_t words = (lshCount - 1) / 64 + 1;
    if (signatures) signatures->assign(size_t(cellCount) * words, 0);
    em2_pair* pairs = nullptr;
    if (used) {
        pairs = pairsFor(cellCount);           // where the pairs go (the mapped SimilarPairs file)
        used->assign(cellCount, 0);
    }
    const int rc = em2_internal_subset_find_similar_pairs4(
        static_cast<const uint64_t*>(toc_.data()), static_cast<const em2_count*>(data_.data()), uint32_t(toc_.objectCount() - 1),
        cellIds, cellCount, static_cast<const uint32_t*>(genes->localIds.data()), uint32_t(genes->localIds.objectCount()),
        geneCount,
        [](void* context) -> const double* {
            Hyperplanes* h = static_cast<Hyperplanes*>(context);
            if (h->drawn.get() != EM2_OK) {
                em2_internal_set_last_error(h->error.c_str());
                return nullptr;
            }
            return h->values.data();
        },
        &hyperplanes, uint32_t(lshCount), signatures ? signatures->data() : nullptr, uint32_t(k), similarityThreshold, pairs,
        used ? used->data() : nullptr);
    if (hyperplanes.drawn.valid()) hyperplanes.drawn.wait();          // (a call that failed before it asked for them)
    if (rc != EM2_OK) fail(rc, em2_last_error());
    timer.stage("hyperplanes (a thread) under the uploads; device: subset, signatures, pairs (incl. transfers)");
}

void Matrix::findSimilarPairs4(const std::string& geneSetName, const std::string& cellSetName,
                               const std::string& similarPairsName, size_t k, double similarityThreshold,
                               size_t lshCount, unsigned int seed) const
{
    // Lsh lsh(tmp-Lsh, subset, lshCount, seed) (ExpressionMatrixLsh.cpp:197), the pair loop and the selection
    // (:200-269).
    // SimilarPairs(directory, name, geneSet, cellSet, k) + copy + sort (ExpressionMatrixLsh.cpp:278-285): the device
    // produces the sorted order and its result is copied straight into the mapped -Pairs file.  tmp-Lsh /
    // tmp-ExpressionMatrixSubset files of the reference are deleted before it returns (:288,
    // ExpressionMatrixSubset.cpp:62-73) and are not created here.
    uint32_t cellCount = 0;
    std::vector<uint32_t> used;
    std::unique_ptr<SimilarPairsWriter> writer;
    runLshPath("findSimilarPairs4", geneSetName, cellSetName, lshCount, seed, cellCount, nullptr, k, similarityThreshold,
               [&](uint32_t cells) {
                   writer.reset(new SimilarPairsWriter(directoryName_, similarPairsName, geneSetName, cellSetName, k, cells));
                   return writer->pairs();
               },
               &used);
    StageTimer timer("findSimilarPairs4");
    writer->finish(used.data());
    writer.reset();
    timer.stage("write files");
}

void Matrix::computeLshSignatures(const std::string& geneSetName, const std::string& cellSetName,
                                  const std::string& lshName, size_t lshCount, unsigned int seed) const
{
    uint32_t cellCount = 0;
    std::vector<uint64_t> signatures;
    runLshPath("computeLshSignatures", geneSetName, cellSetName, lshCount, seed, cellCount, &signatures, 0, 0., nullptr, nullptr);
    writeLsh(directoryName_ + "/Lsh-" + lshName, cellCount, lshCount, signatures.data());       // ExpressionMatrixLsh.cpp:1189
}

void Matrix::analyzeLsh(const std::string& geneSetName, const std::string& cellSetName, size_t lshCount, unsigned int seed,
                        double csvDownsample, const std::string& outputDirectory) const
{
    // ExpressionMatrixLsh.cpp:1253-1283: the subset and its Lsh object (signatures of the same gene set, cell set, seed)
    uint32_t cellCount = 0, geneCount = 0;
    std::vector<uint64_t> toc;
    std::vector<em2_count> data;
    subset(geneSetName, cellSetName, toc, data, geneCount, cellCount);
    std::vector<uint64_t> signatures;
    runLshPath("analyzeLsh", geneSetName, cellSetName, lshCount, seed, cellCount, &signatures, 0, 0., nullptr, nullptr);
    const uint32_t* cellIds = static_cast<const uint32_t*>(cellSet(cellSetName).data());
    const std::string prefix = outputDirectory.empty() ? std::string() : outputDirectory + "/";
    const int rc = em2_analyze_lsh(toc.data(), data.data(), cellCount, geneCount, signatures.data(), uint32_t(lshCount), cellIds, seed,
                                   csvDownsample, (prefix + "Lsh-analysis.csv").c_str(), (prefix + "LSH-analysis-statistics.csv").c_str(),
                                   nullptr, nullptr, nullptr, nullptr, nullptr);
    if (rc != EM2_OK) fail(rc, em2_last_error());
}

void Matrix::findSimilarPairs5(const std::string& geneSetName, const std::string& cellSetName,
                               const std::string& lshName, const std::string& similarPairsName, size_t k,
                               double similarityThreshold, size_t lshSliceLength, size_t bucketOverflow) const
{
    // ExpressionMatrixLsh.cpp:326-351
    const GeneSet& genes = geneSet(geneSetName);
    if (genes.size() == 0) fail(EM2_ERROR_RUNTIME, "Gene set " + geneSetName + " is empty.");
    const MappedFile& cells = cellSet(cellSetName);
    const uint64_t cellCount = cells.objectCount();
    if (cellCount == 0) fail(EM2_ERROR_RUNTIME, "Cell set " + cellSetName + " is empty.");
    uint64_t lshCells = 0, lshCount = 0;
    std::vector<uint64_t> signatures;
    readLsh(directoryName_ + "/Lsh-" + lshName, lshCells, lshCount, signatures);
    if (lshCells != cellCount) {
        fail(EM2_ERROR_RUNTIME, "LSH object " + lshName + " has a number of cells inconsistent with cell set " + cellSetName);
    }
    if (lshSliceLength == 0) {
        // The reference computes lshCount / lshSliceLength here (:355) and dies with SIGFPE.
        fail(EM2_ERROR_INVALID_ARGUMENT, "findSimilarPairs5: lshSliceLength must be positive.");
    }
    if (k > 0xffffffffULL || lshSliceLength > 0xffffffffULL) fail(EM2_ERROR_INVALID_ARGUMENT, "findSimilarPairs5: argument out of range");
    std::vector<em2_pair> pairs(size_t(cellCount) * k);
    std::vector<uint32_t> used(cellCount);
    const int rc = em2_find_similar_pairs5(signatures.data(), uint32_t(cellCount), uint32_t(lshCount), uint32_t(k), similarityThreshold,
                                           uint32_t(lshSliceLength), bucketOverflow, pairs.data(), used.data());
    if (rc != EM2_OK) fail(rc, em2_last_error());
    writeSimilarPairs(directoryName_, similarPairsName, geneSetName, cellSetName, k, uint32_t(cellCount), pairs.data(), used.data());
}

void Matrix::findSimilarPairs7(const std::string& geneSetName, const std::string& cellSetName,
                               const std::string& lshName, const std::string& similarPairsName, size_t k,
                               double similarityThreshold, const std::vector<int32_t>& lshSliceLengths, uint32_t maxCheck,
                               size_t log2BucketCount) const
{
    // ExpressionMatrixLsh.cpp:522-561
    const GeneSet& genes = geneSet(geneSetName);
    if (genes.size() == 0) fail(EM2_ERROR_RUNTIME, "Gene set " + geneSetName + " is empty.");
    const MappedFile& cells = cellSet(cellSetName);
    const uint64_t cellCount = cells.objectCount();
    if (cellCount == 0) fail(EM2_ERROR_RUNTIME, "Cell set " + cellSetName + " is empty.");
    uint64_t lshCells = 0, lshCount = 0;
    std::vector<uint64_t> signatures;
    readLsh(directoryName_ + "/Lsh-" + lshName, lshCells, lshCount, signatures);
    if (lshCells != cellCount) {
        fail(EM2_ERROR_RUNTIME, "LSH object " + lshName + " has a number of cells inconsistent with cell set " + cellSetName);
    }
    if (k > 0xffffffffULL || log2BucketCount > 0xffffffffULL) fail(EM2_ERROR_INVALID_ARGUMENT, "findSimilarPairs7: argument out of range");
    std::vector<em2_pair> pairs(size_t(cellCount) * k);
    std::vector<uint32_t> used(cellCount);
    const int rc = em2_find_similar_pairs7(signatures.data(), uint32_t(cellCount), uint32_t(lshCount), uint32_t(k), similarityThreshold,
                                           lshSliceLengths.data(), uint32_t(lshSliceLengths.size()), maxCheck,
                                           uint32_t(log2BucketCount), pairs.data(), used.data());
    if (rc != EM2_OK) fail(rc, em2_last_error());
    writeSimilarPairs(directoryName_, similarPairsName, geneSetName, cellSetName, k, uint32_t(cellCount), pairs.data(), used.data());
}

void Matrix::removeSimilarPairs(const std::string& similarPairsName) const
{
    // ExpressionMatrixFindSimilarPairs.cpp:126-135: open (with all consistency checks), then remove.
    try {
        SimilarPairsInfo info;
        readSimilarPairs(directoryName_, similarPairsName, info, nullptr, nullptr);
    } catch (const Error&) {
        fail(EM2_ERROR_RUNTIME, "Error removing similar pairs object " + similarPairsName);
    }
    const std::string base = directoryName_ + "/SimilarPairs-" + similarPairsName;
    removeFile(base + "-Pairs");
    removeFile(base + "-CellInfo");
    removeFile(base + "-Info");
}


// ---------------------------------------------------------------------------------------------------------
// SimilarPairs / Lsh files
// ---------------------------------------------------------------------------------------------------------

static const char* const kTemporarySuffix = ".tmp";

SimilarPairsWriter::SimilarPairsWriter(const std::string& directoryName, const std::string& similarPairsName,
                                       const std::string& geneSetName, const std::string& cellSetName, size_t k,
                                       uint32_t cellCount)
    : cellCount_(cellCount), finished_(false)
{
    // accessGeneSet / accessCellSet (SimilarPairs.cpp:97-113)
    GeneSet genes;
    genes.globalIds.openExisting(directoryName + "/GeneSet-" + geneSetName + "-GlobalIds", false, sizeof(uint32_t));
    genes.localIds.openExisting(directoryName + "/GeneSet-" + geneSetName + "-LocalIds", false, sizeof(uint32_t));
    if (!isSorted(genes.genes(), genes.size())) fail(EM2_ERROR_RUNTIME, "Gene set " + geneSetName + " is not sorted.");
    MappedFile cells;
    cells.openExisting(directoryName + "/CellSet-" + cellSetName, false, sizeof(uint32_t));
    if (!isSorted(static_cast<const uint32_t*>(cells.data()), cells.objectCount())) fail(EM2_ERROR_RUNTIME, "Cell set " + cellSetName + " is not sorted.");
    if (cells.objectCount() != cellCount) fail(EM2_ERROR_RUNTIME, "SimilarPairs: cell count is not the size of cell set " + cellSetName);

    // The reference builds its SimilarPairs object only after the pair loop has succeeded
    // (ExpressionMatrixLsh.cpp:278-285); here the files exist while the device works (its result lands straight in
    // the mapped -Pairs file), so they carry temporary names until finish(): a call that fails leaves an existing
    // object of the same name untouched.
    base_ = directoryName + "/SimilarPairs-" + similarPairsName;
    const std::string base = base_;
    removeStale();
    infoFile_.createNew(base + "-Info" + kTemporarySuffix, true, sizeof(SimilarPairsInfoRecord), 1);
    SimilarPairsInfoRecord* info = static_cast<SimilarPairsInfoRecord*>(infoFile_.data());
    info->k = k;
    setStaticString(info->geneSetName, geneSetName);
    info->geneSetHash = hashOf(genes.globalIds, sizeof(uint32_t));
    setStaticString(info->cellSetName, cellSetName);
    info->cellSetHash = hashOf(cells, sizeof(uint32_t));
    pairsFile_.createNew(base + "-Pairs" + kTemporarySuffix, false, sizeof(em2_pair), k * size_t(cellCount));
    cellInfoFile_.createNew(base + "-CellInfo" + kTemporarySuffix, false, sizeof(CellInfoRecord), cellCount);
}

void SimilarPairsWriter::removeStale() const
{
    for (const char* part : {"-Info", "-Pairs", "-CellInfo"}) {
        const std::string path = base_ + part + kTemporarySuffix;
        if (fileExists(path)) ::unlink(path.c_str());
    }
}

SimilarPairsWriter::~SimilarPairsWriter()
{
    if (finished_) return;
    infoFile_.close();
    pairsFile_.close();
    cellInfoFile_.close();
    removeStale();
}

em2_pair* SimilarPairsWriter::pairs() { return static_cast<em2_pair*>(pairsFile_.data()); }

void SimilarPairsWriter::finish(const uint32_t* usedCount)
{
    CellInfoRecord* ci = static_cast<CellInfoRecord*>(cellInfoFile_.data());
    for (uint32_t c = 0; c < cellCount_; c++) {
        ci[c].usedCount = usedCount[c];                      // SimilarPairs::copy, :376
        ci[c].lowestSimilarityIndex = 0xffffffffu;           // constructor values, never updated by copy (:36-40)
        ci[c].lowestSimilarity = FLT_MAX;
    }
    // The three files replace an existing object in this order: the old -Info is removed first (a reader opens -Info
    // first, SimilarPairs.cpp:49-52: while the swap is in progress it finds no object instead of a -Pairs file under an -Info
    // that describes another k), then -Pairs and -CellInfo move into place, -Info last.  A rename that fails leaves the
    // object absent rather than mixed: what has been moved already is removed again.  What remains is the window between
    // the unlink and the last rename, during which the name does not exist; a failure BEFORE this point (the computation)
    // still leaves an existing object untouched.
    (void)::unlink((base_ + "-Info").c_str());
    std::vector<std::string> moved;
    for (const char* part : {"-Pairs", "-CellInfo", "-Info"}) {
        const std::string final = base_ + part;
        if (::rename((final + kTemporarySuffix).c_str(), final.c_str()) != 0) {
            for (const std::string& name : moved) (void)::unlink(name.c_str());
            fail(EM2_ERROR_RUNTIME, "Error renaming " + final + kTemporarySuffix + " to " + final);
        }
        moved.push_back(final);
    }
    finished_ = true;
}

void writeSimilarPairs(const std::string& directoryName, const std::string& similarPairsName,
                       const std::string& geneSetName, const std::string& cellSetName, size_t k,
                       uint32_t cellCount, const em2_pair* pairs, const uint32_t* usedCount)
{
    SimilarPairsWriter writer(directoryName, similarPairsName, geneSetName, cellSetName, k, cellCount);
    if (k && cellCount) std::memcpy(writer.pairs(), pairs, k * size_t(cellCount) * sizeof(em2_pair));
    writer.finish(usedCount);
}

void readSimilarPairs(const std::string& directoryName, const std::string& similarPairsName,
                      SimilarPairsInfo& info, std::vector<em2_pair>* pairs, std::vector<uint32_t>* usedCount)
{
    const std::string base = directoryName + "/SimilarPairs-" + similarPairsName;
    MappedFile infoFile;
    infoFile.openExisting(base + "-Info", true, sizeof(SimilarPairsInfoRecord));
    const SimilarPairsInfoRecord* rec = static_cast<const SimilarPairsInfoRecord*>(infoFile.data());
    info.k = rec->k;
    info.geneSetName = getStaticString(rec->geneSetName);
    info.geneSetHash = rec->geneSetHash;
    info.cellSetName = getStaticString(rec->cellSetName);
    info.cellSetHash = rec->cellSetHash;

    GeneSet genes;
    genes.globalIds.openExisting(directoryName + "/GeneSet-" + info.geneSetName + "-GlobalIds", false, sizeof(uint32_t));
    MappedFile cells;
    cells.openExisting(directoryName + "/CellSet-" + info.cellSetName, false, sizeof(uint32_t));
    MappedFile pairsFile, cellInfoFile;
    pairsFile.openExisting(base + "-Pairs", false, sizeof(em2_pair));
    cellInfoFile.openExisting(base + "-CellInfo", false, sizeof(CellInfoRecord));
    // SimilarPairs.cpp:65-82
    if (hashOf(genes.globalIds, sizeof(uint32_t)) != info.geneSetHash) {
        fail(EM2_ERROR_RUNTIME, "Hash for gene set " + info.geneSetName + " is not consistent with the value at the time SimilarPairs object " + similarPairsName + " was created.");
    }
    if (hashOf(cells, sizeof(uint32_t)) != info.cellSetHash) {
        fail(EM2_ERROR_RUNTIME, "Hash for cell set " + info.cellSetName + " is not consistent with the value at the time SimilarPairs object " + similarPairsName + " was created.");
    }
    if (pairsFile.objectCount() != info.k * cells.objectCount()) {
        fail(EM2_ERROR_RUNTIME, "SimilarPairs object " + similarPairsName + " has similarPairs vector of inconsistent length.");
    }
    if (cellInfoFile.objectCount() != cells.objectCount()) {
        fail(EM2_ERROR_RUNTIME, "SimilarPairs object " + similarPairsName + " has cellInfo vector of inconsistent length.");
    }
    info.cellCount = cells.objectCount();
    if (pairs) {
        const em2_pair* p = static_cast<const em2_pair*>(pairsFile.data());
        pairs->assign(p, p + pairsFile.objectCount());
    }
    if (usedCount) {
        const CellInfoRecord* ci = static_cast<const CellInfoRecord*>(cellInfoFile.data());
        usedCount->resize(cellInfoFile.objectCount());
        for (size_t c = 0; c < usedCount->size(); c++) (*usedCount)[c] = ci[c].usedCount;
    }
}

void writeLsh(const std::string& prefix, uint64_t cellCount, uint64_t lshCount, const uint64_t* signatures)
{
    MappedFile infoFile;
    infoFile.createNew(prefix + "-Info", true, sizeof(LshInfoRecord), 1);             // Lsh.cpp:26-28
    LshInfoRecord* info = static_cast<LshInfoRecord*>(infoFile.data());
    info->lshCount = lshCount;
    info->cellCount = cellCount;
    const uint64_t words = (lshCount - 1) / 64 + 1;
    MappedFile sigFile;
    sigFile.createNew(prefix + "-Signatures", false, sizeof(uint64_t), cellCount * words);   // Lsh.cpp:148
    if (cellCount) std::memcpy(sigFile.data(), signatures, cellCount * words * sizeof(uint64_t));
}

void readLshInfo(const std::string& prefix, uint64_t& cellCount, uint64_t& lshCount)
{
    MappedFile infoFile;
    infoFile.openExisting(prefix + "-Info", true, sizeof(LshInfoRecord));            // Lsh.cpp:52-53
    const LshInfoRecord* info = static_cast<const LshInfoRecord*>(infoFile.data());
    cellCount = info->cellCount;
    lshCount = info->lshCount;
}

void readLsh(const std::string& prefix, uint64_t& cellCount, uint64_t& lshCount, std::vector<uint64_t>& signatures)
{
    readLshInfo(prefix, cellCount, lshCount);
    MappedFile sigFile;
    sigFile.openExisting(prefix + "-Signatures", false, sizeof(uint64_t));
    if (lshCount == 0) fail(EM2_ERROR_RUNTIME, "Lsh object " + prefix + " has lshCount 0.");
    const uint64_t words = (lshCount - 1) / 64 + 1;
    if (sigFile.objectCount() != cellCount * words) fail(EM2_ERROR_RUNTIME, "Lsh object " + prefix + " has a signature vector of inconsistent length.");
    const uint64_t* p = static_cast<const uint64_t*>(sigFile.data());
    signatures.assign(p, p + sigFile.objectCount());
}


// ---------------------------------------------------------------------------------------------------------
// Tooling: directories with exactly the files the path reads.
// ---------------------------------------------------------------------------------------------------------

void addGeneSet(const std::string& directoryName, const std::string& name, const uint32_t* sortedGlobalIds,
                uint32_t count, uint32_t totalGeneCount)
{
    if (!isSorted(sortedGlobalIds, count)) fail(EM2_ERROR_INVALID_ARGUMENT, "addGeneSet: ids must be sorted");
    MappedFile global, local;
    global.createNew(directoryName + "/GeneSet-" + name + "-GlobalIds", false, sizeof(uint32_t), count);
    if (count) std::memcpy(global.data(), sortedGlobalIds, size_t(count) * sizeof(uint32_t));
    // localGeneIdVector is sized by the largest gene id seen (GeneSet::addGene, src/GeneSet.cpp:44-55).
    const uint32_t localSize = count ? sortedGlobalIds[count - 1] + 1 : 0;
    (void)totalGeneCount;
    local.createNew(directoryName + "/GeneSet-" + name + "-LocalIds", false, sizeof(uint32_t), localSize);
    uint32_t* l = static_cast<uint32_t*>(local.data());
    for (uint32_t i = 0; i < localSize; i++) l[i] = kInvalidId;
    for (uint32_t i = 0; i < count; i++) l[sortedGlobalIds[i]] = i;
}

void addCellSet(const std::string& directoryName, const std::string& name, const uint32_t* sortedCellIds, uint32_t count)
{
    if (!isSorted(sortedCellIds, count)) fail(EM2_ERROR_INVALID_ARGUMENT, "addCellSet: ids must be sorted");
    MappedFile f;
    f.createNew(directoryName + "/CellSet-" + name, false, sizeof(uint32_t), count);
    if (count) std::memcpy(f.data(), sortedCellIds, size_t(count) * sizeof(uint32_t));
}

void createDirectoryFromCsr(const std::string& directoryName, uint32_t geneCount, uint32_t cellCount,
                            const uint64_t* toc, const em2_count* data)
{
    if (::mkdir(directoryName.c_str(), 0777) == -1 && errno != EEXIST) fail(EM2_ERROR_IO, "Cannot create directory " + directoryName);
    {
        MappedFile t, d;
        t.createNew(directoryName + "/CellExpressionCounts.toc", false, sizeof(uint64_t), size_t(cellCount) + 1);
        std::memcpy(t.data(), toc, (size_t(cellCount) + 1) * sizeof(uint64_t));
        const uint64_t nnz = toc[cellCount];
        d.createNew(directoryName + "/CellExpressionCounts.data", false, sizeof(em2_count), nnz);
        if (nnz) std::memcpy(d.data(), data, nnz * sizeof(em2_count));
    }
    std::vector<uint32_t> ids(std::max(geneCount, cellCount));
    for (uint32_t i = 0; i < ids.size(); i++) ids[i] = i;
    addGeneSet(directoryName, "AllGenes", ids.data(), geneCount, geneCount);
    addCellSet(directoryName, "AllCells", ids.data(), cellCount);
}

}  // namespace host
}  // namespace em2
