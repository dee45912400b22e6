// em2_matrix_capi.cpp -- C ABI of the ExpressionMatrix-level entry points (include/em2_lsh.h): translates
// em2::host::Error into status codes + em2_last_error().
#include "em2_host.h"

#include <cstring>
#include <new>
#include <string>

// Defined in em2_capi.hip.
extern "C" void em2_internal_set_last_error(const char* message);

struct em2_matrix {
    em2::host::Matrix* impl;
};

namespace {

template <class F> int guarded(F f)
{
    try {
        f();
        return EM2_OK;
    } catch (const em2::host::Error& e) {
        em2_internal_set_last_error(e.message.c_str());
        return e.code;
    } catch (const std::bad_alloc&) {
        em2_internal_set_last_error("out of host memory");
        return EM2_ERROR_RUNTIME;
    } catch (const std::exception& e) {
        em2_internal_set_last_error(e.what());
        return EM2_ERROR_RUNTIME;
    }
}

int nullArgument(const char* function)
{
    em2_internal_set_last_error((std::string(function) + ": null argument").c_str());
    return EM2_ERROR_INVALID_ARGUMENT;
}

}  // namespace

extern "C" {

int em2_matrix_open(const char* directoryName, em2_matrix** matrix)
{
    if (!directoryName || !matrix) return nullArgument("em2_matrix_open");
    *matrix = nullptr;
    return guarded([&] {
        em2::host::Matrix* m = new em2::host::Matrix(directoryName);
        *matrix = new em2_matrix{m};
    });
}

void em2_matrix_close(em2_matrix* matrix)
{
    if (matrix) {
        delete matrix->impl;
        delete matrix;
    }
}

int em2_matrix_find_similar_pairs4(em2_matrix* matrix, const char* geneSetName, const char* cellSetName,
                                   const char* similarPairsName, size_t k, double similarityThreshold,
                                   size_t lshCount, unsigned int seed)
{
    if (!matrix || !geneSetName || !cellSetName || !similarPairsName) return nullArgument("em2_matrix_find_similar_pairs4");
    return guarded([&] {
        matrix->impl->findSimilarPairs4(geneSetName, cellSetName, similarPairsName, k, similarityThreshold, lshCount, seed);
    });
}

int em2_matrix_compute_lsh_signatures(em2_matrix* matrix, const char* geneSetName, const char* cellSetName,
                                      const char* lshName, size_t lshCount, unsigned int seed)
{
    if (!matrix || !geneSetName || !cellSetName || !lshName) return nullArgument("em2_matrix_compute_lsh_signatures");
    return guarded([&] { matrix->impl->computeLshSignatures(geneSetName, cellSetName, lshName, lshCount, seed); });
}

int em2_matrix_analyze_lsh(em2_matrix* matrix, const char* geneSetName, const char* cellSetName, size_t lshCount,
                           unsigned int seed, double csvDownsample, const char* outputDirectory)
{
    if (!matrix || !geneSetName || !cellSetName) return nullArgument("em2_matrix_analyze_lsh");
    return guarded([&] {
        matrix->impl->analyzeLsh(geneSetName, cellSetName, lshCount, seed, csvDownsample, outputDirectory ? outputDirectory : "");
    });
}

int em2_matrix_find_similar_pairs5(em2_matrix* matrix, const char* geneSetName, const char* cellSetName,
                                   const char* lshName, const char* similarPairsName, size_t k,
                                   double similarityThreshold, size_t lshSliceLength, size_t bucketOverflow)
{
    if (!matrix || !geneSetName || !cellSetName || !lshName || !similarPairsName) return nullArgument("em2_matrix_find_similar_pairs5");
    return guarded([&] {
        matrix->impl->findSimilarPairs5(geneSetName, cellSetName, lshName, similarPairsName, k, similarityThreshold,
                                        lshSliceLength, bucketOverflow);
    });
}

int em2_matrix_find_similar_pairs7(em2_matrix* matrix, const char* geneSetName, const char* cellSetName,
                                   const char* lshName, const char* similarPairsName, size_t k,
                                   double similarityThreshold, const int32_t* lshSliceLengths, uint32_t sliceLengthCount,
                                   uint32_t maxCheck, size_t log2BucketCount)
{
    if (!matrix || !geneSetName || !cellSetName || !lshName || !similarPairsName || (!lshSliceLengths && sliceLengthCount)) {
        return nullArgument("em2_matrix_find_similar_pairs7");
    }
    return guarded([&] {
        matrix->impl->findSimilarPairs7(geneSetName, cellSetName, lshName, similarPairsName, k, similarityThreshold,
                                        std::vector<int32_t>(lshSliceLengths, lshSliceLengths + sliceLengthCount), maxCheck,
                                        log2BucketCount);
    });
}

int em2_matrix_remove_similar_pairs(em2_matrix* matrix, const char* similarPairsName)
{
    if (!matrix || !similarPairsName) return nullArgument("em2_matrix_remove_similar_pairs");
    return guarded([&] { matrix->impl->removeSimilarPairs(similarPairsName); });
}

int em2_matrix_subset(em2_matrix* matrix, const char* geneSetName, const char* cellSetName, uint32_t* geneCount,
                      uint32_t* cellCount, uint64_t* nnz, uint64_t* toc, em2_count* data)
{
    if (!matrix || !geneSetName || !cellSetName || !geneCount || !cellCount || !nnz) return nullArgument("em2_matrix_subset");
    return guarded([&] {
        std::vector<uint64_t> t;
        std::vector<em2_count> d;
        matrix->impl->subset(geneSetName, cellSetName, t, d, *geneCount, *cellCount);
        *nnz = d.size();
        if (toc) {
            std::memcpy(toc, t.data(), t.size() * sizeof(uint64_t));
            if (!d.empty() && data) std::memcpy(data, d.data(), d.size() * sizeof(em2_count));
        }
    });
}

int em2_similar_pairs_write(const char* directoryName, const char* similarPairsName, const char* geneSetName,
                            const char* cellSetName, size_t k, uint32_t cellCount, const em2_pair* pairs,
                            const uint32_t* usedCount)
{
    if (!directoryName || !similarPairsName || !geneSetName || !cellSetName || !usedCount || (!pairs && k && cellCount)) {
        return nullArgument("em2_similar_pairs_write");
    }
    return guarded([&] {
        em2::host::writeSimilarPairs(directoryName, similarPairsName, geneSetName, cellSetName, k, cellCount, pairs, usedCount);
    });
}

int em2_similar_pairs_read(const char* directoryName, const char* similarPairsName, uint64_t* k,
                           uint64_t* cellCount, em2_pair* pairs, uint32_t* usedCount)
{
    if (!directoryName || !similarPairsName || !k || !cellCount) return nullArgument("em2_similar_pairs_read");
    return guarded([&] {
        em2::host::SimilarPairsInfo info;
        std::vector<em2_pair> p;
        std::vector<uint32_t> u;
        em2::host::readSimilarPairs(directoryName, similarPairsName, info, pairs ? &p : nullptr, usedCount ? &u : nullptr);
        *k = info.k;
        *cellCount = info.cellCount;
        if (pairs && !p.empty()) std::memcpy(pairs, p.data(), p.size() * sizeof(em2_pair));
        if (usedCount && !u.empty()) std::memcpy(usedCount, u.data(), u.size() * sizeof(uint32_t));
    });
}

int em2_similar_pairs_info(const char* directoryName, const char* similarPairsName, uint64_t* k, uint64_t* cellCount,
                           char* geneSetName, char* cellSetName)
{
    if (!directoryName || !similarPairsName || !k || !cellCount || !geneSetName || !cellSetName) return nullArgument("em2_similar_pairs_info");
    return guarded([&] {
        em2::host::SimilarPairsInfo info;
        em2::host::readSimilarPairs(directoryName, similarPairsName, info, nullptr, nullptr);
        *k = info.k;
        *cellCount = info.cellCount;
        std::memset(geneSetName, 0, 256);
        std::memset(cellSetName, 0, 256);
        std::memcpy(geneSetName, info.geneSetName.data(), info.geneSetName.size());
        std::memcpy(cellSetName, info.cellSetName.data(), info.cellSetName.size());
    });
}

int em2_matrix_cell_set(em2_matrix* matrix, const char* cellSetName, uint32_t* count, uint32_t* ids)
{
    if (!matrix || !cellSetName || !count) return nullArgument("em2_matrix_cell_set");
    return guarded([&] {
        const em2::host::MappedFile& f = matrix->impl->cellSet(cellSetName);
        *count = uint32_t(f.objectCount());
        if (ids && *count) std::memcpy(ids, f.data(), size_t(*count) * sizeof(uint32_t));
    });
}

int em2_lsh_write(const char* directoryName, const char* lshName, uint64_t cellCount, uint64_t lshCount,
                  const uint64_t* signatures)
{
    if (!directoryName || !lshName || (!signatures && cellCount) || lshCount == 0) return nullArgument("em2_lsh_write");
    return guarded([&] {
        em2::host::writeLsh(std::string(directoryName) + "/Lsh-" + lshName, cellCount, lshCount, signatures);
    });
}

int em2_lsh_read(const char* directoryName, const char* lshName, uint64_t* cellCount, uint64_t* lshCount,
                 uint64_t* signatures)
{
    if (!directoryName || !lshName || !cellCount || !lshCount) return nullArgument("em2_lsh_read");
    return guarded([&] {
        const std::string prefix = std::string(directoryName) + "/Lsh-" + lshName;
        if (!signatures) {
            em2::host::readLshInfo(prefix, *cellCount, *lshCount);
        } else {
            std::vector<uint64_t> s;
            em2::host::readLsh(prefix, *cellCount, *lshCount, s);
            if (!s.empty()) std::memcpy(signatures, s.data(), s.size() * sizeof(uint64_t));
        }
    });
}

int em2_tool_create_directory(const char* directoryName, uint32_t geneCount, uint32_t cellCount,
                              const uint64_t* toc, const em2_count* data)
{
    if (!directoryName || !toc) return nullArgument("em2_tool_create_directory");
    return guarded([&] { em2::host::createDirectoryFromCsr(directoryName, geneCount, cellCount, toc, data); });
}

int em2_tool_add_gene_set(const char* directoryName, const char* name, const uint32_t* sortedGlobalIds, uint32_t count)
{
    if (!directoryName || !name || (!sortedGlobalIds && count)) return nullArgument("em2_tool_add_gene_set");
    return guarded([&] { em2::host::addGeneSet(directoryName, name, sortedGlobalIds, count, 0); });
}

int em2_tool_add_cell_set(const char* directoryName, const char* name, const uint32_t* sortedCellIds, uint32_t count)
{
    if (!directoryName || !name || (!sortedCellIds && count)) return nullArgument("em2_tool_add_cell_set");
    return guarded([&] { em2::host::addCellSet(directoryName, name, sortedCellIds, count); });
}

}  // extern "C"
