// ref_layout.cpp -- the byte layout of SimilarPairs-<name>-Info as the reference's OWN class template produces it:
//
//   /root/reference/src/ShortStaticString.hpp   ShortStaticString<255> = StaticString255 (:21, :28-133)
//
// compiled where it lies (oracle/Makefile: -I /root/reference/src) into oracle/_ref/libem2ref_layout.so, next to
// libem2ref.so and under the same rule: TEST INFRASTRUCTURE ONLY, exists only where /root/reference exists, nothing of
// the reference is copied into this repository.
//
// ONE DISCLOSURE, which is why this is a library of its own and not part of ref_components.cpp: the header ends with an
// inline self-test function, testShortStaticString() (:157-214), whose body uses the macro CZI_ASSERT.  The header
// does not include the file that defines it (CZI_ASSERT.hpp), and that file includes boost/lexical_cast.hpp, which this
// image does not have.  The function is never called here, but it has to parse, so this driver defines CZI_ASSERT as a
// no-op before the #include.  Nothing that is pinned depends on it: the class template (members n and s, constructors,
// assignment) contains no CZI_ASSERT.  No Boost header is stood in for.
//
// `Info` below is NOT the reference's (SimilarPairs.hpp, which holds it as a private nested class at :190-203, includes
// MemoryMappedObject.hpp -> CZI_ASSERT.hpp -> Boost): it restates those five member declarations in their order, with the
// reference's StaticString255 as the member type.  What the test pins is therefore sizeof / alignment / byte content of
// StaticString255 and the struct layout the compiler derives from it.

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <limits>
#include <new>

#define CZI_ASSERT(expression) ((void)0)        // see the disclosure above: lets the never-called self test parse
#include "ShortStaticString.hpp"

using namespace ChanZuckerberg::ExpressionMatrix2;

namespace {
// SimilarPairs::Info, src/SimilarPairs.hpp:190-203
class Info {
public:
    size_t k;
    StaticString255 geneSetName;
    uint64_t geneSetHash;
    StaticString255 cellSetName;
    uint64_t cellSetHash;
};
}  // namespace

extern "C" {

uint64_t em2ref_info_size() { return sizeof(Info); }

// offsets of the five members and of the two strings' character arrays
void em2ref_info_offsets(uint64_t* out)
{
    out[0] = offsetof(Info, k);
    out[1] = offsetof(Info, geneSetName);
    out[2] = offsetof(Info, geneSetHash);
    out[3] = offsetof(Info, cellSetName);
    out[4] = offsetof(Info, cellSetHash);
    out[5] = offsetof(StaticString255, n);
    out[6] = offsetof(StaticString255, s);
    out[7] = sizeof(StaticString255);
}

// What SimilarPairs::SimilarPairs does to the mapped object (src/SimilarPairs.cpp:24-29) after
// MemoryMapped::Object<Info>::createNew has run `new(data) T()` on it (src/MemoryMappedObject.hpp:299): value-initialise,
// then assign the five fields.  `out` (em2ref_info_size() bytes) is filled with a pattern first, so that the test also
// sees which bytes the value-initialisation itself defines.  Returns 1 if a name exceeds the capacity (the reference
// throws "ShortStaticString capacity exceeded."), else 0.
int em2ref_make_info(uint64_t k, const char* geneSetName, uint64_t geneSetHash, const char* cellSetName, uint64_t cellSetHash,
                     unsigned char* out)
{
    std::memset(out, 0xAA, sizeof(Info));
    Info* info = new (out) Info();
    try {
        info->k = size_t(k);
        info->geneSetName = std::string(geneSetName);
        info->geneSetHash = geneSetHash;
        info->cellSetName = std::string(cellSetName);
        info->cellSetHash = cellSetHash;
    } catch (std::runtime_error&) {
        return 1;
    }
    return 0;
}

}  // extern "C"
