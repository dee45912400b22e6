// Test program of tests/test_gpu_dist_entry.py: em2_dist_find_similar_pairs4 over a REAL RCCL communicator of one rank (a
// 1-GPU box cannot hold two RCCL ranks), against em2_dev_find_similar_pairs4 on the same signatures.  Exercises the RCCL
// binding and every collective the entry issues (ncclAllGather, ncclAllReduce, grouped ncclSend / ncclRecv to itself).
// usage: em2_dist_rccl <cells> <lshCount> <k>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "em2_lsh.h"

#define CHECK(x) do { if (!(x)) { printf("FAILED %s (line %d): %s\n", #x, __LINE__, em2_last_error()); return 1; } } while (0)

int main(int argc, char** argv)
{
    const uint32_t cells = argc > 1 ? uint32_t(atoi(argv[1])) : 20000, L = argc > 2 ? uint32_t(atoi(argv[2])) : 1024, k = argc > 3 ? uint32_t(atoi(argv[3])) : 20;
    const uint32_t words = (L - 1) / 64 + 1;
    // clustered signatures: 8 centres, every bit flipped with probability 1/8
    std::vector<uint64_t> sig(size_t(cells) * words), centres(8 * words);
    uint64_t state = 88172645463325252ull;
    auto next = [&]() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return state; };
    for (auto& w : centres) w = next();
    for (uint32_t c = 0; c < cells; c++) {
        const uint64_t* centre = centres.data() + size_t(next() % 8) * words;
        for (uint32_t w = 0; w < words; w++) sig[size_t(c) * words + w] = centre[w] ^ (next() & next() & next());
    }
    CHECK(hipSetDevice(0) == hipSuccess);
    ncclUniqueId id;
    ncclComm_t comm;
    CHECK(ncclGetUniqueId(&id) == ncclSuccess);
    CHECK(ncclCommInitRank(&comm, 1, id, 0) == ncclSuccess);

    uint64_t *dLocal = nullptr, *dAll = nullptr;
    em2_pair *dPairs = nullptr, *dPairsRef = nullptr;
    uint32_t *dUsed = nullptr, *dUsedRef = nullptr;
    void *ws = nullptr, *wsRef = nullptr;
    const size_t sigBytes = sig.size() * 8, pairBytes = size_t(cells) * k * sizeof(em2_pair);
    const size_t wsBytes = em2_dist_find_similar_pairs4_workspace(cells, L, k, 0, 1);
    const size_t wsRefBytes = em2_dev_find_similar_pairs4_workspace(cells, cells, L, k);
    CHECK(hipMalloc(&dLocal, sigBytes) == hipSuccess && hipMalloc(&dAll, sigBytes) == hipSuccess);
    CHECK(hipMalloc(&dPairs, pairBytes) == hipSuccess && hipMalloc(&dPairsRef, pairBytes) == hipSuccess);
    CHECK(hipMalloc(&dUsed, cells * 4) == hipSuccess && hipMalloc(&dUsedRef, cells * 4) == hipSuccess);
    CHECK(hipMalloc(&ws, wsBytes) == hipSuccess && hipMalloc(&wsRef, wsRefBytes) == hipSuccess);
    CHECK(hipMemcpy(dLocal, sig.data(), sigBytes, hipMemcpyHostToDevice) == hipSuccess);
    CHECK(hipMemset(dPairs, 0, pairBytes) == hipSuccess && hipMemset(dPairsRef, 0, pairBytes) == hipSuccess);

    const int form = em2_dist_find_similar_pairs4_form(cells, L, k, 1);
    double ms[EM2_DIST_MS_COUNT];
    CHECK(em2_dist_find_similar_pairs4(comm, dLocal, cells, L, k, 0.2, dAll, dPairs, dUsed, ws, wsBytes, nullptr, ms) == EM2_OK);
    CHECK(hipDeviceSynchronize() == hipSuccess);
    CHECK(em2_dev_find_similar_pairs4(dLocal, cells, 0, cells, L, k, 0.2, dPairsRef, dUsedRef, wsRef, wsRefBytes, nullptr) == EM2_OK);
    CHECK(em2_dev_find_similar_pairs4_status(wsRef, cells, k, nullptr) == EM2_OK);
    std::vector<em2_pair> a(size_t(cells) * k), b(size_t(cells) * k);
    std::vector<uint32_t> ua(cells), ub(cells);
    std::vector<uint64_t> all(sig.size());
    CHECK(hipMemcpy(a.data(), dPairs, pairBytes, hipMemcpyDeviceToHost) == hipSuccess);
    CHECK(hipMemcpy(b.data(), dPairsRef, pairBytes, hipMemcpyDeviceToHost) == hipSuccess);
    CHECK(hipMemcpy(ua.data(), dUsed, cells * 4, hipMemcpyDeviceToHost) == hipSuccess);
    CHECK(hipMemcpy(ub.data(), dUsedRef, cells * 4, hipMemcpyDeviceToHost) == hipSuccess);
    CHECK(hipMemcpy(all.data(), dAll, sigBytes, hipMemcpyDeviceToHost) == hipSuccess);
    CHECK(memcmp(all.data(), sig.data(), sigBytes) == 0);
    CHECK(memcmp(ua.data(), ub.data(), cells * 4) == 0);
    CHECK(memcmp(a.data(), b.data(), pairBytes) == 0);
    uint64_t total = 0;
    for (uint32_t u : ua) total += u;
    CHECK(total > 0);
    printf("OK form %d, %llu pairs kept, stage ms: gather %.2f scan %.2f all_reduce %.2f exchange %.2f redistribute %.2f\n", form,
           (unsigned long long)total, ms[0], ms[1], ms[2], ms[3], ms[4]);
    ncclCommDestroy(comm);
    return 0;
}
