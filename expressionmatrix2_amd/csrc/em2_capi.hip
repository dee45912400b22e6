// em2_capi.hip -- the C ABI of include/em2_lsh.h: argument checking, device memory, kernel launches.
// No algorithmic work happens on the host here except the O(geneCount*lshCount) hyperplane generator and the
// O(lshCount) lookup tables; there is no CPU fallback for the device paths.

#include "../../include/em2_lsh.h"

#include "em2_device.h"
#include "em2_tables.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <random>
#include <string>
#include <vector>

namespace {

thread_local std::string lastError;

int fail(int code, const std::string& message)
{
    lastError = message;
    return code;
}

int failHip(hipError_t e, const char* what)
{
    return fail(EM2_ERROR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define EM2_HIP(call)                                   \
    do {                                                \
        hipError_t em2HipError_ = (call);               \
        if (em2HipError_ != hipSuccess) return failHip(em2HipError_, #call); \
    } while (0)

size_t alignUp(size_t x) { return (x + 255u) & ~size_t(255u); }

uint32_t wordCountOf(uint32_t lshCount) { return (lshCount - 1u) / 64u + 1u; }

bool haveDevice()
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}

// Per-device cache of the lookup tables of em2_tables.h.
struct CachedTables {
    int device;
    uint32_t lshCount;
    uint64_t thresholdBits;
    em2::DeviceTables tables;
};
std::mutex cacheMutex;
std::vector<CachedTables> tableCache;
constexpr size_t kTableCacheEntries = 16;
std::vector<void*> retiredBlocks;               // evicted from the cache, freed kRetiredBlocks evictions later (see getDeviceTables)
constexpr size_t kRetiredBlocks = 64;

int getDeviceTables(uint32_t lshCount, double similarityThreshold, em2::DeviceTables& out)
{
    int device = 0;
    EM2_HIP(hipGetDevice(&device));
    uint64_t bits;
    std::memcpy(&bits, &similarityThreshold, sizeof(bits));
    std::lock_guard<std::mutex> lock(cacheMutex);
    for (size_t i = 0; i < tableCache.size(); i++) {
        const CachedTables c = tableCache[i];
        if (c.device == device && c.lshCount == lshCount && c.thresholdBits == bits) {
            // most recently used last
            tableCache.erase(tableCache.begin() + long(i));
            tableCache.push_back(c);
            out = c.tables;
            return EM2_OK;
        }
    }
    em2::SimilarityTables host;
    const char* error = nullptr;
    if (!em2::buildSimilarityTables(lshCount, similarityThreshold, host, &error)) {
        return fail(EM2_ERROR_RUNTIME, error ? error : "similarity table construction failed");
    }
    void* block = nullptr;
    const size_t keyCount = host.keySimilarity.size();
    const size_t bytes0 = alignUp(host.keyOfMismatch.size() * sizeof(uint32_t));
    const size_t bytes1 = alignUp(keyCount * sizeof(int32_t));
    const size_t bytes2 = alignUp(keyCount * sizeof(float));
    EM2_HIP(hipMalloc(&block, bytes0 + bytes1 + bytes2));
    char* base = static_cast<char*>(block);
    EM2_HIP(hipMemcpy(base, host.keyOfMismatch.data(), host.keyOfMismatch.size() * sizeof(uint32_t),
                      hipMemcpyHostToDevice));
    EM2_HIP(hipMemcpy(base + bytes0, host.acceptMaxByKey.data(), keyCount * sizeof(int32_t),
                      hipMemcpyHostToDevice));
    EM2_HIP(hipMemcpy(base + bytes0 + bytes1, host.keySimilarity.data(), keyCount * sizeof(float),
                      hipMemcpyHostToDevice));
    CachedTables c;
    c.device = device;
    c.lshCount = lshCount;
    c.thresholdBits = bits;
    c.tables.keyOfMismatch = reinterpret_cast<const uint32_t*>(base);
    c.tables.acceptMaxByKey = reinterpret_cast<const int32_t*>(base + bytes0);
    c.tables.keySimilarity = reinterpret_cast<const float*>(base + bytes0 + bytes1);
    c.tables.mGlobal = host.mGlobal;
    c.tables.mMaxInitial = host.mMaxInitial;
    c.tables.identityKeys = host.keySimilarity.size() == host.keyOfMismatch.size();
    // At most kTableCacheEntries sets of tables per device stay in the cache (a caller that sweeps thresholds would otherwise
    // grow it without bound): the least recently used one leaves it.  Its block is NOT freed at that moment: a caller on
    // another thread may have fetched these pointers a moment ago and not yet enqueued its kernel (DeviceTables holds device
    // pointers, not the arrays), and hipFree would also synchronise the device inside calls documented as asynchronous.
    // Evicted blocks wait in a retired list; only when kRetiredBlocks more evictions have followed is the oldest one freed
    // (a few KB each: the wait costs nothing, and by then every launch that could hold the pointers has long been enqueued
    // -- hipFree itself waits for the device before it releases memory that enqueued kernels still read).
    size_t onThisDevice = 0;
    for (const CachedTables& other : tableCache) onThisDevice += other.device == device ? 1u : 0u;
    if (onThisDevice >= kTableCacheEntries) {
        for (size_t i = 0; i < tableCache.size(); i++) {
            if (tableCache[i].device != device) continue;
            retiredBlocks.push_back(const_cast<uint32_t*>(tableCache[i].tables.keyOfMismatch));       // the block's base pointer
            tableCache.erase(tableCache.begin() + long(i));
            break;
        }
        if (retiredBlocks.size() > kRetiredBlocks) {
            (void)hipFree(retiredBlocks.front());
            retiredBlocks.erase(retiredBlocks.begin());
        }
    }
    tableCache.push_back(c);
    out = c.tables;
    return EM2_OK;
}

// EM2_TIMING=1: wall time of the stages of the fused host-buffer call on stderr (measurements only).
struct CallTimer {
    bool on = getenv("EM2_TIMING") && getenv("EM2_TIMING")[0] == '1';
    std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    void stage(const char* name)
    {
        if (!on) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[em2 timing]   device call: %s %.1f ms\n", name, std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    }
};

// The host threads this process may keep busy: the hardware's, or the CPU quota of its control group when that is smaller
// (the GPU boxes of the pool show 256 hardware threads under a quota of 16 CPUs; threads beyond the quota get the whole
// process throttled for the rest of the scheduler's period, the thread that talks to the device included).
unsigned usableCpus()
{
    unsigned n = std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {               // cgroup v2: "<quota|max> <period>"
        char quota[32] = {0};
        long period = 0;
        if (std::fscanf(f, "%31s %ld", quota, &period) == 2 && period > 0 && quota[0] != 'm') {
            const long cpus = std::atol(quota) / period;
            if (cpus >= 1 && unsigned(cpus) < n) n = unsigned(cpus);
        }
        std::fclose(f);
    } else {
        long quota = -1, period = 0;                                           // cgroup v1
        if (FILE* q = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (std::fscanf(q, "%ld", &quota) != 1) quota = -1;
            std::fclose(q);
        }
        if (FILE* q = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (std::fscanf(q, "%ld", &period) != 1) period = 0;
            std::fclose(q);
        }
        if (quota > 0 && period > 0 && quota / period >= 1 && unsigned(quota / period) < n) n = unsigned(quota / period);
    }
    return n;
}

// RAII device allocation for the host-buffer entry points.
struct DeviceBuffer {
    void* p = nullptr;
    size_t cachedBytes = 0;          // allocateCached: the block's size in the library's scratch cache
    bool idle = false;               // the owner sets it when the device has finished with the block (a synchronous copy came back)
    ~DeviceBuffer()
    {
        if (!p) return;
        if (cachedBytes && idle) em2::scratchGive(p, cachedBytes);
        else (void)hipFree(p);
    }
    hipError_t allocate(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    // from the scratch cache of the process when it holds a block of that size (em2_fsp5.hip: capped, em2_dev_release_scratch()
    // frees it), and back into it when the call completed
    hipError_t allocateCached(size_t bytes)
    {
        bytes = bytes ? bytes : 1;
        p = em2::scratchTake(bytes, &cachedBytes);
        if (p) return hipSuccess;
        cachedBytes = bytes;
        const auto t0 = std::chrono::steady_clock::now();
        const hipError_t e = hipMalloc(&p, bytes);
        if (getenv("EM2_TIMING")) fprintf(stderr, "[em2 timing] hipMalloc of %zu bytes (not in the scratch cache): %.1f ms, %s\n", bytes,
                                          std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), e == hipSuccess ? "ok" : "FAILED");
        if (e != hipSuccess) {
            (void)hipGetLastError();
            em2::fsp5ReleaseScratch();          // (memory held by the cache may be what is missing)
            return hipMalloc(&p, bytes);
        }
        return e;
    }
    // (the caller knows the device has finished with the block)
    void release()
    {
        if (!p) return;
        if (cachedBytes) em2::scratchGive(p, cachedBytes);
        else (void)hipFree(p);
        p = nullptr;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// a piece of a larger device block
struct Piece {
    void* p = nullptr;
    template <class T> T* as() const { return static_cast<T*>(p); }
};

}  // namespace


extern "C" {

// Used by em2_matrix_capi.cpp (same shared object) to report through the same thread-local slot.
void em2_internal_set_last_error(const char* message) { lastError = message ? message : ""; }

int em2_abi_version(void) { return 1; }

const char* em2_last_error(void) { return lastError.c_str(); }


int em2_lsh_generate_vectors(uint32_t geneCount, uint32_t lshCount, uint32_t seed, double* vectors)
{
    if (!vectors || lshCount == 0) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_lsh_generate_vectors: bad argument");
    // boost::mt19937 has the parameters of std::mt19937.  uniform_01<double> over a 32-bit engine is
    // eng() / 2^32; normal_distribution (Boost <= 1.55) draws two uniforms per PAIR of variates and returns
    // rho*cos(2 pi r1) first, rho*sin(2 pi r1) second, rho = sqrt(-2 log(1 - r2)).
    // The engine is inherently sequential and cheap; the libm calls are neither, so the raw draws are produced
    // first (gene-major, bit-minor draw order, Lsh.cpp:90-101) and turned into variates by all host threads.  Every
    // value and every sum is formed by the same operations in the same order as in the one-thread loop.
    const size_t total = size_t(geneCount) * lshCount;
    const size_t pairCount = (total + 1) / 2;
    std::vector<uint32_t> raw(2 * pairCount);
    unsigned threads = usableCpus();
    if (threads > 32) threads = 32;
    if (threads < 1 || total < (1u << 16)) threads = 1;
    auto parallel = [&](size_t n, const std::function<void(size_t, size_t)>& body) {
        if (threads == 1) {
            body(0, n);
            return;
        }
        std::vector<std::thread> pool;
        const size_t per = (n + threads - 1) / threads;
        for (unsigned t = 0; t < threads; t++) {
            const size_t begin = std::min(n, size_t(t) * per), end = std::min(n, begin + per);
            if (begin < end) pool.emplace_back(body, begin, end);
        }
        for (std::thread& th : pool) th.join();
    };
    const double scale = 1.0 / 4294967296.0;
    const double twoPi = 2.0 * 3.14159265358979323846;
    auto variates = [&](size_t begin, size_t end) {
        for (size_t p = begin; p < end; p++) {
            const double r1 = double(raw[2 * p]) * scale;
            const double r2 = double(raw[2 * p + 1]) * scale;
            const double rho = std::sqrt(-2.0 * std::log(1.0 - r2));
            vectors[2 * p] = rho * std::cos(twoPi * r1);
            if (2 * p + 1 < total) vectors[2 * p + 1] = rho * std::sin(twoPi * r1);
        }
    };
    // The engine, 624 values at a time: the three loops of a state update have no dependence closer than 227 elements and
    // the compiler vectorises them (25 ms for the 3.1e7 draws of 30 000 genes x 1 024 bits where std::mt19937, one value
    // per call, takes 72; tests/test_capi_cpu.py holds it to std::mt19937).  The variates are formed behind it, part by part.
    struct Mt19937Blocks {
        uint32_t s[624];
        explicit Mt19937Blocks(uint32_t seed)
        {
            s[0] = seed;
            for (uint32_t i = 1; i < 624; ++i) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + i;
        }
        static inline uint32_t mix(uint32_t a, uint32_t b)
        {
            const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
            return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        void next(uint32_t* out, size_t count)          // the next count <= 624 values
        {
            for (int i = 0; i < 227; ++i) s[i] = s[i + 397] ^ mix(s[i], s[i + 1]);
            for (int i = 227; i < 623; ++i) s[i] = s[i - 227] ^ mix(s[i], s[i + 1]);
            s[623] = s[396] ^ mix(s[623], s[0]);
            for (size_t i = 0; i < count; ++i) {
                uint32_t y = s[i];
                y ^= y >> 11;
                y ^= (y << 7) & 0x9d2c5680u;
                y ^= (y << 15) & 0xefc60000u;
                y ^= y >> 18;
                out[i] = y;
            }
        }
    };
    {
        // The engine runs on the calling thread; the variates of what it has produced are formed by threads - 1 workers that take
        // ranges of 16 384 pairs from a queue and sleep when it is empty -- never more busy threads than usableCpus(), and none
        // that spins.
        const size_t rawCount = 2 * pairCount;
        constexpr size_t kTaskPairs = size_t(1) << 14;
        std::mutex mutex;
        std::condition_variable wake;
        size_t ready = 0, next = 0;          // pairs whose raw values exist / pairs handed out (both multiples of kTaskPairs or pairCount)
        bool done = false;
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < threads; t++) {
            pool.emplace_back([&]() {
                for (;;) {
                    size_t begin, end;
                    {
                        std::unique_lock<std::mutex> lock(mutex);
                        wake.wait(lock, [&]() { return next < ready || done; });
                        if (next >= ready) return;
                        begin = next;
                        end = std::min(ready, begin + kTaskPairs);
                        next = end;
                    }
                    variates(begin, end);
                }
            });
        }
        Mt19937Blocks engine(seed);
        size_t produced = 0, announced = 0;          // raw values written / pairs announced to the workers
        while (produced < rawCount) {
            const size_t count = std::min(size_t(624), rawCount - produced);
            engine.next(raw.data() + produced, count);
            produced += count;
            const size_t whole = produced == rawCount ? pairCount : (produced / 2) / kTaskPairs * kTaskPairs;
            if (threads > 1 && whole > announced) {
                {
                    std::lock_guard<std::mutex> lock(mutex);
                    ready = whole;
                }
                wake.notify_all();
                announced = whole;
            }
        }
        if (threads == 1) {
            variates(0, pairCount);
        } else {
            // the calling thread helps with what is left, then lets the workers go
            for (;;) {
                size_t begin, end;
                {
                    std::lock_guard<std::mutex> lock(mutex);
                    if (next >= ready) break;
                    begin = next;
                    end = std::min(ready, begin + kTaskPairs);
                    next = end;
                }
                variates(begin, end);
            }
            {
                std::lock_guard<std::mutex> lock(mutex);
                done = true;
            }
            wake.notify_all();
            for (std::thread& th : pool) th.join();
        }
    }
    std::vector<double> sumOfSquares(lshCount, 0.);
    parallel(lshCount, [&](size_t begin, size_t end) {          // per bit: genes in ascending order
        for (size_t g = 0; g < geneCount; g++) {
            const double* row = vectors + g * lshCount;
            for (size_t i = begin; i < end; i++) sumOfSquares[i] += row[i] * row[i];
        }
    });
    for (double& f : sumOfSquares) f = 1. / std::sqrt(f);                    // Lsh.cpp:104-106
    parallel(geneCount, [&](size_t begin, size_t end) {                      // Lsh.cpp:107-111
        for (size_t g = begin; g < end; g++) {
            double* row = vectors + g * lshCount;
            for (size_t i = 0; i < lshCount; i++) row[i] *= sumOfSquares[i];
        }
    });
    return EM2_OK;
}


int em2_lsh_similarity_table(uint32_t lshCount, double* table)
{
    if (!table || lshCount == 0) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_lsh_similarity_table: bad argument");
    em2::computeSimilarityTable(lshCount, table);
    return EM2_OK;
}


uint64_t em2_murmur_hash_64a(const void* key, int len, uint64_t seed)
{
    // MurmurHash64A, Austin Appleby, public domain; 64-bit little-endian form.
    const uint64_t mul = 0xc6a4a7935bd1e995ULL;
    const int shift = 47;
    const unsigned char* bytes = static_cast<const unsigned char*>(key);
    uint64_t hash = seed ^ (uint64_t(len) * mul);
    const int blocks = len / 8;
    for (int b = 0; b < blocks; b++) {
        uint64_t v;
        std::memcpy(&v, bytes + size_t(b) * 8, 8);
        v *= mul;
        v ^= v >> shift;
        v *= mul;
        hash ^= v;
        hash *= mul;
    }
    const unsigned char* tail = bytes + size_t(blocks) * 8;
    const int rest = len & 7;
    for (int i = rest - 1; i >= 0; i--) hash ^= uint64_t(tail[i]) << (8 * i);
    if (rest) hash *= mul;
    hash ^= hash >> shift;
    hash *= mul;
    hash ^= hash >> shift;
    return hash;
}


int em2_device_count(int* count)
{
    if (!count) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_device_count: null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return EM2_OK;
}


int em2_set_device(int device)
{
    if (!haveDevice()) return fail(EM2_ERROR_NO_DEVICE, "no HIP device is visible");
    EM2_HIP(hipSetDevice(device));
    return EM2_OK;
}


// ---------------------------------------------------------------------------------------------------------
// Device-resident entry points.
// ---------------------------------------------------------------------------------------------------------

size_t em2_dev_compute_signatures_workspace(uint32_t cellCount, uint32_t lshCount)
{
    const size_t exact = alignUp(size_t(cellCount) * sizeof(double)) + alignUp(size_t(lshCount) * sizeof(double));
    const size_t screened = lshCount ? em2::projectionScreenedWorkspaceBytes(cellCount, lshCount) : 0;
    return (exact > screened ? exact : screened) + 256;
}


size_t em2_dev_vector_aux_bytes(uint32_t geneCount, uint32_t lshCount)
{
    return alignUp(em2::vectorAuxBytes(geneCount, lshCount));
}


int em2_dev_prepare_vectors(const double* d_vectors, uint32_t geneCount, uint32_t lshCount, void* d_vectorAux, void* stream)
{
    if (!d_vectors || !d_vectorAux || lshCount == 0) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_prepare_vectors: bad argument");
    EM2_HIP(em2::launchPrepareVectors(d_vectors, geneCount, lshCount, d_vectorAux, static_cast<hipStream_t>(stream)));
    return EM2_OK;
}


int em2_dev_compute_signatures(const uint64_t* d_toc, const em2_count* d_data, uint32_t cellCount,
                               uint32_t geneCount, const double* d_vectors, const void* d_vectorAux,
                               uint32_t lshCount, uint64_t* d_signatures, void* d_workspace,
                               size_t workspaceBytes, void* stream)
{
    if (lshCount == 0 || geneCount == 0) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_compute_signatures: lshCount and geneCount must be positive");
    if (cellCount == 0) return EM2_OK;
    if (!d_toc || !d_vectors || !d_signatures || !d_workspace) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_compute_signatures: null pointer");
    if (workspaceBytes < em2_dev_compute_signatures_workspace(cellCount, lshCount)) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_compute_signatures: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    void* ws = reinterpret_cast<void*>(alignUp(reinterpret_cast<size_t>(d_workspace)));
    const em2::CountIn* data = reinterpret_cast<const em2::CountIn*>(d_data);
    if (d_vectorAux && lshCount % 4u == 0u) {
        // screening pass on a float copy of the hyperplanes + exact recomputation of the undecided words
        EM2_HIP(em2::launchProjectionScreened(d_toc, data, cellCount, geneCount, d_vectors, d_vectorAux, lshCount,
                                              d_signatures, ws, s));
        return EM2_OK;
    }
    double* means = static_cast<double*>(ws);
    double* sums = reinterpret_cast<double*>(static_cast<char*>(ws) + alignUp(size_t(cellCount) * sizeof(double)));
    const double* vectorSums = static_cast<const double*>(d_vectorAux);      // the aux block starts with the sums
    if (!vectorSums) {
        EM2_HIP(em2::launchVectorSums(d_vectors, geneCount, lshCount, sums, s));
        vectorSums = sums;
    }
    EM2_HIP(em2::launchCellMeans(d_toc, data, cellCount, geneCount, means, s));
    EM2_HIP(em2::launchProjection(d_toc, data, cellCount, d_vectors, vectorSums, means, lshCount, d_signatures, s));
    return EM2_OK;
}


int em2_dev_compute_signatures_tier(const void* d_workspace, uint32_t cellCount, uint32_t lshCount, int haveVectorAux, int* tier)
{
    if (!tier || lshCount == 0) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_compute_signatures_tier: bad argument");
    *tier = EM2_TIER_EXACT;
    if (!haveVectorAux || lshCount % 4u != 0u || cellCount == 0) return EM2_OK;
    if (lshCount % 64u != 0u) {
        *tier = EM2_TIER_FLOAT;
        return EM2_OK;
    }
    // (the flag the statistics kernel raises when some count is no small integer: word 48 of the counters behind the two
    // per-cell arrays of the workspace, csrc/em2_project.hip: launchProjectionScreened)
    if (!d_workspace) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_compute_signatures_tier: null workspace");
    const char* ws = reinterpret_cast<const char*>(alignUp(reinterpret_cast<size_t>(d_workspace)));
    const size_t a = (size_t(cellCount) * sizeof(double) + 255u) & ~size_t(255u);
    uint32_t notAllInteger = 0;
    // (every stream of the device, non-blocking ones included: the call has no stream of its own, and a copy on the null
    // stream does not order behind a projection launched on a stream created with hipStreamNonBlocking)
    EM2_HIP(hipDeviceSynchronize());
    EM2_HIP(hipMemcpy(&notAllInteger, ws + 2u * a + 48u * sizeof(uint32_t), sizeof(uint32_t), hipMemcpyDeviceToHost));
    *tier = notAllInteger ? EM2_TIER_FIXED16_FLOAT : EM2_TIER_FIXED16_INTEGER;
    return EM2_OK;
}


size_t em2_dev_find_similar_pairs4_workspace(uint32_t cellCount, uint32_t rowCount, uint32_t lshCount, uint32_t k)
{
    if (lshCount == 0) return 0;
    const uint32_t padded = em2::paddedDwords(lshCount);
    size_t bytes = alignUp(size_t(rowCount) * 2u * k * sizeof(em2::Entry));
    if (padded != 2u * wordCountOf(lshCount)) bytes += alignUp(size_t(cellCount) * padded * sizeof(uint32_t));
    bytes += alignUp(em2::fsp4ControlBytes(rowCount));
    bytes += alignUp(em2::fsp4SymmetricBytes(cellCount, rowCount, padded));
    return bytes + 256;
}


int em2_dev_find_similar_pairs4_form(uint32_t cellCount, uint32_t rowCount)
{
    return em2::fsp4UsesSymmetricScan(cellCount, rowCount, 0) ? 1 : 0;
}


int em2_dev_find_similar_pairs4_form_for(uint32_t cellCount, uint32_t rowCount, uint32_t lshCount)
{
    if (lshCount == 0) return 0;
    const uint32_t padded = em2::paddedDwords(lshCount);
    if (!em2::fsp4UsesSymmetricScan(cellCount, rowCount, padded)) return em2::fsp4UsesRowsMatrixScan(cellCount, rowCount, padded) ? 4 : 0;
    return em2::fsp4MatrixFormWanted(padded) ? 3 : 1;
}


int em2_dev_find_similar_pairs4_last_launch(double* values, uint32_t valueCount)
{
    if (!values && valueCount) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_find_similar_pairs4_last_launch: null pointer");
    const em2::Fsp4LaunchInfo info = em2::fsp4LastLaunchInfo();
    const double all[9] = {double(info.form), info.scanKernelMs, info.waveColumnSteps, info.inboxEntries, info.segments, info.fullRowCells,
                           info.matrixPairs, info.matrixKernelMs, info.matrixClockGHz};
    for (uint32_t i = 0; i < valueCount; i++) values[i] = i < 9 ? all[i] : 0.0;
    return EM2_OK;
}


void em2_dev_release_scratch(void) { em2::fsp5ReleaseScratch(); }

int em2_dev_find_similar_pairs5_last_launch(double* values, uint32_t valueCount)
{
    if (!values && valueCount) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_find_similar_pairs5_last_launch: null pointer");
    const em2::Fsp5LaunchInfo info = em2::fsp5LastLaunchInfo();
    const double all[7] = {info.gatheredCandidates, info.cells, info.sliceCount, info.batches, info.filterMs, info.selectMs,
                           info.distinctCandidates};
    for (uint32_t i = 0; i < valueCount; i++) values[i] = i < 7 ? all[i] : 0.0;
    return EM2_OK;
}

int em2_dev_find_similar_pairs4(const uint64_t* d_signatures, uint32_t cellCount, uint32_t rowBegin,
                                uint32_t rowEnd, uint32_t lshCount, uint32_t k, double similarityThreshold,
                                em2_pair* d_pairs, uint32_t* d_usedCount, void* d_workspace,
                                size_t workspaceBytes, void* stream)
{
    if (lshCount == 0) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_find_similar_pairs4: lshCount must be positive");
    if (rowBegin > rowEnd || rowEnd > cellCount) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_find_similar_pairs4: bad row range");
    if (rowBegin == rowEnd) return EM2_OK;
    if (!d_signatures || !d_usedCount || (!d_pairs && k) || !d_workspace) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_find_similar_pairs4: null pointer");
    const uint32_t rows = rowEnd - rowBegin;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (k == 0) {       // keepBest(v, 0) empties every list (src/heap.hpp:116-126)
        EM2_HIP(hipMemsetAsync(d_usedCount, 0, size_t(rows) * sizeof(uint32_t), s));
        return EM2_OK;
    }
    const uint32_t padded = em2::paddedDwords(lshCount);
    if (padded == 0) return fail(EM2_ERROR_UNSUPPORTED, "em2_dev_find_similar_pairs4: lshCount above 4096 is not supported");
    if (k > em2::fsp4MaxK()) return fail(EM2_ERROR_UNSUPPORTED, "em2_dev_find_similar_pairs4: k above " + std::to_string(em2::fsp4MaxK()) + " is not supported");
    if (workspaceBytes < em2_dev_find_similar_pairs4_workspace(cellCount, rows, lshCount, k)) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_find_similar_pairs4: workspace too small");

    em2::DeviceTables tables;
    const int rc = getDeviceTables(lshCount, similarityThreshold, tables);
    if (rc != EM2_OK) return rc;

    char* ws = reinterpret_cast<char*>(alignUp(reinterpret_cast<size_t>(d_workspace)));
    em2::Entry* buffers = reinterpret_cast<em2::Entry*>(ws);
    ws += alignUp(size_t(rows) * 2u * k * sizeof(em2::Entry));
    void* control = ws;
    ws += alignUp(em2::fsp4ControlBytes(rows));
    const uint32_t words = wordCountOf(lshCount);
    const uint32_t* sig32 = reinterpret_cast<const uint32_t*>(d_signatures);
    if (padded != 2u * words) {
        uint32_t* repacked = reinterpret_cast<uint32_t*>(ws);
        EM2_HIP(em2::launchRepackSignatures(d_signatures, cellCount, words, repacked, padded, s));
        sig32 = repacked;
        ws += alignUp(size_t(cellCount) * padded * sizeof(uint32_t));
    }
    void* symmetricWs = em2::fsp4SymmetricBytes(cellCount, rows, padded) ? ws : nullptr;
    EM2_HIP(em2::launchFsp4Scan(sig32, padded, cellCount, rowBegin, rowEnd, k, tables, buffers,
                                reinterpret_cast<em2::PairOut*>(d_pairs), d_usedCount, control, s, symmetricWs));
    return EM2_OK;
}


// ---- sharded symmetric scan: one call per phase, the collectives between them are the caller's ----

static size_t shardedRepackBytes(uint32_t cellCount, uint32_t lshCount)
{
    const uint32_t padded = em2::paddedDwords(lshCount);
    return padded != 2u * wordCountOf(lshCount) ? alignUp(size_t(cellCount) * padded * sizeof(uint32_t)) : 0;
}

int em2_dev_fsp4_sharded_plan(uint32_t cellCount, uint32_t lshCount, uint32_t k, uint32_t rank, uint32_t world,
                              uint64_t* values, uint32_t valueCount)
{
    if (!values && valueCount) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_fsp4_sharded_plan: null pointer");
    uint64_t all[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t padded = lshCount ? em2::paddedDwords(lshCount) : 0;
    if (padded != 0 && k != 0 && k <= em2::fsp4MaxK() && world >= 1 && rank < world) {
        const em2::Fsp4ShardPlan plan = em2::fsp4ShardPlan(cellCount, k, rank, world);
        if (plan.eligible) {
            all[0] = 1;
            all[1] = alignUp(plan.totalBytes) + shardedRepackBytes(cellCount, lshCount) + 256;
            all[2] = plan.offSnap;
            all[3] = plan.offPool;
            all[4] = plan.capLocal;
            all[5] = plan.offGathered;
            all[6] = plan.capGathered;
            all[7] = plan.prefixCells;
            all[8] = plan.ownBlocks;
            all[9] = plan.blocks;
            all[10] = plan.offSorted;
            uint32_t rowBits = 1;
            while ((1ull << rowBits) < uint64_t(cellCount)) ++rowBits;
            all[11] = 13u + rowBits + 6u;
        }
    }
    for (uint32_t i = 0; i < valueCount; i++) values[i] = i < 12 ? all[i] : 0;
    return EM2_OK;
}

int em2_dev_fsp4_sharded_phase(int phase, const uint64_t* d_signatures, uint32_t cellCount, uint32_t lshCount, uint32_t k,
                               double similarityThreshold, uint32_t rank, uint32_t world, em2_pair* d_pairs,
                               uint32_t* d_usedCount, void* d_workspace, size_t workspaceBytes, uint64_t gatheredCount,
                               void* stream)
{
    if (!d_signatures || !d_pairs || !d_usedCount || !d_workspace) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_fsp4_sharded_phase: null pointer");
    if (phase < 0 || phase > 4) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_fsp4_sharded_phase: phase must be 0..4");
    uint64_t values[2] = {0, 0};
    em2_dev_fsp4_sharded_plan(cellCount, lshCount, k, rank, world, values, 2);
    if (!values[0]) return fail(EM2_ERROR_UNSUPPORTED, "em2_dev_fsp4_sharded_phase: this shape is not eligible for the sharded symmetric scan");
    if ((reinterpret_cast<size_t>(d_workspace) & 255u) != 0) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_fsp4_sharded_phase: the workspace must be 256-byte aligned");
    if (workspaceBytes < values[1]) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_fsp4_sharded_phase: workspace too small");
    em2::DeviceTables tables;
    const int rc = getDeviceTables(lshCount, similarityThreshold, tables);
    if (rc != EM2_OK) return rc;
    const em2::Fsp4ShardPlan plan = em2::fsp4ShardPlan(cellCount, k, rank, world);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t padded = em2::paddedDwords(lshCount);
    const uint32_t words = wordCountOf(lshCount);
    const uint32_t* sig32 = reinterpret_cast<const uint32_t*>(d_signatures);
    if (padded != 2u * words) {
        uint32_t* repacked = reinterpret_cast<uint32_t*>(static_cast<char*>(d_workspace) + alignUp(plan.totalBytes));
        if (phase == 0) EM2_HIP(em2::launchRepackSignatures(d_signatures, cellCount, words, repacked, padded, s));
        sig32 = repacked;
    }
    EM2_HIP(em2::launchFsp4ShardPhase(plan, phase, sig32, padded, tables, d_workspace,
                                      static_cast<char*>(d_workspace) + plan.rankBytes,        // gathered / sorted / temp areas
                                      reinterpret_cast<em2::PairOut*>(d_pairs), d_usedCount, gatheredCount, s));
    return EM2_OK;
}

int em2_dev_fsp4_sharded_status(uint32_t cellCount, uint32_t k, uint32_t rank, uint32_t world, const void* d_workspace,
                                void* stream, uint64_t* usedEntries, uint32_t* overflow)
{
    if (!d_workspace || !usedEntries || !overflow) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_fsp4_sharded_status: null pointer");
    const em2::Fsp4ShardPlan plan = em2::fsp4ShardPlan(cellCount, k, rank, world);
    if (!plan.eligible) return fail(EM2_ERROR_UNSUPPORTED, "em2_dev_fsp4_sharded_status: this shape is not eligible for the sharded symmetric scan");
    uint32_t error = 0;
    EM2_HIP(em2::readFsp4ShardStatus(plan, d_workspace, static_cast<hipStream_t>(stream), usedEntries, overflow, &error));
    if (error) return fail(EM2_ERROR_RUNTIME, "findSimilarPairs4: a segment hand-off between waves timed out; the result is incomplete");
    return EM2_OK;
}


int em2_dev_find_similar_pairs4_status(const void* d_workspace, uint32_t rowCount, uint32_t k, void* stream)
{
    if (!d_workspace) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_find_similar_pairs4_status: null workspace");
    if (rowCount == 0 || k == 0) return EM2_OK;
    const char* ws = reinterpret_cast<const char*>(alignUp(reinterpret_cast<size_t>(d_workspace)));
    ws += alignUp(size_t(rowCount) * 2u * k * sizeof(em2::Entry));
    uint32_t error = 0;
    EM2_HIP(em2::readFsp4Error(ws, rowCount, static_cast<hipStream_t>(stream), &error));
    if (error) return fail(EM2_ERROR_RUNTIME, "findSimilarPairs4: a segment hand-off between waves timed out; the result is incomplete");
    return EM2_OK;
}


int em2_dev_find_similar_pairs5(const uint64_t* d_signatures, uint32_t cellCount, uint32_t rowBegin,
                                uint32_t rowEnd, uint32_t lshCount, uint32_t k, double similarityThreshold,
                                uint32_t lshSliceLength, uint64_t bucketOverflow, em2_pair* d_pairs,
                                uint32_t* d_usedCount, void* stream)
{
    if (lshCount == 0) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_find_similar_pairs5: lshCount must be positive");
    if (lshSliceLength == 0 || lshSliceLength > 32) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_find_similar_pairs5: lshSliceLength must be in [1,32]");
    if (rowBegin > rowEnd || rowEnd > cellCount) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_find_similar_pairs5: bad row range");
    if (rowBegin == rowEnd) return EM2_OK;
    if (!d_signatures || !d_usedCount || (!d_pairs && k)) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_find_similar_pairs5: null pointer");
    em2::DeviceTables tables;
    const int rc = getDeviceTables(lshCount, similarityThreshold, tables);
    if (rc != EM2_OK) return rc;
    EM2_HIP(em2::runFsp5(d_signatures, cellCount, rowBegin, rowEnd, lshCount, k, lshSliceLength, bucketOverflow, tables,
                         reinterpret_cast<em2::PairOut*>(d_pairs), d_usedCount, static_cast<hipStream_t>(stream)));
    return EM2_OK;
}


// The argument checks of findSimilarPairs7 in the reference's order (src/ExpressionMatrixLsh.cpp:548-561) and
// Lsh::computeMismatchCountThresholdFromSimilarityThreshold (src/Lsh.hpp:86-95).
static int prepareFsp7(const char* who, uint32_t lshCount, double similarityThreshold, const int32_t* sliceLengths,
                       uint32_t sliceLengthCount, uint32_t k, uint32_t log2BucketCount, uint64_t& mismatchThreshold)
{
    if (lshCount == 0) return fail(EM2_ERROR_INVALID_ARGUMENT, std::string(who) + ": lshCount must be positive");
    if (sliceLengthCount && !sliceLengths) return fail(EM2_ERROR_INVALID_ARGUMENT, std::string(who) + ": null sliceLengths");
    for (uint32_t i = 1; i < sliceLengthCount; i++) {
        if (sliceLengths[i] >= sliceLengths[i - 1]) return fail(EM2_ERROR_RUNTIME, "The slice lengths are not in decreasing order.");
    }
    for (uint32_t i = 0; i < sliceLengthCount; i++) {
        if (sliceLengths[i] > 64) return fail(EM2_ERROR_RUNTIME, "Each slice length can be at most 64 bits.");
    }
    for (uint32_t i = 0; i < sliceLengthCount; i++) {
        // the reference divides lshBitCount by the slice length (:760): zero is a crash there, negatives nonsense
        if (sliceLengths[i] < 1) return fail(EM2_ERROR_INVALID_ARGUMENT, std::string(who) + ": slice lengths must be positive");
    }
    if (log2BucketCount > 40) return fail(EM2_ERROR_UNSUPPORTED, std::string(who) + ": log2BucketCount above 40 is not supported");
    for (uint32_t i = 0; i < sliceLengthCount; i++) {
        if (uint32_t(sliceLengths[i]) < log2BucketCount && sliceLengths[i] > 40) {
            return fail(EM2_ERROR_UNSUPPORTED, std::string(who) + ": directly indexed slices longer than 40 bits are not supported");
        }
    }
    if (k > em2::fsp7MaxK()) return fail(EM2_ERROR_UNSUPPORTED, std::string(who) + ": k above " + std::to_string(em2::fsp7MaxK()) + " is not supported");
    std::vector<double> table(size_t(lshCount) + 1);
    em2::computeSimilarityTable(lshCount, table.data());
    for (size_t m = 0; m < table.size(); m++) {
        if (table[m] < similarityThreshold) {
            mismatchThreshold = uint64_t(m) - 1u;            // size_t arithmetic as in Lsh.hpp:91 (m == 0 wraps)
            return EM2_OK;
        }
    }
    return fail(EM2_ERROR_RUNTIME, "Assertion failed: no mismatch count has a similarity below the similarity threshold (Lsh::computeMismatchCountThresholdFromSimilarityThreshold)");
}

int em2_dev_find_similar_pairs7(const uint64_t* d_signatures, uint32_t cellCount, uint32_t rowBegin, uint32_t rowEnd,
                                uint32_t lshCount, uint32_t k, double similarityThreshold, const int32_t* sliceLengths,
                                uint32_t sliceLengthCount, uint32_t maxCheck, uint32_t log2BucketCount, em2_pair* d_pairs,
                                uint32_t* d_usedCount, void* stream)
{
    uint64_t mismatchThreshold = 0;
    const int prc = prepareFsp7("em2_dev_find_similar_pairs7", lshCount, similarityThreshold, sliceLengths, sliceLengthCount, k,
                                log2BucketCount, mismatchThreshold);
    if (prc != EM2_OK) return prc;
    if (rowBegin > rowEnd || rowEnd > cellCount) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_find_similar_pairs7: bad row range");
    if (rowBegin == rowEnd) return EM2_OK;
    if (!d_signatures || !d_usedCount || (!d_pairs && k)) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_find_similar_pairs7: null pointer");
    em2::DeviceTables tables;
    const int rc = getDeviceTables(lshCount, similarityThreshold, tables);
    if (rc != EM2_OK) return rc;
    EM2_HIP(em2::runFsp7(d_signatures, cellCount, rowBegin, rowEnd, lshCount, k, sliceLengths, sliceLengthCount, maxCheck,
                         log2BucketCount, mismatchThreshold, tables, reinterpret_cast<em2::PairOut*>(d_pairs), d_usedCount,
                         static_cast<hipStream_t>(stream)));
    return EM2_OK;
}


// ---------------------------------------------------------------------------------------------------------
// Host-buffer entry points.
// ---------------------------------------------------------------------------------------------------------

int em2_find_similar_pairs7(const uint64_t* signatures, uint32_t cellCount, uint32_t lshCount, uint32_t k,
                            double similarityThreshold, const int32_t* sliceLengths, uint32_t sliceLengthCount,
                            uint32_t maxCheck, uint32_t log2BucketCount, em2_pair* pairs, uint32_t* usedCount)
{
    uint64_t mismatchThreshold = 0;
    const int prc = prepareFsp7("em2_find_similar_pairs7", lshCount, similarityThreshold, sliceLengths, sliceLengthCount, k,
                                log2BucketCount, mismatchThreshold);
    if (prc != EM2_OK) return prc;
    if (cellCount == 0) return EM2_OK;
    if (!signatures || !usedCount || (!pairs && k)) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_find_similar_pairs7: null pointer");
    if (!haveDevice()) return fail(EM2_ERROR_NO_DEVICE, "em2_find_similar_pairs7: no HIP device is visible (this library has no CPU path)");
    const uint32_t words = wordCountOf(lshCount);
    DeviceBuffer dSig, dPairs, dUsed;
    EM2_HIP(dSig.allocate(size_t(cellCount) * words * sizeof(uint64_t)));
    EM2_HIP(dPairs.allocate(size_t(cellCount) * k * sizeof(em2_pair)));
    EM2_HIP(dUsed.allocate(size_t(cellCount) * sizeof(uint32_t)));
    EM2_HIP(hipMemcpy(dSig.p, signatures, size_t(cellCount) * words * sizeof(uint64_t), hipMemcpyHostToDevice));
    const int rc = em2_dev_find_similar_pairs7(dSig.as<uint64_t>(), cellCount, 0, cellCount, lshCount, k, similarityThreshold,
                                               sliceLengths, sliceLengthCount, maxCheck, log2BucketCount, dPairs.as<em2_pair>(),
                                               dUsed.as<uint32_t>(), nullptr);
    if (rc != EM2_OK) return rc;
    if (k) EM2_HIP(hipMemcpy(pairs, dPairs.p, size_t(cellCount) * k * sizeof(em2_pair), hipMemcpyDeviceToHost));
    EM2_HIP(hipMemcpy(usedCount, dUsed.p, size_t(cellCount) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return EM2_OK;
}


size_t em2_dev_subset_workspace(uint32_t cellCount)
{
    return em2::subsetWorkspaceBytes(cellCount);
}

int em2_dev_subset_count(const uint64_t* d_globalToc, const em2_count* d_globalData, const uint32_t* d_cellIds,
                         uint32_t cellCount, const uint32_t* d_geneLocalIds, uint32_t globalGeneCount, uint64_t* d_toc,
                         void* d_workspace, size_t workspaceBytes, void* stream)
{
    if (!d_globalToc || !d_geneLocalIds || !d_toc || !d_workspace) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_subset_count: null pointer");
    if (workspaceBytes < em2::subsetWorkspaceBytes(cellCount)) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_subset_count: workspace too small");
    EM2_HIP(em2::launchSubsetCount(d_globalToc, reinterpret_cast<const em2::CountIn*>(d_globalData), d_cellIds, cellCount,
                                   d_geneLocalIds, globalGeneCount, d_toc, d_workspace, workspaceBytes,
                                   static_cast<hipStream_t>(stream)));
    return EM2_OK;
}

int em2_dev_subset_fill(const uint64_t* d_globalToc, const em2_count* d_globalData, const uint32_t* d_cellIds,
                        uint32_t cellCount, const uint32_t* d_geneLocalIds, uint32_t globalGeneCount, const uint64_t* d_toc,
                        em2_count* d_data, void* stream)
{
    if (!d_globalToc || !d_geneLocalIds || !d_toc) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_dev_subset_fill: null pointer");
    EM2_HIP(em2::launchSubsetFill(d_globalToc, reinterpret_cast<const em2::CountIn*>(d_globalData), d_cellIds, cellCount,
                                  d_geneLocalIds, globalGeneCount, d_toc, reinterpret_cast<em2::CountIn*>(d_data),
                                  static_cast<hipStream_t>(stream)));
    return EM2_OK;
}


static int subsetFindSimilarPairs4(const uint64_t* globalToc, const em2_count* globalData, uint32_t globalCellCount,
                                   const uint32_t* cellIds, uint32_t cellCount, const uint32_t* geneLocalIds,
                                   uint32_t globalGeneCount, uint32_t geneCount, const double* vectors,
                                   const double* (*vectorsWhenNeeded)(void*), void* vectorsContext, uint32_t lshCount,
                                   uint64_t* signatures, uint32_t k, double similarityThreshold, em2_pair* pairs,
                                   uint32_t* usedCount);

int em2_subset_find_similar_pairs4(const uint64_t* globalToc, const em2_count* globalData, uint32_t globalCellCount,
                                   const uint32_t* cellIds, uint32_t cellCount, const uint32_t* geneLocalIds,
                                   uint32_t globalGeneCount, uint32_t geneCount, const double* vectors, uint32_t lshCount,
                                   uint64_t* signatures, uint32_t k, double similarityThreshold, em2_pair* pairs,
                                   uint32_t* usedCount)
{
    if (!vectors && cellCount && lshCount && geneCount) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_subset_find_similar_pairs4: null pointer");
    return subsetFindSimilarPairs4(globalToc, globalData, globalCellCount, cellIds, cellCount, geneLocalIds, globalGeneCount, geneCount,
                                   vectors, nullptr, nullptr, lshCount, signatures, k, similarityThreshold, pairs, usedCount);
}

// The same call with the hyperplanes delivered WHEN THEY ARE NEEDED (behind the upload of the expression matrix and its subset):
// the facade (em2_host.cpp) draws them on a thread of its own meanwhile -- Lsh::generateLshVectors is 0.15 s of one host core at
// 30 000 genes x 1024 bits, the uploads in front of the projection about as long.  Internal to the library.
int em2_internal_subset_find_similar_pairs4(const uint64_t* globalToc, const em2_count* globalData, uint32_t globalCellCount,
                                            const uint32_t* cellIds, uint32_t cellCount, const uint32_t* geneLocalIds,
                                            uint32_t globalGeneCount, uint32_t geneCount,
                                            const double* (*vectorsWhenNeeded)(void*), void* vectorsContext, uint32_t lshCount,
                                            uint64_t* signatures, uint32_t k, double similarityThreshold, em2_pair* pairs,
                                            uint32_t* usedCount)
{
    if (!vectorsWhenNeeded) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_internal_subset_find_similar_pairs4: null pointer");
    return subsetFindSimilarPairs4(globalToc, globalData, globalCellCount, cellIds, cellCount, geneLocalIds, globalGeneCount, geneCount,
                                   nullptr, vectorsWhenNeeded, vectorsContext, lshCount, signatures, k, similarityThreshold, pairs, usedCount);
}

static int subsetFindSimilarPairs4(const uint64_t* globalToc, const em2_count* globalData, uint32_t globalCellCount,
                                   const uint32_t* cellIds, uint32_t cellCount, const uint32_t* geneLocalIds,
                                   uint32_t globalGeneCount, uint32_t geneCount, const double* vectors,
                                   const double* (*vectorsWhenNeeded)(void*), void* vectorsContext, uint32_t lshCount,
                                   uint64_t* signatures, uint32_t k, double similarityThreshold, em2_pair* pairs,
                                   uint32_t* usedCount)
{
    if (lshCount == 0 || geneCount == 0) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_subset_find_similar_pairs4: lshCount and geneCount must be positive");
    if (cellCount == 0) return EM2_OK;
    if (!globalToc || !geneLocalIds) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_subset_find_similar_pairs4: null pointer");
    const bool wantPairs = usedCount != nullptr;
    if (wantPairs && !pairs && k) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_subset_find_similar_pairs4: null pairs");
    if (!wantPairs && !signatures) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_subset_find_similar_pairs4: nothing to compute");
    if (!haveDevice()) return fail(EM2_ERROR_NO_DEVICE, "em2_subset_find_similar_pairs4: no HIP device is visible (this library has no CPU path)");
    const uint32_t padded = em2::paddedDwords(lshCount);
    if (wantPairs && k) {
        if (padded == 0) return fail(EM2_ERROR_UNSUPPORTED, "em2_subset_find_similar_pairs4: lshCount above 4096 is not supported");
        if (k > em2::fsp4MaxK()) return fail(EM2_ERROR_UNSUPPORTED, "em2_subset_find_similar_pairs4: k above " + std::to_string(em2::fsp4MaxK()) + " is not supported");
    }
    // The rows of the global CSR the cell set needs.  All cells in order (the usual case): the arrays go to the device
    // as they are.  Otherwise the rows are gathered on the host first (a copy of only those rows), still with global
    // gene ids; the gene restriction and the remapping happen on the device either way.
    bool allCells = cellCount == globalCellCount;
    if (cellIds) {
        for (uint32_t i = 0; i < cellCount; i++) {
            if (cellIds[i] >= globalCellCount) return fail(EM2_ERROR_RUNTIME, "em2_subset_find_similar_pairs4: the cell set refers to a cell that does not exist.");
            allCells = allCells && cellIds[i] == i;
        }
    } else if (!allCells) {
        return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_subset_find_similar_pairs4: null cellIds");
    }
    std::vector<uint64_t> rowToc;
    std::vector<em2_count> rowData;
    const uint64_t* srcToc = globalToc;
    const em2_count* srcData = globalData;
    if (!allCells) {
        rowToc.assign(size_t(cellCount) + 1, 0);
        for (uint32_t i = 0; i < cellCount; i++) rowToc[i + 1] = rowToc[i] + (globalToc[cellIds[i] + 1] - globalToc[cellIds[i]]);
        rowData.resize(rowToc[cellCount]);
        for (uint32_t i = 0; i < cellCount; i++) {
            const uint64_t n = rowToc[i + 1] - rowToc[i];
            if (n) std::memcpy(rowData.data() + rowToc[i], globalData + globalToc[cellIds[i]], n * sizeof(em2_count));
        }
        srcToc = rowToc.data();
        srcData = rowData.data();
    }
    const uint64_t srcNnz = srcToc[cellCount];
    if (srcNnz && !srcData) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_subset_find_similar_pairs4: null data");

    const uint32_t words = wordCountOf(lshCount);
    CallTimer timer;
    // ONE device block for the whole call, from the process's scratch cache (em2_fsp5.hip: capped, em2_dev_release_scratch()
    // frees it): the signatures first, then whatever the current phase needs -- the CSR, its subset, the hyperplanes and the
    // projection's workspace; then, in the same place, the result and the scan's workspace.  A hipMalloc of gigabytes took
    // 1.6-4 s in one call of twelve on the boxes of the pool, whichever allocation it hit, so every large buffer of the call has
    // come out of the cache since round 5 -- as a dozen blocks then, 36 GB kept between calls; as one block of the call's peak
    // now (13 GB at a million cells).
    // The pool of deferred candidates is most of the scan's workspace (16 bytes per entry and cell: the pool and its sorted copy).
    // This call starts with room for 512 entries per cell -- the bench's data use 240, 12 GB of workspace at a million cells --
    // and the process comes back for the 1024 of the device-level call (19 GB) once a launch has overflowed the smaller pool
    // (strongly clustered data: the launch then fell back on every row against all columns, correct but twice the work).
    static std::atomic<bool> wantsTheFullPool{false};
    struct PoolSize {
        PoolSize(uint32_t entries) { em2::fsp4SetInboxEntriesPerCell(entries); }
        ~PoolSize() { em2::fsp4SetInboxEntriesPerCell(0u); }
    } poolSize(wantsTheFullPool.load() ? 0u : 512u);
    const size_t subsetWs = em2::subsetWorkspaceBytes(cellCount);
    const size_t wsBytes = em2_dev_compute_signatures_workspace(cellCount, lshCount);
    const size_t auxBytes = lshCount % 4u == 0u ? em2_dev_vector_aux_bytes(geneCount, lshCount) : 0;
    const size_t scanWsBytes = wantPairs && k ? em2_dev_find_similar_pairs4_workspace(cellCount, cellCount, lshCount, k) : 0;
    const size_t sigBytes = alignUp(size_t(cellCount) * words * sizeof(uint64_t));
    const size_t firstPhase = 2u * alignUp((size_t(cellCount) + 1) * sizeof(uint64_t)) + 2u * alignUp(srcNnz * sizeof(em2_count)) +
                              alignUp(size_t(globalGeneCount) * sizeof(uint32_t)) + alignUp(subsetWs) +
                              alignUp(size_t(geneCount) * lshCount * sizeof(double)) + alignUp(wsBytes) + alignUp(auxBytes) + 4096u;
    const size_t secondPhase = wantPairs && k ? alignUp(size_t(cellCount) * k * sizeof(em2_pair)) + alignUp(size_t(cellCount) * sizeof(uint32_t)) +
                                                    alignUp(scanWsBytes) + 4096u : 0;
    DeviceBuffer arena;
    EM2_HIP(arena.allocateCached(sigBytes + (firstPhase > secondPhase ? firstPhase : secondPhase) + 256u));
    size_t arenaAt = alignUp(reinterpret_cast<size_t>(arena.p)) - reinterpret_cast<size_t>(arena.p);
    auto carve = [&](size_t bytes) {
        Piece piece;
        piece.p = static_cast<char*>(arena.p) + arenaAt;
        arenaAt += alignUp(bytes ? bytes : 1);
        return piece;
    };
    const Piece dSig = carve(size_t(cellCount) * words * sizeof(uint64_t));
    const size_t phaseBegin = arenaAt;
    const Piece dSrcToc = carve((size_t(cellCount) + 1) * sizeof(uint64_t)), dSrcData = carve(srcNnz * sizeof(em2_count));
    const Piece dLocal = carve(size_t(globalGeneCount) * sizeof(uint32_t)), dToc = carve((size_t(cellCount) + 1) * sizeof(uint64_t));
    const Piece dSubsetWs = carve(subsetWs);
    EM2_HIP(hipMemcpy(dSrcToc.p, srcToc, (size_t(cellCount) + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    if (srcNnz) EM2_HIP(hipMemcpy(dSrcData.p, srcData, srcNnz * sizeof(em2_count), hipMemcpyHostToDevice));
    if (globalGeneCount) EM2_HIP(hipMemcpy(dLocal.p, geneLocalIds, size_t(globalGeneCount) * sizeof(uint32_t), hipMemcpyHostToDevice));
    std::vector<uint64_t>().swap(rowToc);
    std::vector<em2_count>().swap(rowData);
    timer.stage("allocate + CSR to device");
    EM2_HIP(em2::launchSubsetCount(dSrcToc.as<uint64_t>(), dSrcData.as<em2::CountIn>(), nullptr, cellCount, dLocal.as<uint32_t>(),
                                   globalGeneCount, dToc.as<uint64_t>(), dSubsetWs.p, subsetWs, nullptr));
    uint64_t nnz = 0;
    EM2_HIP(hipMemcpy(&nnz, dToc.as<uint64_t>() + cellCount, sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (nnz > srcNnz) return fail(EM2_ERROR_RUNTIME, "em2_subset_find_similar_pairs4: the subset holds more counts than its source");
    const Piece dData = carve(nnz * sizeof(em2_count));
    EM2_HIP(em2::launchSubsetFill(dSrcToc.as<uint64_t>(), dSrcData.as<em2::CountIn>(), nullptr, cellCount, dLocal.as<uint32_t>(),
                                  globalGeneCount, dToc.as<uint64_t>(), dData.as<em2::CountIn>(), nullptr));
    EM2_HIP(hipStreamSynchronize(nullptr));
    timer.stage("subset");

    // signatures (same steps as em2_compute_signatures, on the device-resident subset)
    const Piece dVectors = carve(size_t(geneCount) * lshCount * sizeof(double)), dWs = carve(wsBytes);
    if (!vectors) {
        vectors = vectorsWhenNeeded(vectorsContext);            // (waits for the thread that draws them)
        if (!vectors) return fail(EM2_ERROR_RUNTIME, em2_last_error());
        timer.stage("wait for the hyperplanes");
    }
    EM2_HIP(hipMemcpy(dVectors.p, vectors, size_t(geneCount) * lshCount * sizeof(double), hipMemcpyHostToDevice));
    void* aux = nullptr;
    if (lshCount % 4u == 0u) {          // (other widths: the exact arithmetic only)
        const Piece dAux = carve(auxBytes);
        const int prc = em2_dev_prepare_vectors(dVectors.as<double>(), geneCount, lshCount, dAux.p, nullptr);
        if (prc != EM2_OK) return prc;
        aux = dAux.p;
    }
    int rc = em2_dev_compute_signatures(dToc.as<uint64_t>(), dData.as<em2_count>(), cellCount, geneCount, dVectors.as<double>(),
                                        aux, lshCount, dSig.as<uint64_t>(), dWs.p, wsBytes, nullptr);
    if (rc != EM2_OK) return rc;
    EM2_HIP(hipStreamSynchronize(nullptr));
    timer.stage("hyperplanes to device + projection");
    if (signatures) EM2_HIP(hipMemcpy(signatures, dSig.p, size_t(cellCount) * words * sizeof(uint64_t), hipMemcpyDeviceToHost));
    arena.idle = true;          // (every use of the block so far has been waited for; a failure below leaves it to the cache all the same)
    if (!wantPairs) return EM2_OK;
    if (k == 0) {
        std::memset(usedCount, 0, size_t(cellCount) * sizeof(uint32_t));
        return EM2_OK;
    }
    // the scan's buffers take the place of everything but the signatures (the stream was synchronised behind the projection)
    arena.idle = false;
    arenaAt = phaseBegin;
    const Piece dPairs = carve(size_t(cellCount) * k * sizeof(em2_pair)), dUsed = carve(size_t(cellCount) * sizeof(uint32_t));
    const Piece dScanWs = carve(scanWsBytes);
    timer.stage("allocate result + workspace");
    // The pages of the result, touched while the device scans.  ExpressionMatrix.findSimilarPairs4 hands over the mapping of a
    // `-Pairs` file it has just created (800 MB at 1M cells, k = 100): the copy at the end of this call took 137-161 ms into those
    // untouched pages and 14 ms into touched ones (same box, hipMemcpy both times; host threads copying in parallel out of pinned
    // buffers made it 230: the page faults of one file do not run side by side).  One host thread writes a zero into every page
    // -- about 125 ms, under the 190 ms of the scan, and only there: started at the top of the call its page faults ran against
    // the device allocations' and releases' changes to the address space, and one call in three took a second longer.  The
    // caller's buffer is the caller's to read only after the call; a failed scan leaves zeros in the pages that were reached.
    struct PageToucher {
        std::thread thread;
        ~PageToucher() { if (thread.joinable()) thread.join(); }
    } toucher;
    if (size_t(cellCount) * k * sizeof(em2_pair) >= (size_t(64) << 20)) {
        volatile char* bytes = reinterpret_cast<volatile char*>(pairs);
        const size_t size = size_t(cellCount) * k * sizeof(em2_pair);
        toucher.thread = std::thread([bytes, size]() {
            for (size_t at = 0; at < size; at += 4096) bytes[at] = 0;
            bytes[size - 1] = 0;
        });
    }
    rc = em2_dev_find_similar_pairs4(dSig.as<uint64_t>(), cellCount, 0, cellCount, lshCount, k, similarityThreshold,
                                     dPairs.as<em2_pair>(), dUsed.as<uint32_t>(), dScanWs.p, scanWsBytes, nullptr);
    if (rc != EM2_OK) return rc;
    rc = em2_dev_find_similar_pairs4_status(dScanWs.p, cellCount, k, nullptr);
    if (rc != EM2_OK) return rc;
    // (all rows in one launch that ended as "rows x all columns on the matrix cores": the symmetric scan's pool overflowed)
    if (em2::fsp4LastLaunchInfo().form == 4 && em2::fsp4UsesSymmetricScan(cellCount, cellCount, em2::paddedDwords(lshCount))) wantsTheFullPool.store(true);
    timer.stage("scan");
    if (toucher.thread.joinable()) toucher.thread.join();
    timer.stage("wait for the result's pages");
    EM2_HIP(hipMemcpy(pairs, dPairs.p, size_t(cellCount) * k * sizeof(em2_pair), hipMemcpyDeviceToHost));
    EM2_HIP(hipMemcpy(usedCount, dUsed.p, size_t(cellCount) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    timer.stage("pairs to host");
    arena.idle = true;          // (the copies above were synchronous: the device is done with the block)
    return EM2_OK;
}


int em2_compute_signatures(const uint64_t* toc, const em2_count* data, uint32_t cellCount, uint32_t geneCount,
                           const double* vectors, uint32_t lshCount, uint64_t* signatures)
{
    if (lshCount == 0 || geneCount == 0) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_compute_signatures: lshCount and geneCount must be positive");
    if (cellCount == 0) return EM2_OK;
    if (!toc || !vectors || !signatures) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_compute_signatures: null pointer");
    if (!haveDevice()) return fail(EM2_ERROR_NO_DEVICE, "em2_compute_signatures: no HIP device is visible (this library has no CPU path)");
    const uint64_t nnz = toc[cellCount];
    if (nnz && !data) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_compute_signatures: null data");
    const uint32_t words = wordCountOf(lshCount);
    DeviceBuffer dToc, dData, dVectors, dSig, dWs, dAux;
    const size_t wsBytes = em2_dev_compute_signatures_workspace(cellCount, lshCount);
    EM2_HIP(dToc.allocate((size_t(cellCount) + 1) * sizeof(uint64_t)));
    EM2_HIP(dData.allocate(nnz * sizeof(em2_count)));
    EM2_HIP(dVectors.allocate(size_t(geneCount) * lshCount * sizeof(double)));
    EM2_HIP(dSig.allocate(size_t(cellCount) * words * sizeof(uint64_t)));
    EM2_HIP(dWs.allocate(wsBytes));
    EM2_HIP(hipMemcpy(dToc.p, toc, (size_t(cellCount) + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    if (nnz) EM2_HIP(hipMemcpy(dData.p, data, nnz * sizeof(em2_count), hipMemcpyHostToDevice));
    EM2_HIP(hipMemcpy(dVectors.p, vectors, size_t(geneCount) * lshCount * sizeof(double), hipMemcpyHostToDevice));
    void* aux = nullptr;
    if (lshCount % 4u == 0u) {          // (other widths: the exact arithmetic only)
        EM2_HIP(dAux.allocate(em2_dev_vector_aux_bytes(geneCount, lshCount)));
        const int prc = em2_dev_prepare_vectors(dVectors.as<double>(), geneCount, lshCount, dAux.p, nullptr);
        if (prc != EM2_OK) return prc;
        aux = dAux.p;
    }
    const int rc = em2_dev_compute_signatures(dToc.as<uint64_t>(), dData.as<em2_count>(), cellCount, geneCount,
                                              dVectors.as<double>(), aux, lshCount, dSig.as<uint64_t>(),
                                              dWs.p, wsBytes, nullptr);
    if (rc != EM2_OK) return rc;
    EM2_HIP(hipStreamSynchronize(nullptr));
    EM2_HIP(hipMemcpy(signatures, dSig.p, size_t(cellCount) * words * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return EM2_OK;
}


int em2_find_similar_pairs4(const uint64_t* signatures, uint32_t cellCount, uint32_t lshCount, uint32_t k,
                            double similarityThreshold, em2_pair* pairs, uint32_t* usedCount)
{
    if (lshCount == 0) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_find_similar_pairs4: lshCount must be positive");
    if (cellCount == 0) return EM2_OK;
    if (!signatures || !usedCount || (!pairs && k)) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_find_similar_pairs4: null pointer");
    if (!haveDevice()) return fail(EM2_ERROR_NO_DEVICE, "em2_find_similar_pairs4: no HIP device is visible (this library has no CPU path)");
    const uint32_t words = wordCountOf(lshCount);
    const size_t wsBytes = em2_dev_find_similar_pairs4_workspace(cellCount, cellCount, lshCount, k);
    DeviceBuffer dSig, dPairs, dUsed, dWs;
    EM2_HIP(dSig.allocate(size_t(cellCount) * words * sizeof(uint64_t)));
    EM2_HIP(dPairs.allocate(size_t(cellCount) * k * sizeof(em2_pair)));
    EM2_HIP(dUsed.allocate(size_t(cellCount) * sizeof(uint32_t)));
    EM2_HIP(dWs.allocate(wsBytes));
    EM2_HIP(hipMemcpy(dSig.p, signatures, size_t(cellCount) * words * sizeof(uint64_t), hipMemcpyHostToDevice));
    const int rc = em2_dev_find_similar_pairs4(dSig.as<uint64_t>(), cellCount, 0, cellCount, lshCount, k,
                                               similarityThreshold, dPairs.as<em2_pair>(), dUsed.as<uint32_t>(),
                                               dWs.p, wsBytes, nullptr);
    if (rc != EM2_OK) return rc;
    EM2_HIP(hipStreamSynchronize(nullptr));
    if (k) EM2_HIP(hipMemcpy(pairs, dPairs.p, size_t(cellCount) * k * sizeof(em2_pair), hipMemcpyDeviceToHost));
    EM2_HIP(hipMemcpy(usedCount, dUsed.p, size_t(cellCount) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return EM2_OK;
}



int em2_find_similar_pairs5(const uint64_t* signatures, uint32_t cellCount, uint32_t lshCount, uint32_t k,
                            double similarityThreshold, uint32_t lshSliceLength, uint64_t bucketOverflow,
                            em2_pair* pairs, uint32_t* usedCount)
{
    if (lshCount == 0) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_find_similar_pairs5: lshCount must be positive");
    if (lshSliceLength == 0 || lshSliceLength > 32) {
        return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_find_similar_pairs5: lshSliceLength must be in [1,32] (the reference divides by zero for 0)");
    }
    if (cellCount == 0) return EM2_OK;
    if (!signatures || !usedCount || (!pairs && k)) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_find_similar_pairs5: null pointer");
    if (!haveDevice()) return fail(EM2_ERROR_NO_DEVICE, "em2_find_similar_pairs5: no HIP device is visible (this library has no CPU path)");
    const uint32_t words = wordCountOf(lshCount);
    em2::DeviceTables tables;
    const int rc = getDeviceTables(lshCount, similarityThreshold, tables);
    if (rc != EM2_OK) return rc;
    DeviceBuffer dSig, dPairs, dUsed;
    EM2_HIP(dSig.allocate(size_t(cellCount) * words * sizeof(uint64_t)));
    EM2_HIP(dPairs.allocate(size_t(cellCount) * k * sizeof(em2_pair)));
    EM2_HIP(dUsed.allocate(size_t(cellCount) * sizeof(uint32_t)));
    EM2_HIP(hipMemcpy(dSig.p, signatures, size_t(cellCount) * words * sizeof(uint64_t), hipMemcpyHostToDevice));
    EM2_HIP(em2::runFsp5(dSig.as<uint64_t>(), cellCount, 0, cellCount, lshCount, k, lshSliceLength, bucketOverflow,
                         tables, dPairs.as<em2::PairOut>(), dUsed.as<uint32_t>(), nullptr));
    EM2_HIP(hipStreamSynchronize(nullptr));
    if (k) EM2_HIP(hipMemcpy(pairs, dPairs.p, size_t(cellCount) * k * sizeof(em2_pair), hipMemcpyDeviceToHost));
    EM2_HIP(hipMemcpy(usedCount, dUsed.p, size_t(cellCount) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return EM2_OK;
}



// pairs / usedCount on the host (em2_cell_graph_edges) or already on the device (em2_dev_cell_graph_edges: the
// SimilarPairs of a device-resident findSimilarPairs4 go straight into the graph, 0.8 GB less over PCIe at 1M cells)
static int cellGraphEdges(const char* what, bool pairsOnDevice, const em2_pair* pairs, const uint32_t* usedCount,
                          uint32_t similarPairsCellCount, uint32_t k, const uint32_t* similarPairsCellSet,
                          const uint32_t* graphCellSet, uint32_t graphCellCount, double similarityThreshold,
                          uint32_t maxConnectivity, uint32_t* edgeVertex0, uint32_t* edgeVertex1, float* edgeSimilarity,
                          uint64_t* edgeCount)
{
    if (!edgeCount) return fail(EM2_ERROR_INVALID_ARGUMENT, std::string(what) + ": null edgeCount");
    *edgeCount = 0;
    // CellGraph.cpp:101 tests pairs.size() == maxConnectivity after a push_back, so 0 never matches and means
    // "no limit"; a vertex never selects more than the k stored pairs either way.
    if (maxConnectivity == 0 || maxConnectivity > k) maxConnectivity = k;
    if (graphCellCount == 0 || maxConnectivity == 0) return EM2_OK;
    if (!usedCount || !similarPairsCellSet || !graphCellSet || (!pairs && k && similarPairsCellCount) || !edgeVertex0 || !edgeVertex1 || !edgeSimilarity) {
        return fail(EM2_ERROR_INVALID_ARGUMENT, std::string(what) + ": null pointer");
    }
    if (!haveDevice()) return fail(EM2_ERROR_NO_DEVICE, std::string(what) + ": no HIP device is visible (this library has no CPU path)");
    for (uint32_t i = 1; i < similarPairsCellCount; i++) {
        if (similarPairsCellSet[i - 1] > similarPairsCellSet[i]) return fail(EM2_ERROR_RUNTIME, std::string(what) + ": the SimilarPairs cell set is not sorted.");
    }
    // vertexTable of the reference (cell id -> vertex): the graph cell set sorted by id + the vertex of each entry.
    std::vector<uint32_t> order(graphCellCount);
    for (uint32_t i = 0; i < graphCellCount; i++) order[i] = i;
    // (a cell set in ascending order -- AllCells, any set as stored -- needs no sort: 1M ids cost it some 25 ms)
    const bool graphWasSorted = std::is_sorted(graphCellSet, graphCellSet + graphCellCount);
    if (!graphWasSorted) {
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return graphCellSet[a] < graphCellSet[b]; });
    }
    std::vector<uint32_t> sortedIds(graphCellCount);
    for (uint32_t i = 0; i < graphCellCount; i++) sortedIds[i] = graphCellSet[order[i]];
    for (uint32_t i = 1; i < graphCellCount; i++) {
        if (sortedIds[i] == sortedIds[i - 1]) return fail(EM2_ERROR_INVALID_ARGUMENT, std::string(what) + ": duplicate cell id in the graph cell set");
    }
    const size_t slots = size_t(graphCellCount) * maxConnectivity;
    DeviceBuffer dPairs, dUsed, dSp, dGraph, dSorted, dOrder, dE0, dE1, dEs;
    if (!pairsOnDevice) {
        EM2_HIP(dPairs.allocate(size_t(similarPairsCellCount) * k * sizeof(em2_pair)));
        EM2_HIP(dUsed.allocate(size_t(similarPairsCellCount) * sizeof(uint32_t)));
    }
    // The three output arrays may be host or device memory (documented capacity graphCellCount * maxConnectivity): device
    // arrays are written in place, host arrays through device buffers.
    auto onDevice = [](const void* p) {
        hipPointerAttribute_t attributes;
        if (hipPointerGetAttributes(&attributes, p) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        return attributes.type == hipMemoryTypeDevice;
    };
    const bool outputsOnDevice = onDevice(edgeVertex0) && onDevice(edgeVertex1) && onDevice(edgeSimilarity);
    if (!outputsOnDevice) {
        EM2_HIP(dE0.allocate(slots * sizeof(uint32_t)));
        EM2_HIP(dE1.allocate(slots * sizeof(uint32_t)));
        EM2_HIP(dEs.allocate(slots * sizeof(float)));
    }
    uint32_t* out0 = outputsOnDevice ? edgeVertex0 : dE0.as<uint32_t>();
    uint32_t* out1 = outputsOnDevice ? edgeVertex1 : dE1.as<uint32_t>();
    float* outSimilarity = outputsOnDevice ? edgeSimilarity : dEs.as<float>();
    // (sets of consecutive ids need no search on the device)
    const bool spConsecutive = similarPairsCellCount && similarPairsCellSet[similarPairsCellCount - 1] - similarPairsCellSet[0] == similarPairsCellCount - 1 &&
                               std::adjacent_find(similarPairsCellSet, similarPairsCellSet + similarPairsCellCount) == similarPairsCellSet + similarPairsCellCount;
    const bool graphConsecutive = sortedIds[graphCellCount - 1] - sortedIds[0] == graphCellCount - 1;
    if (similarPairsCellCount) {
        if (!pairsOnDevice) {
            if (k) EM2_HIP(hipMemcpy(dPairs.p, pairs, size_t(similarPairsCellCount) * k * sizeof(em2_pair), hipMemcpyHostToDevice));
            EM2_HIP(hipMemcpy(dUsed.p, usedCount, size_t(similarPairsCellCount) * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
    }
    const em2::PairOut* devicePairs = pairsOnDevice ? reinterpret_cast<const em2::PairOut*>(pairs) : dPairs.as<em2::PairOut>();
    const uint32_t* deviceUsed = pairsOnDevice ? usedCount : dUsed.as<uint32_t>();
    // The cell sets go to the device unless they are arithmetic there: consecutive ids (the SimilarPairs set), consecutive ids in
    // ascending order (the graph's: AllCells and every stored set without gaps) -- three 4 MB uploads and four allocations less
    // at a million cells.
    const bool graphArithmetic = graphConsecutive && graphWasSorted;
    if (!spConsecutive && similarPairsCellCount) {
        EM2_HIP(dSp.allocate(size_t(similarPairsCellCount) * sizeof(uint32_t)));
        EM2_HIP(hipMemcpy(dSp.p, similarPairsCellSet, size_t(similarPairsCellCount) * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    if (!graphArithmetic) {
        EM2_HIP(dGraph.allocate(size_t(graphCellCount) * sizeof(uint32_t)));
        EM2_HIP(dOrder.allocate(size_t(graphCellCount) * sizeof(uint32_t)));
        EM2_HIP(hipMemcpy(dGraph.p, graphCellSet, size_t(graphCellCount) * sizeof(uint32_t), hipMemcpyHostToDevice));
        EM2_HIP(hipMemcpy(dOrder.p, order.data(), size_t(graphCellCount) * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    if (!graphConsecutive) {
        EM2_HIP(dSorted.allocate(size_t(graphCellCount) * sizeof(uint32_t)));
        EM2_HIP(hipMemcpy(dSorted.p, sortedIds.data(), size_t(graphCellCount) * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    uint64_t count = 0;
    EM2_HIP(em2::runCellGraphEdges(devicePairs, deviceUsed, similarPairsCellCount, k, spConsecutive ? nullptr : dSp.as<uint32_t>(),
                                   graphArithmetic ? nullptr : dGraph.as<uint32_t>(), graphConsecutive ? nullptr : dSorted.as<uint32_t>(),
                                   graphArithmetic ? nullptr : dOrder.as<uint32_t>(), graphCellCount,
                                   similarityThreshold, maxConnectivity, out0, out1, outSimilarity, &count, nullptr, spConsecutive,
                                   similarPairsCellCount ? similarPairsCellSet[0] : 0u, graphConsecutive, sortedIds[0]));
    if (count && !outputsOnDevice) {
        EM2_HIP(hipMemcpy(edgeVertex0, dE0.p, count * sizeof(uint32_t), hipMemcpyDeviceToHost));
        EM2_HIP(hipMemcpy(edgeVertex1, dE1.p, count * sizeof(uint32_t), hipMemcpyDeviceToHost));
        EM2_HIP(hipMemcpy(edgeSimilarity, dEs.p, count * sizeof(float), hipMemcpyDeviceToHost));
    }
    *edgeCount = count;
    return EM2_OK;
}

int em2_cell_graph_edges(const em2_pair* pairs, const uint32_t* usedCount, uint32_t similarPairsCellCount, uint32_t k,
                         const uint32_t* similarPairsCellSet, const uint32_t* graphCellSet, uint32_t graphCellCount,
                         double similarityThreshold, uint32_t maxConnectivity, uint32_t* edgeVertex0,
                         uint32_t* edgeVertex1, float* edgeSimilarity, uint64_t* edgeCount)
{
    return cellGraphEdges("em2_cell_graph_edges", false, pairs, usedCount, similarPairsCellCount, k, similarPairsCellSet, graphCellSet,
                          graphCellCount, similarityThreshold, maxConnectivity, edgeVertex0, edgeVertex1, edgeSimilarity, edgeCount);
}

int em2_dev_cell_graph_edges(const em2_pair* d_pairs, const uint32_t* d_usedCount, uint32_t similarPairsCellCount, uint32_t k,
                             const uint32_t* similarPairsCellSet, const uint32_t* graphCellSet, uint32_t graphCellCount,
                             double similarityThreshold, uint32_t maxConnectivity, uint32_t* edgeVertex0,
                             uint32_t* edgeVertex1, float* edgeSimilarity, uint64_t* edgeCount)
{
    return cellGraphEdges("em2_dev_cell_graph_edges", true, d_pairs, d_usedCount, similarPairsCellCount, k, similarPairsCellSet,
                          graphCellSet, graphCellCount, similarityThreshold, maxConnectivity, edgeVertex0, edgeVertex1, edgeSimilarity,
                          edgeCount);
}

int em2_analyze_lsh(const uint64_t* toc, const em2_count* data, uint32_t cellCount, uint32_t geneCount,
                    const uint64_t* signatures, uint32_t lshCount, const uint32_t* globalCellIds, uint32_t seed,
                    double csvDownsample, const char* pairsCsvPath, const char* statisticsCsvPath,
                    uint64_t* sum0, double* sum1, double* sum2, double* exactSimilarity, double* lshSimilarity)
{
    if (lshCount == 0 || geneCount == 0) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_analyze_lsh: lshCount and geneCount must be positive");
    if (cellCount == 0) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_analyze_lsh: the cell set is empty");
    if (!toc || !signatures || !globalCellIds || !pairsCsvPath) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_analyze_lsh: null pointer");
    if (!haveDevice()) return fail(EM2_ERROR_NO_DEVICE, "em2_analyze_lsh: no HIP device is visible (this library has no CPU path)");
    const uint64_t nnz = toc[cellCount];
    if (nnz && !data) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_analyze_lsh: null data");
    for (uint64_t i = 0; i < nnz; ++i) {
        if (data[i].gene >= geneCount) return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_analyze_lsh: a local gene id is not below geneCount");
    }
    const uint32_t words = wordCountOf(lshCount);

    // ExpressionMatrixSubset::computeSums (src/ExpressionMatrixSubset.cpp:47-58)
    std::vector<double> sums(2 * size_t(cellCount), 0.);
    for (uint32_t c = 0; c < cellCount; ++c) {
        double s1 = 0., s2 = 0.;
        for (uint64_t i = toc[c]; i < toc[c + 1]; ++i) {
            const float count = data[i].count;
            s1 += count;
            s2 += count * count;
        }
        sums[2 * size_t(c)] = s1;
        sums[2 * size_t(c) + 1] = s2;
    }

    em2::AnalyzeLshState* state = em2::analyzeLshBegin(lshCount, seed, pairsCsvPath);
    if (!state) return fail(EM2_ERROR_RUNTIME, std::string("em2_analyze_lsh: cannot open ") + pairsCsvPath);
    struct Closer {
        em2::AnalyzeLshState*& s;
        ~Closer() { if (s) em2::analyzeLshEnd(s, 1, nullptr, nullptr, nullptr, nullptr); }
    } closer{state};

    // rows in chunks of about 16M pairs: 192 MB of device output per chunk, walked by the host in order
    const uint64_t chunkPairs = 1ull << 24;
    uint32_t maxRows = 0;
    uint64_t maxPairs = 0;
    for (uint32_t begin = 0; begin + 1 < cellCount;) {
        uint32_t end = begin;
        uint64_t pairs = 0;
        while (end + 1 < cellCount && (end == begin || pairs + (cellCount - 1 - end) <= chunkPairs)) pairs += cellCount - 1 - end++;
        if (end - begin > maxRows) maxRows = end - begin;
        if (pairs > maxPairs) maxPairs = pairs;
        begin = end;
    }
    DeviceBuffer dToc, dData, dSig, dProducts, dMismatches, dScratch;
    EM2_HIP(dToc.allocate((size_t(cellCount) + 1) * sizeof(uint64_t)));
    EM2_HIP(dData.allocate(nnz * sizeof(em2_count)));
    EM2_HIP(dSig.allocate(size_t(cellCount) * words * sizeof(uint64_t)));
    EM2_HIP(dProducts.allocate(maxPairs * sizeof(double)));
    EM2_HIP(dMismatches.allocate(maxPairs * sizeof(uint32_t)));
    EM2_HIP(dScratch.allocate(em2::analyzeScratchBytes(geneCount, maxRows)));
    EM2_HIP(hipMemcpy(dToc.p, toc, (size_t(cellCount) + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    if (nnz) EM2_HIP(hipMemcpy(dData.p, data, nnz * sizeof(em2_count), hipMemcpyHostToDevice));
    EM2_HIP(hipMemcpy(dSig.p, signatures, size_t(cellCount) * words * sizeof(uint64_t), hipMemcpyHostToDevice));
    std::vector<double> products(maxPairs);
    std::vector<uint32_t> mismatches(maxPairs);
    uint64_t done = 0;
    for (uint32_t begin = 0; begin + 1 < cellCount;) {
        uint32_t end = begin;
        uint64_t pairs = 0;
        while (end + 1 < cellCount && (end == begin || pairs + (cellCount - 1 - end) <= chunkPairs)) pairs += cellCount - 1 - end++;
        EM2_HIP(em2::launchAnalyzePairs(dToc.as<uint64_t>(), dData.as<em2::CountIn>(), cellCount, geneCount, dSig.as<uint64_t>(), words,
                                        begin, end, dScratch.p, dProducts.as<double>(), dMismatches.as<uint32_t>(), nullptr));
        EM2_HIP(hipStreamSynchronize(nullptr));
        EM2_HIP(hipMemcpy(products.data(), dProducts.p, pairs * sizeof(double), hipMemcpyDeviceToHost));
        EM2_HIP(hipMemcpy(mismatches.data(), dMismatches.p, pairs * sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (!em2::analyzeLshRows(state, sums.data(), cellCount, geneCount, globalCellIds, begin, end, products.data(), mismatches.data(),
                                 csvDownsample, exactSimilarity ? exactSimilarity + done : nullptr,
                                 lshSimilarity ? lshSimilarity + done : nullptr)) {
            return fail(EM2_ERROR_RUNTIME, "em2_analyze_lsh: Assertion failed: bin < binCount (a pair of cells with exact similarity 1, or a "
                                           "cell without variance; src/ExpressionMatrixLsh.cpp:1322)");
        }
        done += pairs;
        begin = end;
    }
    em2::AnalyzeLshState* finished = state;
    state = nullptr;
    if (!em2::analyzeLshEnd(finished, lshCount, statisticsCsvPath, sum0, sum1, sum2)) {
        return fail(EM2_ERROR_RUNTIME, std::string("em2_analyze_lsh: cannot write ") + (statisticsCsvPath ? statisticsCsvPath : ""));
    }
    return EM2_OK;
}


static int labelPropagation(bool edgesOnDevice, const uint32_t* vertexCellIds, uint32_t vertexCount, const uint32_t* edgeVertex0,
                            const uint32_t* edgeVertex1, const float* edgeSimilarity, uint64_t edgeCount, uint64_t seed,
                            uint64_t stableIterationCountThreshold, uint64_t maxIterationCount, uint32_t* clusterIds,
                            uint64_t* iterationCount)
{
    if (iterationCount) *iterationCount = 0;
    if (vertexCount == 0) return EM2_OK;
    if (!vertexCellIds || !clusterIds || (edgeCount && (!edgeVertex0 || !edgeVertex1 || !edgeSimilarity))) {
        return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_cell_graph_label_propagation: null pointer");
    }
    // (edges on the device are what em2_dev_cell_graph_edges wrote there: valid by construction, not read here)
    for (uint64_t e = 0; e < (edgesOnDevice ? 0 : edgeCount); e++) {
        if (edgeVertex0[e] >= vertexCount || edgeVertex1[e] >= vertexCount) {
            return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_cell_graph_label_propagation: an edge names a vertex that does not exist");
        }
        if (edgeVertex0[e] == edgeVertex1[e]) {
            return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_cell_graph_label_propagation: an edge joins a vertex to itself");
        }
    }
    // The vector the reference shuffles: the vertices in the order of its std::map vertexTable (CellGraph.cpp:484-489).
    std::vector<uint32_t> byCellId(vertexCount);
    for (uint32_t v = 0; v < vertexCount; v++) byCellId[v] = v;
    if (!std::is_sorted(vertexCellIds, vertexCellIds + vertexCount)) {       // (vertices in cell-id order need no sort)
        std::stable_sort(byCellId.begin(), byCellId.end(), [&](uint32_t a, uint32_t b) { return vertexCellIds[a] < vertexCellIds[b]; });
    }
    for (uint32_t i = 1; i < vertexCount; i++) {
        if (vertexCellIds[byCellId[i]] == vertexCellIds[byCellId[i - 1]]) {
            return fail(EM2_ERROR_INVALID_ARGUMENT, "em2_cell_graph_label_propagation: duplicate cell id among the vertices");
        }
    }
    if (!haveDevice()) return fail(EM2_ERROR_NO_DEVICE, "em2_cell_graph_label_propagation: no HIP device is visible (this library has no CPU path)");
    uint64_t iterations = 0;
    uint32_t error = 0;
    CallTimer timer;
    EM2_HIP(em2::runLabelPropagation(vertexCellIds, vertexCount, edgeVertex0, edgeVertex1, edgeSimilarity, edgeCount,
                                     byCellId.data(), seed, stableIterationCountThreshold, maxIterationCount, clusterIds,
                                     &iterations, &error, nullptr));
    if (error == 1) return fail(EM2_ERROR_RUNTIME, "em2_cell_graph_label_propagation: a wave waited too long for an earlier vertex (is the GPU shared?)");
    if (error != 0) return fail(EM2_ERROR_RUNTIME, "em2_cell_graph_label_propagation: the cluster tables outgrew their arena");
    if (iterationCount) *iterationCount = iterations;
    timer.stage("label propagation");

    // CellGraph.cpp:561-596: clusters renumbered from 0 by decreasing size; equal sizes by decreasing label
    // (std::greater on (size, label)).
    // A label is the cell id of a vertex: count the vertices per label through the vertex that owns the id (the sorted
    // cell ids are at hand), instead of sorting a million labels.
    struct Cluster {
        uint64_t size;
        uint32_t label;
    };
    std::vector<uint32_t> sortedCellIds(vertexCount);
    for (uint32_t i = 0; i < vertexCount; i++) sortedCellIds[i] = vertexCellIds[byCellId[i]];
    const bool contiguous = uint64_t(sortedCellIds.back()) - sortedCellIds.front() + 1u == vertexCount;
    std::vector<uint32_t> owner(vertexCount);            // position of the label of vertex v among the sorted cell ids
    std::vector<uint32_t> sizeOf(vertexCount, 0u);
    for (uint32_t v = 0; v < vertexCount; v++) {
        const uint32_t label = clusterIds[v];
        const uint32_t at = contiguous ? label - sortedCellIds.front()
                                       : uint32_t(std::lower_bound(sortedCellIds.begin(), sortedCellIds.end(), label) - sortedCellIds.begin());
        if (at >= vertexCount || sortedCellIds[at] != label) return fail(EM2_ERROR_RUNTIME, "em2_cell_graph_label_propagation: a label is no cell id of the graph");
        owner[v] = at;
        ++sizeOf[at];
    }
    std::vector<Cluster> clusters;
    for (uint32_t at = 0; at < vertexCount; at++) {
        if (sizeOf[at]) clusters.push_back(Cluster{uint64_t(sizeOf[at]), sortedCellIds[at]});
    }
    std::sort(clusters.begin(), clusters.end(), [](const Cluster& a, const Cluster& b) {
        return a.size != b.size ? a.size > b.size : a.label > b.label;
    });
    std::vector<uint32_t>& clusterOf = sizeOf;           // reused: position among the sorted cell ids -> new cluster id
    for (uint32_t i = 0; i < clusters.size(); i++) {
        const uint32_t at = contiguous ? clusters[i].label - sortedCellIds.front()
                                       : uint32_t(std::lower_bound(sortedCellIds.begin(), sortedCellIds.end(), clusters[i].label) - sortedCellIds.begin());
        clusterOf[at] = i;
    }
    for (uint32_t v = 0; v < vertexCount; v++) clusterIds[v] = clusterOf[owner[v]];
    timer.stage("cluster renumbering");
    return EM2_OK;
}

int em2_cell_graph_label_propagation(const uint32_t* vertexCellIds, uint32_t vertexCount, const uint32_t* edgeVertex0,
                                     const uint32_t* edgeVertex1, const float* edgeSimilarity, uint64_t edgeCount,
                                     uint64_t seed, uint64_t stableIterationCountThreshold,
                                     uint64_t maxIterationCount, uint32_t* clusterIds, uint64_t* iterationCount)
{
    return labelPropagation(false, vertexCellIds, vertexCount, edgeVertex0, edgeVertex1, edgeSimilarity, edgeCount, seed,
                            stableIterationCountThreshold, maxIterationCount, clusterIds, iterationCount);
}

int em2_dev_cell_graph_label_propagation(const uint32_t* vertexCellIds, uint32_t vertexCount, const uint32_t* d_edgeVertex0,
                                         const uint32_t* d_edgeVertex1, const float* d_edgeSimilarity, uint64_t edgeCount,
                                         uint64_t seed, uint64_t stableIterationCountThreshold,
                                         uint64_t maxIterationCount, uint32_t* clusterIds, uint64_t* iterationCount)
{
    return labelPropagation(true, vertexCellIds, vertexCount, d_edgeVertex0, d_edgeVertex1, d_edgeSimilarity, edgeCount, seed,
                            stableIterationCountThreshold, maxIterationCount, clusterIds, iterationCount);
}

}  // extern "C"
