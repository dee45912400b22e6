// em2_project.hip -- LSH signature projection on gfx950, bit-identical to
// ExpressionMatrixSubset::computeSums (src/ExpressionMatrixSubset.cpp:47-58, sum1 only) and
// Lsh::computeCellLshSignatures (src/Lsh.cpp:118-224).
//
// Exactness: the sign of a sequential FP64 sum decides each bit, so the order of operations is part of
// the contract:  sp = (-mean) * S[i];  then for every stored count of the cell in stored order
// sp = sp + (double(count) * U[g][i])  with the product rounded before the add (the reference is built for
// SSE4.2: no FMA).  This file is compiled with -ffp-contract=off and uses __dmul_rn/__dadd_rn.
// No MFMA: an MFMA contraction would change the summation order.
//
// Mapping: one lane = one signature bit of one cell; the 64 lanes of a wave are 64 consecutive bits, so
// every expression count costs the wave one coalesced 512-byte read of the gene's hyperplane row segment and
// the finished 64 bits leave as one ballot -> one uint64 store (MSB-first, src/BitSet.hpp:48-62).
// The cell's (gene, count) stream is wave-uniform and comes through the scalar unit.  grid.y walks 256-bit
// column chunks of the hyperplane matrix so the chunk in use (geneCount x 256 x 8 B) stays cache resident.

// Screening (launchProjectionScreened).  The kernel above is bound by the hyperplane gathers (8 bytes per count and
// bit).  Only the SIGN of each scalar product is needed, so a cheaper first pass decides almost every bit:
//   a_i = (-mean)*S_i + sum_j x_j * float(U[g_j][i])      float copy of U = half the bytes; FP64 fma accumulation
// differs from the reference's sequentially rounded value sp_i by at most
//   E_i = (1.01*2^-24 + (n+2)*2^-52) * (|mean|*|S_i| + (sum_j |x_j|) * max_g |U[g][i]|)
// (2^-24: rounding of U to float; (n+2)*2^-52: the two FP64 chains round differently; the bracket bounds the sum of
// the magnitudes of all terms).  If |a_i| > E_i the reference's sign is the sign of a_i.  Otherwise the 64-bit
// word holding bit i goes on a work list and is recomputed by the exact kernel's arithmetic (a few 1e-4 of the
// words on the benchmark data).  The result is therefore still bit-identical to the sequential FP64 reference.

#include "em2_device.h"

#include <cstdlib>
#include <cstring>

namespace em2 {
namespace {

typedef const __attribute__((address_space(4))) uint64_t* ScalarPtr64;

constexpr int kCellsPerBlock = 64;          // (16 / 32 / 64 / 128 / 256 at 1M cells: the projection 34.3 / 33.9 / 33.6 / 33.4 / 33.5 ms)

__global__ void __launch_bounds__(256)
cellMeansKernel(const uint64_t* __restrict__ toc, const CountIn* __restrict__ data, uint32_t cellCount,
                uint32_t geneCount, double* __restrict__ means)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cellCount) return;
    double sum1 = 0.;
    const uint64_t end = toc[c + 1];
    for (uint64_t j = toc[c]; j < end; ++j) {
        sum1 = __dadd_rn(sum1, double(data[j].count));      // ExpressionMatrixSubset.cpp:52-55
    }
    means[c] = sum1 / double(geneCount);                     // Lsh.cpp:168
}

__global__ void __launch_bounds__(256)
vectorSumsKernel(const double* __restrict__ vectors, uint32_t geneCount, uint32_t lshCount,
                 double* __restrict__ sums)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lshCount) return;
    double s = 0.;
    for (uint32_t g = 0; g < geneCount; ++g) {
        s = __dadd_rn(s, vectors[size_t(g) * lshCount + i]);  // Lsh.cpp:137-144
    }
    sums[i] = s;
}

__global__ void __launch_bounds__(256)
projectionKernel(const uint64_t* __restrict__ toc, const CountIn* __restrict__ data, uint32_t cellCount,
                 const double* __restrict__ vectors, const double* __restrict__ vectorSums,
                 const double* __restrict__ means, uint32_t lshCount, uint32_t wordCount,
                 uint64_t* __restrict__ signatures)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t word = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (word >= wordCount) return;
    const uint32_t bit = word * 64u + lane;
    const bool bitValid = bit < lshCount;
    const double* column = vectors + (bitValid ? bit : 0u);
    const double s = bitValid ? vectorSums[bit] : 0.;
    ScalarPtr64 entries = (ScalarPtr64)(uintptr_t)data;

    const uint32_t cellBegin = blockIdx.x * kCellsPerBlock;
    const uint32_t cellEnd = min(cellBegin + kCellsPerBlock, cellCount);
    for (uint32_t c = cellBegin; c < cellEnd; ++c) {
        const uint64_t jBegin = toc[c];
        const uint64_t jEnd = toc[c + 1];
        double sp = __dmul_rn(-means[c], s);                 // Lsh.cpp:180-182
        uint64_t j = jBegin;
        for (; j + 4 <= jEnd; j += 4) {
            double u[4], x[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint64_t e = entries[j + t];          // {gene (low 32), count bits (high 32)}
                u[t] = column[size_t(uint32_t(e)) * lshCount];
                x[t] = double(__uint_as_float(uint32_t(e >> 32)));
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) sp = __dadd_rn(sp, __dmul_rn(x[t], u[t]));   // Lsh.cpp:188-198
        }
        for (; j < jEnd; ++j) {
            const uint64_t e = entries[j];
            const double u = column[size_t(uint32_t(e)) * lshCount];
            const double x = double(__uint_as_float(uint32_t(e >> 32)));
            sp = __dadd_rn(sp, __dmul_rn(x, u));
        }
        const uint64_t mask = __builtin_amdgcn_ballot_w64(bitValid && sp > 0.);      // Lsh.cpp:201-206
        if (lane == 0u) signatures[size_t(c) * wordCount + word] = __brevll(mask);  // bit i -> 63-(i&63)
    }
}


// Per-cell inputs of the screening pass: mean (as cellMeansKernel: the counts added in stored order, every addition
// rounded) and sum of |count|.  One wave per cell, coalesced.  The lanes' partial sums in any order equal the sequential
// sum whenever no addition can round at all: every count is a multiple of u = 2^e (e = the smallest exponent of a count's
// last mantissa bit) and sum|count| < 2^53 u, so every partial sum is a multiple of u below 2^53 u, which a double holds
// exactly.  That is the case for integer counts and for any float counts within ~2^20 of each other; a cell where it is
// not is summed again by one lane in stored order.
__global__ void __launch_bounds__(256)
cellStatsKernel(const uint64_t* __restrict__ toc, const CountIn* __restrict__ data, uint32_t cellCount,
                uint32_t geneCount, double* __restrict__ means, double* __restrict__ sumAbs, uint32_t* __restrict__ notAllInteger)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= cellCount) return;
    const uint64_t begin = toc[c], end = toc[c + 1];
    double sum1 = 0., abs1 = 0.;
    int lowest = 1000;                              // exponent of the last mantissa bit of the smallest count seen
    bool integers = true;                           // every count an integer of at most 15 bits (the integer first tier)
    for (uint64_t j = begin + lane; j < end; j += 64u) {
        const float value = data[j].count;
        const double x = double(value);
        sum1 += x;
        abs1 += fabs(x);
        integers = integers && value == truncf(value) && fabsf(value) <= 32767.f;
        const int exponentField = int((__float_as_uint(value) >> 23) & 0xffu);
        if (value != 0.f) lowest = min(lowest, (exponentField ? exponentField : 1) - 150);
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        sum1 += __shfl_xor(sum1, d, 64);
        abs1 += __shfl_xor(abs1, d, 64);
        lowest = min(lowest, __shfl_xor(lowest, d, 64));
    }
    // (abs1 is exact under the same condition; a value that is not finite fails the test)
    const bool exact = lowest == 1000 || abs1 < ldexp(1., 53 + lowest);
    if (!exact) {
        sum1 = 0.;
        abs1 = 0.;
        for (uint64_t j = begin; j < end; ++j) {    // (all lanes the same walk: uniform loads)
            const double x = double(data[j].count);
            sum1 = __dadd_rn(sum1, x);
            abs1 += fabs(x);
        }
    }
    // (the integer tier accumulates count * q, |q| <= 32767, in 32 bits: sum|count| <= 65535 keeps every partial sum in range)
    const bool integerCell = __builtin_amdgcn_ballot_w64(!integers) == 0ull && abs1 <= 65535.;
    if (lane == 0u) {
        means[c] = sum1 / double(geneCount);
        sumAbs[c] = abs1 * (1. + 1e-12);             // upper bound of the exact sum of magnitudes
        if (!integerCell && notAllInteger) atomicOr(notAllInteger, 1u);
    }
}

// aux layout: [lshCount doubles: S_i][lshCount doubles: max_g |U[g][i]|][geneCount*lshCount floats: float(U)]
__global__ void __launch_bounds__(256)
vectorStatsKernel(const double* __restrict__ vectors, uint32_t geneCount, uint32_t lshCount, double* __restrict__ sums,
                  double* __restrict__ maxAbs)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lshCount) return;
    double s = 0., m = 0.;
    // (sixteen loads in flight, the additions in gene order: with one load per addition the kernel sat in memory latency
    // 30,000 times per column, 12 ms)
    uint32_t g = 0;
    for (; g + 16u <= geneCount; g += 16u) {
        double u[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) u[q] = vectors[size_t(g + uint32_t(q)) * lshCount + i];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            s = __dadd_rn(s, u[q]);                               // Lsh.cpp:137-144
            m = fmax(m, fabs(u[q]));
        }
    }
    for (; g < geneCount; ++g) {
        const double u = vectors[size_t(g) * lshCount + i];
        s = __dadd_rn(s, u);
        m = fmax(m, fabs(u));
    }
    sums[i] = s;
    maxAbs[i] = m;
}

__global__ void __launch_bounds__(256)
vectorsToFloatKernel(const double* __restrict__ vectors, uint32_t geneCount, uint32_t lshCount, bool sliceMajor,
                     float* __restrict__ out)
{
    const uint64_t count = uint64_t(geneCount) * lshCount;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += uint64_t(gridDim.x) * blockDim.x) {
        uint64_t to = i;
        if (sliceMajor) {       // [slice of 32 bits][gene][32]: a slice is one contiguous geneCount x 128 B block
            const uint32_t gene = uint32_t(i / lshCount);
            const uint32_t bit = uint32_t(i % lshCount);
            to = (uint64_t(bit >> 5) * geneCount + gene) * 32u + (bit & 31u);
        }
        out[to] = float(vectors[i]);
    }
}

// The float copy of the hyperplanes is slice-major whenever the signature is a whole number of 32-bit slices.
__host__ __device__ inline bool floatCopyIsSliceMajor(uint32_t lshCount) { return lshCount % 32u == 0u; }

// A 16-bit fixed-point copy of the hyperplanes for the first screening tier, whenever the signature is a whole number
// of 64-bit words: q[slice of 64 bits][gene][64] = round(U[g][i] / scale_i), scale_i = max_g |U[g][i]| / 32767, so that
// |U[g][i] - q * scale_i| <= scale_i / 2.  A slice is one contiguous geneCount x 128 B block, like the float copy's.
__host__ __device__ inline bool haveQuantizedCopy(uint32_t lshCount) { return lshCount % 64u == 0u; }
constexpr uint32_t kQuantizedInFlight = 4;       // entries per lane in flight in the 16-bit tier (8, in the earlier form of its loop: 124 registers, four waves per SIMD instead of five, 44 ms instead of 40.6)

__global__ void __launch_bounds__(256)
vectorsToQuantizedKernel(const double* __restrict__ vectors, uint32_t geneCount, uint32_t lshCount,
                         const double* __restrict__ maxAbs, int16_t* __restrict__ out, double* __restrict__ scales)
{
    const uint64_t count = uint64_t(geneCount) * lshCount;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += uint64_t(gridDim.x) * blockDim.x) {
        const uint32_t gene = uint32_t(i / lshCount);
        const uint32_t bit = uint32_t(i % lshCount);
        const double scale = maxAbs[bit] / 32767.;
        if (gene == 0u) scales[bit] = scale;
        double q = scale > 0. ? rint(vectors[i] / scale) : 0.;
        q = q > 32767. ? 32767. : (q < -32767. ? -32767. : q);
        // (inside a gene's 128-byte line the word's bits are TRANSPOSED as an 8 x 8 matrix: position 8 s + t holds bit 8 t + s, so
        // that the lane that ends up with position 8 s + t of the first tier's sums -- lane 8 t + s -- holds the bit of its own number)
        out[(uint64_t(bit >> 6) * geneCount + gene) * 64u + (((bit & 7u) << 3) | ((bit >> 3) & 7u))] = int16_t(int(q));
    }
}

// First screening tier on the 16-bit copy, XCD-sliced like projectionScreenSlicedKernel: blockIdx.x & 7 picks the 64-bit
// word of the signature a block works on, one 128-byte line per entry holds the word's 64 hyperplane values, so the tier
// gathers HALF the lines of the float tier.  lane = (entry group g = lane / 8, 8 bits sub = lane % 8), one 16-byte load
// per lane.  The products count * q are exact in double (24 + 15 bits), the sum is scaled once at the end.  Its value
// differs from the reference's sequentially rounded sum by at most
//     (n + 12) 2^-52 (|mean| |S_i| + sum|x| max|U_i|)   (rounding on both sides)   +   0.501 scale_i sum|x|   (quantisation)
// and a bit whose |value| exceeds that has the reference's sign.  The quantisation term is ~250 times the float copy's,
// so this tier leaves several per cent of the 64-bit words undecided (the float tier: a few 1e-4); those go to the work
// list like the float tier's, for the exact arithmetic.
// DIAG (EM2_PROJECTION_DIAG, measurements only, wrong results, nothing listed for the later tiers): 1 = the gathers
// without the arithmetic, 2 = the arithmetic without the row gathers.
// INTEGER: the tier for matrices whose counts are all small integers (what expression COUNTS are, and the benchmark's): the
// products count * q are formed and summed exactly in 32-bit integers, one v_mad_i32_i16 per product (the high half of a
// register through op_sel) where the float form needs 1.5 instructions (two conversions and a packed fma per two products);
// the chunking, its single-precision term of the bound and the overflow case go away.  Both instantiations are launched;
// the statistics kernel has set a device flag when some cell does not qualify, and the one whose turn it is not leaves at once.
template <int DIAG = 0, bool INTEGER = false>
__global__ void __launch_bounds__(256, 4)
projectionScreenQuantizedKernel(const uint64_t* __restrict__ toc, const CountIn* __restrict__ data, uint32_t cellCount,
                                uint32_t geneCount, const int16_t* __restrict__ quantized, const double* __restrict__ scales,
                                const double* __restrict__ vectorSums, const double* __restrict__ vectorMaxAbs,
                                const double* __restrict__ means, const double* __restrict__ sumAbs, uint32_t lshCount,
                                uint32_t wordCount, uint64_t* __restrict__ signatures, uint64_t* __restrict__ workList,
                                uint32_t* __restrict__ workCount, uint64_t* __restrict__ bitList, uint32_t* __restrict__ bitCount,
                                const uint32_t* __restrict__ notAllInteger)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t word = blockIdx.y * 8u + (blockIdx.x & 7u);
    if (word >= wordCount) return;
    if (notAllInteger && (*notAllInteger != 0u) == INTEGER) return;          // (uniform: the other instantiation's launch)
    const uint32_t sub = lane & 7u;
    const uint32_t group = lane >> 3;
    // the word's slice of the 16-bit copy (wave-uniform: a scalar base) and the lane's 16 bytes of a gene's 128-byte line
    const char* slice = reinterpret_cast<const char*>(quantized + size_t(word) * geneCount * 64u);
    const uint32_t laneBytes = sub * 16u;
    // After the gathers a lane finishes ONE bit of the word, the bit of its own number (see the reduction below), with the
    // three constants of that hyperplane column (sum, largest magnitude, scale).  (Until round 5 every lane
    // finished the eight bits of its 16 bytes, all eight entry groups the same eight: 24 constants per lane, which waited in
    // LDS -- 64 bytes apart, lanes sub and sub + 4 on the same banks: 4.6e9 bank-conflict cycles per launch at 1M cells,
    // profiles/r04_pmc_bench_1Mcells_1gpu.json -- and eight times the double-precision work.)
    // They wait in LDS (lane l reads element l: no two lanes on a bank) behind an index the compiler cannot see through: held
    // in registers across the gather loop they cost the float form its fifth wave per SIMD (98 registers).
    __shared__ double bitConstants[3][64];
    if (threadIdx.x < 64u) {
        bitConstants[0][threadIdx.x] = vectorSums[word * 64u + threadIdx.x];
        bitConstants[1][threadIdx.x] = vectorMaxAbs[word * 64u + threadIdx.x];
        bitConstants[2][threadIdx.x] = scales[word * 64u + threadIdx.x];
    }
    __syncthreads();
    const double* allConstants = &bitConstants[0][0];
    const uint64_t* entries = reinterpret_cast<const uint64_t*>(data);
    const uint32_t cellBegin = (blockIdx.x >> 3) * kCellsPerBlock;
    const uint32_t cellEnd = min(cellBegin + kCellsPerBlock, cellCount);
    const uint32_t firstCell = uint32_t(__builtin_amdgcn_readfirstlane(int(cellBegin + wave)));      // (the wave's cells are uniform)
    for (uint32_t c = firstCell; c < cellEnd; c += 4u) {
        const uint64_t jBegin = toc[c];
        const uint64_t jEnd = toc[c + 1];
        const uint32_t entryCount = uint32_t(jEnd - jBegin);
        const uint64_t* cellEntries = entries + jBegin;
        double a[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
        typedef float Float2 __attribute__((ext_vector_type(2)));
        // The products of up to sixteen entries are summed in single precision first (packed: two bits per instruction) and
        // that chunk sum goes into the double accumulator: a sixteenth of the conversions and double additions, which run
        // at half rate.  A product count * q has 39 significant bits, so every fma of a chunk rounds (2^-24 relative): at
        // most 16 * 2^-24 * sum|count * q| in all, 6 % of the quantisation term of the bound below.
        // The 8 entry groups of the wave take every 8th entry; kQuantizedInFlight entries per lane are in flight; an
        // entry past the cell's end is the cell's first one with count 0 (its products are exact zeros), so there is no
        // remainder loop.  Addresses are 32-bit offsets from scalar bases (one shift-or per entry).
        // The entries: one coalesced load per 64 of them (lane l takes entry first + l; the next 64 are loaded while these
        // are used), handed to the groups through the LDS crossbar (ds_bpermute: group g, slot q of half h reads lane
        // 32 h + 8 q + g) -- eight lanes loading the same 8 bytes cost the vector memory path as much as 64 different ones,
        // a third of what the rows cost it, and that path is what bounds this kernel (the gathers alone: 34 of 38 ms).
        // (the integer form needs 76 registers where the float form needs 94: eight entries per lane in flight still leave five
        // waves per SIMD -- 40 row loads in flight per SIMD instead of 24)
        constexpr uint32_t kInFlight = INTEGER ? 2u * kQuantizedInFlight : kQuantizedInFlight;
        constexpr uint32_t kPart = kInFlight * 8u;                       // entries of the wave per part
        constexpr uint32_t kHalves = 64u / kPart;                        // parts per 64 loaded entries
        static_assert(kQuantizedInFlight == 4u && kHalves * kPart == 64u, "one or two parts per 64 entries");
        const uint32_t sourceLane = group << 2;                          // (byte address of a lane for ds_bpermute)
        uint64_t nextEntry = 0;
#define EM2_LOAD_ENTRIES(first_)                                                                                              \
        {                                                                                                                     \
            const uint32_t index_ = (first_) + lane;                                                                          \
            const uint64_t e_ = *reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(cellEntries) +               \
                                                                   (index_ < entryCount ? index_ * 8u : 0u));                 \
            nextEntry = index_ < entryCount ? e_ : uint64_t(uint32_t(e_));           /* count 0 past the end */               \
        }
        if (entryCount) EM2_LOAD_ENTRIES(0u)
        Float2 chunk[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
        int exact[8] = {0, 0, 0, 0, 0, 0, 0, 0};          // INTEGER: the eight sums of count * q, exact
        uint32_t partsInChunk = 0;
        for (uint32_t first = 0; first < entryCount; first += 64u) {
            const int entryGene = int(uint32_t(nextEntry)), entryCountBits = int(uint32_t(nextEntry >> 32));
            if (first + 64u < entryCount) EM2_LOAD_ENTRIES(first + 64u)
#pragma unroll
            for (uint32_t half = 0; half < kHalves; ++half) {
                if (half && first + kPart >= entryCount) break;          // (uniform)
                uint4 u[kInFlight];
                float x[kInFlight];
                uint32_t gene[kInFlight];
#pragma unroll
                for (uint32_t q = 0; q < kInFlight; ++q) {
                    const int from = int(sourceLane + 4u * (kPart * half + 8u * q));
                    gene[q] = uint32_t(__builtin_amdgcn_ds_bpermute(from, entryGene));
                    x[q] = __int_as_float(__builtin_amdgcn_ds_bpermute(from, entryCountBits));
                }
#pragma unroll
                for (uint32_t q = 0; q < kInFlight; ++q) {
                    if (DIAG == 2) u[q] = uint4{gene[q], gene[q] + 1u, gene[q] + 2u, gene[q] + 3u};
                    else u[q] = *reinterpret_cast<const uint4*>(slice + ((gene[q] << 7) | laneBytes));
                }
                const bool lastPart = first + (half + 1u) * kPart >= entryCount;
                if (DIAG == 1 && INTEGER) {
#pragma unroll
                    for (uint32_t q = 0; q < kInFlight; ++q) exact[q & 7u] += int((u[q].x ^ u[q].y ^ u[q].z ^ u[q].w) & 0xffu) + int(x[q]);
                    continue;
                }
                if (DIAG == 1) {
#pragma unroll
                    for (uint32_t q = 0; q < kInFlight; ++q) {
                        chunk[q & 3u].x += __uint_as_float((u[q].x ^ u[q].y ^ u[q].z ^ u[q].w) & 0x3fffffffu);
                        chunk[q & 3u].y += x[q];
                    }
                    if (++partsInChunk == 4u || lastPart) {
#pragma unroll
                        for (int m = 0; m < 4; ++m) a[2 * m] += double(chunk[m].x) + double(chunk[m].y);
                        partsInChunk = 0;
                    }
                    continue;
                }
                if (INTEGER) {
#pragma unroll
                    for (uint32_t q = 0; q < kInFlight; ++q) {
                        const uint32_t w[4] = {u[q].x, u[q].y, u[q].z, u[q].w};
                        const int count = int(x[q]);          // (an integer of at most 15 bits: the statistics kernel checked)
#pragma unroll
                        for (int m = 0; m < 4; ++m) {
                            asm("v_mad_i32_i16 %0, %1, %2, %0" : "+v"(exact[2 * m]) : "v"(w[m]), "v"(count));
                            asm("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"(exact[2 * m + 1]) : "v"(w[m]), "v"(count));
                        }
                    }
                    continue;
                }
#pragma unroll
                for (uint32_t q = 0; q < kInFlight; ++q) {
                    const uint32_t w[4] = {u[q].x, u[q].y, u[q].z, u[q].w};
                    const Float2 xx = {x[q], x[q]};
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const Float2 qf = {float(int(w[m] << 16) >> 16), float(int(w[m]) >> 16)};
                        chunk[m] = __builtin_elementwise_fma(xx, qf, chunk[m]);
                    }
                }
                if (++partsInChunk == 4u || lastPart) {          // (uniform) sixteen entries per lane at most
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        a[2 * m] += double(chunk[m].x);
                        a[2 * m + 1] += double(chunk[m].y);
                        chunk[m] = Float2{0.f, 0.f};
                    }
                    partsInChunk = 0;
                }
            }
        }
#undef EM2_LOAD_ENTRIES
        // The eight entry groups' sums, TRANSPOSED on the way: a lane holds eight sums (positions 8 sub .. 8 sub + 7 of the line)
        // over its group's entries; in three exchanges (group bit 0, 1, 2) it gives away half of what it still holds and adds
        // what its partner gives -- 4 + 2 + 1 values cross instead of 3 x 8 -- and ends with the one sum of position
        // 8 sub + group over ALL entries, which by the copy's order inside a line is bit 8 group + sub = the lane's own number.
        // (The additions of a bit happen in the order of the butterfly this replaces; the integer form is exact anyway.)
        double mine;
        if (INTEGER) {
            int four[4], two[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int give = (group & 1u) ? exact[2 * i] : exact[2 * i + 1];
                four[i] = ((group & 1u) ? exact[2 * i + 1] : exact[2 * i]) + __shfl_xor(give, 8, 64);          // t = 2 i + (group & 1)
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int give = (group & 2u) ? four[2 * i] : four[2 * i + 1];
                two[i] = ((group & 2u) ? four[2 * i + 1] : four[2 * i]) + __shfl_xor(give, 16, 64);            // t = 4 i + (group & 3)
            }
            const int give = (group & 4u) ? two[0] : two[1];
            mine = double(((group & 4u) ? two[1] : two[0]) + __shfl_xor(give, 32, 64));                        // t = group
        } else {
            double four[4], two[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double give = (group & 1u) ? a[2 * i] : a[2 * i + 1];
                four[i] = ((group & 1u) ? a[2 * i + 1] : a[2 * i]) + __shfl_xor(give, 8, 64);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const double give = (group & 2u) ? four[2 * i] : four[2 * i + 1];
                two[i] = ((group & 2u) ? four[2 * i + 1] : four[2 * i]) + __shfl_xor(give, 16, 64);
            }
            const double give = (group & 4u) ? two[0] : two[1];
            mine = ((group & 4u) ? two[1] : two[0]) + __shfl_xor(give, 32, 64);
        }
        const double mean = means[c];
        const double n = double(jEnd - jBegin);
        const double factor = (n + 12.) * 2.220446049250313e-16 * 1.000001;
        const double absMean = fabs(mean);
        const double absX = sumAbs[c];
        uint32_t constantOfLane = lane;
        asm volatile("" : "+v"(constantOfLane));          // (not loop-invariant as far as the compiler knows)
        const double sumOfBit = allConstants[constantOfLane], maxOfBit = allConstants[64u + constantOfLane],
                     scaleOfBit = allConstants[128u + constantOfLane];
        const double total = __fma_rn(mine, scaleOfBit, __dmul_rn(-mean, sumOfBit));
        // (+ the single-precision chunks: 16 * 2^-24 * sum|count * q| * scale <= 9.6e-7 * sum|x| * max|U_i|, and a
        // subnormal slack for them)
        // (INTEGER: the sum of count * q is exact, that term and the subnormal slack are not needed)
        const double bound = factor * (absMean * fabs(sumOfBit) + absX * maxOfBit) + 0.501 * scaleOfBit * absX +
                             (INTEGER ? 0. : 9.6e-7 * absX * maxOfBit + n * 1.5e-45 * scaleOfBit) + 1e-300;
        // (a single-precision chunk that overflowed makes the total infinite: undecided as well)
        const bool ambiguous = !(fabs(total) > bound) || !(fabs(total) <= 1.7976931348623157e308);
        // lane l = bit l of the word, first bit most significant
        const uint64_t w = __brevll(__builtin_amdgcn_ballot_w64(total > 0.));
        // A word with undecided bits goes to a later tier.  Almost always it is ONE bit (a cell has 1.1 undecided bits in 1.07
        // words on the benchmark data): such a word is listed by that bit, for the per-bit form of the float tier (one float
        // per count instead of the 64 of the whole word); anything else is listed as a word.
        const uint64_t ambiguousMask = __builtin_amdgcn_ballot_w64(ambiguous);
        if (lane == 0u) {
            signatures[size_t(c) * wordCount + word] = w;
            if (ambiguousMask != 0ull && DIAG == 0) {
                if ((ambiguousMask & (ambiguousMask - 1ull)) == 0ull && bitList) {
                    const uint32_t bit = word * 64u + uint32_t(__builtin_ctzll(ambiguousMask));
                    bitList[atomicAdd(bitCount, 1u)] = (uint64_t(c) << 32) | bit;
                } else {
                    workList[atomicAdd(workCount, 1u)] = (uint64_t(c) << 32) | word;
                }
            }
        }
    }
}

// Screening pass: one wave = 256 consecutive bits of one cell (4 per lane, one 16-byte load per count).
__global__ void __launch_bounds__(256)
projectionScreenKernel(const uint64_t* __restrict__ toc, const CountIn* __restrict__ data, uint32_t cellCount,
                       uint32_t geneCount, const float* __restrict__ vectors32, const double* __restrict__ vectorSums,
                       const double* __restrict__ vectorMaxAbs, const double* __restrict__ means,
                       const double* __restrict__ sumAbs, uint32_t lshCount, uint32_t wordCount,
                       uint64_t* __restrict__ signatures, uint64_t* __restrict__ workList,
                       uint32_t* __restrict__ workCount)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t chunk = blockIdx.y * 4u + (threadIdx.x >> 6);          // 256-bit chunk of the signature
    const uint32_t bit0 = chunk * 256u + lane * 4u;
    if (chunk * 256u >= lshCount) return;
    const bool valid = bit0 < lshCount;                                     // lshCount % 4 == 0 on this path
    const bool sliceMajor = floatCopyIsSliceMajor(lshCount);
    const uint32_t safeBit = valid ? bit0 : 0u;
    const float* column = vectors32 + (sliceMajor ? size_t(safeBit >> 5) * geneCount * 32u + (safeBit & 31u) : size_t(safeBit));
    const size_t rowStride = sliceMajor ? 32u : lshCount;
    double s[4], mx[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        s[t] = valid ? vectorSums[bit0 + t] : 0.;
        mx[t] = valid ? vectorMaxAbs[bit0 + t] : 0.;
    }
    ScalarPtr64 entries = (ScalarPtr64)(uintptr_t)data;
    const uint32_t word = chunk * 4u + (lane >> 4);                         // the 64-bit word this lane's bits are in

    const uint32_t cellBegin = blockIdx.x * kCellsPerBlock;
    const uint32_t cellEnd = min(cellBegin + kCellsPerBlock, cellCount);
    for (uint32_t c = cellBegin; c < cellEnd; ++c) {
        const uint64_t jBegin = toc[c];
        const uint64_t jEnd = toc[c + 1];
        const double mean = means[c];
        double a[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = __dmul_rn(-mean, s[t]);
        uint64_t j = jBegin;
        for (; j + 4 <= jEnd; j += 4) {
            float4 u[4];
            double x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint64_t e = entries[j + q];
                u[q] = *reinterpret_cast<const float4*>(column + size_t(uint32_t(e)) * rowStride);
                x[q] = double(__uint_as_float(uint32_t(e >> 32)));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[0] = __fma_rn(x[q], double(u[q].x), a[0]);
                a[1] = __fma_rn(x[q], double(u[q].y), a[1]);
                a[2] = __fma_rn(x[q], double(u[q].z), a[2]);
                a[3] = __fma_rn(x[q], double(u[q].w), a[3]);
            }
        }
        for (; j < jEnd; ++j) {
            const uint64_t e = entries[j];
            const float4 u = *reinterpret_cast<const float4*>(column + size_t(uint32_t(e)) * rowStride);
            const double x = double(__uint_as_float(uint32_t(e >> 32)));
            a[0] = __fma_rn(x, double(u.x), a[0]);
            a[1] = __fma_rn(x, double(u.y), a[1]);
            a[2] = __fma_rn(x, double(u.z), a[2]);
            a[3] = __fma_rn(x, double(u.w), a[3]);
        }
        // error bound and decision
        const double n = double(jEnd - jBegin);
        const double factor = (1.01 * 5.9604644775390625e-08 + (n + 2.) * 2.220446049250313e-16) * 1.000001;
        const double absMean = fabs(mean);
        const double absX = sumAbs[c];
        uint32_t nibble = 0;
        bool ambiguous = false;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double bound = factor * (absMean * fabs(s[t]) + absX * mx[t]) + absX * 1.5e-45 + 1e-300;   // + float subnormal slack
            ambiguous |= valid && !(fabs(a[t]) > bound);
            nibble |= (valid && a[t] > 0.) ? (8u >> t) : 0u;                 // first bit most significant
        }
        // 16 lanes x 4 bits -> one MSB-first 64-bit word
        uint64_t w = uint64_t(nibble) << (60u - 4u * (lane & 15u));
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const uint32_t lo = uint32_t(__shfl_xor(int(uint32_t(w)), d, 16));
            const uint32_t hi = uint32_t(__shfl_xor(int(uint32_t(w >> 32)), d, 16));
            w |= uint64_t(lo) | (uint64_t(hi) << 32);
        }
        const uint64_t ambMask = __builtin_amdgcn_ballot_w64(ambiguous);
        if ((lane & 15u) == 0u && word < wordCount) {
            signatures[size_t(c) * wordCount + word] = w;
            if (((ambMask >> (lane & 48u)) & 0xffffull) != 0ull) {
                const uint32_t slot = atomicAdd(workCount, 1u);
                workList[slot] = (uint64_t(c) << 32) | word;
            }
        }
    }
}

// Screening pass, XCD-sliced form.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share one),
// so blockIdx.x & 7 picks the 32-bit slice of the signature a block works on: every XCD's 4 MB L2 then serves ONE
// slice of the float hyperplanes (geneCount x 128 B: 3.8 MB at 30k genes) instead of competing for the whole matrix
// (123 MB, served from the Infinity Cache at about half the L2 rate).  A block is 16 cells x 32 bits; a wave takes
// one cell at a time, 8 entries per step: lane = (entry group g = lane/8, 4 bits sub = lane%8), one 16-byte load per
// lane, one 128-byte line per entry.  The partial sums of the 8 groups are added by a butterfly; the screening bound
// covers the different order of the additions.  Requires lshCount % 32 == 0.
// Measured at 1M cells x 30k genes x 1024 bits (projection ms per step): one block per 1024 bits 151; this form
// 110 with a gene-major float copy, 77 with the slice-major copy; non-temporal loads of the CSR entries +8; 16-bit
// slices (EM2_PROJECTION=sliced16, half a cache line per entry) 111.
template <int BITS>       // bits per slice: 32 (one 128-byte line per entry) or 16
__global__ void __launch_bounds__(256)
projectionScreenSlicedKernel(const uint64_t* __restrict__ toc, const CountIn* __restrict__ data, uint32_t cellCount,
                             uint32_t geneCount, const float* __restrict__ vectors32, const double* __restrict__ vectorSums,
                             const double* __restrict__ vectorMaxAbs, const double* __restrict__ means,
                             const double* __restrict__ sumAbs, uint32_t lshCount, uint32_t wordCount,
                             uint64_t* __restrict__ signatures, uint64_t* __restrict__ workList,
                             uint32_t* __restrict__ workCount)
{
    constexpr uint32_t SUB = BITS / 4;             // lanes per entry (4 bits each)
    constexpr uint32_t GROUPS = 64u / SUB;         // entries per wave step
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t bit0 = blockIdx.y * (8u * BITS) + (blockIdx.x & 7u) * BITS;
    if (bit0 >= lshCount) return;
    const uint32_t sub = lane % SUB;
    const uint32_t group = lane / SUB;
    const uint32_t myBit = bit0 + sub * 4u;
    // slice-major float copy: the slice is one contiguous geneCount x 128 B block (a gene-major copy puts the lines
    // of a slice 4 KB apart, which leaves most of the L2's sets unused by it)
    const float* column = vectors32 + size_t(myBit >> 5) * geneCount * 32u + (myBit & 31u);
    double s[4], mx[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        s[t] = vectorSums[myBit + t];
        mx[t] = vectorMaxAbs[myBit + t];
    }
    const uint64_t* entries = reinterpret_cast<const uint64_t*>(data);
    const uint32_t word = bit0 >> 6;
    const uint32_t cellBegin = (blockIdx.x >> 3) * kCellsPerBlock;
    const uint32_t cellEnd = min(cellBegin + kCellsPerBlock, cellCount);
    for (uint32_t c = cellBegin + wave; c < cellEnd; c += 4u) {
        const uint64_t jBegin = toc[c];
        const uint64_t jEnd = toc[c + 1];
        double a[4] = {0., 0., 0., 0.};
        uint64_t j = jBegin + group;
        for (; j + 3u * GROUPS < jEnd; j += 4u * GROUPS) {      // 4 entries per lane in flight
            float4 u[4];
            double x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint64_t e = entries[j + GROUPS * q];
                u[q] = *reinterpret_cast<const float4*>(column + size_t(uint32_t(e)) * 32u);
                x[q] = double(__uint_as_float(uint32_t(e >> 32)));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[0] = __fma_rn(x[q], double(u[q].x), a[0]);
                a[1] = __fma_rn(x[q], double(u[q].y), a[1]);
                a[2] = __fma_rn(x[q], double(u[q].z), a[2]);
                a[3] = __fma_rn(x[q], double(u[q].w), a[3]);
            }
        }
        for (; j < jEnd; j += GROUPS) {
            const uint64_t e = entries[j];
            const float4 u = *reinterpret_cast<const float4*>(column + size_t(uint32_t(e)) * 32u);
            const double x = double(__uint_as_float(uint32_t(e >> 32)));
            a[0] = __fma_rn(x, double(u.x), a[0]);
            a[1] = __fma_rn(x, double(u.y), a[1]);
            a[2] = __fma_rn(x, double(u.z), a[2]);
            a[3] = __fma_rn(x, double(u.w), a[3]);
        }
#pragma unroll
        for (int d = SUB; d < 64; d <<= 1) {
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] += __shfl_xor(a[t], d, 64);
        }
        const double mean = means[c];
        const double n = double(jEnd - jBegin);
        // n products and n + 5 additions per bit in some order: (n + 8) half-ulps cover them
        const double factor = (1.01 * 5.9604644775390625e-08 + (n + 8.) * 2.220446049250313e-16) * 1.000001;
        const double absMean = fabs(mean);
        const double absX = sumAbs[c];
        uint32_t nibble = 0;
        bool ambiguous = false;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double total = a[t] + __dmul_rn(-mean, s[t]);
            const double bound = factor * (absMean * fabs(s[t]) + absX * mx[t]) + absX * 1.5e-45 + 1e-300;
            ambiguous |= !(fabs(total) > bound);
            nibble |= (total > 0.) ? (8u >> t) : 0u;
        }
        uint32_t piece = nibble << (BITS - 4u - 4u * sub);      // first bit most significant
#pragma unroll
        for (int d = 1; d < int(SUB); d <<= 1) piece |= uint32_t(__shfl_xor(int(piece), d, SUB));
        const uint64_t ambMask = __builtin_amdgcn_ballot_w64(ambiguous);
        if (lane == 0u) {
            // the slice's position inside its MSB-first 64-bit word, as a little-endian sub-word store
            const size_t wordIndex = size_t(c) * wordCount + word;
            if (BITS == 32) {
                reinterpret_cast<uint32_t*>(signatures)[wordIndex * 2u + (1u - ((bit0 >> 5) & 1u))] = piece;
            } else {
                reinterpret_cast<uint16_t*>(signatures)[wordIndex * 4u + (3u - ((bit0 >> 4) & 3u))] = uint16_t(piece);
            }
            if ((ambMask & ((1ull << SUB) - 1ull)) != 0ull) {
                const uint32_t slot = atomicAdd(workCount, 1u);
                workList[slot] = (uint64_t(c) << 32) | word;
            }
        }
    }
}

// Second screening tier: the (cell, word) items the 16-bit tier could not decide, on the float copy, one wave per item
// (lane = bit of the word: two 128-byte lines per entry).  Sequential FP64 accumulation in stored order, so the bound of
// projectionScreenKernel applies ((n + 2) half-ulps).  What is still undecided goes to a second list, for the exact
// arithmetic.  Requires the slice-major float copy (lshCount % 32 == 0).
__global__ void __launch_bounds__(256)
projectionScreenItemsKernel(const uint64_t* __restrict__ toc, const CountIn* __restrict__ data, uint32_t geneCount,
                            const float* __restrict__ vectors32, const double* __restrict__ vectorSums,
                            const double* __restrict__ vectorMaxAbs, const double* __restrict__ means,
                            const double* __restrict__ sumAbs, uint32_t lshCount, uint32_t wordCount,
                            uint64_t* __restrict__ signatures, const uint64_t* __restrict__ workList,
                            const uint32_t* __restrict__ workCount, uint64_t* __restrict__ nextList,
                            uint32_t* __restrict__ nextCount)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);
    const uint32_t count = *workCount;
    ScalarPtr64 entries = (ScalarPtr64)(uintptr_t)data;
    for (uint32_t item = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); item < count; item += waves) {
        const uint64_t it = workList[item];
        const uint32_t c = uint32_t(it >> 32);
        const uint32_t word = uint32_t(it);
        const uint32_t bit = word * 64u + lane;           // < lshCount: the signature is whole words here
        const float* column = vectors32 + size_t(bit >> 5) * geneCount * 32u + (bit & 31u);
        const double s = vectorSums[bit];
        const uint64_t jBegin = toc[c];
        const uint64_t jEnd = toc[c + 1];
        const double mean = means[c];
        double a = __dmul_rn(-mean, s);
        uint64_t j = jBegin;
        for (; j + 8u <= jEnd; j += 8u) {            // eight loads in flight, the additions in stored order
            double u[8], x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint64_t e = entries[j + q];
                u[q] = double(column[size_t(uint32_t(e)) * 32u]);
                x[q] = double(__uint_as_float(uint32_t(e >> 32)));
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) a = __fma_rn(x[q], u[q], a);
        }
        for (; j < jEnd; ++j) {
            const uint64_t e = entries[j];
            const double u = double(column[size_t(uint32_t(e)) * 32u]);
            const double x = double(__uint_as_float(uint32_t(e >> 32)));
            a = __fma_rn(x, u, a);
        }
        const double n = double(jEnd - jBegin);
        const double factor = (1.01 * 5.9604644775390625e-08 + (n + 2.) * 2.220446049250313e-16) * 1.000001;
        const double absX = sumAbs[c];
        const double bound = factor * (fabs(mean) * fabs(s) + absX * vectorMaxAbs[bit]) + absX * 1.5e-45 + 1e-300;
        const bool ambiguous = !(fabs(a) > bound);
        const uint64_t mask = __builtin_amdgcn_ballot_w64(a > 0.);
        const uint64_t ambMask = __builtin_amdgcn_ballot_w64(ambiguous);
        if (lane == 0u) {
            signatures[size_t(c) * wordCount + word] = __brevll(mask);
            if (ambMask != 0ull) nextList[atomicAdd(nextCount, 1u)] = it;
        }
    }
}

// The same tier for the words in which ONE bit is undecided, listed by that bit: eight lanes per item, each taking every eighth
// count of the item's cell with eight gathers of a single float in flight -- 4 bytes per count where the word form above reads
// two 128-byte lines; the eight partial sums are added by a butterfly (the bound's (n + 10) half-ulps cover the order of the
// additions, as in the sliced kernel's).  (One lane per item walked its cell alone: 33 dependent rounds of gathers per item,
// 3.3 ms at a million cells; eight lanes per item in list order: 3.0 ms, and 22 GB over the fabric -- every gather a 64-byte
// sector of a 123 MB table.)  The items are taken SLICE BY SLICE, every XCD its own slices: blockIdx & 7 picks the words w with
// w & 7 == that (as in the first tier, whose blocks listed them), and for each of their 32-bit slices in turn the group's waves
// go through the list, 64 items at a time, and take those of the slice -- a slice of the float copy is 128 bytes per gene,
// 3.84 MB at 30 000 genes: it stays in the XCD's L2 while its items are worked on.  The list is read once per slice of a
// group (9 MB per reading at a million cells).  A bit that stays undecided sends its word to the exact tier's list, any other
// goes into the signature (nobody else writes that word: a word is on exactly one of the two lists).
constexpr uint32_t kProjectionBitsBlocks = 2048;       // the grid of projectionScreenBitsKernel: a multiple of 8 (a group per XCD)
__global__ void __launch_bounds__(256)
projectionScreenBitsKernel(const uint64_t* __restrict__ toc, const CountIn* __restrict__ data, uint32_t geneCount,
                           const float* __restrict__ vectors32, const double* __restrict__ vectorSums,
                           const double* __restrict__ vectorMaxAbs, const double* __restrict__ means,
                           const double* __restrict__ sumAbs, uint32_t wordCount, uint64_t* __restrict__ signatures,
                           const uint64_t* __restrict__ bitList, const uint32_t* __restrict__ bitCount,
                           uint64_t* __restrict__ nextList, uint32_t* __restrict__ nextCount)
{
    const uint32_t count = *bitCount;
    const uint64_t* entries = reinterpret_cast<const uint64_t*>(data);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t sub = lane & 7u, slot = lane >> 3;
    const uint32_t group = blockIdx.x & 7u;
    // (kProjectionBitsBlocks: the grid is a multiple of 8 blocks, one group of blocks per XCD; a grid below 8 would leave the
    // stride at 0)
    const uint32_t wavesInGroup = max(1u, gridDim.x >> 3) * (blockDim.x >> 6);
    const uint32_t waveInGroup = (blockIdx.x >> 3) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t chunks = (count + 63u) / 64u;
    for (uint32_t slice = 2u * group; slice < 2u * wordCount; slice += (slice & 1u) ? 15u : 1u) {
      for (uint32_t chunk = waveInGroup; chunk < chunks; chunk += wavesInGroup) {
        const uint32_t index = chunk * 64u + lane;
        const uint64_t mine = index < count ? bitList[index] : 0ull;
        // the chunk's items of this slice, eight at a time: slot k takes the k-th of those still to do.  (Two of a chunk's 64 on
        // average: most slots idle -- collecting eight in LDS first changes nothing, 1.95 ms either way: the gathers of single
        // floats, 2.9e8 sectors per launch, run at the rate of the L2s.)
        uint64_t todo = __builtin_amdgcn_ballot_w64(index < count && (uint32_t(mine) >> 5) == slice);
        while (todo != 0ull) {
            uint64_t rest = todo;
            uint32_t source = 64u;
            for (uint32_t q = 0; q <= slot && rest != 0ull; ++q) {
                source = q == slot ? uint32_t(__builtin_ctzll(rest)) : 64u;
                rest &= rest - 1ull;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) todo &= todo - (todo != 0ull ? 1ull : 0ull);
            const bool live = source < 64u;
            const uint32_t from = live ? source : 0u;
            const uint64_t it = (uint64_t(uint32_t(__shfl(int(uint32_t(mine >> 32)), int(from), 64))) << 32) |
                                uint64_t(uint32_t(__shfl(int(uint32_t(mine)), int(from), 64)));
            const uint32_t c = uint32_t(it >> 32);
            const uint32_t bit = uint32_t(it);
            const float* column = vectors32 + size_t(bit >> 5) * geneCount * 32u + (bit & 31u);
            const uint64_t jBegin = live ? toc[c] : 0ull;
            const uint64_t jEnd = live ? toc[c + 1] : 0ull;
            double a = 0.;
            uint64_t j = jBegin + sub;
            for (; j + 56u < jEnd; j += 64u) {            // eight loads in flight per lane, 64 counts of the cell per round
                double u[8], x[8];
    #pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const uint64_t e = entries[j + 8u * uint32_t(q)];
                    u[q] = double(column[size_t(uint32_t(e)) * 32u]);
                    x[q] = double(__uint_as_float(uint32_t(e >> 32)));
                }
    #pragma unroll
                for (int q = 0; q < 8; ++q) a = __fma_rn(x[q], u[q], a);
            }
            for (; j < jEnd; j += 8u) {
                const uint64_t e = entries[j];
                a = __fma_rn(double(__uint_as_float(uint32_t(e >> 32))), double(column[size_t(uint32_t(e)) * 32u]), a);
            }
    #pragma unroll
            for (int d = 1; d < 8; d <<= 1) a += __shfl_xor(a, d, 64);
            if (!live || sub != 0u) continue;
            const double s = vectorSums[bit];
            const double mean = means[c];
            a += __dmul_rn(-mean, s);
            const double n = double(jEnd - jBegin);
            const double factor = (1.01 * 5.9604644775390625e-08 + (n + 10.) * 2.220446049250313e-16) * 1.000001;
            const double absX = sumAbs[c];
            const double bound = factor * (fabs(mean) * fabs(s) + absX * vectorMaxAbs[bit]) + absX * 1.5e-45 + 1e-300;
            const uint32_t word = bit >> 6;
            if (!(fabs(a) > bound)) {
                nextList[atomicAdd(nextCount, 1u)] = (uint64_t(c) << 32) | word;
            } else {
                const uint64_t mask = 1ull << (63u - (bit & 63u));          // first bit most significant (src/BitSet.hpp:48-62)
                uint64_t* target = signatures + size_t(c) * wordCount + word;
                *target = a > 0. ? (*target | mask) : (*target & ~mask);
            }
        }
      }
    }
}

// Exact recomputation of the listed (cell, word) items: the arithmetic of projectionKernel, one wave per item.
__global__ void __launch_bounds__(256)
projectionExactItemsKernel(const uint64_t* __restrict__ toc, const CountIn* __restrict__ data,
                           const double* __restrict__ vectors, const double* __restrict__ vectorSums,
                           const double* __restrict__ means, uint32_t lshCount, uint32_t wordCount,
                           uint64_t* __restrict__ signatures, const uint64_t* __restrict__ workList,
                           const uint32_t* __restrict__ workCount)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);
    const uint32_t count = *workCount;
    ScalarPtr64 entries = (ScalarPtr64)(uintptr_t)data;
    for (uint32_t item = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); item < count; item += waves) {
        const uint64_t it = workList[item];
        const uint32_t c = uint32_t(it >> 32);
        const uint32_t word = uint32_t(it);
        const uint32_t bit = word * 64u + lane;
        const bool bitValid = bit < lshCount;
        const double* column = vectors + (bitValid ? bit : 0u);
        const double s = bitValid ? vectorSums[bit] : 0.;
        const uint64_t jBegin = toc[c];
        const uint64_t jEnd = toc[c + 1];
        double sp = __dmul_rn(-means[c], s);
        for (uint64_t j = jBegin; j < jEnd; ++j) {
            const uint64_t e = entries[j];
            const double u = column[size_t(uint32_t(e)) * lshCount];
            const double x = double(__uint_as_float(uint32_t(e >> 32)));
            sp = __dadd_rn(sp, __dmul_rn(x, u));
        }
        const uint64_t mask = __builtin_amdgcn_ballot_w64(bitValid && sp > 0.);
        if (lane == 0u) signatures[size_t(c) * wordCount + word] = __brevll(mask);
    }
}

}  // namespace


hipError_t launchCellMeans(const uint64_t* toc, const CountIn* data, uint32_t cellCount, uint32_t geneCount,
                           double* means, hipStream_t stream)
{
    if (cellCount == 0) return hipSuccess;
    cellMeansKernel<<<dim3((cellCount + 255u) / 256u), dim3(256), 0, stream>>>(toc, data, cellCount, geneCount,
                                                                                means);
    return hipGetLastError();
}

hipError_t launchVectorSums(const double* vectors, uint32_t geneCount, uint32_t lshCount, double* sums,
                            hipStream_t stream)
{
    if (lshCount == 0) return hipSuccess;
    vectorSumsKernel<<<dim3((lshCount + 63u) / 64u), dim3(64), 0, stream>>>(vectors, geneCount, lshCount, sums);
    return hipGetLastError();
}

hipError_t launchProjection(const uint64_t* toc, const CountIn* data, uint32_t cellCount, const double* vectors,
                            const double* vectorSums, const double* means, uint32_t lshCount,
                            uint64_t* signatures, hipStream_t stream)
{
    if (cellCount == 0 || lshCount == 0) return hipSuccess;
    const uint32_t wordCount = (lshCount - 1u) / 64u + 1u;
    const dim3 grid((cellCount + kCellsPerBlock - 1u) / kCellsPerBlock, (wordCount + 3u) / 4u);
    projectionKernel<<<grid, dim3(256), 0, stream>>>(toc, data, cellCount, vectors, vectorSums, means, lshCount,
                                                     wordCount, signatures);
    return hipGetLastError();
}

size_t vectorAuxBytes(uint32_t geneCount, uint32_t lshCount)
{
    size_t bytes = 2u * size_t(lshCount) * sizeof(double) + size_t(geneCount) * lshCount * sizeof(float);
    // the 16-bit copy and its scales (first screening tier)
    if (haveQuantizedCopy(lshCount)) bytes += size_t(lshCount) * sizeof(double) + size_t(geneCount) * lshCount * sizeof(int16_t);
    return bytes;
}

hipError_t launchPrepareVectors(const double* vectors, uint32_t geneCount, uint32_t lshCount, void* aux, hipStream_t stream)
{
    if (lshCount == 0) return hipSuccess;
    double* sums = static_cast<double*>(aux);
    double* maxAbs = sums + lshCount;
    float* vectors32 = reinterpret_cast<float*>(maxAbs + lshCount);
    vectorStatsKernel<<<dim3((lshCount + 63u) / 64u), dim3(64), 0, stream>>>(vectors, geneCount, lshCount, sums, maxAbs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const uint64_t count = uint64_t(geneCount) * lshCount;
    if (count) {
        uint64_t blocks = (count + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        vectorsToFloatKernel<<<dim3(uint32_t(blocks)), dim3(256), 0, stream>>>(vectors, geneCount, lshCount,
                                                                               floatCopyIsSliceMajor(lshCount), vectors32);
        if (haveQuantizedCopy(lshCount)) {
            double* scales = reinterpret_cast<double*>(vectors32 + count);
            int16_t* quantized = reinterpret_cast<int16_t*>(scales + lshCount);
            vectorsToQuantizedKernel<<<dim3(uint32_t(blocks)), dim3(256), 0, stream>>>(vectors, geneCount, lshCount, maxAbs, quantized, scales);
        }
    }
    return hipGetLastError();
}

size_t projectionScreenedWorkspaceBytes(uint32_t cellCount, uint32_t lshCount)
{
    const size_t wordCount = (size_t(lshCount) - 1u) / 64u + 1u;
    const size_t a = (size_t(cellCount) * sizeof(double) + 255u) & ~size_t(255u);
    // work list: one slot per (cell, 16-bit slice), the sliced screen can list a word once per slice
    return 2u * a + 256u + ((4u * size_t(cellCount) * wordCount * sizeof(uint64_t) + 255u) & ~size_t(255u));
}

hipError_t launchProjectionScreened(const uint64_t* toc, const CountIn* data, uint32_t cellCount, uint32_t geneCount,
                                    const double* vectors, const void* aux, uint32_t lshCount, uint64_t* signatures,
                                    void* workspace, hipStream_t stream)
{
    if (cellCount == 0 || lshCount == 0) return hipSuccess;
    const uint32_t wordCount = (lshCount - 1u) / 64u + 1u;
    const double* sums = static_cast<const double*>(aux);
    const double* maxAbs = sums + lshCount;
    const float* vectors32 = reinterpret_cast<const float*>(maxAbs + lshCount);
    const size_t a = (size_t(cellCount) * sizeof(double) + 255u) & ~size_t(255u);
    char* ws = static_cast<char*>(workspace);
    double* means = reinterpret_cast<double*>(ws);
    double* sumAbs = reinterpret_cast<double*>(ws + a);
    uint32_t* workCount = reinterpret_cast<uint32_t*>(ws + 2u * a);
    uint64_t* workList = reinterpret_cast<uint64_t*>(ws + 2u * a + 256u);
    hipError_t e = hipMemsetAsync(workCount, 0, 256, stream);
    if (e != hipSuccess) return e;
    // workCount[48]: set by the statistics when some cell's counts are no small integers (then the float first tier runs)
    cellStatsKernel<<<dim3((cellCount + 3u) / 4u), dim3(256), 0, stream>>>(toc, data, cellCount, geneCount, means, sumAbs, workCount + 48);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    // The first tier by shape: a whole number of 64-bit words -- the 16-bit fixed-point copy (half the gathers), what it cannot
    // decide to the float tier on the listed words and bits, the rest of that to the exact arithmetic; a whole number of 32-bit
    // slices -- the XCD-sliced float form; anything else -- one block per 1024 bits on the float copy.
    const bool sliced = lshCount % 32u == 0u;
    const bool quantizedTier = haveQuantizedCopy(lshCount);
    if (quantizedTier) {
        const double* scales = reinterpret_cast<const double*>(vectors32 + size_t(geneCount) * lshCount);
        const int16_t* quantized = reinterpret_cast<const int16_t*>(scales + lshCount);
        const uint32_t cellBlocks = (cellCount + kCellsPerBlock - 1u) / kCellsPerBlock;
        const dim3 grid(cellBlocks * 8u, (wordCount + 7u) / 8u);
#ifdef EM2_DIAG
        const char* diagText = getenv("EM2_PROJECTION_DIAG");
        const int diag = diagText ? atoi(diagText) : 0;
        // (1 / 2: the float form's gathers alone / arithmetic alone; 3 / 4: the same of the integer form)
        auto kernel = diag == 1 ? &projectionScreenQuantizedKernel<1> : (diag == 2 ? &projectionScreenQuantizedKernel<2> :
                      (diag == 3 ? &projectionScreenQuantizedKernel<1, true> : (diag == 4 ? &projectionScreenQuantizedKernel<2, true> : &projectionScreenQuantizedKernel<0>)));
        const bool diagOff = diag == 0;
#else
        auto kernel = &projectionScreenQuantizedKernel<0>;
        const bool diagOff = true;
#endif
        // (the work area holds 4 slots per word: [0, CW) the words for the float tier, [CW, 2 CW) the single bits for its per-bit
        // form, [2 CW, 4 CW) what the two leave to the exact tier)
        const bool perBit = true;
        const bool integerTier = diagOff;
        kernel<<<grid, dim3(256), 0, stream>>>(toc, data, cellCount, geneCount, quantized, scales, sums, maxAbs, means,
                                                                        sumAbs, lshCount, wordCount, signatures, workList, workCount,
                                                                        perBit ? workList + size_t(cellCount) * wordCount : nullptr, workCount + 32,
                                                                        integerTier ? workCount + 48 : nullptr);
        if (integerTier) {
            e = hipGetLastError();
            if (e != hipSuccess) return e;
            projectionScreenQuantizedKernel<0, true><<<grid, dim3(256), 0, stream>>>(
                toc, data, cellCount, geneCount, quantized, scales, sums, maxAbs, means, sumAbs, lshCount, wordCount, signatures, workList,
                workCount, perBit ? workList + size_t(cellCount) * wordCount : nullptr, workCount + 32, workCount + 48);
        }
    } else if (sliced) {
        e = hipMemsetAsync(signatures, 0, size_t(cellCount) * wordCount * sizeof(uint64_t), stream);    // halves nobody owns
        if (e != hipSuccess) return e;
        const uint32_t cellBlocks = (cellCount + kCellsPerBlock - 1u) / kCellsPerBlock;
        const dim3 grid(cellBlocks * 8u, (lshCount + 255u) / 256u);
        projectionScreenSlicedKernel<32><<<grid, dim3(256), 0, stream>>>(toc, data, cellCount, geneCount, vectors32, sums, maxAbs, means, sumAbs,
                                                                         lshCount, wordCount, signatures, workList, workCount);
    } else {
        const dim3 grid((cellCount + kCellsPerBlock - 1u) / kCellsPerBlock, (lshCount + 1023u) / 1024u);
        projectionScreenKernel<<<grid, dim3(256), 0, stream>>>(toc, data, cellCount, geneCount, vectors32, sums, maxAbs, means, sumAbs,
                                                               lshCount, wordCount, signatures, workList, workCount);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (quantizedTier) {
        // second tier on the float copy for what the 16-bit tier left; its own leftovers form a second list (the work
        // area holds 4 slots per word: the second list starts at slot cellCount * wordCount)
        uint64_t* nextList = workList + 2u * size_t(cellCount) * wordCount;
        uint32_t* nextCount = workCount + 16;
        projectionScreenItemsKernel<<<dim3(2048), dim3(256), 0, stream>>>(toc, data, geneCount, vectors32, sums, maxAbs, means, sumAbs, lshCount,
                                                                          wordCount, signatures, workList, workCount, nextList, nextCount);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        static_assert(kProjectionBitsBlocks % 8u == 0u && kProjectionBitsBlocks >= 8u, "projectionScreenBitsKernel: whole groups of 8 blocks");
        projectionScreenBitsKernel<<<dim3(kProjectionBitsBlocks), dim3(256), 0, stream>>>(toc, data, geneCount, vectors32, sums, maxAbs, means, sumAbs, wordCount,
                                                                         signatures, workList + size_t(cellCount) * wordCount, workCount + 32,
                                                                         nextList, nextCount);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        workList = nextList;
        workCount = nextCount;
    }
    projectionExactItemsKernel<<<dim3(1024), dim3(256), 0, stream>>>(toc, data, vectors, sums, means, lshCount, wordCount,
                                                                     signatures, workList, workCount);
    return hipGetLastError();
}

}  // namespace em2
