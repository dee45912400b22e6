// Microbenchmark: the generated tile step of the matrix-core walk (csrc/em2_matrix_step_asm.h) alone, back to back,
// 4 waves per block, 1 or 2 blocks per CU, no DMA, no barriers:
//   plain    32 MFMAs + their 16 fragment reads from LDS through the register ring (what the step's structure costs against
//            the raw MFMA rate of tools/ubench_mfma_issue.hip);
//   tests    the same with the test of the previous tile between the MFMAs, bounds that nothing passes;
//   events   the same with random +-1 operands and bounds at `sigmas` standard deviations of the dot product (sigma = 32):
//            3.0 gives the scan's rate of about four records per wave and tile; the stubs store their records.
//   hipcc --offload-arch=gfx950 -O2 -I expressionmatrix2_amd/csrc -o ubench_matrix_step tools/ubench_matrix_step.hip
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"          // (hip calls whose status nobody reads: a microbenchmark)
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "em2_matrix_step_asm.h"
// The 0 / 1 steps (EM2_MATRIX_ZERO_ONE) restart the accumulators from row + column terms: zeros here (a block of 256 bytes
// behind the waves' bounds), so the record rates follow the bounds as they do with the +-1 steps.
#if EM2_MATRIX_ZERO_ONE
#define UBENCH_TERM_OPERANDS(base) , "s"(base), "v"(0.f), "v"(0.f)
#else
#define UBENCH_TERM_OPERANDS(base)
#endif
#ifndef UBENCH_TILE_BIT
#define UBENCH_TILE_BIT 0u          // (-DUBENCH_TILE_BIT=0x80000000u for EM2_GEN_STUB=pend: a record word that is never 0)
#endif

// LDS: 4 tiles (64 KB), then per wave: rowDot float[64] (256 B) + bounds float[4][32] (512 B) + terms float[2][32] (256 B)
__global__ void __launch_bounds__(256) stepLoop(int rounds, int mode, float bound, const unsigned* tiles, const unsigned* rows,
                                                unsigned long long* logs, unsigned* counts, float* out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (unsigned i = threadIdx.x; i < 65536 / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = tiles[i];
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* walk = reinterpret_cast<float*>(lds + 65536 + wave * 1024);
    walk[lane] = bound;
    for (int i = lane; i < 128; i += 64) walk[64 + i] = bound;
    walk[192 + lane] = 0.f;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)lds);
    const unsigned walkLds = unsigned(__builtin_amdgcn_readfirstlane(int(base + 65536 + wave * 1024)));
    auto uniform64 = [](unsigned long long v) {
        return (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(int(unsigned(v))) |
               ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(int(unsigned(v >> 32))) << 32);
    };
    const unsigned long long rowAddress = uniform64(reinterpret_cast<unsigned long long>(rows) + size_t(blockIdx.x * 4 + wave) * 32768);
    const unsigned long long logBase = uniform64(reinterpret_cast<unsigned long long>(logs) + size_t(blockIdx.x * 4 + wave) * 64 * 4096 * EM2_MATRIX_RECORD_BYTES);
    asm volatile(EM2_MATRIX_LOAD_ROWS : : "s"(rowAddress) : EM2_MATRIX_STEP_CLOBBERS);
    unsigned offset = lane * 4096 * EM2_MATRIX_RECORD_BYTES, offset1 = lane * 4096 * EM2_MATRIX_RECORD_BYTES + 2048 * EM2_MATRIX_RECORD_BYTES;
    asm volatile(EM2_MATRIX_SET_RECORD_OFFSETS : : "v"(offset), "v"(offset1) : EM2_MATRIX_OWNED_REGISTERS);
    unsigned long long scratch[5];
    unsigned recordOffset = offset, recordOffset1 = offset1;
    for (int r = 0; r < rounds; r++) {
        const unsigned t0 = unsigned(__builtin_amdgcn_readfirstlane(int(base + ((2 * r) & 3) * 16384)));
        const unsigned t1 = unsigned(__builtin_amdgcn_readfirstlane(int(base + ((2 * r + 1) & 3) * 16384)));
        if (mode == 0) {
            asm volatile(EM2_MATRIX_STEP_X : : "s"(t0) : EM2_MATRIX_STEP_CLOBBERS);
            asm volatile(EM2_MATRIX_STEP_Y : : "s"(t1) : EM2_MATRIX_STEP_CLOBBERS);
#ifdef EM2_MATRIX_PAIR_TESTING
        } else if (mode == 2) {
            // the two steps as one statement (tiles of a pair are adjacent in LDS: pairs 0/1 and 2/3)
            const unsigned pairBase = unsigned(__builtin_amdgcn_readfirstlane(int(base + (r & 1) * 32768)));
            asm volatile(EM2_MATRIX_PAIR_TESTING
                         : "=v"(recordOffset), "=v"(recordOffset1), "=&s"(scratch[0]), "=&s"(scratch[1]), "=&s"(scratch[2]), "=&s"(scratch[3]), "=&s"(scratch[4])
                         : "s"(pairBase), "s"(walkLds + 256), "s"(walkLds + 256 + 128), "s"(walkLds), "s"(logBase), "s"(unsigned(r) * 64u),
                           "s"(unsigned(r) * 64u + 32u)
                         : EM2_MATRIX_STEP_CLOBBERS);
            if ((r & 15) == 15) {
                counts[(blockIdx.x * 4 + wave) * 64 + lane] += (recordOffset - offset) / EM2_MATRIX_RECORD_BYTES + (recordOffset1 - offset1) / EM2_MATRIX_RECORD_BYTES;
                asm volatile(EM2_MATRIX_SET_RECORD_OFFSETS : : "v"(offset), "v"(offset1) : EM2_MATRIX_OWNED_REGISTERS);
            }
#endif
        } else {
            asm volatile(EM2_MATRIX_STEP_X_TESTING_Y
                         : "=v"(recordOffset), "=v"(recordOffset1), "=&s"(scratch[0]), "=&s"(scratch[1]), "=&s"(scratch[2]), "=&s"(scratch[3]), "=&s"(scratch[4])
                         : "s"(t0), "s"(walkLds + 256), "s"(walkLds), "s"(logBase), "s"(unsigned(r) * 64u | UBENCH_TILE_BIT)
                           UBENCH_TERM_OPERANDS(walkLds + 768)
                         : EM2_MATRIX_STEP_CLOBBERS);
            asm volatile(EM2_MATRIX_STEP_Y_TESTING_X
                         : "=v"(recordOffset), "=v"(recordOffset1), "=&s"(scratch[0]), "=&s"(scratch[1]), "=&s"(scratch[2]), "=&s"(scratch[3]), "=&s"(scratch[4])
                         : "s"(t1), "s"(walkLds + 256 + 128), "s"(walkLds), "s"(logBase), "s"(unsigned(r) * 64u + 32u | UBENCH_TILE_BIT)
                           UBENCH_TERM_OPERANDS(walkLds + 768 + 128)
                         : EM2_MATRIX_STEP_CLOBBERS);
            if ((r & 15) == 15) {       // (the log of a lane holds 4096 records: start over)
                counts[(blockIdx.x * 4 + wave) * 64 + lane] += (recordOffset - offset) / EM2_MATRIX_RECORD_BYTES + (recordOffset1 - offset1) / EM2_MATRIX_RECORD_BYTES;
                asm volatile(EM2_MATRIX_SET_RECORD_OFFSETS : : "v"(offset), "v"(offset1) : EM2_MATRIX_OWNED_REGISTERS);
            }
        }
    }
    float acc;
    asm volatile("s_nop 15\n s_nop 15\n v_mov_b32 %0, v64" : "=v"(acc));
    if (acc == 12345.f) out[0] = acc;
}

static void run(const char* name, int blocksPerCu, int rounds, int mode, float bound, const unsigned* tiles, const unsigned* rows,
                unsigned long long* logs, unsigned* counts)
{
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
#ifdef UBENCH_EXTRA_LDS
    const size_t ldsBytes = 65536 + 4 * 1024 + 4096;          // (a spare 4 KB behind the walk blocks: EM2_GEN_STUB=dsw writes there)
#else
    const size_t ldsBytes = 65536 + 4 * 1024;
#endif
    hipFuncSetAttribute(reinterpret_cast<const void*>(&stepLoop), hipFuncAttributeMaxDynamicSharedMemorySize, int(ldsBytes));
    stepLoop<<<cus * blocksPerCu, 256, ldsBytes>>>(16, mode, bound, tiles, rows, logs, counts, out);
    hipDeviceSynchronize();
    hipMemset(counts, 0, size_t(cus) * 2 * 4 * 64 * 4);
    hipEventRecord(a);
    stepLoop<<<cus * blocksPerCu, 256, ldsBytes>>>(rounds, mode, bound, tiles, rows, logs, counts, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned> host(size_t(cus) * 2 * 4 * 64);
    hipMemcpy(host.data(), counts, host.size() * 4, hipMemcpyDeviceToHost);
    double records = 0;
    for (unsigned c : host) records += c;
    const double waveSteps = double(rounds) * 2.0 * 4.0 * blocksPerCu * cus;
    const double mfmaPerSimd = double(rounds) * 64.0 * blocksPerCu;
    const double flops = mfmaPerSimd * 4.0 * cus * 131072.0;
    printf("%-28s %d block(s) per CU: %.2f ms, %.1f cycles (2.4 GHz) per MFMA per SIMD, %.2f PFLOP/s, %.2f records per wave and tile\n", name,
           blocksPerCu, ms, ms * 1e6 / mfmaPerSimd * 2.4, flops / (ms * 1e-3) / 1e15, records / waveSteps);
    hipFree(out);
}

// ubench_matrix_step [codeA0 codeA1 codeB0 codeB1]: the FP4 (E2M1) codes that stand for a signature bit 0 / 1 in the column tiles
// (A) and in the rows (B); default 2 10 2 10 = +1 / -1 on both sides, the product's encoding.  (Round 4: does another pair of
// values -- 0 / 1, +-0.5, +-2 ... -- cost the matrix pipe less power, i.e. hold a higher clock?  Only "plain" is meaningful then.)
int main(int argc, char** argv)
{
    const unsigned codeA0 = argc > 4 ? unsigned(atoi(argv[1])) : 0x2u, codeA1 = argc > 4 ? unsigned(atoi(argv[2])) : 0xAu;
    const unsigned codeB0 = argc > 4 ? unsigned(atoi(argv[3])) : 0x2u, codeB1 = argc > 4 ? unsigned(atoi(argv[4])) : 0xAu;
    printf("codes: columns %x / %x, rows %x / %x\n", codeA0, codeA1, codeB0, codeB1);
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const size_t waves = size_t(cus) * 2 * 4;
    std::vector<unsigned> tiles(65536 / 4), rows(waves * 32768 / 4);
    unsigned long long state = 88172645463325252ull;
    auto next = [&]() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return state; };
    auto nibbles = [&](unsigned c0, unsigned c1) { unsigned w = 0; const unsigned long long r = next(); for (int n = 0; n < 8; n++) w |= (((r >> n) & 1u) ? c1 : c0) << (4 * n); return w; };
    for (auto& w : tiles) w = nibbles(codeA0, codeA1);
    for (auto& w : rows) w = nibbles(codeB0, codeB1);
    unsigned *dTiles, *dRows, *dCounts;
    unsigned long long* dLogs;
    hipMalloc(&dTiles, 65536);
    hipMalloc(&dRows, rows.size() * 4);
    hipMalloc(&dLogs, size_t(waves) * 64 * 4096 * EM2_MATRIX_RECORD_BYTES);
    hipMalloc(&dCounts, waves * 64 * 4);
    hipMemcpy(dTiles, tiles.data(), 65536, hipMemcpyHostToDevice);
    hipMemcpy(dRows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice);
    const int rounds = 20000;
    run("plain", 1, rounds, 0, 0.f, dTiles, dRows, dLogs, dCounts);
    run("plain", 2, rounds, 0, 0.f, dTiles, dRows, dLogs, dCounts);
    if (argc == 6) return 0;          // (a fifth argument alone: the plain runs only)
    // (argv[5], argv[6]: mean and standard deviation of the dot product under these codes, for the bounds of the event runs)
    const float mean = argc > 6 ? float(atof(argv[5])) : 0.f, sigma = argc > 6 ? float(atof(argv[6])) : 32.f;
    run("tests, nothing passes", 2, rounds, 1, 1e9f, dTiles, dRows, dLogs, dCounts);
    run("pair: tests, nothing passes", 2, rounds, 2, 1e9f, dTiles, dRows, dLogs, dCounts);
    // (3.28 sigma: about one record per wave and tile, the rate of the bench's data in round 6's diagnostic build)
    const float sigmas[] = {3.5f, 3.28f, 3.0f, 2.5f};
    for (float s : sigmas) {
        char name[64];
        snprintf(name, sizeof(name), "events, bound %.1f sigma", s);
        run(name, 2, rounds, 1, mean + sigma * s, dTiles, dRows, dLogs, dCounts);
        snprintf(name, sizeof(name), "pair: events, %.1f sigma", s);
        run(name, 2, rounds, 2, mean + sigma * s, dTiles, dRows, dLogs, dCounts);
    }
    return 0;
}
