// Calibration of rocprofv3's FETCH_SIZE for the scan kernel's access pattern: every wave streams a disjoint,
// contiguous region once through the scalar unit (s_load_dwordx16), nothing is re-read, the buffer (2 GiB) is far
// larger than the Infinity Cache.  Known byte count = the buffer size.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_sload tools/ubench_sload_fetch.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -o c -- /tmp/ubench_sload
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef const __attribute__((address_space(4))) uint32_t* ScalarPtr;

__global__ void __launch_bounds__(256) streamScalar(const uint32_t* __restrict__ data, uint64_t dwordsPerWave, uint32_t* out)
{
    const uint64_t wave = uint64_t(blockIdx.x) * 4u + uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)));
    ScalarPtr p = (ScalarPtr)(uintptr_t)data + wave * dwordsPerWave;
    uint32_t acc = 0;
    for (uint64_t i = 0; i < dwordsPerWave; i += 16) {
#pragma unroll
        for (int w = 0; w < 16; ++w) acc ^= p[i + w];
    }
    if (acc == 0x12345678u) out[0] = acc;        // keeps the loads alive
}

int main()
{
    const uint64_t bytes = 2ull << 30;
    const uint32_t waves = 256u * 16u * 8u;                       // 32768 waves
    const uint64_t dwordsPerWave = bytes / 4 / waves;             // 16 Ki dwords = 64 KiB per wave
    uint32_t* d = nullptr;
    uint32_t* out = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
    (void)hipMemset(d, 1, bytes);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) streamScalar<<<waves / 4, 256>>>(d, dwordsPerWave, out);
    (void)hipDeviceSynchronize();
    printf("streamed %llu bytes per launch through s_load_dwordx16\n", (unsigned long long)bytes);
    return 0;
}
