// Microbenchmark: how fast does one SIMD take v_mfma_f32_32x32x64_f8f6f4 (FP4 operands) when nothing else is in the way?
// One block of 64 * wavesPerSimd * 4 threads per CU, every wave issues `rounds` x 32 MFMAs into two accumulators
// (alternating, as the scan's tile step does), operands constant, no memory traffic.  Prints cycles per MFMA per SIMD
// (s_memtime is a constant-rate counter; the core clock is taken from the wall time) for 1 and 2 waves per SIMD, and the
// same with 4 independent VALU instructions between the MFMAs of a k-step (the tests of the scan).
//   hipcc --offload-arch=gfx950 -O2 -o ubench_mfma_issue tools/ubench_mfma_issue.hip && ./ubench_mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int VALU>
__global__ void __launch_bounds__(512) mfmaLoop(int rounds, float* out)
{
    float acc = 0.f;
    asm volatile(
        "v_mov_b32 v40, 0x22222222\n v_mov_b32 v41, 0x22222222\n v_mov_b32 v42, 0x22222222\n v_mov_b32 v43, 0x22222222\n"
        "v_mov_b32 v44, 0xaaaaaaaa\n v_mov_b32 v45, 0x22222222\n v_mov_b32 v46, 0xaaaaaaaa\n v_mov_b32 v47, 0x22222222\n"
        "v_mov_b32 v20, 1.0\n v_mov_b32 v21, 2.0\n"
        ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v20", "v21");
    for (int r = 0; r < rounds; r++) {
#define KSTEP(first)                                                                                                   \
        asm volatile("v_mfma_f32_32x32x64_f8f6f4 v[64:79], v[40:43], v[44:47], " first " cbsz:4 blgp:4\n"              \
                     "v_mfma_f32_32x32x64_f8f6f4 v[80:95], v[40:43], v[44:47], " first " cbsz:4 blgp:4" :::            \
                     "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79",  \
                     "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95"); \
        if (VALU) asm volatile("v_min_f32 v22, v20, v21\n v_min_f32 v23, v21, v20\n v_cmp_le_f32_e64 s[20:21], v22, v20\n v_cmp_le_f32_e64 s[22:23], v23, v21" ::: "v22", "v23", "s20", "s21", "s22", "s23");
        KSTEP("0")
#undef KSTEP
#define KSTEP2                                                                                                         \
        asm volatile("v_mfma_f32_32x32x64_f8f6f4 v[64:79], v[40:43], v[44:47], v[64:79] cbsz:4 blgp:4\n"              \
                     "v_mfma_f32_32x32x64_f8f6f4 v[80:95], v[40:43], v[44:47], v[80:95] cbsz:4 blgp:4" :::            \
                     "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79",  \
                     "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95"); \
        if (VALU) asm volatile("v_min_f32 v22, v20, v21\n v_min_f32 v23, v21, v20\n v_cmp_le_f32_e64 s[20:21], v22, v20\n v_cmp_le_f32_e64 s[22:23], v23, v21" ::: "v22", "v23", "s20", "s21", "s22", "s23");
        KSTEP2 KSTEP2 KSTEP2 KSTEP2 KSTEP2 KSTEP2 KSTEP2 KSTEP2 KSTEP2 KSTEP2 KSTEP2 KSTEP2 KSTEP2 KSTEP2 KSTEP2
#undef KSTEP2
    }
    asm volatile("s_nop 15\n s_nop 15\n v_mov_b32 %0, v64" : "=v"(acc));
    if (acc == 12345.f) out[0] = acc;
}

template <int VALU>
static void run(int wavesPerSimd, int rounds)
{
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int threads = 64 * 4 * wavesPerSimd;
    mfmaLoop<VALU><<<cus, threads>>>(10, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    mfmaLoop<VALU><<<cus, threads>>>(rounds, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double mfmaPerSimd = double(rounds) * 32.0 * wavesPerSimd;
    const double flops = mfmaPerSimd * 4.0 * cus * 131072.0;
    printf("%d wave(s) per SIMD, %s: %.2f ms, %.1f ns per MFMA per SIMD = %.1f cycles at 2.4 GHz, %.2f PFLOP/s\n", wavesPerSimd,
           VALU ? "4 VALU per k-step" : "MFMA only", ms, ms * 1e6 / mfmaPerSimd, ms * 1e6 / mfmaPerSimd * 2.4, flops / (ms * 1e-3) / 1e15);
    hipFree(out);
}

int main()
{
    const int rounds = 20000;
    run<0>(1, rounds);
    run<0>(2, rounds);
    run<1>(1, rounds);
    run<1>(2, rounds);
    return 0;
}
