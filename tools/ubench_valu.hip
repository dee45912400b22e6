// ubench_valu.hip -- what is the real instruction roofline of XOR+popcount on this chip?
// Measures v_xor_b32(sgpr,vgpr)+v_bcnt_u32_b32 throughput at 1,2,4,8 waves per SIMD, with the column operand
// (a) held in SGPRs (no memory), (b) streamed with s_load_dwordx16 like the scan kernel.
// Build & run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef const __attribute__((address_space(4))) uint32_t* ScalarPtr;

__global__ void __launch_bounds__(256) kRegs(const uint32_t* __restrict__ sig, uint32_t iters, uint32_t* out)
{
    uint32_t r[32];
    const uint32_t* rp = sig + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) % 4096 * 32;
#pragma unroll
    for (int w = 0; w < 32; ++w) r[w] = rp[w];
    ScalarPtr p = (ScalarPtr)(uintptr_t)sig;
    uint32_t c[32];
#pragma unroll
    for (int w = 0; w < 32; ++w) c[w] = p[w];
    uint32_t best = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        uint32_t m = 0;
#pragma unroll
        for (int w = 0; w < 32; ++w) m += __builtin_popcount(r[w] ^ c[w]);
        best += (m <= 3u) ? 1u : 0u;
        // make the SGPR operands loop-variant without memory traffic
#pragma unroll
        for (int w = 0; w < 32; w += 8) c[w] += i;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = best;
}

__global__ void __launch_bounds__(256) kStream(const uint32_t* __restrict__ sig, uint32_t cols, uint32_t* out)
{
    uint32_t r[32];
    const uint32_t* rp = sig + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) % 4096 * 32;
#pragma unroll
    for (int w = 0; w < 32; ++w) r[w] = rp[w];
    ScalarPtr p = (ScalarPtr)(uintptr_t)sig;
    uint32_t chunk[2][32];
#pragma unroll
    for (int w = 0; w < 32; ++w) chunk[0][w] = p[w];
    uint32_t best = 0;
    for (uint32_t col = 0; col < cols; col += 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_sched_barrier(0);
            ScalarPtr pn = (col + s + 1 < cols) ? p + 32 : p;
#pragma unroll
            for (int w = 0; w < 32; ++w) chunk[(s + 1) & 1][w] = pn[w];
            p = pn;
            __builtin_amdgcn_sched_barrier(0);
            uint32_t m = 0;
#pragma unroll
            for (int w = 0; w < 32; ++w) m += __builtin_popcount(r[w] ^ chunk[s & 1][w]);
            best += (m <= 3u) ? 1u : 0u;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = best;
}

template <int N> __device__ __forceinline__ uint32_t bcast16(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x150 + N, 0xf, 0xf, false);
}
__device__ __forceinline__ void bcntAcc(uint32_t& m, uint32_t x)
{
    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(m) : "v"(x));
}
template <int W> struct AccDpp {
    static __device__ __forceinline__ void run(const uint32_t (&r)[32], uint32_t c0, uint32_t c1, uint32_t& m)
    {
        bcntAcc(m, r[W] ^ bcast16<(W >> 1)>((W & 1) ? c1 : c0));
        AccDpp<W + 1>::run(r, c0, c1, m);
    }
};
template <> struct AccDpp<32> {
    static __device__ __forceinline__ void run(const uint32_t (&)[32], uint32_t, uint32_t, uint32_t&) {}
};

// column operand in 2 VGPRs (lane l holds dwords 2(l&15), 2(l&15)+1), broadcast with DPP row_newbcast
__global__ void __launch_bounds__(256) kDppRegs(const uint32_t* __restrict__ sig, uint32_t iters, uint32_t* out)
{
    uint32_t r[32];
    const uint32_t* rp = sig + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) % 4096 * 32;
#pragma unroll
    for (int w = 0; w < 32; ++w) r[w] = rp[w];
    uint32_t c0 = rp[3], c1 = rp[5];
    uint32_t best = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        uint32_t m = 0;
        AccDpp<0>::run(r, c0, c1, m);
        best += (m <= 3u) ? 1u : 0u;
        c0 += i; c1 ^= i;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = best;
}

template <int DEPTH>
__global__ void __launch_bounds__(256) kDppStream(const uint32_t* __restrict__ sig, uint32_t cols, uint32_t* out)
{
    uint32_t r[32];
    const uint32_t* rp = sig + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) % 4096 * 32;
#pragma unroll
    for (int w = 0; w < 32; ++w) r[w] = rp[w];
    const uint32_t* base = sig + (threadIdx.x & 15u) * 2u;
    uint2 q[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) q[d] = *reinterpret_cast<const uint2*>(base + (size_t)d * 32);
    uint32_t best = 0;
    for (uint32_t col = 0; col < cols; col += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const uint2 cur = q[d];
            uint32_t nc = col + d + DEPTH;
            nc = nc < cols ? nc : cols - 1;
            q[d] = *reinterpret_cast<const uint2*>(base + (size_t)nc * 32);
            uint32_t m = 0;
            AccDpp<0>::run(r, cur.x, cur.y, m);
            best += (m <= 3u) ? 1u : 0u;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = best;
}

int main()
{
    const uint32_t cells = 1u << 20;
    std::vector<uint32_t> h((size_t)cells * 32);
    uint64_t s = 88172645463325252ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)s; }
    uint32_t *d, *o;
    hipMalloc(&d, h.size() * 4);
    hipMalloc(&o, 256 * 4 * 8 * 64 * 4 * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int wavesPerSimd[] = {1, 2, 4, 5, 8};
    for (int mode = 0; mode < 5; ++mode) {
        for (int wps : wavesPerSimd) {
            const int blocks = 256 * wps;      // 256 CUs x wps blocks of 4 waves
            const uint32_t n = 200000u;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) kRegs<<<blocks, 256>>>(d, n, o);
                else if (mode == 1) kStream<<<blocks, 256>>>(d, n, o);
                else if (mode == 2) kDppRegs<<<blocks, 256>>>(d, n, o);
                else if (mode == 3) kDppStream<4><<<blocks, 256>>>(d, n, o);
                else kDppStream<8><<<blocks, 256>>>(d, n, o);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const double waveInstr = (double)n * 64.0;                 // xor+bcnt per wave
            const double perSimd = waveInstr * wps;                    // instructions issued per SIMD
            const double cyc = ms * 1e-3 * 2.4e9 / perSimd;
            printf("%s waves/SIMD=%d  %.3f ms  -> %.2f cycles(@2.4GHz) per wave-instruction per SIMD; "
                   "%.3e comparisons/s\n", mode == 0 ? "sgpr-regs  " : mode == 1 ? "sgpr-stream" : mode == 2 ? "dpp-regs   " : mode == 3 ? "dpp-stream4" : "dpp-stream8", wps, ms, cyc,
                   (double)n * 64.0 * 4 * wps * 256 / (ms * 1e-3));
        }
    }
    return 0;
}
