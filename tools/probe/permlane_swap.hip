// What v_permlane16_swap / v_permlane32_swap return on gfx950 when both operands hold the lane id.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/permlane_swap tools/probe/permlane_swap.hip && /tmp/permlane_swap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out)
{
    const unsigned x = threadIdx.x;
    auto r16 = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    auto r32 = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    out[x] = r16[0];
    out[64 + x] = r16[1];
    out[128 + x] = r32[0];
    out[192 + x] = r32[1];
}
int main()
{
    unsigned* d;
    unsigned h[256];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"permlane16_swap [0]", "permlane16_swap [1]", "permlane32_swap [0]", "permlane32_swap [1]"};
    for (int r = 0; r < 4; r++) {
        printf("%s:", names[r]);
        for (int i = 0; i < 64; i += 8) printf(" %u", h[64 * r + i]);
        printf("\n");
    }
    return 0;
}
