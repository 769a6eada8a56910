// What v_permlane32_swap / v_permlane16_swap do, lane by lane (gfx950).
// Build: hipcc --offload-arch=gfx950 -O2 tools/permlane_swap_test.hip -o tools/permlane_swap_test
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *o)
{
    const unsigned x = threadIdx.x, y = 100 + threadIdx.x;
    const u32x2 a = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    const u32x2 b = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    o[threadIdx.x] = a[0]; o[64 + threadIdx.x] = a[1]; o[128 + threadIdx.x] = b[0]; o[192 + threadIdx.x] = b[1];
}
int main()
{
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[4] = {"permlane32_swap[0]", "permlane32_swap[1]", "permlane16_swap[0]", "permlane16_swap[1]"};
    for (int q = 0; q < 4; ++q) {
        printf("%s: rows (lane 0 of each 16-lane row):", names[q]);
        for (int r = 0; r < 4; ++r) printf(" %u", h[64 * q + 16 * r]);
        printf("\n");
    }
    return 0;
}
