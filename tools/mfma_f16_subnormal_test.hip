// Does v_mfma_f32_16x16x16_f16 honour subnormal fp16 inputs?  A = subnormal (2^-20), B = 1024:
// exact product 2^-10.  Flushed inputs give 0.
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_f16_subnormal_test.hip -o tools/mfma_f16_subnormal_test
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float *out, float av, float bv)
{
    f16x4 a = {(_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0}, b = a;
    if ((threadIdx.x >> 4) == 0) { a[0] = (_Float16)av; b[0] = (_Float16)bv; }
    const f32x4 c = {0.f, 0.f, 0.f, 0.f};
    const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
    out[threadIdx.x] = d[0];
}
int main()
{
    float *d; hipMalloc(&d, 64 * sizeof(float)); float h[64];
    const float cases[][2] = {{9.5367431640625e-07f, 1024.f}, {5.9604644775390625e-08f, 16384.f}, {6.103515625e-05f, 1.f}, {3.0517578125e-05f, 2.f}};
    for (auto &c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c[0], c[1]);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("a = %g (fp16 %s), b = %g: D = %g, exact %g\n", c[0], c[0] < 6.103515625e-05f ? "subnormal" : "normal", c[1], h[0], c[0] * c[1]);
    }
    return 0;
}
