// Does v_mfma_f32_16x16x32_bf16 add C before or after the products?  Products that cancel
// exactly, C = 1e-30: D == 1e-30 means C survives (added last or carried exactly), D == 0 means
// it was absorbed by a large partial sum first.
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_c_order_test.hip -o tools/mfma_c_order_test
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(float *out, int pattern)
{
    const int lane = threadIdx.x, g = lane >> 4;
    // bf16 constants
    const short one = 0x3f80, mone = (short)0xbf80, half = 0x3f00, big = 0x4780 /* 65536 */, mbig = (short)0xc780;
    bf16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    if (pattern == 0) {            // +1*1 and -1*1 in the same lane group
        if (g == 0) { a[0] = one; b[0] = one; a[1] = mone; b[1] = one; }
    } else if (pattern == 1) {     // cancelling pair split across lane groups 0 and 3 (k = 0 and k = 24)
        if (g == 0) { a[0] = one; b[0] = one; }
        if (g == 3) { a[0] = mone; b[0] = one; }
    } else if (pattern == 2) {     // large terms: 65536 * 65536 cancels, in different groups
        if (g == 1) { a[3] = big; b[3] = big; }
        if (g == 2) { a[5] = mbig; b[5] = big; }
    } else if (pattern == 3) {     // the d2 shape: 0.25 - 0.5 + 0.25 spread over three groups
        if (g == 0) { a[0] = half; b[0] = half; }
        if (g == 1) { a[0] = mone; b[0] = half; }
        if (g == 3) { a[0] = half; b[0] = half; }
    } else {                       // many terms: every k slot +x then -x alternating
        for (int k = 0; k < 8; ++k) { a[k] = (k & 1) ? mone : one; b[k] = one; }
    }
    const f32x4 c = {1e-30f, 1e-30f, 1e-30f, 1e-30f};
    const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    out[lane * 4 + 0] = d[0]; out[lane * 4 + 1] = d[1]; out[lane * 4 + 2] = d[2]; out[lane * 4 + 3] = d[3];
}

int main()
{
    float *d; hipMalloc(&d, 256 * sizeof(float));
    float h[256];
    for (int p = 0; p < 5; ++p) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, p);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int n_c = 0, n_zero = 0, n_other = 0;
        for (int i = 0; i < 256; ++i) { if (h[i] == 1e-30f) ++n_c; else if (h[i] == 0.f) ++n_zero; else ++n_other; }
        printf("pattern %d: D == C in %d entries, D == 0 in %d, other in %d (first other %g)\n", p, n_c, n_zero, n_other, h[0]);
    }
    return 0;
}
