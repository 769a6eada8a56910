// Latency of dependent chains on one wave of gfx950 (what the register-resident build's diagonal block is made of):
// v_fma_f64, v_rsq_f64, v_readlane -> VALU, v_cndmask, v_mfma_f64_16x16x4_f64 (accumulator chain and operand chain).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_f64 tools/ubench_f64.hip && ./tools/ubench_f64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int KIND>
__global__ void k(double *out, unsigned long long *cyc, double seed, int n)
{
    double x = seed + threadIdx.x * 1e-9, y = seed * 0.5;
    double4_t acc = {x, y, x, y};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        if (KIND == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) x = fma(x, 1.0000001, y);
        } else if (KIND == 1) {
#pragma unroll
            for (int q = 0; q < 16; ++q) x = __builtin_amdgcn_rsq(x) + 1.5;
        } else if (KIND == 2) {
#pragma unroll
            for (int q = 0; q < 16; ++q) x = fma(readlane_f64(x, q), 1.0000001, y);
        } else if (KIND == 3) {
#pragma unroll
            for (int q = 0; q < 16; ++q) x = (threadIdx.x == (unsigned)q) ? y : x + 1.0;
        } else if (KIND == 4) {
#pragma unroll
            for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0);
        } else if (KIND == 5) {
#pragma unroll
            for (int q = 0; q < 16; ++q) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, (double4_t){0, 0, 0, 0}, 0, 0, 0); x = acc[0] * 1e-3; }
        } else if (KIND == 6) {
#pragma unroll
            for (int q = 0; q < 16; ++q) x = fma(x, 1.0000001, y) , y = fma(y, 0.9999999, 0.25);      // two independent chains
        } else if (KIND == 7) {
#pragma unroll
            for (int q = 0; q < 16; ++q) x = __builtin_amdgcn_rcp(x) + 1.5;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[KIND] = t1 - t0;
    out[threadIdx.x + 64 * KIND] = x + y + acc[0] + acc[1] + acc[2] + acc[3];
}
int main()
{
    double *out; unsigned long long *cyc;
    hipMalloc(&out, 64 * 16 * 8); hipMalloc(&cyc, 16 * 8);
    const int n = 1000;
    for (int rep = 0; rep < 2; ++rep) {
        k<0><<<1, 64>>>(out, cyc, 1.25, n); k<1><<<1, 64>>>(out, cyc, 1.25, n); k<2><<<1, 64>>>(out, cyc, 1.25, n); k<3><<<1, 64>>>(out, cyc, 1.25, n);
        k<4><<<1, 64>>>(out, cyc, 1.25, n); k<5><<<1, 64>>>(out, cyc, 1.25, n); k<6><<<1, 64>>>(out, cyc, 1.25, n); k<7><<<1, 64>>>(out, cyc, 1.25, n);
        hipDeviceSynchronize();
    }
    unsigned long long h[16];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[8] = {"dependent v_fma_f64", "dependent v_rsq_f64 + v_add_f64", "2 x v_readlane + v_fma_f64 (dependent)", "v_cmp + 2 x v_cndmask + v_add_f64 (dependent)",
                            "v_mfma_f64_16x16x4 accumulator chain", "v_mfma_f64_16x16x4 -> v_mul_f64 -> operand of the next", "two independent v_fma_f64 chains (per pair)", "dependent v_rcp_f64 + v_add_f64"};
    for (int q = 0; q < 8; ++q) printf("%-60s %7.1f cycles per link\n", names[q], (double)h[q] / (16.0 * n));
    return 0;
}
