// ubench_f64_waves.hip -- v_mfma_f64_16x16x4_f64 throughput on ONE CU as the register build uses it: waves per SIMD x
// accumulator chains per wave.  Cycles (s_memtime) for the whole workgroup divided by the matrix instructions ONE SIMD issued.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_f64_waves tools/ubench_f64_waves.hip && tools/ubench_f64_waves
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int CHAINS, int WITH_LDS>
__global__ void k(double *out, unsigned long long *cyc, double seed, int n)
{
    __shared__ double lds[16 * 17 * 4];
    for (int e = threadIdx.x; e < 16 * 17 * 4; e += blockDim.x) lds[e] = seed + e * 1e-9;
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    double4_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double a = seed + lane * 1e-6, b = seed - lane * 1e-6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
        // one "tile": four matrix instructions on one accumulator (CHAINS = 1), or on two / four accumulators in turn
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            double x = a, y = b;
            if (WITH_LDS) { x = lds[c * 17 + 4 * s + g + ((it & 3) * 272)]; y = lds[c * 17 + 4 * s + g + (((it + 1) & 3) * 272)]; }
#pragma unroll
            for (int q = 0; q < CHAINS; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[q], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    double r = 0;
    for (int q = 0; q < 4; ++q) r += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int CHAINS, int WITH_LDS>
static void run(int threads, const char *what, double *out, unsigned long long *cyc)
{
    const int n = 2000;
    hipLaunchKernelGGL((k<CHAINS, WITH_LDS>), dim3(1), dim3(threads), 0, 0, out, cyc, 1.0, n);
    unsigned long long h = 0;
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double per_simd = (double)n * 4 * CHAINS * (threads / 64) / 4.0;      // matrix instructions one SIMD issued (waves spread over 4 SIMDs)
    printf("%-86s %6.1f memtime ticks per matrix instruction per SIMD\n", what, (double)h / (threads >= 256 ? per_simd : (double)n * 4 * CHAINS));
}

int main()
{
    double *out; unsigned long long *cyc;
    if (hipMalloc(&out, 8 * 1024) != hipSuccess || hipMalloc(&cyc, 8) != hipSuccess) { printf("no device\n"); return 1; }
    run<1, 0>(64, "1 wave on the CU, one accumulator chain", out, cyc);
    run<2, 0>(64, "1 wave on the CU, two accumulators in turn", out, cyc);
    run<1, 0>(256, "1 wave per SIMD, one chain each", out, cyc);
    run<1, 0>(512, "2 waves per SIMD, one chain each", out, cyc);
    run<2, 0>(512, "2 waves per SIMD, two accumulators in turn each", out, cyc);
    run<1, 1>(512, "2 waves per SIMD, one chain each, operands from LDS (8 ds_read_b64 per four instructions)", out, cyc);
    run<2, 1>(512, "2 waves per SIMD, two accumulators in turn, operands from LDS", out, cyc);
    run<1, 1>(256, "1 wave per SIMD, one chain, operands from LDS", out, cyc);
    return 0;
}
