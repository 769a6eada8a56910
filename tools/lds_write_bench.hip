// lds_write_bench.hip -- cost of one lane (vs all lanes) storing a 256-byte row into LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double2_t __attribute__((ext_vector_type(2)));
__device__ unsigned long long g_out[64];
template <int MODE>
__global__ void k(double *sink, int sel_lane, int off_bytes)
{
    __shared__ __attribute__((aligned(16))) char smem[16384];
    const int lane = threadIdx.x & 63;
    double a[32];
    for (int c = 0; c < 32; ++c) a[c] = threadIdx.x + c;
    unsigned long long acc = 0;
    for (int it = 0; it < 64; ++it) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 0) {            // one lane, 16 x b128
            if (lane == sel_lane) {
                double2_t *dst = (double2_t *)(smem + off_bytes + (it & 1) * 4096);
#pragma unroll
                for (int c = 0; c < 16; ++c) dst[c] = (double2_t){a[2 * c], a[2 * c + 1]};
            }
        } else if (MODE == 1) {     // one lane, 32 x b64
            if (lane == sel_lane) {
                volatile double *dst = (volatile double *)(smem + off_bytes + (it & 1) * 4096);
#pragma unroll
                for (int c = 0; c < 32; ++c) dst[c] = a[c];
            }
        } else if (MODE == 2) {     // all lanes, 16 x b128 each (own 256-byte row, 64 rows)
            double2_t *dst = (double2_t *)(smem + lane * 256);
#pragma unroll
            for (int c = 0; c < 16; ++c) dst[c] = (double2_t){a[2 * c], a[2 * c + 1]};
        } else if (MODE == 3) {     // 32 lanes, one b64 each: the row spread over lanes
            if (lane < 32) ((double *)(smem + off_bytes + (it & 1) * 4096))[lane] = a[0];
        } else if (MODE == 4) {     // one lane, 64 x b32
            if (lane == sel_lane) {
                volatile float *dst = (volatile float *)(smem + off_bytes + (it & 1) * 4096);
#pragma unroll
                for (int c = 0; c < 32; ++c) { dst[2 * c] = (float)a[c]; dst[2 * c + 1] = (float)a[c]; }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        acc += t1 - t0;
        for (int c = 0; c < 32; ++c) a[c] += 1.0;
    }
    if (threadIdx.x == 0) g_out[MODE] = acc / 64;
    sink[threadIdx.x] = a[3] + ((double *)smem)[threadIdx.x & 7];
}
int main()
{
    double *sink; hipMalloc(&sink, 8 * 1024);
    for (int threads : {64, 320}) {
        hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, sink, 0, 0);
        hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, sink, 0, 0);
        hipLaunchKernelGGL(k<2>, dim3(1), dim3(threads), 0, 0, sink, 0, 0);
        hipLaunchKernelGGL(k<3>, dim3(1), dim3(threads), 0, 0, sink, 0, 0);
        hipLaunchKernelGGL(k<4>, dim3(1), dim3(threads), 0, 0, sink, 0, 0);
        hipDeviceSynchronize();
        unsigned long long o[64]; hipMemcpyFromSymbol(o, HIP_SYMBOL(g_out), sizeof(o));
        printf("threads %d: one lane 16xb128 %llu cyc | one lane 32xb64 %llu | all lanes 16xb128 %llu | 32 lanes 1xb64 %llu | one lane 64xb32 %llu (incl ~40 stamp)\n",
               threads, o[0], o[1], o[2], o[3], o[4]);
    }
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, sink, 17, 8);
    hipDeviceSynchronize();
    unsigned long long o[64]; hipMemcpyFromSymbol(o, HIP_SYMBOL(g_out), sizeof(o));
    printf("one lane 16xb128 at +8 bytes (misaligned): %llu cyc\n", o[0]);
    return 0;
}
