// ubench_mfma16.hip -- what the fp16 matrix pipe of gfx950 sustains in the shapes the shared-rig evaluation
// kernel uses: v_mfma_f32_16x16x32_f16 alone (independent accumulators, all operands in VGPRs), and with
// vector work, logarithms and LDS operand reads interleaved between the matrix instructions.
// Build:  hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma16.hip -o tools/ubench_mfma16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int ITERS = 1024;
constexpr int NACC = 16;

// per iteration: NACC matrix instructions on NACC different accumulators, each followed by NV v_fma_f32,
// NL v_log_f32 and ND ds_read_b128 that nothing waits for until the end of the iteration
template <int NV, int NL, int ND, bool AGPR = false>
__global__ __launch_bounds__(256) void k_mix(float *out, float a, float b)
{
    __shared__ uint4 lds[1024];
    for (int q = threadIdx.x; q < 1024; q += 256) lds[q] = make_uint4(q, q + 1, q + 2, q + 3);
    __syncthreads();
    f32x4 acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 A = {0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, B = {0x38003800u, 0x38003800u, 0x38003800u, 0x38003800u};
    float v0 = a, v1 = b, v2 = a + 1, v3 = b + 1, l0 = a + 2, l1 = a + 3;
    u32x4 d = {0, 0, 0, 0};
    const unsigned addr = (threadIdx.x & 63) * 16;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
            if (AGPR) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[q]) : "v"(A), "v"(B));
            else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(A), "v"(B));
            if (NV >= 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(a), "v"(b));
            if (NV >= 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v1) : "v"(a), "v"(b));
            if (NV >= 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v2) : "v"(a), "v"(b));
            if (NV >= 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(a), "v"(b));
            if (NL >= 1 && (NL >= 2 || (q & 1) == 0)) asm volatile("v_log_f32 %0, %0" : "+v"(l0));
            if (NL >= 3) asm volatile("v_log_f32 %0, %0" : "+v"(l1));
            if (ND >= 1 && (q % (4 / ND)) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr));
        }
        if (ND >= 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int q = 1; q < NACC; ++q) s += acc[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + v0 + v1 + v2 + v3 + l0 + l1 + (float)d[0];
}

// the 32 x 32 x 16 form: twice the flops per instruction -- twice the pipe time, and the SAME issue-port time
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NACC32 = 8;
template <int NV, int NA = NACC32>
__global__ __launch_bounds__(256) void k_mix32(float *out, float a, float b)
{
    f32x16 acc[NACC32];
#pragma unroll
    for (int q = 0; q < NACC32; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    u32x4 A = {0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, B = {0x38003800u, 0x38003800u, 0x38003800u, 0x38003800u};
    float v0 = a, v1 = b, v2 = a + 1, v3 = b + 1, v4 = a + 2, v5 = b + 2, v6 = a + 3, v7 = b + 3;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int q = 0; q < NACC32; ++q) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[q % NA]) : "v"(A), "v"(B));
            if (NV >= 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(a), "v"(b));
            if (NV >= 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v1) : "v"(a), "v"(b));
            if (NV >= 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v2) : "v"(a), "v"(b));
            if (NV >= 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(a), "v"(b));
            if (NV >= 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v4) : "v"(a), "v"(b));
            if (NV >= 6) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v5) : "v"(a), "v"(b));
            if (NV >= 7) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v6) : "v"(a), "v"(b));
            if (NV >= 8) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v7) : "v"(a), "v"(b));
        }
    }
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
#pragma unroll
    for (int q = 0; q < NACC32; ++q) s += acc[q][0] + acc[q][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

struct Bench { const char *name; void (*fn)(float *, float, float); int nacc = NACC; double flop = 16384.0; double pipe = 16.0; };

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs; cycles are per matrix instruction and SIMD at 2.4 GHz (the pipe's own time is 16; 32 for the 32 x 32 x 16 form)\n", prop.gcnArchName, ncu);
    float *out;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * ncu * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const Bench benches[] = {
        {"mfma 16x16x32 f16 alone", k_mix<0, 0, 0>},
        {"+ 1 v_fma per mfma", k_mix<1, 0, 0>},
        {"+ 2 v_fma per mfma", k_mix<2, 0, 0>},
        {"+ 3 v_fma per mfma", k_mix<3, 0, 0>},
        {"+ 4 v_fma per mfma", k_mix<4, 0, 0>},
        {"+ 1 v_log per 2 mfma", k_mix<0, 1, 0>},
        {"+ 1 v_log per mfma", k_mix<0, 2, 0>},
        {"+ 1 v_log + 2 v_fma per mfma", k_mix<2, 2, 0>},
        {"+ 1 ds_read_b128 per 4 mfma", k_mix<0, 0, 1>},
        {"+ 1 ds_read_b128 per 2 mfma", k_mix<0, 0, 2>},
        {"+ 1 v_log + 2 v_fma + ds_read/4", k_mix<2, 2, 1>},
        {"32x32x16 f16 alone", k_mix32<0>, NACC32, 32768.0, 32.0},
        {"32x32x16, 2 accumulators in turn", k_mix32<0, 2>, NACC32, 32768.0, 32.0},
        {"32x32x16, 1 accumulator", k_mix32<0, 1>, NACC32, 32768.0, 32.0},
        {"32x32x16, 2 acc + 4 v_fma", k_mix32<4, 2>, NACC32, 32768.0, 32.0},
        {"32x32x16 + 4 v_fma per mfma", k_mix32<4>, NACC32, 32768.0, 32.0},
        {"32x32x16 + 6 v_fma per mfma", k_mix32<6>, NACC32, 32768.0, 32.0},
        {"32x32x16 + 8 v_fma per mfma", k_mix32<8>, NACC32, 32768.0, 32.0},
        {"acc in AGPRs: alone", k_mix<0, 0, 0, true>},
        {"acc in AGPRs: + 2 v_fma", k_mix<2, 0, 0, true>},
        {"acc in AGPRs: + 3 v_fma", k_mix<3, 0, 0, true>},
        {"acc in AGPRs: + 4 v_fma", k_mix<4, 0, 0, true>},
        {"acc in AGPRs: + 1 v_log + 2 v_fma", k_mix<2, 2, 0, true>},
    };
    for (const Bench &b : benches) {
        for (int wps = 1; wps <= 2; ++wps) {
            const int blocks = ncu * wps;
            hipLaunchKernelGGL(b.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(b.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            const double mf = (double)blocks * 4 * ITERS * b.nacc;            // wave-level matrix instructions
            const double per_simd_per_ns = mf / (best * 1e6) / (ncu * 4);
            printf("%-34s waves/SIMD %d  %8.3f ms  %6.2f cycles per mfma per SIMD  = %5.1f%% of the pipe, %7.1f TFLOP/s\n", b.name, wps, best,
                   2.4 / per_simd_per_ns, 100.0 * b.pipe / (2.4 / per_simd_per_ns), mf * b.flop / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
