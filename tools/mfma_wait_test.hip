// mfma_wait_test.hip -- how many wait states a VALU read of a matrix instruction's result needs on gfx950, measured: the
// consumer is issued W wait states (W x "s_nop 0") behind the matrix instruction and compared bitwise with the same pair
// 40 wait states apart.  hipcc (ROCm 7.2) pads 8 wait states behind v_mfma_f32_16x16x32_f16 and v_mfma_f32_16x16x16_f16 and
// 12 behind v_mfma_f32_32x32x16_f16 (tools: compile a two-line kernel with -S); this finds where the differences stop.
//   hipcc --offload-arch=gfx950 -O2 -o tools/mfma_wait_test tools/mfma_wait_test.hip && tools/mfma_wait_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define STR2(x) #x
#define STR(x) STR2(x)
#define CLOB "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", \
             "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75"
#define SETUP "v_mov_b32 v48, 0x3c003c00\n\tv_mov_b32 v49, 0x3c003c00\n\tv_mov_b32 v50, 0x3c003c00\n\tv_mov_b32 v51, 0x3c003c00\n\t" \
              "v_cvt_pk_f16_f32 v52, %1, %2\n\tv_mov_b32 v53, v52\n\tv_mov_b32 v54, v52\n\tv_mov_b32 v55, v52\n\t.rept 12\n\ts_nop 0\n\t.endr\n\t"
#define NOPS_W ".rept %3\n\ts_nop 0\n\t.endr\n\t"

// KIND 0: 16x16x32 f16 -> v_add_f32; 1: 16x16x16 f16 -> v_add_f32; 2: 32x32x16 f16 -> v_add_f32; 3: 16x16x32 f16 -> v_pk_add_f32
template <int KIND, int W>
__device__ __forceinline__ void pair(float x, float y, float &tight, float &padded)
{
    if constexpr (KIND == 0) {
        asm volatile(SETUP "v_mfma_f32_16x16x32_f16 v[44:47], v[48:51], v[52:55], 0\n\t" NOPS_W "v_add_f32 %0, v44, v47\n\t.rept 40\n\ts_nop 0\n\t.endr" : "=v"(tight) : "v"(x), "v"(y), "n"(W) : CLOB);
        asm volatile(SETUP "v_mfma_f32_16x16x32_f16 v[44:47], v[48:51], v[52:55], 0\n\t" NOPS_W "v_add_f32 %0, v44, v47\n\t.rept 40\n\ts_nop 0\n\t.endr" : "=v"(padded) : "v"(x), "v"(y), "n"(40) : CLOB);
    } else if constexpr (KIND == 1) {
        asm volatile(SETUP "v_mfma_f32_16x16x16_f16 v[44:47], v[48:49], v[52:53], 0\n\t" NOPS_W "v_add_f32 %0, v44, v47\n\t.rept 40\n\ts_nop 0\n\t.endr" : "=v"(tight) : "v"(x), "v"(y), "n"(W) : CLOB);
        asm volatile(SETUP "v_mfma_f32_16x16x16_f16 v[44:47], v[48:49], v[52:53], 0\n\t" NOPS_W "v_add_f32 %0, v44, v47\n\t.rept 40\n\ts_nop 0\n\t.endr" : "=v"(padded) : "v"(x), "v"(y), "n"(40) : CLOB);
    } else if constexpr (KIND == 2) {
        asm volatile(SETUP "v_mfma_f32_32x32x16_f16 v[60:75], v[48:51], v[52:55], 0\n\t" NOPS_W "v_add_f32 %0, v60, v75\n\t.rept 40\n\ts_nop 0\n\t.endr" : "=v"(tight) : "v"(x), "v"(y), "n"(W) : CLOB);
        asm volatile(SETUP "v_mfma_f32_32x32x16_f16 v[60:75], v[48:51], v[52:55], 0\n\t" NOPS_W "v_add_f32 %0, v60, v75\n\t.rept 40\n\ts_nop 0\n\t.endr" : "=v"(padded) : "v"(x), "v"(y), "n"(40) : CLOB);
    } else if constexpr (KIND == 4 || KIND == 5) {
        // the probe behind a burst of independent matrix instructions (the matrix pipe busy, both waves of the SIMD at it):
        // KIND 4: six 16x16x32 in front of a 16x16x32 probe; KIND 5: three 32x32x16 in front of a 16x16x32 probe
#define BURST4 "v_mfma_f32_16x16x32_f16 v[60:63], v[48:51], v[52:55], 0\n\tv_mfma_f32_16x16x32_f16 v[64:67], v[48:51], v[52:55], 0\n\t" \
               "v_mfma_f32_16x16x32_f16 v[68:71], v[48:51], v[52:55], 0\n\tv_mfma_f32_16x16x32_f16 v[72:75], v[48:51], v[52:55], 0\n\t" \
               "v_mfma_f32_16x16x32_f16 v[60:63], v[48:51], v[52:55], v[60:63]\n\tv_mfma_f32_16x16x32_f16 v[64:67], v[48:51], v[52:55], v[64:67]\n\t"
#define BURST5 "v_mfma_f32_32x32x16_f16 v[60:75], v[48:51], v[52:55], 0\n\tv_mfma_f32_32x32x16_f16 v[60:75], v[48:51], v[52:55], v[60:75]\n\t" \
               "v_mfma_f32_32x32x16_f16 v[60:75], v[48:51], v[52:55], v[60:75]\n\t"
        if constexpr (KIND == 4) {
            asm volatile(SETUP BURST4 "v_mfma_f32_16x16x32_f16 v[44:47], v[48:51], v[52:55], 0\n\t" NOPS_W "v_add_f32 %0, v44, v47\n\t.rept 40\n\ts_nop 0\n\t.endr" : "=v"(tight) : "v"(x), "v"(y), "n"(W) : CLOB);
            asm volatile(SETUP BURST4 "v_mfma_f32_16x16x32_f16 v[44:47], v[48:51], v[52:55], 0\n\t" NOPS_W "v_add_f32 %0, v44, v47\n\t.rept 40\n\ts_nop 0\n\t.endr" : "=v"(padded) : "v"(x), "v"(y), "n"(60) : CLOB);
        } else {
            asm volatile(SETUP BURST5 "v_mfma_f32_16x16x32_f16 v[44:47], v[48:51], v[52:55], 0\n\t" NOPS_W "v_add_f32 %0, v44, v47\n\t.rept 40\n\ts_nop 0\n\t.endr" : "=v"(tight) : "v"(x), "v"(y), "n"(W) : CLOB);
            asm volatile(SETUP BURST5 "v_mfma_f32_16x16x32_f16 v[44:47], v[48:51], v[52:55], 0\n\t" NOPS_W "v_add_f32 %0, v44, v47\n\t.rept 40\n\ts_nop 0\n\t.endr" : "=v"(padded) : "v"(x), "v"(y), "n"(60) : CLOB);
        }
    } else {
        asm volatile(SETUP "v_mfma_f32_16x16x32_f16 v[44:47], v[48:51], v[52:55], 0\n\t" NOPS_W "v_pk_add_f32 v[40:41], v[44:45], v[46:47]\n\t.rept 40\n\ts_nop 0\n\t.endr\n\tv_add_f32 %0, v40, v41" : "=v"(tight) : "v"(x), "v"(y), "n"(W) : CLOB);
        asm volatile(SETUP "v_mfma_f32_16x16x32_f16 v[44:47], v[48:51], v[52:55], 0\n\t" NOPS_W "v_pk_add_f32 v[40:41], v[44:45], v[46:47]\n\t.rept 40\n\ts_nop 0\n\t.endr\n\tv_add_f32 %0, v40, v41" : "=v"(padded) : "v"(x), "v"(y), "n"(40) : CLOB);
    }
}

template <int KIND, int W>
__global__ __launch_bounds__(512) void k_test(unsigned long long *diffs, int iters)
{
    unsigned s = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
    unsigned long long n = 0;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const float u0 = (float)(s >> 8) * (1.f / 16777216.f);
        s = s * 1664525u + 1013904223u;
        const float u1 = (float)(s >> 8) * (1.f / 16777216.f);
        float t, p;
        pair<KIND, W>(u0 + 0.5f, u1 - 0.5f, t, p);
        n += __float_as_uint(t) != __float_as_uint(p);
    }
    if (n) atomicAdd(diffs, n);
}

template <int KIND, int W>
static void run(unsigned long long *d, int iters)
{
    (void)hipMemset(d, 0, 8);
    hipLaunchKernelGGL((k_test<KIND, W>), dim3(1024), dim3(512), 0, 0, d, iters);
    unsigned long long h = 0;
    (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    static const char *names[] = {"v_mfma_f32_16x16x32_f16 -> v_add_f32   ", "v_mfma_f32_16x16x16_f16 -> v_add_f32   ", "v_mfma_f32_32x32x16_f16 -> v_add_f32   ",
                                  "v_mfma_f32_16x16x32_f16 -> v_pk_add_f32", "6 x 16x16x32 then 16x16x32 -> v_add_f32", "3 x 32x32x16 then 16x16x32 -> v_add_f32"};
    printf("%s  %2d wait states: %12llu differences in %.2e pairs\n", names[KIND], W, h, 1024.0 * 512.0 * iters);
    fflush(stdout);
}

template <int KIND, int... Ws>
static void sweep(unsigned long long *d, int iters) { (run<KIND, Ws>(d, iters), ...); }

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    unsigned long long *d = nullptr;
    if (hipMalloc(&d, 8) != hipSuccess) { printf("no device\n"); return 1; }
    sweep<1, 5, 6, 7, 8>(d, iters);
    sweep<0, 6, 7, 8, 9, 10>(d, iters);
    sweep<3, 6, 7, 8, 9>(d, iters);
    sweep<2, 10, 11, 12, 13, 14>(d, iters);
    sweep<4, 6, 7, 8, 9, 10, 12, 16>(d, iters);
    sweep<5, 6, 7, 8, 9, 10, 12, 16>(d, iters);
    (void)hipFree(d);
    return 0;
}
