// pk_hazard_test.hip -- directed experiments on gfx950 for the launch-to-launch differences that packed fp32 arithmetic
// gave under in-flight matrix instructions in the Gaussian shared-rig kernel.
// The static audit (tools/isa_hazard_scan.py) finds every register dependence of that kernel padded as LLVM's gfx940/gfx950
// tables ask, and the same write-after-read patterns on matrix operands in the kernels that are bitwise repeatable -- so
// what is left are pairs the tables do not list.  Each test issues such a pair back to back inside ONE asm block (fixed
// registers, no wait states between producer and consumer, a matrix instruction -- on whatever its registers hold -- in
// flight in front of it) and, in the same
// iteration and on the same inputs, the padded form (s_nop 7 between); outputs are compared bitwise.  A pair the hardware
// interlocks gives 0 differences; test 0 is the positive control: v_exp_f32 -> VALU use with no wait state, which the ISA
// guide lists as needing one.
//   hipcc --offload-arch=gfx950 -O2 -o tools/pk_hazard_test tools/pk_hazard_test.hip && tools/pk_hazard_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define MFMA_IN_FLIGHT "v_mfma_f32_16x16x32_f16 v[44:47], v[48:51], v[52:55], v[44:47]\n\t"
#define CLOBBERS "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57"

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int TEST>
__device__ __forceinline__ void pair(f32x2 a, f32x2 b, f32x2 c, float &tight, float &padded)
{
    if constexpr (TEST == 0) {
        // positive control: transcendental -> dependent VALU, zero wait states (the guide asks for one)
        asm volatile(MFMA_IN_FLIGHT "v_exp_f32 v42, %1\n\tv_add_f32 %0, v42, %2\n\ts_nop 7" : "=v"(tight) : "v"(a.x), "v"(b.x) : CLOBBERS);
        asm volatile(MFMA_IN_FLIGHT "v_exp_f32 v42, %1\n\ts_nop 7\n\tv_add_f32 %0, v42, %2\n\ts_nop 7" : "=v"(padded) : "v"(a.x), "v"(b.x) : CLOBBERS);
    } else if constexpr (TEST == 1) {
        // packed fma -> transcendental reads of BOTH halves of its result, zero wait states
        asm volatile(MFMA_IN_FLIGHT "v_pk_fma_f32 v[40:41], %1, %2, %3\n\tv_exp_f32 v42, v40\n\tv_exp_f32 v43, v41\n\ts_nop 7\n\tv_add_f32 %0, v42, v43\n\ts_nop 7"
                     : "=v"(tight) : "v"(a), "v"(b), "v"(c) : CLOBBERS);
        asm volatile(MFMA_IN_FLIGHT "v_pk_fma_f32 v[40:41], %1, %2, %3\n\ts_nop 7\n\tv_exp_f32 v42, v40\n\ts_nop 7\n\tv_exp_f32 v43, v41\n\ts_nop 7\n\tv_add_f32 %0, v42, v43\n\ts_nop 7"
                     : "=v"(padded) : "v"(a), "v"(b), "v"(c) : CLOBBERS);
    } else if constexpr (TEST == 2) {
        // packed fma -> plain VALU reads of both halves (conversion to fp16, as the split into pieces does)
        asm volatile(MFMA_IN_FLIGHT "v_pk_fma_f32 v[40:41], %1, %2, %3\n\tv_cvt_pk_f16_f32 v42, v40, v41\n\ts_nop 7\n\tv_mov_b32 %0, v42\n\ts_nop 7"
                     : "=v"(tight) : "v"(a), "v"(b), "v"(c) : CLOBBERS);
        asm volatile(MFMA_IN_FLIGHT "v_pk_fma_f32 v[40:41], %1, %2, %3\n\ts_nop 7\n\tv_cvt_pk_f16_f32 v42, v40, v41\n\ts_nop 7\n\tv_mov_b32 %0, v42\n\ts_nop 7"
                     : "=v"(padded) : "v"(a), "v"(b), "v"(c) : CLOBBERS);
    } else if constexpr (TEST == 3) {
        // transcendental pair -> packed consumer of the pair, zero wait states beyond the one the guide asks for
        asm volatile(MFMA_IN_FLIGHT "v_exp_f32 v40, %1\n\tv_exp_f32 v41, %2\n\ts_nop 0\n\tv_pk_add_f32 v[42:43], v[40:41], %3\n\ts_nop 7\n\tv_add_f32 %0, v42, v43\n\ts_nop 7"
                     : "=v"(tight) : "v"(a.x), "v"(a.y), "v"(b) : CLOBBERS);
        asm volatile(MFMA_IN_FLIGHT "v_exp_f32 v40, %1\n\tv_exp_f32 v41, %2\n\ts_nop 7\n\tv_pk_add_f32 v[42:43], v[40:41], %3\n\ts_nop 7\n\tv_add_f32 %0, v42, v43\n\ts_nop 7"
                     : "=v"(padded) : "v"(a.x), "v"(a.y), "v"(b) : CLOBBERS);
    } else if constexpr (TEST == 4) {
        // packed write of a pair whose halves are the SOURCES of transcendentals issued just before (write-after-read)
        asm volatile(MFMA_IN_FLIGHT "v_mov_b32 v40, %1\n\tv_mov_b32 v41, %2\n\ts_nop 7\n\tv_exp_f32 v42, v40\n\tv_exp_f32 v43, v41\n\tv_pk_mul_f32 v[40:41], %3, %3\n\ts_nop 7\n\tv_add_f32 %0, v42, v43\n\ts_nop 7"
                     : "=v"(tight) : "v"(a.x), "v"(a.y), "v"(b) : CLOBBERS);
        asm volatile(MFMA_IN_FLIGHT "v_mov_b32 v40, %1\n\tv_mov_b32 v41, %2\n\ts_nop 7\n\tv_exp_f32 v42, v40\n\tv_exp_f32 v43, v41\n\ts_nop 7\n\tv_pk_mul_f32 v[40:41], %3, %3\n\ts_nop 7\n\tv_add_f32 %0, v42, v43\n\ts_nop 7"
                     : "=v"(padded) : "v"(a.x), "v"(a.y), "v"(b) : CLOBBERS);
    } else if constexpr (TEST == 5) {
        // scalar write of ONE source of a transcendental issued just before (the write-after-read DESIGN 4.1c reports)
        asm volatile(MFMA_IN_FLIGHT "v_mov_b32 v40, %1\n\ts_nop 7\n\tv_exp_f32 v42, v40\n\tv_mul_f32 v40, %2, %2\n\ts_nop 7\n\tv_add_f32 %0, v42, v40\n\ts_nop 7"
                     : "=v"(tight) : "v"(a.x), "v"(b.x) : CLOBBERS);
        asm volatile(MFMA_IN_FLIGHT "v_mov_b32 v40, %1\n\ts_nop 7\n\tv_exp_f32 v42, v40\n\ts_nop 7\n\tv_mul_f32 v40, %2, %2\n\ts_nop 7\n\tv_add_f32 %0, v42, v40\n\ts_nop 7"
                     : "=v"(padded) : "v"(a.x), "v"(b.x) : CLOBBERS);
    } else if constexpr (TEST == 6) {
        // packed result -> matrix instruction operand with the two wait states LLVM gives a VALU producer
        asm volatile("v_mov_b32 v48, 0x3c003c00\n\tv_mov_b32 v49, 0x3c003c00\n\tv_mov_b32 v50, 0x3c003c00\n\tv_mov_b32 v51, 0x3c003c00\n\tv_pk_mul_f32 v[52:53], %1, %2\n\tv_pk_mul_f32 v[54:55], %2, %1\n\ts_nop 1\n\tv_mfma_f32_16x16x32_f16 v[44:47], v[48:51], v[52:55], 0\n\ts_nop 7\n\ts_nop 7\n\tv_add_f32 %0, v44, v45\n\ts_nop 7"
                     : "=v"(tight) : "v"(a), "v"(b) : CLOBBERS);
        asm volatile("v_mov_b32 v48, 0x3c003c00\n\tv_mov_b32 v49, 0x3c003c00\n\tv_mov_b32 v50, 0x3c003c00\n\tv_mov_b32 v51, 0x3c003c00\n\tv_pk_mul_f32 v[52:53], %1, %2\n\tv_pk_mul_f32 v[54:55], %2, %1\n\ts_nop 7\n\tv_mfma_f32_16x16x32_f16 v[44:47], v[48:51], v[52:55], 0\n\ts_nop 7\n\ts_nop 7\n\tv_add_f32 %0, v44, v45\n\ts_nop 7"
                     : "=v"(padded) : "v"(a), "v"(b) : CLOBBERS);
    } else if constexpr (TEST == 8) {
        // the shipped thin-plate split (split_pair_f16): v_cvt_pk_f16_f32, then v_fma_mixlo_f16 and v_fma_mixhi_f16 writing
        // the two halves of ONE register back to back, then a reader -- partial-register writes with zero wait states
        asm volatile(MFMA_IN_FLIGHT "v_cvt_pk_f16_f32 v40, %1, %2\n\tv_fma_mixlo_f16 v42, v40, -1.0, %1 op_sel_hi:[1,0,0]\n\t"
                     "v_fma_mixhi_f16 v42, v40, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_xor_b32 %0, v42, v40\n\ts_nop 7"
                     : "=v"(tight) : "v"(a.x), "v"(a.y) : CLOBBERS);
        asm volatile(MFMA_IN_FLIGHT "v_cvt_pk_f16_f32 v40, %1, %2\n\ts_nop 7\n\tv_fma_mixlo_f16 v42, v40, -1.0, %1 op_sel_hi:[1,0,0]\n\ts_nop 7\n\t"
                     "v_fma_mixhi_f16 v42, v40, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 7\n\tv_xor_b32 %0, v42, v40\n\ts_nop 7"
                     : "=v"(padded) : "v"(a.x), "v"(a.y) : CLOBBERS);
    } else if constexpr (TEST == 9) {
        // write-after-write: a transcendental's destination overwritten by the next instruction
        asm volatile(MFMA_IN_FLIGHT "v_exp_f32 v42, %1\n\tv_mov_b32 v42, %2\n\ts_nop 7\n\ts_nop 7\n\tv_mov_b32 %0, v42\n\ts_nop 7" : "=v"(tight) : "v"(a.x), "v"(b.x) : CLOBBERS);
        asm volatile(MFMA_IN_FLIGHT "v_exp_f32 v42, %1\n\ts_nop 7\n\tv_mov_b32 v42, %2\n\ts_nop 7\n\ts_nop 7\n\tv_mov_b32 %0, v42\n\ts_nop 7" : "=v"(padded) : "v"(a.x), "v"(b.x) : CLOBBERS);
    } else if constexpr (TEST == 10) {
        // matrix instruction result read by a packed instruction after the wait states LLVM's table gives a 4-pass result (7)
        asm volatile("v_mov_b32 v48, 0x3c003c00\n\tv_mov_b32 v49, 0x3c003c00\n\tv_mov_b32 v50, 0x3c003c00\n\tv_mov_b32 v51, 0x3c003c00\n\t"
                     "v_cvt_pk_f16_f32 v52, %1, %2\n\tv_mov_b32 v53, v52\n\tv_mov_b32 v54, v52\n\tv_mov_b32 v55, v52\n\ts_nop 7\n\t"
                     "v_mfma_f32_16x16x32_f16 v[44:47], v[48:51], v[52:55], 0\n\ts_nop 6\n\tv_pk_add_f32 v[40:41], v[44:45], v[46:47]\n\ts_nop 7\n\tv_add_f32 %0, v40, v41\n\ts_nop 7"
                     : "=v"(tight) : "v"(a.x), "v"(a.y) : CLOBBERS);
        asm volatile("v_mov_b32 v48, 0x3c003c00\n\tv_mov_b32 v49, 0x3c003c00\n\tv_mov_b32 v50, 0x3c003c00\n\tv_mov_b32 v51, 0x3c003c00\n\t"
                     "v_cvt_pk_f16_f32 v52, %1, %2\n\tv_mov_b32 v53, v52\n\tv_mov_b32 v54, v52\n\tv_mov_b32 v55, v52\n\ts_nop 7\n\t"
                     "v_mfma_f32_16x16x32_f16 v[44:47], v[48:51], v[52:55], 0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\tv_pk_add_f32 v[40:41], v[44:45], v[46:47]\n\ts_nop 7\n\tv_add_f32 %0, v40, v41\n\ts_nop 7"
                     : "=v"(padded) : "v"(a.x), "v"(a.y) : CLOBBERS);
    } else {
        // packed write of the B operand of a matrix instruction issued just before (write-after-read on an operand)
        asm volatile("v_mov_b32 v48, 0x3c003c00\n\tv_mov_b32 v49, 0x3c003c00\n\tv_mov_b32 v50, 0x3c003c00\n\tv_mov_b32 v51, 0x3c003c00\n\tv_mov_b32 v52, %1\n\tv_mov_b32 v53, %2\n\tv_mov_b32 v54, %1\n\tv_mov_b32 v55, %2\n\ts_nop 7\n\tv_mfma_f32_16x16x32_f16 v[44:47], v[48:51], v[52:55], 0\n\t"
                     "v_pk_mul_f32 v[52:53], %3, %3\n\tv_pk_mul_f32 v[54:55], %3, %3\n\ts_nop 7\n\ts_nop 7\n\tv_add_f32 %0, v44, v45\n\ts_nop 7"
                     : "=v"(tight) : "v"(a.x), "v"(a.y), "v"(b) : CLOBBERS);
        asm volatile("v_mov_b32 v48, 0x3c003c00\n\tv_mov_b32 v49, 0x3c003c00\n\tv_mov_b32 v50, 0x3c003c00\n\tv_mov_b32 v51, 0x3c003c00\n\tv_mov_b32 v52, %1\n\tv_mov_b32 v53, %2\n\tv_mov_b32 v54, %1\n\tv_mov_b32 v55, %2\n\ts_nop 7\n\tv_mfma_f32_16x16x32_f16 v[44:47], v[48:51], v[52:55], 0\n\ts_nop 7\n\ts_nop 7\n\t"
                     "v_pk_mul_f32 v[52:53], %3, %3\n\tv_pk_mul_f32 v[54:55], %3, %3\n\ts_nop 7\n\ts_nop 7\n\tv_add_f32 %0, v44, v45\n\ts_nop 7"
                     : "=v"(padded) : "v"(a.x), "v"(a.y), "v"(b) : CLOBBERS);
    }
}

template <int TEST>
__global__ __launch_bounds__(512) void k_test(unsigned long long *diffs, int iters)
{
    unsigned s = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
    unsigned long long n = 0;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const float u0 = (float)(s >> 8) * (1.f / 16777216.f);
        s = s * 1664525u + 1013904223u;
        const float u1 = (float)(s >> 8) * (1.f / 16777216.f);
        const f32x2 a = {u0 - 0.5f, u1 - 0.5f}, b = {u1 + 0.25f, u0 + 0.75f}, c = {u0 * u1, u1 - u0};
        float t, p;
        pair<TEST>(a, b, c, t, p);
        n += __float_as_uint(t) != __float_as_uint(p);
    }
    if (n) atomicAdd(diffs, n);
}

template <int TEST>
static void run(const char *what, unsigned long long *d_diffs, int iters)
{
    hipMemset(d_diffs, 0, 8);
    hipLaunchKernelGGL(k_test<TEST>, dim3(1024), dim3(512), 0, 0, d_diffs, iters);      // 2 waves per SIMD on every CU
    unsigned long long h = 0;
    hipMemcpy(&h, d_diffs, 8, hipMemcpyDeviceToHost);
    printf("test %d  %-98s %12llu differences in %.2e pairs\n", TEST, what, h, 1024.0 * 512.0 * iters);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    unsigned long long *d = nullptr;
    if (hipMalloc(&d, 8) != hipSuccess) { printf("no device\n"); return 1; }
    run<0>("CONTROL  v_exp_f32 -> dependent v_add_f32, 0 wait states (guide: 1)", d, iters);
    run<1>("v_pk_fma_f32 -> v_exp_f32 of both halves, 0 wait states", d, iters);
    run<2>("v_pk_fma_f32 -> v_cvt_pk_f16_f32 of both halves, 0 wait states", d, iters);
    run<3>("v_exp_f32 x 2 -> v_pk_add_f32 of the pair, 1 wait state", d, iters);
    run<4>("v_exp_f32 x 2 then v_pk_mul_f32 OVERWRITING their sources, 0 wait states (write-after-read)", d, iters);
    run<5>("v_exp_f32 then v_mul_f32 OVERWRITING its source, 0 wait states (write-after-read)", d, iters);
    run<6>("v_pk_mul_f32 x 2 -> matrix instruction B operand, 2 wait states (LLVM's VALU rule)", d, iters);
    run<7>("matrix instruction then v_pk_mul_f32 OVERWRITING its B operand, 0 wait states", d, iters);
    run<8>("v_cvt_pk_f16_f32 -> v_fma_mixlo_f16 -> v_fma_mixhi_f16 (same register) -> reader, 0 wait states (shipped split)", d, iters);
    run<9>("v_exp_f32 then v_mov_b32 to the SAME destination, 0 wait states (write-after-write)", d, iters);
    run<10>("matrix instruction (8 passes) -> v_pk_add_f32 of its result after 7 wait states", d, iters);
    hipFree(d);
    return 0;
}
