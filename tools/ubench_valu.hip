// ubench_valu.hip -- instruction issue rates on gfx950 that decide the shape of the
// evaluation kernel: fp32 fma vs packed fma vs transcendentals vs MFMA, alone and mixed.
// Build:  hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
// Output: one line per (kernel, waves/SIMD): wave-instructions per ns chip-wide and the
//         implied cycles per wave-instruction per SIMD at the measured clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int ITERS = 2048;

#define REP8(x) x x x x x x x x

// each kernel: 8 independent accumulators, body = 32 instructions per iteration (or as noted)
__global__ void k_fma(float *out, float a, float b)
{
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
}

__global__ void k_fma_sgpr(float *out, float a, float b)
{
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            asm volatile("v_fmac_f32 %0, %8, %1\n v_fmac_f32 %1, %8, %2\n v_fmac_f32 %2, %8, %3\n v_fmac_f32 %3, %8, %4\n"
                         "v_fmac_f32 %4, %8, %5\n v_fmac_f32 %5, %8, %6\n v_fmac_f32 %6, %8, %7\n v_fmac_f32 %7, %8, %0\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "s"(a), "s"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
}

__global__ void k_pk_fma(float *out, float a, float b)
{
    f32x2 r0 = {(float)threadIdx.x, 1.f}, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f, r4 = r0 + 4.f, r5 = r0 + 5.f, r6 = r0 + 6.f, r7 = r0 + 7.f;
    f32x2 va = {a, a}, vb = {b, b};
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(va), "v"(vb));
    }
    f32x2 s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

#define TRANS_KERNEL(NAME, OP)                                                                        \
    __global__ void NAME(float *out, float a, float b)                                                \
    {                                                                                                 \
        float r0 = threadIdx.x + 1.5f, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
        for (int i = 0; i < ITERS; ++i) {                                                             \
            _Pragma("unroll") for (int u = 0; u < 4; ++u)                                             \
                asm volatile(OP " %0, %0\n " OP " %1, %1\n " OP " %2, %2\n " OP " %3, %3\n"           \
                             OP " %4, %4\n " OP " %5, %5\n " OP " %6, %6\n " OP " %7, %7\n"           \
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)); \
        }                                                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + a + b;   \
    }
TRANS_KERNEL(k_log, "v_log_f32")
TRANS_KERNEL(k_exp, "v_exp_f32")
TRANS_KERNEL(k_sqrt, "v_sqrt_f32")

// the evaluation loop's mix per pair: 10 full-rate VALU + 1 transcendental (x8 pairs = 88 instr)
__global__ void k_mix10_1(float *out, float a, float b)
{
    float r0 = threadIdx.x + 1.5f, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    float t0 = 1, t1 = 2, t2 = 3, t3 = 4, t4 = 5, t5 = 6, t6 = 7, t7 = 8;
    for (int i = 0; i < ITERS; ++i) {
#define PAIR(R, T)                                                                                   \
        asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n" \
                     "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_log_f32 %1, %0\n"        \
                     "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n" \
                     "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n"                           \
                     : "+v"(R), "+v"(T) : "v"(a), "v"(b));
        PAIR(r0, t0) PAIR(r1, t1) PAIR(r2, t2) PAIR(r3, t3) PAIR(r4, t4) PAIR(r5, t5) PAIR(r6, t6) PAIR(r7, t7)
#undef PAIR
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + t0 + t1 + t2 + t3 + t4 + t5 + t6 + t7;
}

// same arithmetic packed two pairs at a time: 5 pk + 2 trans + ... = per 2 pairs: 10 pk-VALU + 2 trans
__global__ void k_mixpk(float *out, float a, float b)
{
    f32x2 r0 = {(float)threadIdx.x + 1.5f, 2.5f}, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f;
    f32x2 t0 = {1, 2}, t1 = {2, 3}, t2 = {3, 4}, t3 = {4, 5};
    float l0 = 1.5f, l1 = 2.5f, l2 = 3.5f, l3 = 4.5f, m0 = 1.25f, m1 = 2.25f, m2 = 3.25f, m3 = 4.25f;
    f32x2 va = {a, a}, vb = {b, b};
    for (int i = 0; i < ITERS; ++i) {
#define PAIR2(R, T, LA, LB)                                                                                  \
        asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n" \
                     "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n"                    \
                     "v_log_f32 %2, %2\n v_log_f32 %3, %3\n"                                         \
                     "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %0, %1, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n" \
                     "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n"                    \
                     : "+v"(R), "+v"(T), "+v"(LA), "+v"(LB) : "v"(va), "v"(vb));
        PAIR2(r0, t0, l0, m0) PAIR2(r1, t1, l1, m1) PAIR2(r2, t2, l2, m2) PAIR2(r3, t3, l3, m3)
#undef PAIR2
    }
    f32x2 s = r0 + r1 + r2 + r3 + t0 + t1 + t2 + t3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + l0 + l1 + l2 + l3 + m0 + m1 + m2 + m3;
}

// 7 VALU + 1 trans + one 4x4x1 MFMA doing the 3-wide accumulate
__global__ void k_mix_mfma(float *out, float a, float b)
{
    float r0 = threadIdx.x + 1.5f, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3;
    float t0 = 1, t1 = 2, t2 = 3, t3 = 4;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < ITERS; ++i) {
#define PAIRM(R, T, C)                                                                               \
        asm volatile("v_fma_f32 %0, %0, %3, %4\n v_fma_f32 %0, %0, %3, %4\n v_fma_f32 %0, %0, %3, %4\n" \
                     "v_fma_f32 %0, %0, %3, %4\n v_fma_f32 %0, %0, %3, %4\n v_fma_f32 %0, %0, %3, %4\n" \
                     "v_log_f32 %1, %0\n v_fma_f32 %0, %1, %3, %4\n s_nop 1\n"                        \
                     "v_mfma_f32_4x4x1_16b_f32 %2, %3, %0, %2\n"                                      \
                     : "+v"(R), "+v"(T), "+v"(C) : "v"(a), "v"(b));
        PAIRM(r0, t0, c0) PAIRM(r1, t1, c1) PAIRM(r2, t2, c2) PAIRM(r3, t3, c3)
        PAIRM(r0, t0, c0) PAIRM(r1, t1, c1) PAIRM(r2, t2, c2) PAIRM(r3, t3, c3)
#undef PAIRM
    }
    f32x4 s = c0 + c1 + c2 + c3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + r0 + r1 + r2 + r3 + t0 + t1 + t2 + t3;
}

__global__ void k_mfma4(float *out, float a, float b)
{
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %8, %9, %0\n v_mfma_f32_4x4x1_16b_f32 %1, %8, %9, %1\n"
                         "v_mfma_f32_4x4x1_16b_f32 %2, %8, %9, %2\n v_mfma_f32_4x4x1_16b_f32 %3, %8, %9, %3\n"
                         "v_mfma_f32_4x4x1_16b_f32 %4, %8, %9, %4\n v_mfma_f32_4x4x1_16b_f32 %5, %8, %9, %5\n"
                         "v_mfma_f32_4x4x1_16b_f32 %6, %8, %9, %6\n v_mfma_f32_4x4x1_16b_f32 %7, %8, %9, %7\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
    }
    f32x4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ void k_fma64(float *out, float a, float b)
{
    double r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    double da = a, db = b;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                         "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(da), "v"(db));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7);
}

__global__ void k_mfma64(float *out, float a, float b)
{
    f64x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double da = a, db = b;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %4, %5, %0\n v_mfma_f64_16x16x4_f64 %1, %4, %5, %1\n"
                         "v_mfma_f64_16x16x4_f64 %2, %4, %5, %2\n v_mfma_f64_16x16x4_f64 %3, %4, %5, %3\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(da), "v"(db));
    }
    f64x4 s = c0 + c1 + c2 + c3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(s[0] + s[1] + s[2] + s[3]);
}

// ---- SGPR-vs-VGPR uniform operands -------------------------------------------------
__global__ void k_fma_s0(float *out, float a, float b)   // v_fma_f32 v, s, v, v : independent accumulators
{
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    float vb = b;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            asm volatile("v_fma_f32 %0, %8, %0, %9\n v_fma_f32 %1, %8, %1, %9\n v_fma_f32 %2, %8, %2, %9\n v_fma_f32 %3, %8, %3, %9\n"
                         "v_fma_f32 %4, %8, %4, %9\n v_fma_f32 %5, %8, %5, %9\n v_fma_f32 %6, %8, %6, %9\n v_fma_f32 %7, %8, %7, %9\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "s"(a), "v"(vb));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
}

__global__ void k_pk_fma_s(float *out, float a, float b)   // v_pk_fma_f32 v, v, s[pair] op_sel_hi, v
{
    f32x2 r0 = {(float)threadIdx.x, 1.f}, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f, r4 = r0 + 4.f, r5 = r0 + 5.f, r6 = r0 + 6.f, r7 = r0 + 7.f;
    f32x2 sa = {a, b};
    f32x2 vb = {b, b};
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %1, %8, %9 op_sel_hi:[1,0,1]\n"
                         "v_pk_fma_f32 %2, %2, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %3, %8, %9 op_sel_hi:[1,0,1]\n"
                         "v_pk_fma_f32 %4, %4, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %5, %5, %8, %9 op_sel_hi:[1,0,1]\n"
                         "v_pk_fma_f32 %6, %6, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %7, %7, %8, %9 op_sel_hi:[1,0,1]\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "s"(sa), "v"(vb));
    }
    f32x2 s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

// the evaluation kernel's exact per-centre sequence for two packed register slots (24 VALU),
// uniform operands in SGPRs (as the SCALAR variant issues them) ...
#define EVAL_SEQ(C01, C23, W01, W23, BIAS)                                                                   \
    asm volatile(                                                                                            \
        "v_pk_add_f32 %6, %0, " C01 " op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"                           \
        "v_pk_add_f32 %7, %3, " C01 " op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"                           \
        "v_pk_add_f32 %8, %1, " C01 " op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"                              \
        "v_pk_add_f32 %9, %4, " C01 " op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"                              \
        "v_pk_fma_f32 %6, %6, %6, " BIAS " op_sel_hi:[1,1,0]\n"                                              \
        "v_pk_fma_f32 %7, %7, %7, " BIAS " op_sel_hi:[1,1,0]\n"                                              \
        "v_pk_add_f32 %10, %2, " C23 " op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"                          \
        "v_pk_add_f32 %11, %5, " C23 " op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"                          \
        "v_pk_fma_f32 %6, %8, %8, %6\n v_pk_fma_f32 %7, %9, %9, %7\n"                                        \
        "v_pk_fma_f32 %6, %10, %10, %6\n v_pk_fma_f32 %7, %11, %11, %7\n"                                    \
        "v_log_f32 %12, %13\n v_log_f32 %13, %12\n v_log_f32 %14, %15\n v_log_f32 %15, %14\n"                \
        "v_pk_mul_f32 %8, %6, %8\n v_pk_mul_f32 %9, %7, %9\n"                                                \
        "v_pk_fma_f32 %16, %8, " W01 ", %16 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %17, %9, " W01 ", %17 op_sel_hi:[1,0,1]\n" \
        "v_pk_fma_f32 %18, %8, " W01 ", %18 op_sel:[0,1,0]\n v_pk_fma_f32 %19, %9, " W01 ", %19 op_sel:[0,1,0]\n"       \
        "v_pk_fma_f32 %20, %8, " W23 ", %20 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %21, %9, " W23 ", %21 op_sel_hi:[1,0,1]\n" \
        : "+v"(px0), "+v"(py0), "+v"(pz0), "+v"(px1), "+v"(py1), "+v"(pz1), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), \
          "+v"(t4), "+v"(t5), "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) \
        : "s"(c01), "s"(c23), "s"(w01), "s"(w23), "s"(bias), "v"(vc01), "v"(vc23), "v"(vw01), "v"(vw23), "v"(vbias))

#define EVAL_DECL                                                                                            \
    f32x2 px0 = {(float)threadIdx.x, 1.f}, py0 = px0 + 1.f, pz0 = px0 + 2.f, px1 = px0 + 3.f, py1 = px0 + 4.f, pz1 = px0 + 5.f; \
    f32x2 t0 = px0, t1 = px0, t2 = px0, t3 = px0, t4 = px0, t5 = px0;                                        \
    float l0 = 1.5f, l1 = 2.5f, l2 = 3.5f, l3 = 4.5f;                                                        \
    f32x2 a0 = {0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0;                                          \
    f32x2 c01 = {a, b}, c23 = {b, a}, w01 = {a, a}, w23 = {b, b}, bias = {1e-37f, 1e-37f};                   \
    f32x2 vc01 = c01, vc23 = c23, vw01 = w01, vw23 = w23, vbias = bias;

__global__ void k_evalseq_sgpr(float *out, float a, float b)
{
    EVAL_DECL
    for (int i = 0; i < ITERS; ++i) { EVAL_SEQ("%22", "%23", "%24", "%25", "%26"); EVAL_SEQ("%22", "%23", "%24", "%25", "%26"); }
    f32x2 s = a0 + a1 + a2 + a3 + a4 + a5 + t0 + t1;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + l0 + l1 + l2 + l3;
}
// ... and the same sequence with the uniform operands in VGPR pairs
__global__ void k_evalseq_vgpr(float *out, float a, float b)
{
    EVAL_DECL
    for (int i = 0; i < ITERS; ++i) { EVAL_SEQ("%27", "%28", "%29", "%30", "%31"); EVAL_SEQ("%27", "%28", "%29", "%30", "%31"); }
    f32x2 s = a0 + a1 + a2 + a3 + a4 + a5 + t0 + t1;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + l0 + l1 + l2 + l3;
}

// ---- does a 16x16x4 f32 MFMA run in the shadow of VALU work? ---------------------------
// per iteration: NM MFMAs (each produces 4 d2 per lane) + for each: 4 v_log + 2 v_pk_mul + 6 v_pk_fma
template <bool WITH_MFMA, bool WITH_VALU>
__global__ void k_mfma16_mix(float *out, float a, float b)
{
    f32x4 c0 = {1, 2, 3, 4}, d0 = c0, d1 = c0;
    f32x2 acc0 = {0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0, acc4 = acc0, acc5 = acc0;
    f32x2 w0 = {a, b}, w1 = {b, a}, w2 = {a, a}, w3 = {b, b}, w4 = {a, 1}, w5 = {b, 1};
    float va = a + threadIdx.x, vb = b;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (WITH_MFMA) {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %3, %4\n" : "=v"(d0) : "0"(d0), "v"(va), "v"(vb), "v"(c0));
            }
            if (WITH_VALU) {
                // VALU work on the OTHER buffer (software-pipelined: independent of the MFMA just issued)
                asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n"
                             : "+v"(d1[0]), "+v"(d1[1]), "+v"(d1[2]), "+v"(d1[3]));
                f32x2 t01 = {d1[0], d1[1]}, t23 = {d1[2], d1[3]};
                asm volatile("v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n"
                             "v_pk_fma_f32 %2, %0, %8, %2\n v_pk_fma_f32 %3, %0, %9, %3\n v_pk_fma_f32 %4, %0, %10, %4\n"
                             "v_pk_fma_f32 %5, %1, %11, %5\n v_pk_fma_f32 %6, %1, %12, %6\n v_pk_fma_f32 %7, %1, %13, %7\n"
                             : "+v"(t01), "+v"(t23), "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(acc4), "+v"(acc5)
                             : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "v"(w5));
            }
            f32x4 tmp = d0; d0 = d1; d1 = tmp;
        }
    }
    f32x2 s = acc0 + acc1 + acc2 + acc3 + acc4 + acc5;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + d0[0] + d1[1];
}

struct Bench { const char *name; void (*fn)(float *, float, float); int instr_per_iter; const char *note; };

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs, clockRate %d kHz\n", prop.gcnArchName, ncu, prop.clockRate);
    float *out;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * ncu * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const Bench benches[] = {
        {"v_fma_f32", k_fma, 32, "full-rate VALU"},
        {"v_fmac_f32 sgpr-src", k_fma_sgpr, 32, "SGPR operand, 1-deep dependency ring"},
        {"v_fma_f32 v,s,v,v", k_fma_s0, 32, "SGPR src0, independent accumulators"},
        {"v_pk_fma_f32 sgpr op_sel", k_pk_fma_s, 32, "SGPR pair broadcast by op_sel_hi"},
        {"eval sequence, SGPR uniforms", k_evalseq_sgpr, 48, "24 VALU per centre for 2 packed slots, x2"},
        {"eval sequence, VGPR uniforms", k_evalseq_vgpr, 48, "same, uniforms in VGPR pairs"},
        {"v_pk_fma_f32", k_pk_fma, 32, "2 fma per lane per instr"},
        {"v_log_f32", k_log, 32, "transcendental"},
        {"v_exp_f32", k_exp, 32, "transcendental"},
        {"v_sqrt_f32", k_sqrt, 32, "transcendental"},
        {"mix 10 fma + 1 log", k_mix10_1, 88, "the evaluation loop's mix, 8 pairs"},
        {"mix 10 pk_fma + 2 log", k_mixpk, 48, "same, packed: 8 pairs = 4 x (10 pk + 2 log)"},
        {"mix 7 fma + log + mfma4x4x1", k_mix_mfma, 72, "8 pairs, accumulate on the matrix pipe (9 issue slots/pair)"},
        {"v_mfma_f32_4x4x1", k_mfma4, 32, "512 flop per instr"},
        {"mfma16x16x4f32 alone", k_mfma16_mix<true, false>, 2, "2 MFMA per iteration"},
        {"valu tile work alone", k_mfma16_mix<false, true>, 24, "2 x (4 log + 2 pk_mul + 6 pk_fma)"},
        {"mfma16x16x4f32 + valu tile", k_mfma16_mix<true, true>, 26, "2 x (1 MFMA + 12 VALU): overlap?"},
        {"v_fma_f64", k_fma64, 32, ""},
        {"v_mfma_f64_16x16x4", k_mfma64, 32, "2048 flop per instr"},
    };
    for (const Bench &b : benches) {
        for (int wps = 1; wps <= 8; wps *= 2) {
            const int blocks = ncu * wps;   // 256 threads = 4 waves = 1 wave per SIMD per block
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
            const double winstr = (double)blocks * 4 * ITERS * b.instr_per_iter;   // wave-instructions
            const double per_simd_per_ns = winstr / (best * 1e6) / (ncu * 4);
            printf("%-30s waves/SIMD %d  %8.3f ms  %7.3f winstr/ns/SIMD  -> %6.2f cycles/winstr/SIMD @2.4GHz  (%s)\n",
                   b.name, wps, best, per_simd_per_ns, 2.4 / per_simd_per_ns, b.note);
        }
    }
    return 0;
}
