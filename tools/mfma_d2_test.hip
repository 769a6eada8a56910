// mfma_d2_test.hip -- experiment: squared distances of 16 vertices x 16 centres from ONE
// v_mfma_f32_16x16x32_bf16, to fp32 accuracy, by splitting every fp32 coordinate exactly
// into three bf16 pieces (hi + mid + lo, truncation splits are exact).
//
//   d2[i][j] = |x_j|^2 (C-in)  +  sum_comp  c_i * (-2 x_j)  +  |c_i|^2 * 1
//   per component 6 of the 9 cross products are kept (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo,
//   lo*hi; the dropped ones are < 2^-24 relative), |c|^2 takes 3 slots: 21 of K = 32.
//
// Reports (1) max abs / rel error of d2 against fp64, (2) whether the bf16 MFMA runs in the
// shadow of the VALU work of the evaluation loop (4 log + 2 pk_mul + 6 pk_fma per tile).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_d2_test.hip -o tools/mfma_d2_test
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, unsigned short &hi, unsigned short &mid, unsigned short &lo)
{
    const unsigned u = __float_as_uint(x);
    const float fh = __uint_as_float(u & 0xffff0000u);
    const float r1 = x - fh;                                   // exact
    const unsigned u1 = __float_as_uint(r1);
    const float fm = __uint_as_float(u1 & 0xffff0000u);
    const float r2 = r1 - fm;                                  // exact, <= 8 significant bits left
    hi = (unsigned short)(u >> 16);
    mid = (unsigned short)(u1 >> 16);
    lo = (unsigned short)(__float_as_uint(r2) >> 16);
}

// operand of lane (g = lane>>4, i = lane&15): 8 bf16 for k = 8g..8g+7
__device__ __forceinline__ bf16x8 centre_operand(const float *c /*16 x 3*/, int lane)
{
    const int g = lane >> 4, i = lane & 15;
    bf16x8 a = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned short h, m, l;
    if (g < 3) {
        split3(c[3 * i + g], h, m, l);
        a[0] = h; a[1] = h; a[2] = m; a[3] = m; a[4] = h; a[5] = l;
    } else {
        const float cc = fmaf(c[3 * i + 2], c[3 * i + 2], fmaf(c[3 * i + 1], c[3 * i + 1], c[3 * i] * c[3 * i]));
        split3(cc, h, m, l);
        a[0] = h; a[1] = m; a[2] = l;
    }
    return a;
}
__device__ __forceinline__ bf16x8 vertex_operand(const float *x /*16 x 3*/, int lane)
{
    const int g = lane >> 4, j = lane & 15;
    bf16x8 b = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned short h, m, l;
    if (g < 3) {
        split3(-2.f * x[3 * j + g], h, m, l);
        b[0] = h; b[1] = m; b[2] = h; b[3] = m; b[4] = l; b[5] = h;
    } else {
        b[0] = 0x3f80; b[1] = 0x3f80; b[2] = 0x3f80;          // bf16 1.0
    }
    return b;
}

// one wave = one vertex tile; writes d2[(tile*16 + j) * M + centre]
__global__ void k_d2(const float *X, const float *C, int M, float *out)
{
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const float *x = X + 48 * tile;
    const bf16x8 b = vertex_operand(x, lane);
    const int j = lane & 15;
    const float xx = fmaf(x[3 * j + 2], x[3 * j + 2], fmaf(x[3 * j + 1], x[3 * j + 1], x[3 * j] * x[3 * j]));
    const f32x4 cin = {xx, xx, xx, xx};
    for (int ct = 0; ct < M / 16; ++ct) {
        const bf16x8 a = centre_operand(C + 48 * ct, lane);
        const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, cin, 0, 0, 0);
        // D: col = lane & 15 (vertex j), row = 4 * (lane >> 4) + r (centre)
        for (int r = 0; r < 4; ++r)
            out[(size_t)(tile * 16 + j) * M + ct * 16 + 4 * (lane >> 4) + r] = d[r];
    }
}

constexpr int ITERS = 2048;
template <bool WITH_MFMA, bool WITH_VALU>
__global__ void k_overlap(float *out, float a, float b)
{
    f32x4 c0 = {1, 2, 3, 4}, d0 = c0, d1 = c0;
    bf16x8 va = {0x3f80, 0x3f00, 0x3e80, 0x3f80, 0, 0, 0, 0}, vb = va;
    va[4] = (short)threadIdx.x;
    f32x2 acc0 = {0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0, acc4 = acc0, acc5 = acc0;
    f32x2 w0 = {a, b}, w1 = {b, a}, w2 = {a, a}, w3 = {b, b}, w4 = {a, 1}, w5 = {b, 1};
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (WITH_MFMA)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %4\n" : "=v"(d0) : "0"(d0), "v"(va), "v"(vb), "v"(c0));
            if (WITH_VALU) {
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

typedef float f32x16 __attribute__((ext_vector_type(16)));
// incremental cost of ONE matrix instruction inside a VALU-dense stream:
// per iteration 48 independent v_pk_fma_f32 plus KIND: 0 nothing, 1 one 16x16x32 bf16, 2 one 32x32x16 bf16,
// 3 two 16x16x32, 4 one 16x16x32 placed mid-stream
template <int KIND>
__global__ void k_incr(float *out, float a, float b)
{
    f32x2 r0 = {(float)threadIdx.x, 1.f}, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f, r4 = r0 + 4.f, r5 = r0 + 5.f, r6 = r0 + 6.f, r7 = r0 + 7.f;
    f32x2 va = {a, a}, vb = {b, b};
    bf16x8 ma = {0x3f80, 0x3f00, 0x3e80, 0x3f80, 0, 0, 0, 0}, mb = ma;
    f32x4 d4 = {0, 0, 0, 0}, d4b = d4;
    f32x16 d16 = {0};
    for (int i = 0; i < ITERS; ++i) {
#define PK8 asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n" \
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n" \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(va), "v"(vb));
        if (KIND == 1 || KIND == 3) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n" : "+v"(d4) : "v"(ma), "v"(mb));
        if (KIND == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n" : "+v"(d16) : "v"(ma), "v"(mb));
        PK8 PK8 PK8
        if (KIND == 3 || KIND == 4) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n" : "+v"(d4b) : "v"(ma), "v"(mb));
        PK8 PK8 PK8
#undef PK8
    }
    f32x2 s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + d4[0] + d4b[1] + d16[3];
}

int main()
{
    // ---- accuracy ----
    const int N = 4096, M = 256;
    std::vector<float> X(3 * N), C(3 * M);
    srand(7);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto &v : X) v = rnd();
    for (auto &v : C) v = rnd();
    for (int i = 0; i < 64; ++i)            // some vertices sit on / next to centres
        for (int k = 0; k < 3; ++k) X[3 * i + k] = C[3 * (i % M) + k] + (i < 32 ? 0.f : 1e-3f * rnd());
    float *dX, *dC, *dO;
    CHECK(hipMalloc(&dX, X.size() * 4)); CHECK(hipMalloc(&dC, C.size() * 4)); CHECK(hipMalloc(&dO, (size_t)N * M * 4));
    CHECK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_d2, dim3(N / 16 / 4), dim3(256), 0, 0, dX, dC, M, dO);
    CHECK(hipDeviceSynchronize());
    std::vector<float> O((size_t)N * M);
    CHECK(hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost));
    double max_abs = 0, max_abs_direct = 0, max_rel_far = 0; int neg = 0;
    for (int v = 0; v < N; ++v)
        for (int c = 0; c < M; ++c) {
            double ref = 0; float dir = 0;
            for (int k = 0; k < 3; ++k) {
                const double d = (double)X[3 * v + k] - (double)C[3 * c + k];
                ref += d * d;
                const float df = X[3 * v + k] - C[3 * c + k];
                dir = fmaf(df, df, dir);
            }
            const double e = fabs((double)O[(size_t)v * M + c] - ref);
            max_abs = fmax(max_abs, e);
            max_abs_direct = fmax(max_abs_direct, fabs((double)dir - ref));
            if (ref > 0.25) max_rel_far = fmax(max_rel_far, e / ref);
            if (O[(size_t)v * M + c] < 0) ++neg;
        }
    printf("d2 via bf16x3 MFMA: max abs err %.3e (direct fp32: %.3e), max rel err for d2 > 0.25: %.3e, negatives: %d\n",
           max_abs, max_abs_direct, max_rel_far, neg);

    // ---- overlap ----
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    float *out; CHECK(hipMalloc(&out, sizeof(float) * 256 * ncu * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct { const char *name; void (*fn)(float *, float, float); } ks[] = {
        {"bf16 mfma 16x16x32 alone", k_overlap<true, false>},
        {"valu tile work alone    ", k_overlap<false, true>},
        {"bf16 mfma + valu tile   ", k_overlap<true, true>},
    };
    for (auto &k : ks)
        for (int wps = 1; wps <= 8; wps *= 2) {
            const int blocks = ncu * wps;
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            const double tiles = (double)blocks * 4 * ITERS * 2;     // per wave: 2 tiles per iteration
            printf("%s waves/SIMD %d: %8.3f ms  -> %6.1f units (1/2.4 ns) per tile per SIMD\n", k.name, wps, best,
                   best * 1e6 * 2.4 / (tiles / (ncu * 4)));
        }
    struct { const char *name; void (*fn)(float *, float, float); } ki[] = {
        {"48 pk_fma                      ", k_incr<0>}, {"48 pk_fma + 1 mfma16x16x32    ", k_incr<1>},
        {"48 pk_fma + 1 mfma32x32x16    ", k_incr<2>}, {"48 pk_fma + 2 mfma16x16x32    ", k_incr<3>},
        {"48 pk_fma + 1 mfma16x16x32 mid", k_incr<4>},
    };
    for (auto &k : ki)
        for (int wps = 2; wps <= 8; wps *= 2) {
            const int blocks = ncu * wps;
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            printf("%s waves/SIMD %d: %8.3f ms -> %6.1f units per iteration per SIMD-wave-slot\n", k.name, wps, best,
                   best * 1e6 * 2.4 / ((double)wps * ITERS));
        }
    return 0;
}
