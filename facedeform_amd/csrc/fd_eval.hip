// fd_eval.hip -- per-vertex RBF evaluation with the reference's fused epilogue.
//
// Replaces the loop body of SOP_FaceDeform::cookMySop,
// reference src/SOP_FaceDeform.cpp:404-439 (gate, rbfcalc, project_to_tangents,
// fall-off, write-back) with one gfx950 kernel.
//
// Mapping to the hardware
//   * one lane owns V vertices (registers: position + 3 accumulators each); a
//     256-thread workgroup covers 256*V consecutive vertices, so every global
//     access of a wave is one contiguous 768 B (P) or 256 B (dist2/falloff) span;
//   * the control data is wave-uniform.  Variant SCALAR streams the 32 B centre
//     records through the scalar unit (s_load_dwordx8 -> SGPR operands: no VGPRs,
//     no LDS bandwidth, no VALU cost); variant LDS stages a tile of records in
//     LDS once per workgroup and broadcast-reads it (ds_read_b128, one address
//     for all 64 lanes);
//   * per (vertex, centre) pair: 3 sub, 3 fma (d2), one transcendental
//     (v_log_f32 / v_exp_f32 / v_sqrt_f32), 1 mul, 3 fma.  The kernel constant
//     (0.5*ln2 for thin-plate, ...) is folded into the weights by the pack kernel;
//   * fp32 partial sums are folded into fp64 accumulators every 64 centres, which
//     bounds the accumulation error independently of M.
//   * FP64 variant: same structure, all arithmetic in fp64.
// Built with -ffp-contract=off: every fused multiply-add is written out, so a vertex
// gets the same bits whichever lane / register slot it lands in (range splits are
// bit-identical) and the fp32 epilogue rounds like the reference's unfused CPU code.
#include <cstdio>
#include <cstdlib>

#include <type_traits>

#include "fd_internal.h"

namespace fd {

namespace {

constexpr int kBlock = 256;
constexpr int kChunk = 64;  // centres per fp32 partial sum
constexpr unsigned kNumCU = 256;      // MI355X
constexpr int kDefaultVariant = 102;  // packed lanes, scalar-loaded records, V = 4

typedef const __attribute__((address_space(4))) Rec32 *ConstRec32;
typedef const __attribute__((address_space(4))) Rec64 *ConstRec64;

struct EvalParams {
    int64_t N;
    const float *P_in;
    float *P_out;
    const float *dist2;
    float *falloff_out;
    const float *tu, *tv, *nrm;
    float radius2, falloffrate;
    int Mpad;
    const Rec32 *rec32;
    const Rec64 *rec64;
    const MfmaTile *tiles;
    const MfmaTileH *tiles16;
    const DevModel *model;
};

// per-frame parameters of a batched launch (blockIdx.y picks the frame); travels as a kernel argument
struct EvalBatch {
    EvalParams p[kMaxBatch];
};
static_assert(sizeof(EvalBatch) <= 4000, "the table must fit the kernel argument segment");

// ---- kernels phi'(d2) (constant factors live in the packed weights) ----------
template <int KIND>
__device__ __forceinline__ float phi32(float d2, float s)
{
    if constexpr (KIND == FD_KERNEL_THIN_PLATE) {
        return d2 * __builtin_amdgcn_logf(d2);        // d2*log2(d2); d2 carries +1e-37
    } else if constexpr (KIND == FD_KERNEL_GAUSSIAN || KIND == FD_KERNEL_GAUSSIAN_QNN) {
        return __builtin_amdgcn_exp2f(d2 * s);        // s = -log2(e)/R_j^2
    } else if constexpr (KIND == FD_KERNEL_BIHARMONIC) {
        return __builtin_amdgcn_sqrtf(d2);
    } else {
        return d2 * __builtin_amdgcn_sqrtf(d2);
    }
}

// Natural logarithm of a positive, normal double to ~1 ulp-and-a-half without the library's
// table walk: x = m * 2^e with m in [1/sqrt2, sqrt2), ln m = 2 atanh(s), s = (m - 1) / (m + 1),
// |s| <= 0.1716, eleven odd terms (the next one is below 1e-17).  About half the instructions of
// ocml's log; the fp64 evaluation spends most of its time here.
__device__ __forceinline__ double fast_log_pos(double x)
{
    double m = __builtin_amdgcn_frexp_mant(x);               // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    if (m < 0.70710678118654752) { m *= 2.0; e -= 1; }
    const double num = m - 1.0, den = m + 1.0;
    // division by Newton on the hardware reciprocal (den in [1.7, 2.42))
    double r = __builtin_amdgcn_rcp(den);
    r = fma(fma(-den, r, 1.0), r, r);
    r = fma(fma(-den, r, 1.0), r, r);
    double s0 = num * r;
    s0 = fma(fma(-den, s0, num), r, s0);                     // one correction of the quotient
    const double z = s0 * s0;
    double p = 1.0 / 21.0;
    p = fma(p, z, 1.0 / 19.0);
    p = fma(p, z, 1.0 / 17.0);
    p = fma(p, z, 1.0 / 15.0);
    p = fma(p, z, 1.0 / 13.0);
    p = fma(p, z, 1.0 / 11.0);
    p = fma(p, z, 1.0 / 9.0);
    p = fma(p, z, 1.0 / 7.0);
    p = fma(p, z, 1.0 / 5.0);
    p = fma(p, z, 1.0 / 3.0);
    const double lnm = fma(s0 * z, 2.0 * p, 2.0 * s0);       // 2 s (1 + z p)
    return fma((double)e, 0.69314718055994530942, lnm);
}

template <int KIND>
__device__ __forceinline__ double phi64(double d2, double s)
{
    if constexpr (KIND == FD_KERNEL_THIN_PLATE) {
        // d2 is a sum of squares of fp32 differences: zero or >= 2^-298, never subnormal
        return d2 > 0.0 ? d2 * fast_log_pos(d2) : 0.0;       // weights carry the 0.5
    } else if constexpr (KIND == FD_KERNEL_GAUSSIAN || KIND == FD_KERNEL_GAUSSIAN_QNN) {
        return exp(d2 * s);                           // s = -1/R_j^2
    } else if constexpr (KIND == FD_KERNEL_BIHARMONIC) {
        return sqrt(d2);
    } else {
        return d2 * sqrt(d2);
    }
}

// thin-plate only: keeps log2 finite at d2 == 0 (phi -> -1.2e-35, i.e. 0) at no cost,
// because it rides in the first fma of the distance.
template <int KIND>
__device__ __forceinline__ constexpr float d2_bias()
{
    return KIND == FD_KERNEL_THIN_PLATE ? 1e-37f : 0.f;
}

// ---- fp32 epilogue, same operation order as the reference -------------------
__device__ __forceinline__ void normalize3(float &x, float &y, float &z)
{
    const float l2 = x * x + y * y + z * z;
    if (l2 > 0.f) {
        const float inv = 1.f / sqrtf(l2);
        x *= inv; y *= inv; z *= inv;
    }
}

// reference src/SOP_FaceDeform.hpp:28-41
__device__ __forceinline__ void project_to_tangents(const float u[3], const float v[3],
                                                    const float n[3], float d[3])
{
    float g[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) g[i][j] = u[i] * u[j] + v[i] * v[j] + n[i] * n[j];
    float a1[3], a2[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        a1[j] = u[0] * g[0][j] + u[1] * g[1][j] + u[2] * g[2][j];
        a2[j] = v[0] * g[0][j] + v[1] * g[1][j] + v[2] * g[2][j];
    }
    normalize3(a1[0], a1[1], a1[2]);
    normalize3(a2[0], a2[1], a2[2]);
    const float da1 = d[0] * a1[0] + d[1] * a1[1] + d[2] * a1[2];
    const float da2 = d[0] * a2[0] + d[1] * a2[1] + d[2] * a2[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) d[c] = a1[c] * da1 + a2[c] * da2;
}

// gate (:405-410) is decided by the caller; this is :415-438 for one vertex
__device__ __forceinline__ void epilogue_store(const EvalParams &p, int64_t i, const float pos[3],
                                               float disp[3], float dist2)
{
    if (p.tu) {
        float u[3] = {p.tu[3 * i], p.tu[3 * i + 1], p.tu[3 * i + 2]};
        float v[3] = {p.tv[3 * i], p.tv[3 * i + 1], p.tv[3 * i + 2]};
        float n[3] = {p.nrm[3 * i], p.nrm[3 * i + 1], p.nrm[3 * i + 2]};
        normalize3(u[0], u[1], u[2]);
        normalize3(v[0], v[1], v[2]);
        normalize3(n[0], n[1], n[2]);
        project_to_tangents(u, v, n, disp);
    }
    // :423-424.  Without a dist2 attribute the value is 0: min(0 / r2, 1) = +-0 and pow(1, rate) = 1
    // for every rate (C99), so the library powf -- a few dozen instructions per vertex -- is
    // skipped; r2 == 0 (0/0) keeps the general path.
    float falloff = 1.f;
    if (p.dist2 != nullptr || !(p.radius2 != 0.f)) {
        falloff = fminf(dist2 / p.radius2, 1.f);
        falloff = powf(1.f - falloff, p.falloffrate);
    }
    if (p.falloff_out) p.falloff_out[i] = falloff;
    p.P_out[3 * i] = pos[0] + disp[0] * falloff;
    p.P_out[3 * i + 1] = pos[1] + disp[1] * falloff;
    p.P_out[3 * i + 2] = pos[2] + disp[2] * falloff;
}

// ---- fp32 evaluation ----------------------------------------------------------
// LaneT = float: one vertex per register; LaneT = f32x2: two vertices per register
// pair, arithmetic on v_pk_{add,mul,fma}_f32 (measured 1.29x the issue rate of the
// scalar mix on gfx950, tools/ubench_valu.hip).  Wave-uniform operands are splat by
// the instruction's op_sel bits, so they cost no extra registers or moves.
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <typename T> struct Lanes;
template <> struct Lanes<float> {
    static constexpr int W = 1;
    static __device__ __forceinline__ float splat(float x) { return x; }
    static __device__ __forceinline__ float get(float v, int) { return v; }
    static __device__ __forceinline__ void set(float &v, int, float x) { v = x; }
};
template <> struct Lanes<f32x2> {
    static constexpr int W = 2;
    static __device__ __forceinline__ f32x2 splat(float x) { return (f32x2){x, x}; }
    static __device__ __forceinline__ float get(f32x2 v, int k) { return k ? v.y : v.x; }
    static __device__ __forceinline__ void set(f32x2 &v, int k, float x) { if (k) v.y = x; else v.x = x; }
};

__device__ __forceinline__ float vfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ f32x2 vfma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

template <int KIND>
__device__ __forceinline__ f32x2 phi32(f32x2 d2, float s)
{
    if constexpr (KIND == FD_KERNEL_THIN_PLATE) {
        const f32x2 l = {__builtin_amdgcn_logf(d2.x), __builtin_amdgcn_logf(d2.y)};
        return d2 * l;
    } else if constexpr (KIND == FD_KERNEL_GAUSSIAN || KIND == FD_KERNEL_GAUSSIAN_QNN) {
        const f32x2 e = d2 * s;
        return (f32x2){__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
    } else if constexpr (KIND == FD_KERNEL_BIHARMONIC) {
        return (f32x2){__builtin_amdgcn_sqrtf(d2.x), __builtin_amdgcn_sqrtf(d2.y)};
    } else {
        const f32x2 r = {__builtin_amdgcn_sqrtf(d2.x), __builtin_amdgcn_sqrtf(d2.y)};
        return d2 * r;
    }
}

// V vertices per lane in Q = V / W registers; vertex v sits in register v / W, component v % W
template <int KIND, int V, bool USE_LDS, typename LaneT, int SHARE = 1>
__device__ __forceinline__ void deform32_body(const EvalParams &p)
{
    using L = Lanes<LaneT>;
    constexpr int W = L::W;
    constexpr int Q = V / W;
    static_assert(V % W == 0, "V must be a multiple of the lane width");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * (kBlock * V);

    // positions are held in normalised coordinates x' = (x - x0) * inv_s (inv_s is a power of
    // two); the raw position is re-read in the epilogue
    const float nx = p.model->norm32[0], ny = p.model->norm32[1], nz = p.model->norm32[2];
    const float inv_s = p.model->norm32[3];
    LaneT px[Q], py[Q], pz[Q];
    float d2v[V];
    bool live[V];
    bool any_live = false;
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int64_t i = base + v * kBlock + tid;
        const int64_t ic = i < p.N ? i : p.N - 1;
        L::set(px[v / W], v % W, (p.P_in[3 * ic] - nx) * inv_s);
        L::set(py[v / W], v % W, (p.P_in[3 * ic + 1] - ny) * inv_s);
        L::set(pz[v / W], v % W, (p.P_in[3 * ic + 2] - nz) * inv_s);
        d2v[v] = p.dist2 ? p.dist2[ic] : 0.f;
        live[v] = (i < p.N) && !(d2v[v] > p.radius2);   // gate on squares, :402,:408
        any_live |= live[v];
    }
    const bool built = p.model->terminationtype == 1;

    double accx[V], accy[V], accz[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        // polynomial part first: C0 + L.x' + q |x'|^2 per output (affine term of the weights,
        // plus thin-plate's change-of-unit correction)
        const float *a = p.model->poly32;
        const float x = L::get(px[v / W], v % W), y = L::get(py[v / W], v % W), z = L::get(pz[v / W], v % W);
        const float xx = __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x));
        accx[v] = (double)__builtin_fmaf(a[4], xx, __builtin_fmaf(a[3], z, __builtin_fmaf(a[2], y, __builtin_fmaf(a[1], x, a[0]))));
        accy[v] = (double)__builtin_fmaf(a[9], xx, __builtin_fmaf(a[8], z, __builtin_fmaf(a[7], y, __builtin_fmaf(a[6], x, a[5]))));
        accz[v] = (double)__builtin_fmaf(a[14], xx, __builtin_fmaf(a[13], z, __builtin_fmaf(a[12], y, __builtin_fmaf(a[11], x, a[10]))));
    }

    // wave-uniform skip: every vertex of this wave is gated out or out of range
    const bool wave_work = __any(any_live) && built;

    if constexpr (USE_LDS) {
        // stage all records once per workgroup (16 B per lane, coalesced)
        const int n16 = p.Mpad * 2;
        const float4 *src = reinterpret_cast<const float4 *>(p.rec32);
        float4 *dst = reinterpret_cast<float4 *>(smem);
        for (int q = tid; q < n16; q += kBlock) dst[q] = src[q];
        __syncthreads();
    }

    if (wave_work) {
        const LaneT bias = L::splat(d2_bias<KIND>());
        LaneT ax[Q], ay[Q], az[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) ax[q] = ay[q] = az[q] = L::splat(0.f);

        struct Ctr { float cx, cy, cz, s, wx, wy, wz; };
        auto fetch = [&](int j) -> Ctr {
            Ctr c;
            if constexpr (USE_LDS) {
                const float4 *r = reinterpret_cast<const float4 *>(smem) + 2 * j;
                const float4 r0 = r[0], r1 = r[1];
                c.cx = r0.x; c.cy = r0.y; c.cz = r0.z; c.s = r0.w;
                c.wx = r1.x; c.wy = r1.y; c.wz = r1.z;
            } else {
                ConstRec32 r = (ConstRec32)(uintptr_t)(p.rec32 + j);
                c.cx = r->cx; c.cy = r->cy; c.cz = r->cz; c.s = r->s;
                c.wx = r->wx; c.wy = r->wy; c.wz = r->wz;
            }
            return c;
        };
        // Software pipeline with two register sets: while group A is consumed the records of
        // group B are already requested, and vice versa, so a wave never sits on s_waitcnt
        // with an empty pipe (the waves of a small launch run in lockstep and would all
        // stall together).  Mpad is a multiple of 2 * kGroup.
        constexpr int kGroup = 2;            // 2 sets x 2 records x 7 SGPRs; 4 spills scalars
        static_assert(kRecPad % (2 * kGroup) == 0, "record padding must cover two groups");
        // fold the fp32 partial sums into fp64 (every kChunk centres and at the end)
        auto flush = [&]() {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                accx[v] += (double)L::get(ax[v / W], v % W);
                accy[v] += (double)L::get(ay[v / W], v % W);
                accz[v] += (double)L::get(az[v / W], v % W);
            }
#pragma unroll
            for (int q = 0; q < Q; ++q) ax[q] = ay[q] = az[q] = L::splat(0.f);
        };
        // (records travel by value: an array passed by reference becomes an LDS-backed alloca)
        auto consume1 = [&](const Ctr g) {
            {
                const float cx = g.cx, cy = g.cy, cz = g.cz, s = g.s;
                const float wx = g.wx, wy = g.wy, wz = g.wz;
                // stage by stage across the Q register slots: Q independent dependency
                // chains interleave, so no stage waits on (or pads for) its predecessor
                LaneT dx[Q], dy[Q], dz[Q], d2[Q], t[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) dx[q] = px[q] - cx;
#pragma unroll
                for (int q = 0; q < Q; ++q) dy[q] = py[q] - cy;
#pragma unroll
                for (int q = 0; q < Q; ++q) dz[q] = pz[q] - cz;
#pragma unroll
                for (int q = 0; q < Q; ++q) d2[q] = vfma(dx[q], dx[q], bias);
#pragma unroll
                for (int q = 0; q < Q; ++q) d2[q] = vfma(dy[q], dy[q], d2[q]);
#pragma unroll
                for (int q = 0; q < Q; ++q) d2[q] = vfma(dz[q], dz[q], d2[q]);
#pragma unroll
                for (int q = 0; q < Q; ++q) t[q] = phi32<KIND>(d2[q], s);
#pragma unroll
                for (int q = 0; q < Q; ++q) ax[q] = vfma(t[q], L::splat(wx), ax[q]);
#pragma unroll
                for (int q = 0; q < Q; ++q) ay[q] = vfma(t[q], L::splat(wy), ay[q]);
#pragma unroll
                for (int q = 0; q < Q; ++q) az[q] = vfma(t[q], L::splat(wz), az[q]);
            }
        };
        // Multilayer Gaussian model (records centre-major: the SHARE layers of one centre are
        // consecutive): the squared distances of a centre are formed once and every layer takes
        // its own exponent of them -- 3 + 4 SHARE issue slots per centre instead of 7 SHARE.
        LaneT sd2[Q];
        auto consume2 = [&](const Ctr g0, const Ctr g1, bool new_centre) {
            if (new_centre) {
                LaneT dx[Q], dy[Q], dz[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) dx[q] = px[q] - g0.cx;
#pragma unroll
                for (int q = 0; q < Q; ++q) dy[q] = py[q] - g0.cy;
#pragma unroll
                for (int q = 0; q < Q; ++q) dz[q] = pz[q] - g0.cz;
#pragma unroll
                for (int q = 0; q < Q; ++q) sd2[q] = vfma(dx[q], dx[q], bias);
#pragma unroll
                for (int q = 0; q < Q; ++q) sd2[q] = vfma(dy[q], dy[q], sd2[q]);
#pragma unroll
                for (int q = 0; q < Q; ++q) sd2[q] = vfma(dz[q], dz[q], sd2[q]);
            }
            LaneT t0[Q], t1[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) t0[q] = phi32<KIND>(sd2[q], g0.s);
#pragma unroll
            for (int q = 0; q < Q; ++q) t1[q] = phi32<KIND>(sd2[q], g1.s);
#pragma unroll
            for (int q = 0; q < Q; ++q) ax[q] = vfma(t0[q], L::splat(g0.wx), ax[q]);
#pragma unroll
            for (int q = 0; q < Q; ++q) ay[q] = vfma(t0[q], L::splat(g0.wy), ay[q]);
#pragma unroll
            for (int q = 0; q < Q; ++q) az[q] = vfma(t0[q], L::splat(g0.wz), az[q]);
#pragma unroll
            for (int q = 0; q < Q; ++q) ax[q] = vfma(t1[q], L::splat(g1.wx), ax[q]);
#pragma unroll
            for (int q = 0; q < Q; ++q) ay[q] = vfma(t1[q], L::splat(g1.wy), ay[q]);
#pragma unroll
            for (int q = 0; q < Q; ++q) az[q] = vfma(t1[q], L::splat(g1.wz), az[q]);
        };
        static_assert(kGroup == 2 && kRecPad % 8 == 0, "four stages of two records per iteration");
        static_assert(SHARE == 1 || SHARE == 2 || SHARE == 4 || SHARE == 8, "layers that share a centre's distances");
        // Four stages per iteration; each requests the next pair of records before it consumes
        // the current pair.  Scalar loads return out of order, so every wait on them is
        // lgkmcnt(0): the wait for the pair about to be consumed comes BEFORE the next request,
        // never after.  The loads stay compiler-visible on purpose: hand-issued (inline asm)
        // s_loads whose results cross the loop back-edge get copied / their registers reused
        // before the data lands (seen in the ISA: an in-flight destination reused as an address
        // -> memory fault).  hipcc sinks only the loop-carried request to the latch, so three
        // of the four stages overlap their fetch with arithmetic.
#define FD_STAGE(N0, N1, JJ, C0, C1, ST)                                      \
        if constexpr (!USE_LDS) __builtin_amdgcn_s_waitcnt(0xc07f);           \
        N0 = fetch(JJ); N1 = fetch((JJ) + 1);                                 \
        __builtin_amdgcn_sched_barrier(0);                                    \
        if constexpr (SHARE == 1) { consume1(C0); consume1(C1); }             \
        else consume2(C0, C1, (2 * (ST)) % SHARE == 0);
        Ctr a0 = fetch(0), a1 = fetch(1), b0, b1, c0, c1, d0, d1;
        for (int j = 0; j < p.Mpad; j += 8) {
            FD_STAGE(b0, b1, j + 2, a0, a1, 0)
            FD_STAGE(c0, c1, j + 4, b0, b1, 1)
            FD_STAGE(d0, d1, j + 6, c0, c1, 2)
            const int jn = (j + 8 < p.Mpad) ? j + 8 : j;   // the last pass re-reads its own records
            FD_STAGE(a0, a1, jn, d0, d1, 3)
            if (((j + 8) & (kChunk - 1)) == 0 || j + 8 >= p.Mpad) flush();
        }
#undef FD_STAGE
    }

#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int64_t i = base + v * kBlock + tid;
        if (i >= p.N) continue;
        const float pos[3] = {p.P_in[3 * i], p.P_in[3 * i + 1], p.P_in[3 * i + 2]};
        if (!live[v] || !built) {
            if (p.P_out != p.P_in) {
                p.P_out[3 * i] = pos[0]; p.P_out[3 * i + 1] = pos[1]; p.P_out[3 * i + 2] = pos[2];
            }
            continue;
        }
        float disp[3] = {(float)accx[v], (float)accy[v], (float)accz[v]};
        epilogue_store(p, i, pos, disp, d2v[v]);
    }
}

template <int KIND, int V, bool USE_LDS, typename LaneT, int SHARE = 1>
__global__ __launch_bounds__(kBlock) void k_deform32(const EvalParams p)
{
    deform32_body<KIND, V, USE_LDS, LaneT, SHARE>(p);
}

// several frames in one launch (the default variant only: packed lanes, scalar-loaded records, V = 4)
template <int KIND>
__global__ __launch_bounds__(kBlock) void k_deform32_batch(const EvalBatch args)
{
    deform32_body<KIND, 4, false, f32x2>(args.p[blockIdx.y]);
}

// ---- thin-plate evaluation with d2 on the bf16 matrix pipe --------------------------------
// The six VALU operations per pair that build d2 (3 sub, 3 fma) are the largest slice of the
// all-VALU kernel after the logarithm.  Here one v_mfma_f32_16x16x32_bf16 produces the 256
// squared distances of a 16-centre x 16-vertex tile:
//     d2[i][j] = sum_k A[i][k] * B[k][j] = |x'_j|^2 - 2 x'_j . c'_i + |c'_i|^2
// with every fp32 value split exactly into three bf16 pieces (k_pack_tiles packs the centre
// side; the vertex side is split here).  K = 32 holds, per lane group g: the six cross products
// of coordinate g (g < 3), and for g = 3 the three pieces of |c'|^2 against 1 and 1 against the
// three pieces of |x'|^2.  Accuracy equals the direct fp32 form (measured 1.4e-6 absolute on
// [-1,1]^3 against 1.0e-6; tools/mfma_d2_test.hip).  The accumulator layout -- vertex on the
// lane (col = lane & 15), centres 4*(lane>>4)+r in the four registers -- leaves the reduction
// over centres in-lane; the four lane groups are summed once at the end.
// VALU per tile and lane: 4 log + 4 mul + 6 pk_fma instead of 20 packed ops + 4 log.
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTileChunk = 24;   // centre tiles staged in LDS at a time (24 * 1280 B = 30 KiB)

// d2 * log2|d2| with the DX9 multiply (0 * anything = 0): rounding can leave a vertex that sits
// on a centre at exactly zero or at a tiny negative d2; the first gives 0 * -inf = 0 here and the
// second an error of order 1e-7 * 23, the size of the rounding of d2 itself.  No clamp needed.
extern "C" __device__ float fd_fmul_legacy(float, float) __asm("llvm.amdgcn.fmul.legacy");
__device__ __forceinline__ float d2_log_d2(float d)
{
    return fd_fmul_legacy(d, __builtin_amdgcn_logf(__builtin_fabsf(d)));
}

// Inputs of one vertex group as a lane holds them: position + dist2 of the one vertex per tile
// quartet whose epilogue this lane runs (tile g of the quartet, column j).  Every vertex is
// loaded exactly once (the B-operand slots get their coordinate from these lanes by shuffle):
// with page-locked host arrays the loads cross the host link.
template <int TV>
struct GroupIn {
    float pos[TV / 4][3];
    float d2v[TV / 4];
};

template <int TV>
__device__ __forceinline__ GroupIn<TV> load_group(const EvalParams &p, int64_t vbase, int g, int j)
{
    GroupIn<TV> in;
#pragma unroll
    for (int q = 0; q < TV / 4; ++q) {
        const int64_t vi = vbase + 16 * (4 * q + g) + j;
        const int64_t vc = vi < p.N ? vi : p.N - 1;
        in.pos[q][0] = p.P_in[3 * vc]; in.pos[q][1] = p.P_in[3 * vc + 1]; in.pos[q][2] = p.P_in[3 * vc + 2];
        in.d2v[q] = p.dist2 ? p.dist2[vc] : 0.f;
    }
    return in;
}

// x + y across lane halves / rows: with X = value of tile a and Y = value of tile b,
// swap32_sum gives rows 0,1 = X.row g + X.row g+2 and rows 2,3 = the same of Y;
// swap16_sum(S, T) then gives row 0 = S.row0 + S.row1, row 1 = T.row0 + T.row1, row 2 =
// S.row2 + S.row3, row 3 = T.row2 + T.row3.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float swap32_sum(float x, float y)
{
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap16_sum(float x, float y)
{
    const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// HALF = false: bf16 x 3 pieces, K = 32 (MfmaTile).  HALF = true: fp16 x 2 pieces, K = 16
// (MfmaTileH): half the matrix-pipe time and operand bytes, about twice the d2 rounding error.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int TV, bool HALF>
__device__ __forceinline__ void deform32_tps_mfma_body(const EvalParams &p, int ngroups)
{
    static_assert(TV % 4 == 0, "a lane group finishes one tile of every quartet");
    using Tile = typename std::conditional<HALF, MfmaTileH, MfmaTile>::type;
    using Operand = typename std::conditional<HALF, f16x4, bf16x8>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Tile *lds_tiles = reinterpret_cast<const Tile *>(smem);
    const Tile *gl_tiles;
    if constexpr (HALF) gl_tiles = p.tiles16; else gl_tiles = p.tiles;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, j = lane & 15;
    const int ntiles = p.Mpad / 16;
    const float n0 = p.model->norm32[0], n1 = p.model->norm32[1], n2 = p.model->norm32[2];
    const float inv_s = p.model->norm32[3];
    const bool built = p.model->terminationtype == 1;
    const bool resident = ntiles <= kTileChunk;     // the whole model fits: stage it once
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int ct0, int nct) {
        const uint4 *src = reinterpret_cast<const uint4 *>(gl_tiles + ct0);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        const int n16 = nct * (int)(sizeof(Tile) / 16);
        __syncthreads();
        for (int q = tid; q < n16; q += kBlock) dst[q] = src[q];
        __syncthreads();
    };

    // a block walks vertex groups blockIdx.x, blockIdx.x + gridDim.x, ... (64 * TV vertices each);
    // the next group's inputs are in flight while this one is computed
    int grp = blockIdx.x;
    GroupIn<TV> nxt = load_group<TV>(p, ((int64_t)grp * 4 + wave) * (16 * TV), g, j);
    if (resident) stage(0, ntiles);

    for (; grp < ngroups; grp += gridDim.x) {
        const int64_t vbase = ((int64_t)grp * 4 + wave) * (16 * TV);
        const GroupIn<TV> in = nxt;
        {
            const int gn = grp + (int)gridDim.x < ngroups ? grp + (int)gridDim.x : grp;
            nxt = load_group<TV>(p, ((int64_t)gn * 4 + wave) * (16 * TV), g, j);
        }

        // B operand (vertex side) of every vertex tile of this wave
        // this lane's own vertices (tile g of every quartet, column j), normalised once: the
        // B-operand slots of all lane groups fetch from here by shuffle, and the epilogue reuses it
        float npos[TV / 4][3], xx_own[TV / 4], m2pos[TV / 4][3];
#pragma unroll
        for (int q = 0; q < TV / 4; ++q) {
            npos[q][0] = (in.pos[q][0] - n0) * inv_s;
            npos[q][1] = (in.pos[q][1] - n1) * inv_s;
            npos[q][2] = (in.pos[q][2] - n2) * inv_s;
            xx_own[q] = __builtin_fmaf(npos[q][2], npos[q][2], __builtin_fmaf(npos[q][1], npos[q][1], npos[q][0] * npos[q][0]));
#pragma unroll
            for (int c = 0; c < 3; ++c) m2pos[q][c] = -2.f * npos[q][c];
        }
        Operand bop[TV];
#pragma unroll
        for (int t = 0; t < TV; ++t) {
            // vertex (tile t, column j) sits in lane group t & 3 of quartet t / 4
            const int srcl = 16 * (t & 3) + j;
            const float c0 = __shfl(m2pos[t / 4][0], srcl), c1 = __shfl(m2pos[t / 4][1], srcl),
                        c2 = __shfl(m2pos[t / 4][2], srcl), xs = __shfl(xx_own[t / 4], srcl);
            // lane group g < 3 carries -2 x'_g, group 3 carries |x'|^2
            const float v2 = g == 0 ? c0 : (g == 1 ? c1 : (g == 2 ? c2 : xs));
            if constexpr (HALF) {
                const _Float16 h = (_Float16)v2;
                const _Float16 l = (_Float16)(v2 - (float)h);
                const _Float16 one = (_Float16)1.0f;
                f16x4 b;
                if (g < 3) b = (f16x4){h, l, h, l};          // against {c_hi, c_hi, c_lo, c_lo}
                else b = (f16x4){one, one, h, l};            // against {|c|^2_hi, |c|^2_lo, 1, 1}
                bop[t] = b;
            } else {
                const unsigned u = __float_as_uint(v2);
                const float r1 = v2 - __uint_as_float(u & 0xffff0000u);
                const unsigned u1 = __float_as_uint(r1);
                const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
                const short h = (short)(u >> 16), m = (short)(u1 >> 16), l = (short)(__float_as_uint(r2) >> 16);
                const short one = (short)0x3f80;
                bf16x8 b;
                if (g < 3) b = (bf16x8){h, m, h, m, l, h, 0, 0};
                else b = (bf16x8){one, one, one, h, m, l, 0, 0};
                bop[t] = b;
            }
        }
        bool lane_live = false;
#pragma unroll
        for (int q = 0; q < TV / 4; ++q)
            lane_live |= (vbase + 16 * (4 * q + g) + j < p.N) && !(in.d2v[q] > p.radius2);
        const bool wave_work = __any(lane_live) && built;

        f32x2 acc[TV][3];
        float acc2[TV][3];
#pragma unroll
        for (int t = 0; t < TV; ++t)
#pragma unroll
            for (int c = 0; c < 3; ++c) { acc[t][c] = (f32x2){0.f, 0.f}; if (!wave_work) acc2[t][c] = 0.f; }

        for (int ct0 = 0; ct0 < ntiles; ct0 += kTileChunk) {
            const int nct = ntiles - ct0 < kTileChunk ? ntiles - ct0 : kTileChunk;
            if (!resident) stage(ct0, nct);
            if (wave_work) {
                for (int ct = 0; ct < nct; ++ct) {
                    const Tile &tile = lds_tiles[ct];
                    const Operand aop = *reinterpret_cast<const Operand *>(&tile.a[lane][0]);
                    const float4 w0 = *reinterpret_cast<const float4 *>(&tile.w[g][0]);
                    const float4 w1 = *reinterpret_cast<const float4 *>(&tile.w[g][4]);
                    const float4 w2 = *reinterpret_cast<const float4 *>(&tile.w[g][8]);
                    const f32x2 wA[3] = {(f32x2){w0.x, w0.y}, (f32x2){w0.z, w0.w}, (f32x2){w1.x, w1.y}};   // rows 0,1
                    const f32x2 wB[3] = {(f32x2){w1.z, w1.w}, (f32x2){w2.x, w2.y}, (f32x2){w2.z, w2.w}};   // rows 2,3
                    f32x4 d[TV];
#pragma unroll
                    for (int t = 0; t < TV; ++t) {
                        if constexpr (HALF) d[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(aop, bop[t], zero4, 0, 0, 0);
                        else d[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aop, bop[t], zero4, 0, 0, 0);
                    }
#pragma unroll
                    for (int t = 0; t < TV; ++t) {
                        const f32x2 tA = {d2_log_d2(d[t][0]), d2_log_d2(d[t][1])};
                        const f32x2 tB = {d2_log_d2(d[t][2]), d2_log_d2(d[t][3])};
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            acc[t][c] = vfma(tA, wA[c], acc[t][c]);
                            acc[t][c] = vfma(tB, wB[c], acc[t][c]);
                        }
                    }
                }
                // second-level fp32 sums: a run is at most 4 * kTileChunk = 96 terms per slot.
                // (the first chunk assigns; the accumulators are only cleared if another chunk follows)
                const bool first = ct0 == 0, more = ct0 + kTileChunk < ntiles;
#pragma unroll
                for (int t = 0; t < TV; ++t)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float run = acc[t][c].x + acc[t][c].y;
                        acc2[t][c] = first ? run : acc2[t][c] + run;
                        if (more) acc[t][c] = (f32x2){0.f, 0.f};
                    }
            }
        }

        // sum over the four lane groups; afterwards every lane of column j holds the total of
        // vertex j, and lane group g finishes tile g of every quartet
#pragma unroll
        for (int q = 0; q < TV / 4; ++q) {
            // v_permlane32_swap + add folds rows g and g + 2 of two values at once, v_permlane16_swap
            // + add finishes: lane group g ends up with the total of tile g (tools/permlane_swap_test.hip)
            float mine[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float s02 = swap32_sum(acc2[4 * q + 0][c], acc2[4 * q + 2][c]);
                const float s13 = swap32_sum(acc2[4 * q + 1][c], acc2[4 * q + 3][c]);
                mine[c] = swap16_sum(s02, s13);
            }
            const int64_t i = vbase + 16 * (4 * q + g) + j;
            if (i >= p.N) continue;
            const float pos[3] = {in.pos[q][0], in.pos[q][1], in.pos[q][2]};
            const float d2v = in.d2v[q];
            if (d2v > p.radius2 || !built) {
                if (p.P_out != p.P_in) {
                    p.P_out[3 * i] = pos[0]; p.P_out[3 * i + 1] = pos[1]; p.P_out[3 * i + 2] = pos[2];
                }
                continue;
            }
            const float x = npos[q][0], y = npos[q][1], z = npos[q][2];
            const float xx = xx_own[q];
            const float *a = p.model->poly32;
            float disp[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float poly = __builtin_fmaf(a[5 * c + 4], xx, __builtin_fmaf(a[5 * c + 3], z,
                                     __builtin_fmaf(a[5 * c + 2], y, __builtin_fmaf(a[5 * c + 1], x, a[5 * c]))));
                disp[c] = poly + mine[c];
            }
            epilogue_store(p, i, pos, disp, d2v);
        }
    }
}

template <int TV, bool HALF>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_deform32_tps_mfma(const EvalParams p, int ngroups)
{
    deform32_tps_mfma_body<TV, HALF>(p, ngroups);
}

// The same evaluation for several models in ONE launch: blockIdx.y picks the model and its
// vertex arrays (the table travels as a kernel argument).  A 1M-vertex launch spends ~12 % of
// its time ramping up and draining; launches that are 8-32x larger do not (measured: 56 us per
// frame alone, 50 us per frame when evaluations overlap).
template <int TV, bool HALF>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_deform32_tps_mfma_batch(const EvalBatch args, int ngroups)
{
    deform32_tps_mfma_body<TV, HALF>(args.p[blockIdx.y], ngroups);
}

// ---- fp64 evaluation ----------------------------------------------------------
template <int KIND, int V>
__global__ __launch_bounds__(kBlock) void k_deform64(const EvalParams p)
{
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * (kBlock * V);
    float pxf[V], pyf[V], pzf[V], d2v[V];
    bool live[V];
    bool any_live = false;
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int64_t i = base + v * kBlock + tid;
        const int64_t ic = i < p.N ? i : p.N - 1;
        pxf[v] = p.P_in[3 * ic];
        pyf[v] = p.P_in[3 * ic + 1];
        pzf[v] = p.P_in[3 * ic + 2];
        d2v[v] = p.dist2 ? p.dist2[ic] : 0.f;
        live[v] = (i < p.N) && !(d2v[v] > p.radius2);
        any_live |= live[v];
    }
    const bool built = p.model->terminationtype == 1;
    double accx[V], accy[V], accz[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const double *a = p.model->affine64;
        const double x = pxf[v], y = pyf[v], z = pzf[v];
        accx[v] = fma(a[3], z, fma(a[2], y, fma(a[1], x, a[0])));
        accy[v] = fma(a[7], z, fma(a[6], y, fma(a[5], x, a[4])));
        accz[v] = fma(a[11], z, fma(a[10], y, fma(a[9], x, a[8])));
    }
    if (__any(any_live) && built) {
#pragma unroll 2
        for (int j = 0; j < p.Mpad; ++j) {
            ConstRec64 r = (ConstRec64)(uintptr_t)(p.rec64 + j);
            const double cx = r->cx, cy = r->cy, cz = r->cz, s = r->s;
            const double wx = r->wx, wy = r->wy, wz = r->wz;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const double dx = (double)pxf[v] - cx;
                const double dy = (double)pyf[v] - cy;
                const double dz = (double)pzf[v] - cz;
                const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
                const double t = phi64<KIND>(d2, s);
                accx[v] = fma(t, wx, accx[v]);
                accy[v] = fma(t, wy, accy[v]);
                accz[v] = fma(t, wz, accz[v]);
            }
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int64_t i = base + v * kBlock + tid;
        if (i >= p.N) continue;
        const float pos[3] = {pxf[v], pyf[v], pzf[v]};
        if (!live[v] || !built) {
            if (p.P_out != p.P_in) {
                p.P_out[3 * i] = pos[0]; p.P_out[3 * i + 1] = pos[1]; p.P_out[3 * i + 2] = pos[2];
            }
            continue;
        }
        float disp[3] = {(float)accx[v], (float)accy[v], (float)accz[v]};
        epilogue_store(p, i, pos, disp, d2v[v]);
    }
}

// ---- frames that share the mesh AND the rest rig: the contraction on the matrix pipe ------------------
// The frames of an animated shot, the blendshapes of one head (BASELINE configs 4 and 2-as-benchmarked):
// the same vertices against the same centres, only the deltas -- hence the weights -- differ.  Then
//     Delta_f(x_v) = poly_f(x_v) + sum_j phi(|x_v - c_j|^2) w_f[j]
// is Phi (N x M) times W (M x 3F): phi is formed ONCE per (vertex, centre) -- d2 on the matrix pipe as
// in k_deform32_tps_mfma, then one v_log_f32 and one multiply -- and the 3F-wide contraction, which is
// what costs 24 of the 38 vector instructions per 4 pairs in the one-frame kernel, becomes
// v_mfma_f32_16x16x32_f16 work: north_star's "N x M evaluation recast as a dense GEMM-like contraction".
// fp16 has 11 significant bits, so both operands go in as two pieces (hi = RN16(v), lo = RN16(v - hi):
// 22 bits) and a product is three instructions, hi*hi + hi*lo + lo*hi (the dropped lo*lo is 2^-22
// relative); accumulation is fp32 inside the matrix pipe.  Per-frame weights are scaled by a power of
// two so that their largest piece sits at 2^13 (fp16 range), undone exactly in the epilogue.
//
// Layout: a workgroup of 8 waves (two per SIMD) keeps the centre tiles and the weight tiles of a chunk
// of centres in LDS -- the whole model at M = 256, F = 32: 12 + 128 KiB -- and walks vertex groups of
// 512; a wave owns 64 vertices = 4 vertex tiles of 16.  Per 32 centres (one K block) and vertex tile:
// two d2 instructions, 8 log + 8 multiplies + the fp16 split per lane, and that lane's 8 phi values
// ARE its B operand (the k-slot <-> centre map is a free choice as long as the weight tiles use the
// same one: slot 8g + s = centre 32 kb + 16 (s >> 2) + 4 g + (s & 3)).  An output tile is 16 rows
// x 16 vertices with rows = 4 frames x (x, y, z, unused): lane group g' of the accumulator then holds
// all three components of frame 4 T + g' for its vertex and writes them as 12 contiguous bytes.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int kSharedThreads = 512;
constexpr size_t kSharedLdsBudget = 158 * 1024;   // of 160 KiB (one workgroup per CU)

// Two row layouts of the output tiles (16 rows x 16 vertices each):
//   padded (up to 12 frames): tile T holds frames 4 T .. 4 T + 3, row = 4 (frame - 4 T) + component, one row in
//     four unused -- ceil(F / 4) tiles;
//   dense (13 frames and more): frames come in blocks of 16 and a block is three tiles, one per component:
//     tile 3 B + c holds component c of frames 16 B .. 16 B + 15, row = frame - 16 B -- 3 ceil(F / 16) tiles, no
//     unused rows at F = 16, 32 (6 tiles instead of 8 at 32 frames: a quarter fewer matrix instructions).
// Either way the 4 x 4 transpose across lane groups in the epilogue leaves every lane with all rows of
// every tile for ONE vertex.
constexpr bool shared_dense(int nF) { return nF > 12; }
constexpr int shared_tiles(int nF) { return shared_dense(nF) ? 3 * ((nF + 15) / 16) : (nF + 3) / 4; }
constexpr int shared_slots(int nT, bool dense) { return dense ? nT / 3 * 16 : nT * 4; }     // frame records

struct SharedFrame {              // one per frame slot (nT * 4), written by k_pack_shared
    float inv_scale;              // 2^-k: undoes the scaling of the frame's weights and polynomial
    int built;                    // terminationtype == 1
    int pad[2];
    float *P_out, *falloff_out;   // the frame's outputs (read from LDS inside the frame loop: 64 pointers
                                  // as kernel arguments end up hoisted into SGPRs all at once and spilled)
};
static_assert(sizeof(SharedFrame) == 32, "frame record");

struct SharedOut {                // per-frame outputs (kernel argument)
    float *P_out[kMaxBatch];
    float *falloff_out[kMaxBatch];
};

struct SharedParams {
    int64_t N;
    const float *P_in;
    const float *dist2;
    const float *tu, *tv, *nrm;
    float radius2, falloffrate;
    int ntiles;                   // centre tiles (Mpad / 16)
    int nkb;                      // K blocks of 32 centres = ceil(ntiles / 2)
    int nF, nT;                   // frames, output tiles (4 frames each)
    int kchunk;                   // K blocks staged in LDS at a time
    const MfmaTileH *ctiles;      // centre tiles of the shared rest rig (any one context's)
    const DevModel *model0;       // normalisation of the shared rest rig
    const uint4 *wtiles;          // [nkb][nT][2 (hi, lo)][64 lanes] x 16 B, then the polynomial tiles [nT][64 lanes] x 16 B
    const SharedFrame *frames;    // [nT * 4]
    int dbg;                      // FD_SHARED_DBG (diagnostics, tests/tools/shared_eval_timing.py): 1 = no stores, 2 = no K loop
    int fast;                     // no dist2, no tangent frames, every frame slot in use and built, fd_falloff wanted everywhere:
                                  // full vertex groups take the branch-free epilogue (below)
    int stagger;                  // waves 4..7 start this many x 8192 cycles late (resident model only)
    unsigned long long *stamps;   // diagnostics (FD_SHARED_STAMPS): shader-clock shares of the phases, per wave of workgroup 0
};

struct SharedSlots {              // the models of the frames (kernel argument of the pack kernel)
    const Rec32 *rec32[kMaxBatch];
    const DevModel *model[kMaxBatch];
};

// weight tiles, polynomial tiles and frame records from the solved models.  grid (nkb, nT), 256 threads (the first 64 write the tile).
// The polynomial part of a frame (DevModel::poly32: C0 + L.x' + q |x'|^2 per output) rides in the
// same matrix product as five more "centres" whose phi are (1, x', y', z', |x'|^2): one K = 32
// instruction per output tile and vertex tile holds all three split products -- lane group 0
// pairs hi x hi, group 1 lo(vertex) x hi(coefficient), group 2 hi(vertex) x lo(coefficient).
__global__ __launch_bounds__(256) void k_pack_shared(const SharedSlots slots, const SharedOut out, int nF, int Mpad, int dense,
                                                      uint4 *wtiles, SharedFrame *frames)
{
    const int kb = blockIdx.x, T = blockIdx.y, nT = gridDim.y, nkb = gridDim.x;
    const int lane = threadIdx.x & 63, g = lane >> 4, rho = lane & 15;
    // row rho of tile T: frame f0 + fi, component c
    const int nfr = dense ? 16 : 4, f0 = dense ? 16 * (T / 3) : 4 * T;
    const int fi = dense ? rho : rho >> 2, c = dense ? T % 3 : rho & 3;
    // scale of each of this tile's frames: largest |weight| or |polynomial coefficient| to [2^13, 2^14).
    // 256 / nfr lanes per frame, all of a frame's records requested at once.
    __shared__ float s_scale[16];
    {
        const int per = 256 / nfr, q = threadIdx.x / per, l = threadIdx.x % per;
        const int f = f0 + q;
        float m = 0.f;
        if (f < nF) {
            const Rec32 *r = slots.rec32[f];
            for (int j = l; j < Mpad; j += per) m = fmaxf(m, fmaxf(fabsf(r[j].wx), fmaxf(fabsf(r[j].wy), fabsf(r[j].wz))));
            if (l < 15) m = fmaxf(m, fabsf(slots.model[f]->poly32[l]));
        }
        for (int off = per / 2; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));      // per is 16 or 64: inside a wave
        if (l == 0) {
            int k = 0;
            if (m > 0.f && m < INFINITY) k = 13 - (__builtin_amdgcn_frexp_expf(m) - 1);
            k = k < -100 ? -100 : (k > 100 ? 100 : k);
            s_scale[q] = ldexpf(1.f, k);
            if (kb == 0 && (!dense || T % 3 == 0)) {
                SharedFrame fr;
                fr.inv_scale = ldexpf(1.f, -k);
                fr.built = (f < nF && slots.model[f]->terminationtype == 1) ? 1 : 0;
                fr.pad[0] = fr.pad[1] = 0;
                fr.P_out = f < nF ? out.P_out[f] : nullptr;
                fr.falloff_out = f < nF ? out.falloff_out[f] : nullptr;
                frames[f] = fr;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int f = f0 + fi;
    const float sc = s_scale[fi];
    f16x8 hi, lo;
#pragma unroll
    for (int sidx = 0; sidx < 8; ++sidx) {
        const int centre = 32 * kb + 16 * (sidx >> 2) + 4 * g + (sidx & 3);
        float w = 0.f;
        if (f < nF && c < 3 && centre < Mpad) {
            const Rec32 r = slots.rec32[f][centre];
            w = (c == 0 ? r.wx : (c == 1 ? r.wy : r.wz)) * sc;
        }
        const _Float16 h = (_Float16)w;
        hi[sidx] = h;
        lo[sidx] = (_Float16)(w - (float)h);
    }
    uint4 *dst = wtiles + ((size_t)kb * nT + T) * 128;
    dst[lane] = __builtin_bit_cast(uint4, hi);
    dst[64 + lane] = __builtin_bit_cast(uint4, lo);
    if (kb == 0) {
        // polynomial tile of output tile T: k-slot s < 5 of lane group g carries coefficient s
        // ({C0, Lx, Ly, Lz, q}) of row rho -- hi piece in groups 0 and 1, lo piece in group 2
        f16x8 pt;
#pragma unroll
        for (int sidx = 0; sidx < 8; ++sidx) {
            float w = 0.f;
            if (f < nF && c < 3 && sidx < 5 && g < 3) w = slots.model[f]->poly32[5 * c + sidx] * sc;
            const _Float16 h = (_Float16)w;
            pt[sidx] = g == 2 ? (_Float16)(w - (float)h) : h;
        }
        wtiles[(size_t)nkb * nT * 128 + (size_t)T * 64 + lane] = __builtin_bit_cast(uint4, pt);
    }
}

// fp32 pair -> its two fp16 pieces, packed: hi = RN16(v), lo = RN16(v - hi).  One v_cvt_pk_f16_f32 and
// two mixed-precision fmas that subtract the fp16 piece from the fp32 value and round the
// remainder to fp16 in the same instruction (v_fma_mixlo/hi_f16 write one half of the destination
// and keep the other) -- three instructions for two values, no unpacking, no repacking.
__device__ __forceinline__ void split_pair_f16(float v0, float v1, unsigned &hi, unsigned &lo)
{
    const f16x2 hh = __builtin_convertvector((f32x2){v0, v1}, f16x2);
    hi = __builtin_bit_cast(unsigned, hh);
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(v0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(v1));
    lo = l;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
// one vertex position as a single 12-byte store (dword-aligned: global_store_dwordx3)
struct __attribute__((packed, aligned(4))) Pos3 { float x, y, z; };
__device__ __forceinline__ void store_pos3(Pos3 FD_GLOBAL *dst, float x, float y, float z)
{
    dst->x = x; dst->y = y; dst->z = z;      // member-wise: a struct assignment through an address-space pointer does not compile on the host pass
}

template <int NT, bool DENSE>
__global__ __launch_bounds__(kSharedThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_deform32_tps_shared(const SharedParams p, int ngroups)
{
    constexpr int TV = 4;                        // vertex tiles per wave
    constexpr int kSlots = shared_slots(NT, DENSE);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: [frame records kSlots][polynomial tiles NT*64 x 16 B][centre tiles kchunk*2][weight tiles kchunk*NT*2*64 x 16 B]
    SharedFrame *s_frames = reinterpret_cast<SharedFrame *>(smem);
    uint4 *s_poly = reinterpret_cast<uint4 *>(smem + sizeof(SharedFrame) * (size_t)kSlots);
    MfmaTileH *s_ct = reinterpret_cast<MfmaTileH *>(s_poly + NT * 64);
    uint4 *s_w = reinterpret_cast<uint4 *>(reinterpret_cast<char *>(s_ct) + sizeof(MfmaTileH) * (size_t)(2 * p.kchunk));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform, and the compiler should know it
    const int g = lane >> 4, j = lane & 15;
    const float n0 = p.model0->norm32[0], n1 = p.model0->norm32[1], n2 = p.model0->norm32[2];
    const float inv_s = p.model0->norm32[3];
    const bool resident = p.nkb <= p.kchunk;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int kb0, int nk) {
        __syncthreads();
        {   // centre tiles 2 kb0 .. 2 (kb0 + nk) - 1; beyond ntiles: zeros
            const uint4 *src = reinterpret_cast<const uint4 *>(p.ctiles + 2 * kb0);
            uint4 *dst = reinterpret_cast<uint4 *>(s_ct);
            const int per = (int)(sizeof(MfmaTileH) / 16);
            const int have = (p.ntiles - 2 * kb0) * per;
            for (int q = tid; q < 2 * nk * per; q += kSharedThreads) dst[q] = q < have ? src[q] : make_uint4(0u, 0u, 0u, 0u);
        }
        {
            const uint4 *src = p.wtiles + (size_t)kb0 * NT * 128;
            const int n16 = nk * NT * 128;
            for (int q = tid; q < n16; q += kSharedThreads) s_w[q] = src[q];
        }
        __syncthreads();
    };

    // The frame records (scale, status, output pointers: 8 dwords x 4 NT frames) live across the
    // lanes of NT / 2 registers for the whole kernel; the epilogue picks a frame's scalars out with
    // v_readlane -- no memory round trip per frame.  (From LDS every frame paid an LDS read behind
    // the other wave's operand traffic; through the scalar cache 800 cycles per pair of frames; as
    // kernel arguments the compiler hoists 32 x 6 scalars above the K loop and spills them.)
    constexpr int kTabRegs = (kSlots * 8 + 63) / 64;
    unsigned tab[kTabRegs];
#pragma unroll
    for (int q = 0; q < kTabRegs; ++q) {
        const int idx = 64 * q + lane;
        tab[q] = idx < kSlots * 8 ? reinterpret_cast<const unsigned *>(p.frames)[idx] : 0u;
    }
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.frames);
        uint4 *dst = reinterpret_cast<uint4 *>(s_frames);
        for (int q = tid; q < kSlots * (int)(sizeof(SharedFrame) / 16); q += kSharedThreads) dst[q] = src[q];
        const uint4 *psrc = p.wtiles + (size_t)p.nkb * NT * 128;
        for (int q = tid; q < NT * 64; q += kSharedThreads) s_poly[q] = psrc[q];
    }
    if (resident) {
        stage(0, p.nkb);
        // The two waves of a SIMD (w and w + 4) run the same program: left alone they reach their
        // logarithm phase together and their matrix phase together, and each phase then has one
        // pipe idle.  A start-up delay for the second half puts one wave's vector work beside
        // the other's matrix work (no barrier follows while the model is resident).
        if (wave >= 4) {
            __builtin_amdgcn_s_sleep(12);
            for (int q = 0; q < (p.stagger & 0xff); ++q) __builtin_amdgcn_s_sleep(127);
        }
        // experiment: workgroups start in four phases ((stagger >> 8) x 8128 cycles apart)
        for (int q = 0; q < (p.stagger >> 8) * (int)(blockIdx.x & 3); ++q) __builtin_amdgcn_s_sleep(127);
    } else {
        __syncthreads();
    }

    const bool stamp = p.stamps != nullptr && blockIdx.x == 0;
    unsigned long long st_prev = 0, st_acc[5] = {0, 0, 0, 0, 0};
#define FD_SSTAMP(K) if (stamp) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[K] += t_ - st_prev; st_prev = t_; __builtin_amdgcn_sched_barrier(0); }
    if (stamp) st_prev = __builtin_amdgcn_s_memtime();
    // Inputs of a vertex group as lane (g, j) holds them: the four vertices (vt, j).  The NEXT group's
    // are requested before this group's stores go out: vector memory operations of a wave retire in
    // issue order, so a load queued behind the epilogue's 64 stores would wait for all of them.
    struct GroupRaw { float p[TV][3]; float d2[TV]; };
    auto load_raw = [&](int grp_, auto fastTag) {
        constexpr bool FAST = decltype(fastTag)::value;
        GroupRaw r;
        const int64_t vb = ((int64_t)grp_ * (kSharedThreads / 64) + wave) * (16 * TV);
#pragma unroll
        for (int t = 0; t < TV; ++t) {
            const int64_t vi = vb + 16 * t + j;
            const int64_t vc = vi < p.N ? vi : p.N - 1;
            r.p[t][0] = p.P_in[3 * vc]; r.p[t][1] = p.P_in[3 * vc + 1]; r.p[t][2] = p.P_in[3 * vc + 2];
            if constexpr (FAST) r.d2[t] = 0.f; else r.d2[t] = p.dist2 ? p.dist2[vc] : 0.f;
        }
        return r;
    };
    // "These loaded registers are needed now": placed in straight-line code right after a group's stores, it lets the
    // compiler count exactly how many stores follow the loads and wait with that count; consumed for the first time at
    // the top of the next iteration -- where the path from the prologue joins -- it would have to assume none.
    auto settle = [&](const GroupRaw &r) {
        asm volatile("" :: "v"(r.p[0][0]), "v"(r.p[0][1]), "v"(r.p[0][2]), "v"(r.p[1][0]), "v"(r.p[1][1]), "v"(r.p[1][2]),
                           "v"(r.p[2][0]), "v"(r.p[2][1]), "v"(r.p[2][2]), "v"(r.p[3][0]), "v"(r.p[3][1]), "v"(r.p[3][2]));
    };
    GroupRaw nxt = load_raw(blockIdx.x < (unsigned)ngroups ? (int)blockIdx.x : 0, std::false_type{});
    settle(nxt);
    // One vertex group (512 vertices of the workgroup, 64 of this wave).  FAST: the group is full and the
    // launch has no gate, fall-off, tangent frames or unbuilt frames -- the epilogue is then straight-line
    // code in which every lane issues every store.  That matters beyond the instruction count: with no
    // branch that could skip a store, the compiler KNOWS 64 stores follow the next group's loads and waits
    // for those loads with vmcnt(63); with conditional stores it has to assume none were issued, waits with
    // vmcnt(0), and every wave sits out the drain of its own 32 KB of stores (all waves at once: the whole
    // chip alternated between a matrix phase with HBM idle and a store phase with the pipes idle).
    auto do_group = [&](int grp, auto fastTag) {
        constexpr bool FAST = decltype(fastTag)::value;
        const int64_t vbase = ((int64_t)grp * (kSharedThreads / 64) + wave) * (16 * TV);
        // The two waves of a SIMD (w, w + 4) take the higher issue priority in turn, group by group: left to the
        // default (oldest first) waves 0..3 finish all their groups a quarter of the kernel early and the others run
        // the rest with nobody to fill their stalls.
        if ((((grp / (int)gridDim.x) ^ (wave >> 2)) & 1) != 0) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
        const GroupRaw cur = nxt;
        // this lane's own vertex in the epilogue is (vt = g, j): one of the four it has loaded
        auto pick = [&](float a0, float a1, float a2, float a3) {        // two levels of v_cndmask, no branches
            const float lo = (g & 1) ? a1 : a0, hi = (g & 1) ? a3 : a2;
            return (g & 2) ? hi : lo;
        };
        const float pos[3] = {pick(cur.p[0][0], cur.p[1][0], cur.p[2][0], cur.p[3][0]), pick(cur.p[0][1], cur.p[1][1], cur.p[2][1], cur.p[3][1]),
                              pick(cur.p[0][2], cur.p[1][2], cur.p[2][2], cur.p[3][2])};
        const float own_d2 = pick(cur.d2[0], cur.d2[1], cur.d2[2], cur.d2[3]);
        // every lane group holds vertex (vt, j): the d2 operand needs one coordinate of it per lane
        // group, the polynomial operand all of them
        f16x4 bop[TV];
        f32x4 acc[NT][TV];
        bool lane_live = false;
#pragma unroll
        for (int t = 0; t < TV; ++t) {
            const int64_t vi = vbase + 16 * t + j;
            const float x = (cur.p[t][0] - n0) * inv_s, y = (cur.p[t][1] - n1) * inv_s, z = (cur.p[t][2] - n2) * inv_s;
            const float d2v = cur.d2[t];
            if constexpr (FAST) lane_live = true; else lane_live |= (vi < p.N) && !(d2v > p.radius2);
            const float xx = __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x));
            const float v2 = g == 0 ? -2.f * x : (g == 1 ? -2.f * y : (g == 2 ? -2.f * z : xx));
            const _Float16 h = (_Float16)v2;
            const _Float16 l = (_Float16)(v2 - (float)h);
            const _Float16 one = (_Float16)1.0f;
            bop[t] = g < 3 ? (f16x4){h, l, h, l} : (f16x4){one, one, h, l};
            // polynomial operand: k-slots {1, x', y', z', |x'|^2}: hi pieces in lane groups 0 and 2,
            // lo pieces in group 1 (against the coefficients' hi), nothing in group 3
            unsigned xyh, xyl, zxh, zxl;
            split_pair_f16(x, y, xyh, xyl);
            split_pair_f16(z, xx, zxh, zxl);
            u32x4 pb;
            if (g == 1) pb = (u32x4){xyl << 16, (xyl >> 16) | (zxl << 16), zxl >> 16, 0u};                  // {0, xl, yl, zl, xxl}
            else pb = (u32x4){0x3c00u | (xyh << 16), (xyh >> 16) | (zxh << 16), zxh >> 16, 0u};            // {1, xh, yh, zh, xxh}
            if (g == 3) pb = (u32x4){0u, 0u, 0u, 0u};
            const f16x8 pbv = __builtin_bit_cast(f16x8, pb);
#pragma unroll
            for (int T = 0; T < NT; ++T)
                acc[T][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, s_poly[T * 64 + lane]), pbv, zero4, 0, 0, 0);
        }
        const bool wave_work = FAST ? true : __any(lane_live);
        if constexpr (FAST) {
            // The next group's positions are requested HERE, a whole K loop before the stores of this group: a wave
            // has at most 63 vector-memory operations in flight and they retire in order, so loads requested just
            // before the epilogue's stores would hold up the last of them until their own data (queued behind a
            // chip-wide burst of stores) has come back -- every epilogue then lasts one loaded-memory round trip.
            const int gn = grp + (int)gridDim.x;
            nxt = load_raw(gn < ngroups ? gn : grp, fastTag);
        }
        FD_SSTAMP(0)

        // phi of K block kb (32 centres) for the wave's four vertex tiles, split into fp16 pieces: the B operands
        auto phi_block = [&](int kb, u32x4 (&xh)[TV], u32x4 (&xl)[TV]) {
            const f16x4 aopA = *reinterpret_cast<const f16x4 *>(&s_ct[2 * kb].a[lane][0]);
            const f16x4 aopB = *reinterpret_cast<const f16x4 *>(&s_ct[2 * kb + 1].a[lane][0]);
#pragma unroll
            for (int t = 0; t < TV; ++t) {
                const f32x4 da = __builtin_amdgcn_mfma_f32_16x16x16f16(aopA, bop[t], zero4, 0, 0, 0);
                const f32x4 db = __builtin_amdgcn_mfma_f32_16x16x16f16(aopB, bop[t], zero4, 0, 0, 0);
                unsigned h, l;
                split_pair_f16(d2_log_d2(da[0]), d2_log_d2(da[1]), h, l); xh[t][0] = h; xl[t][0] = l;
                split_pair_f16(d2_log_d2(da[2]), d2_log_d2(da[3]), h, l); xh[t][1] = h; xl[t][1] = l;
                split_pair_f16(d2_log_d2(db[0]), d2_log_d2(db[1]), h, l); xh[t][2] = h; xl[t][2] = l;
                split_pair_f16(d2_log_d2(db[2]), d2_log_d2(db[3]), h, l); xh[t][3] = h; xl[t][3] = l;
            }
        };
        // acc += W(kb) x phi(kb): three split products per output tile and vertex tile
        auto contract = [&](int kb, const u32x4 (&xh)[TV], const u32x4 (&xl)[TV]) {
            const uint4 *wk = s_w + (size_t)kb * NT * 128 + lane;
#pragma unroll
            for (int T = 0; T < NT; ++T) {
                const f16x8 ah = __builtin_bit_cast(f16x8, wk[T * 128]), al = __builtin_bit_cast(f16x8, wk[T * 128 + 64]);
#pragma unroll
                for (int t = 0; t < TV; ++t) {
                    const f16x8 vh = __builtin_bit_cast(f16x8, xh[t]), vl = __builtin_bit_cast(f16x8, xl[t]);
                    acc[T][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, vh, acc[T][t], 0, 0, 0);
                    acc[T][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, vh, acc[T][t], 0, 0, 0);
                    acc[T][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, vl, acc[T][t], 0, 0, 0);
                }
            }
        };
        for (int kb0 = 0; kb0 < p.nkb; kb0 += p.kchunk) {
            const int nk = p.nkb - kb0 < p.kchunk ? p.nkb - kb0 : p.kchunk;
            if (!resident) stage(kb0, nk);
            if ((!FAST && !wave_work) || (p.dbg & 2)) continue;
            // Software pipeline over the K blocks: while the matrix pipe contracts block kb with the weights, the
            // vector unit forms phi of block kb + 1 (two d2 instructions per vertex tile, 8 logarithms, 8 multiplies
            // and the fp16 split per lane).  Inside one wave the two would otherwise run back to back -- the
            // logarithms with the matrix pipe idle, then 72 matrix instructions with the vector unit idle -- and two
            // waves per SIMD running the same program do not interleave well enough to hide either.
            u32x4 bh[TV], bl[TV];
            phi_block(0, bh, bl);
            for (int kb = 0; kb + 1 < nk; ++kb) {
                u32x4 nbh[TV], nbl[TV];
                phi_block(kb + 1, nbh, nbl);
                contract(kb, bh, bl);
                // issue order inside this block: one matrix instruction, then the vector work that fits under it
#pragma unroll
                for (int q = 0; q < 32; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);      // transcendental
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);      // VALU
                }
#pragma unroll
                for (int q = 32; q < 8 + NT * TV * 3; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                }
                // (unrolling by two with the buffers swapped instead of these 32 copies spills: 109 registers)
#pragma unroll
                for (int t = 0; t < TV; ++t) { bh[t] = nbh[t]; bl[t] = nbl[t]; }
            }
            contract(nk - 1, bh, bl);
        }

        FD_SSTAMP(1)
        // ---- epilogue.  The accumulators hold, in lane group g, frame 4 T + g for the four vertex
        // tiles; a 4 x 4 transpose across the lane groups (two v_permlane32_swap + two
        // v_permlane16_swap per four registers) turns that into vertex tile g for the four frames
        // of the tile: every lane then owns ONE vertex (vbase + lane), does the per-vertex work
        // (gate, fall-off, tangent axes) once, and a frame's 64 positions leave as one contiguous
        // 768-byte store.  All transposes first, in place (acc[T][k][c] becomes row 4 k + c of tile T for
        // this lane's vertex: padded layout frame 4 T + k, component c; dense layout component T % 3 of
        // frame 16 (T / 3) + 4 k + c), while the wave is still converged.
#pragma unroll
        for (int T = 0; T < NT; ++T) {
#pragma unroll
            for (int c = 0; c < (DENSE ? 4 : 3); ++c) {
                // X_k[g] = (rows 4 g .., vertex tile k)  ->  Y_k[g] = (rows 4 k .., vertex tile g)
                const u32x2 s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[T][0][c]), __float_as_uint(acc[T][2][c]), false, false);
                const u32x2 s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[T][1][c]), __float_as_uint(acc[T][3][c]), false, false);
                const u32x2 y01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
                const u32x2 y23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
                acc[T][0][c] = __uint_as_float(y01[0]); acc[T][1][c] = __uint_as_float(y01[1]);
                acc[T][2][c] = __uint_as_float(y23[0]); acc[T][3][c] = __uint_as_float(y23[1]);
            }
        }
        // the reference's order: gate -> tangent projection -> fall-off -> add (src/SOP_FaceDeform.cpp:405-438)
        const int64_t i = vbase + lane;
        const bool inb = i < p.N;
        const int64_t ic = inb ? i : p.N - 1;
        const bool gated = own_d2 > p.radius2;
        const unsigned off12 = 12u * (unsigned)lane, off4 = 4u * (unsigned)lane;   // byte offsets inside the wave's 64-vertex window
        if constexpr (!FAST) {
            const int gn = grp + (int)gridDim.x;
            nxt = load_raw(gn < ngroups ? gn : grp, fastTag);
        }
        FD_SSTAMP(2)
        if constexpr (FAST) {
            // gate open, fall-off 1 (pow(1 - 0, rate): no dist2 attribute), no tangent frames: P + d * 1.  The sum the
            // matrix pipe holds is 2^k d: one fma with the exact 2^-k gives the same bits as (d * 1) + P.
            // 32 position stores + 16 fall-off stores (two frames each) = 48 operations per group: with the
            // next group's loads they fit the 63 a wave may have in flight, so the epilogue never waits for an
            // acknowledgement.  fd_falloff of frames (fs, fs + 1): lanes 0..31 write vertices 2 l, 2 l + 1 of frame
            // fs, lanes 32..63 the same of frame fs + 1 (8 bytes each: two 256-byte rows per instruction).
            const f32x2 ones = {1.f, 1.f};
            const unsigned off8 = 8u * (unsigned)(lane & 31);
#pragma unroll
            for (int fs = 0; fs < kSlots; ++fs) {
                const float inv = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)tab[(8 * fs) / 64], (8 * fs) % 64));
                const uint64_t pout = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)tab[(8 * fs + 5) / 64], (8 * fs + 5) % 64) << 32) |
                                      (unsigned)__builtin_amdgcn_readlane((int)tab[(8 * fs + 4) / 64], (8 * fs + 4) % 64);
                float d0, d1, d2c;
                if constexpr (DENSE) {
                    const int B = fs / 16, k = (fs % 16) / 4, r = fs % 4;
                    d0 = acc[3 * B][k][r]; d1 = acc[3 * B + 1][k][r]; d2c = acc[3 * B + 2][k][r];
                } else {
                    const int T = fs / 4, k = fs % 4;
                    d0 = acc[T][k][0]; d1 = acc[T][k][1]; d2c = acc[T][k][2];
                }
                Pos3 FD_GLOBAL *dstP = (Pos3 FD_GLOBAL *)((char FD_GLOBAL *)(pout + 12ull * (uint64_t)vbase) + off12);
                if (fs % 2 == 0) {
                    const uint64_t fa = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)tab[(8 * fs + 7) / 64], (8 * fs + 7) % 64) << 32) |
                                        (unsigned)__builtin_amdgcn_readlane((int)tab[(8 * fs + 6) / 64], (8 * fs + 6) % 64);
                    const int fs1 = fs + 1 < kSlots ? fs + 1 : fs;
                    const uint64_t fb = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)tab[(8 * fs1 + 7) / 64], (8 * fs1 + 7) % 64) << 32) |
                                        (unsigned)__builtin_amdgcn_readlane((int)tab[(8 * fs1 + 6) / 64], (8 * fs1 + 6) % 64);
                    const uint64_t fo = (lane < 32 ? fa : fb) + 4ull * (uint64_t)vbase;
                    *(f32x2 FD_GLOBAL *)((char FD_GLOBAL *)fo + off8) = ones;
                }
                store_pos3(dstP, __builtin_fmaf(d0, inv, pos[0]), __builtin_fmaf(d1, inv, pos[1]), __builtin_fmaf(d2c, inv, pos[2]));
            }
            settle(nxt);
            FD_SSTAMP(3)
            return;
        }
        if (inb && gated) {
            // B2: a gated vertex keeps its position (and no fd_falloff entry is written)
            for (int f = 0; f < p.nF; ++f) {
                float *dstp = s_frames[f].P_out;
                if (dstp != p.P_in) store_pos3((Pos3 FD_GLOBAL *)as_global(dstp) + i, pos[0], pos[1], pos[2]);
            }
        }
        float fall = 1.f;
        float a1[3] = {0.f, 0.f, 0.f}, a2[3] = {0.f, 0.f, 0.f};
        const bool doit = inb && !gated;
        if (doit) {
            if (p.dist2 != nullptr || !(p.radius2 != 0.f)) {
                const float q = fminf(own_d2 / p.radius2, 1.f);
                fall = powf(1.f - q, p.falloffrate);
            }
            if (p.tu) {
                // project_to_tangents (src/SOP_FaceDeform.hpp:28-41): the two axes depend on the vertex only
                float u[3] = {p.tu[3 * ic], p.tu[3 * ic + 1], p.tu[3 * ic + 2]};
                float v[3] = {p.tv[3 * ic], p.tv[3 * ic + 1], p.tv[3 * ic + 2]};
                float n[3] = {p.nrm[3 * ic], p.nrm[3 * ic + 1], p.nrm[3 * ic + 2]};
                normalize3(u[0], u[1], u[2]);
                normalize3(v[0], v[1], v[2]);
                normalize3(n[0], n[1], n[2]);
                float gm[3][3];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) gm[r][c] = u[r] * u[c] + v[r] * v[c] + n[r] * n[c];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    a1[c] = u[0] * gm[0][c] + u[1] * gm[1][c] + u[2] * gm[2][c];
                    a2[c] = v[0] * gm[0][c] + v[1] * gm[1][c] + v[2] * gm[2][c];
                }
                normalize3(a1[0], a1[1], a1[2]);
                normalize3(a2[0], a2[1], a2[2]);
            }
        }
#pragma unroll
        for (int fs = 0; fs < kSlots; ++fs) {
            {
                const int f = fs;                    // wave-uniform, compile-time after unrolling
                if (f >= p.nF) continue;
                // word w of frame f sits in lane (8 f + w) % 64 of tab[(8 f + w) / 64]
                const float inv = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)tab[(8 * f) / 64], (8 * f) % 64));
                const bool built = __builtin_amdgcn_readlane((int)tab[(8 * f + 1) / 64], (8 * f + 1) % 64) != 0;
                const uint64_t pout = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)tab[(8 * f + 5) / 64], (8 * f + 5) % 64) << 32) |
                                      (unsigned)__builtin_amdgcn_readlane((int)tab[(8 * f + 4) / 64], (8 * f + 4) % 64);
                const uint64_t fout = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)tab[(8 * f + 7) / 64], (8 * f + 7) % 64) << 32) |
                                      (unsigned)__builtin_amdgcn_readlane((int)tab[(8 * f + 6) / 64], (8 * f + 6) % 64);
                if (!doit) continue;
                // scalar base (the wave's window of the frame's arrays) + a 32-bit lane offset
                Pos3 FD_GLOBAL *dstP = (Pos3 FD_GLOBAL *)((char FD_GLOBAL *)(pout + 12ull * (uint64_t)vbase) + off12);
                if (!built) {
                    if (pout != (uint64_t)p.P_in) store_pos3(dstP, pos[0], pos[1], pos[2]);
                    continue;
                }
                // 2^-k is exact: disp is the sum the matrix pipe accumulated, polynomial included
                float disp[3];
                if constexpr (DENSE) {
                    const int B = fs / 16, k = (fs % 16) / 4, r = fs % 4;
                    disp[0] = acc[3 * B][k][r] * inv; disp[1] = acc[3 * B + 1][k][r] * inv; disp[2] = acc[3 * B + 2][k][r] * inv;
                } else {
                    const int T = fs / 4, k = fs % 4;
                    disp[0] = acc[T][k][0] * inv; disp[1] = acc[T][k][1] * inv; disp[2] = acc[T][k][2] * inv;
                }
                if (p.dbg & 1) continue;         // diagnostics: everything but the stores
                if (p.tu) {
                    const float da1 = disp[0] * a1[0] + disp[1] * a1[1] + disp[2] * a1[2];
                    const float da2 = disp[0] * a2[0] + disp[1] * a2[1] + disp[2] * a2[2];
#pragma unroll
                    for (int c = 0; c < 3; ++c) disp[c] = a1[c] * da1 + a2[c] * da2;
                }
                if (fout) *(float FD_GLOBAL *)((char FD_GLOBAL *)(fout + 4ull * (uint64_t)vbase) + off4) = fall;
                store_pos3(dstP, pos[0] + disp[0] * fall, pos[1] + disp[1] * fall, pos[2] + disp[2] * fall);
            }
        }
        FD_SSTAMP(3)
    };
    int grp = blockIdx.x;
    // a frame whose build failed passes the mesh through: general path (the status words sit in the frame table)
    bool built_here = true;
#pragma unroll
    for (int q = 0; q < kTabRegs; ++q) {
        const int idx = 64 * q + lane;
        if (idx < kSlots * 8 && (idx & 7) == 1) built_here = built_here && tab[q] != 0u;
    }
    if (p.fast && __all(built_here)) {
        const int nfull = (int)(p.N / kSharedThreads);       // groups in which every wave's 64 vertices exist
        for (; grp < nfull; grp += gridDim.x) do_group(grp, std::true_type{});
    }
    for (; grp < ngroups; grp += gridDim.x) do_group(grp, std::false_type{});
    if (stamp && lane == 0) {
        for (int q = 0; q < 4; ++q) p.stamps[wave * 8 + q] = st_acc[q];
    }
#undef FD_SSTAMP
}

template <int KIND>
hipError_t launch_kind(const DeformArgs &a, const EvalParams &p, hipStream_t stream)
{
    if (a.N <= 0) return hipSuccess;
    if (a.precision == FD_EVAL_FP64) {
        constexpr int V = 2;
        const int64_t per = (int64_t)kBlock * V;
        const unsigned grid = (unsigned)((a.N + per - 1) / per);
        hipLaunchKernelGGL((k_deform64<KIND, V>), dim3(grid), dim3(kBlock), 0, stream, p);
        return hipGetLastError();
    }
    // variant = lanes * 100 + source * 10 + log2(V):  lanes 0 scalar / 1 packed,
    // source 0 scalar-loaded records / 1 LDS-staged records.  0 = the default below.
    int variant = a.variant > 0 ? a.variant : kDefaultVariant;
    // thin-plate with at least four centre tiles: d2 on the matrix pipe (variant 200) is the
    // faster kernel (C2 68 vs 80 us, C3 429 vs 540 us); below that its per-group set-up shows
    if (a.variant <= 0 && KIND == FD_KERNEL_THIN_PLATE && a.tiles != nullptr && a.Mpad >= 64) {
        static const bool bf16_tiles = getenv("FD_MFMA_BF16") != nullptr;
        variant = (a.tiles16 != nullptr && !bf16_tiles) ? 202 : 200;
    }
    if (variant == 202 && a.tiles16 == nullptr) variant = 200;
    if (variant == 200 || variant == 202) {
        if constexpr (KIND == FD_KERNEL_THIN_PLATE) {
            constexpr int TV = 4;
            const int64_t per = (int64_t)kBlock / 64 * 16 * TV;     // vertices per workgroup
            const int64_t ngroups = (a.N + per - 1) / per;
            // at most ~8 workgroups per CU in the grid; beyond that a workgroup walks several
            // vertex groups and stages a resident model only once
            static const int64_t max_grid = [] {
                const char *e = getenv("FD_MFMA_GRID");
                const long v = e ? atol(e) : 0;
                return (int64_t)(v > 0 ? v : 2048);
            }();
            const int64_t rounds = (ngroups + max_grid - 1) / max_grid;
            const unsigned grid = (unsigned)((ngroups + rounds - 1) / rounds);
            const int ntiles = a.Mpad / 16;
            const size_t nres = (size_t)(ntiles < kTileChunk ? ntiles : kTileChunk);
            if (variant == 202)
                hipLaunchKernelGGL((k_deform32_tps_mfma<TV, true>), dim3(grid), dim3(kBlock), sizeof(MfmaTileH) * nres, stream,
                                   p, (int)ngroups);
            else
                hipLaunchKernelGGL((k_deform32_tps_mfma<TV, false>), dim3(grid), dim3(kBlock), sizeof(MfmaTile) * nres, stream,
                                   p, (int)ngroups);
            return hipGetLastError();
        } else {
            variant = kDefaultVariant;   // the matrix-pipe path exists for thin-plate only
        }
    }
    if constexpr (KIND == FD_KERNEL_GAUSSIAN) {
        // multilayer model, default kernel: the layers of a centre share its distances (the
        // records come centre-major from k_pack, layers a multiple of the share)
        if (a.variant <= 0 && a.layers >= 2 && a.layers % 2 == 0) {
            const int64_t per = (int64_t)kBlock * 4;
            const unsigned grid = (unsigned)((a.N + per - 1) / per);
            const unsigned share = (grid + kNumCU - 1) / kNumCU;
            const bool bal = getenv("FD_NO_BALANCE") == nullptr;
            const size_t dyn = (bal && share >= 3 && share <= 8) ? (((160u * 1024u) / share) & ~1023u) : 0;
            if (a.layers % 8 == 0)
                hipLaunchKernelGGL((k_deform32<KIND, 4, false, f32x2, 8>), dim3(grid), dim3(kBlock), dyn, stream, p);
            else if (a.layers % 4 == 0)
                hipLaunchKernelGGL((k_deform32<KIND, 4, false, f32x2, 4>), dim3(grid), dim3(kBlock), dyn, stream, p);
            else
                hipLaunchKernelGGL((k_deform32<KIND, 4, false, f32x2, 2>), dim3(grid), dim3(kBlock), dyn, stream, p);
            return hipGetLastError();
        }
    }
    const size_t lds_bytes = (size_t)a.Mpad * sizeof(Rec32);
    if ((variant / 10) % 10 == 1 && lds_bytes > 64 * 1024) variant -= 10;   // one LDS tile must hold them all
    // Even placement: with every workgroup resident at once the dispatcher may stack 5 on one
    // CU and 3 on another, and the kernel then lasts as long as the fullest CU.  Reserving
    // 160 KiB / ceil(grid / 256) of LDS per workgroup caps every CU at the even share.
    const bool balance = getenv("FD_NO_BALANCE") == nullptr;
#define FD_LAUNCH(VV, LDS, LT)                                                                      \
    do {                                                                                             \
        const int64_t per = (int64_t)kBlock * (VV);                                                  \
        const unsigned grid = (unsigned)((a.N + per - 1) / per);                                     \
        size_t dyn = (LDS) ? lds_bytes : 0;                                                          \
        const unsigned share = (grid + kNumCU - 1) / kNumCU;                                         \
        if (!(LDS) && balance && share >= 3 && share <= 8) dyn = ((160u * 1024u) / share) & ~1023u;  \
        hipLaunchKernelGGL((k_deform32<KIND, VV, LDS, LT>), dim3(grid), dim3(kBlock), dyn, stream, p); \
        return hipGetLastError();                                                                    \
    } while (0)
    switch (variant) {
    case 1: FD_LAUNCH(2, false, float);
    case 2: FD_LAUNCH(4, false, float);
    case 3: FD_LAUNCH(8, false, float);
    case 11: FD_LAUNCH(2, true, float);
    case 12: FD_LAUNCH(4, true, float);
    case 13: FD_LAUNCH(8, true, float);
    case 101: FD_LAUNCH(2, false, f32x2);
    case 102: FD_LAUNCH(4, false, f32x2);
    case 103: FD_LAUNCH(8, false, f32x2);
    case 111: FD_LAUNCH(2, true, f32x2);
    case 112: FD_LAUNCH(4, true, f32x2);
    case 113: FD_LAUNCH(8, true, f32x2);
    default: return hipErrorInvalidValue;
    }
#undef FD_LAUNCH
}

}  // namespace

hipError_t launch_deform(const DeformArgs &a, hipStream_t stream)
{
    EvalParams p;
    p.N = a.N;
    p.P_in = a.P_in; p.P_out = a.P_out;
    p.dist2 = a.dist2; p.falloff_out = a.falloff_out;
    p.tu = a.tu; p.tv = a.tv; p.nrm = a.nrm;
    p.radius2 = a.radius2; p.falloffrate = a.falloffrate;
    p.Mpad = a.Mpad;
    p.rec32 = a.rec32; p.rec64 = a.rec64; p.tiles = a.tiles; p.tiles16 = a.tiles16;
    p.model = a.model;
    switch (a.kind) {
    case FD_KERNEL_GAUSSIAN:
    case FD_KERNEL_GAUSSIAN_QNN: return launch_kind<FD_KERNEL_GAUSSIAN>(a, p, stream);
    case FD_KERNEL_THIN_PLATE: return launch_kind<FD_KERNEL_THIN_PLATE>(a, p, stream);
    case FD_KERNEL_BIHARMONIC: return launch_kind<FD_KERNEL_BIHARMONIC>(a, p, stream);
    case FD_KERNEL_CUBIC: return launch_kind<FD_KERNEL_CUBIC>(a, p, stream);
    default: return hipErrorInvalidValue;
    }
}

static EvalParams make_params(const DeformArgs &a)
{
    EvalParams p;
    p.N = a.N;
    p.P_in = a.P_in; p.P_out = a.P_out;
    p.dist2 = a.dist2; p.falloff_out = a.falloff_out;
    p.tu = a.tu; p.tv = a.tv; p.nrm = a.nrm;
    p.radius2 = a.radius2; p.falloffrate = a.falloffrate;
    p.Mpad = a.Mpad;
    p.rec32 = a.rec32; p.rec64 = a.rec64; p.tiles = a.tiles; p.tiles16 = a.tiles16;
    p.model = a.model;
    return p;
}

// n evaluations: one launch when every one of them would take the default thin-plate
// matrix-pipe kernel on equally sized inputs, the single launches otherwise.  Either way each
// model's result is bit-identical to launch_deform on its own.
hipError_t launch_deform_batch(const DeformArgs *a, int n, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    // same default fp32 kernel for all, equally sized inputs?
    bool same = n > 1 && n <= kMaxBatch;
    for (int i = 0; i < n && same; ++i)
        same = a[i].precision == FD_EVAL_FP32 && a[i].variant <= 0 && a[i].N == a[0].N && a[i].N > 0 &&
               a[i].Mpad == a[0].Mpad && a[i].kind == a[0].kind &&
               !(a[i].layers >= 2 && a[i].layers % 2 == 0);     // shared-distance multilayer kernel: single launches (same bits as fd_deform)
    static const bool bf16_tiles = getenv("FD_MFMA_BF16") != nullptr;
    bool mfma = same && a[0].kind == FD_KERNEL_THIN_PLATE && a[0].Mpad >= 64 && !bf16_tiles;
    for (int i = 0; i < n && mfma; ++i) mfma = a[i].tiles16 != nullptr;
    const bool valu = same && !mfma && !(a[0].kind == FD_KERNEL_THIN_PLATE && a[0].Mpad >= 64);
    if (!mfma && !valu) {
        for (int i = 0; i < n; ++i) {
            const hipError_t e = launch_deform(a[i], stream);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    EvalBatch args;
    for (int i = 0; i < n; ++i) args.p[i] = make_params(a[i]);
    for (int i = n; i < kMaxBatch; ++i) args.p[i] = args.p[0];
    if (valu) {
        const int64_t per = (int64_t)kBlock * 4;
        const dim3 grid((unsigned)((a[0].N + per - 1) / per), (unsigned)n);
        switch (a[0].kind) {
        case FD_KERNEL_GAUSSIAN:
        case FD_KERNEL_GAUSSIAN_QNN:
            hipLaunchKernelGGL((k_deform32_batch<FD_KERNEL_GAUSSIAN>), grid, dim3(kBlock), 0, stream, args); break;
        case FD_KERNEL_THIN_PLATE:
            hipLaunchKernelGGL((k_deform32_batch<FD_KERNEL_THIN_PLATE>), grid, dim3(kBlock), 0, stream, args); break;
        case FD_KERNEL_BIHARMONIC:
            hipLaunchKernelGGL((k_deform32_batch<FD_KERNEL_BIHARMONIC>), grid, dim3(kBlock), 0, stream, args); break;
        case FD_KERNEL_CUBIC:
            hipLaunchKernelGGL((k_deform32_batch<FD_KERNEL_CUBIC>), grid, dim3(kBlock), 0, stream, args); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    constexpr int TV = 4;
    const int64_t per = (int64_t)kBlock / 64 * 16 * TV;
    const int64_t ngroups = (a[0].N + per - 1) / per;
    const int64_t rounds = (ngroups + 2047) / 2048;
    const unsigned grid = (unsigned)((ngroups + rounds - 1) / rounds);
    const int ntiles = a[0].Mpad / 16;
    const size_t nres = (size_t)(ntiles < kTileChunk ? ntiles : kTileChunk);
    hipLaunchKernelGGL((k_deform32_tps_mfma_batch<TV, true>), dim3(grid, (unsigned)n), dim3(kBlock), sizeof(MfmaTileH) * nres, stream,
                       args, (int)ngroups);
    return hipGetLastError();
}

// Frames of one mesh and one rest rig (SharedDeformArgs): pack the weight tiles, then one launch.
hipError_t launch_deform_shared(const SharedDeformArgs &a, hipStream_t stream)
{
    if (a.N <= 0 || a.nF <= 0) return hipSuccess;
    if (a.nF > kMaxBatch || a.Mpad % 16 != 0) return hipErrorInvalidValue;
    const int ntiles = a.Mpad / 16, nkb = (ntiles + 1) / 2;
    const bool dense = shared_dense(a.nF);
    const int nT = shared_tiles(a.nF);
    SharedSlots slots{};
    SharedOut out{};
    for (int f = 0; f < kMaxBatch; ++f) {
        const int q = f < a.nF ? f : 0;
        slots.rec32[f] = a.rec32[q]; slots.model[f] = a.model[q];
        out.P_out[f] = a.P_out[q]; out.falloff_out[f] = a.falloff_out ? a.falloff_out[q] : nullptr;
    }
    hipLaunchKernelGGL(k_pack_shared, dim3(nkb, nT), dim3(256), 0, stream, slots, out, a.nF, a.Mpad, dense ? 1 : 0, (uint4 *)a.wtiles,
                       (SharedFrame *)a.frames);
    SharedParams p{};
    p.N = a.N; p.P_in = a.P_in; p.dist2 = a.dist2; p.tu = a.tu; p.tv = a.tv; p.nrm = a.nrm;
    p.radius2 = a.radius2; p.falloffrate = a.falloffrate;
    p.ntiles = ntiles; p.nkb = nkb; p.nF = a.nF; p.nT = nT;
    p.ctiles = a.ctiles; p.model0 = a.model[0];
    p.wtiles = (const uint4 *)a.wtiles; p.frames = (const SharedFrame *)a.frames;
    { static const char *e = getenv("FD_SHARED_DBG"); p.dbg = e ? atoi(e) : 0; }
    {
        static const bool no_fast = getenv("FD_SHARED_NO_FAST") != nullptr;       // A/B: general epilogue everywhere
        bool fast = !no_fast && (p.dbg & 1) == 0 && a.dist2 == nullptr && a.tu == nullptr && a.radius2 > 0.f && a.falloff_out != nullptr &&
                    a.nF == shared_slots(nT, dense);
        for (int f = 0; fast && f < a.nF; ++f) fast = a.falloff_out[f] != nullptr && a.P_out[f] != nullptr;
        p.fast = fast ? 1 : 0;
    }
    { static const char *e = getenv("FD_SHARED_STAGGER"); p.stagger = e ? atoi(e) : 0; }
    static unsigned long long *d_stamps = nullptr;
    static const bool want_stamps = getenv("FD_SHARED_STAMPS") != nullptr;
    if (want_stamps && !d_stamps) (void)hipMalloc((void **)&d_stamps, 64 * sizeof(unsigned long long));
    p.stamps = want_stamps ? d_stamps : nullptr;
    { static const bool e = getenv("FD_SHARED_STAMPS_GENERAL") != nullptr; if (want_stamps && e) p.fast = 0; }
    const size_t fixed = sizeof(SharedFrame) * (size_t)shared_slots(nT, dense) + (size_t)nT * 64 * 16;
    const size_t per_kb = 2 * sizeof(MfmaTileH) + (size_t)nT * 128 * 16;
    int kchunk = (int)((kSharedLdsBudget - fixed) / per_kb);
    if (kchunk < 1) return hipErrorInvalidValue;
    if (kchunk > nkb) kchunk = nkb;
    p.kchunk = kchunk;
    const size_t lds = fixed + per_kb * (size_t)kchunk;
    const int64_t per = kSharedThreads / 64 * 64;          // vertices per workgroup and group
    const int64_t ngroups = (a.N + per - 1) / per;
    // One persistent workgroup per CU (150 KiB of LDS, two 240-register waves per SIMD: nothing else
    // fits beside it).  FD_SHARED_CUS < 256 leaves the other CUs to whatever runs on other streams --
    // the builds of the next frames in a pipeline (bench.py).
    static const int64_t max_wgs = [] {
        const char *e = getenv("FD_SHARED_CUS");
        const long v = e ? atol(e) : 0;
        return (int64_t)(v > 0 && v < (long)kNumCU ? v : (long)kNumCU);
    }();
    const unsigned grid = (unsigned)(ngroups < max_wgs ? ngroups : max_wgs);
#define FD_SHARED_CASE(NTV, DNS)                                                                                     \
    {                                                                                                                \
        static bool attr_set = false;                                                                                \
        if (!attr_set) {                                                                                             \
            hipError_t e = hipFuncSetAttribute((const void *)k_deform32_tps_shared<NTV, DNS>,                       \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);              \
            if (e != hipSuccess) return e;                                                                           \
            attr_set = true;                                                                                         \
        }                                                                                                            \
        hipLaunchKernelGGL((k_deform32_tps_shared<NTV, DNS>), dim3(grid), dim3(kSharedThreads), lds, stream, p, (int)ngroups); \
    }
    if (dense) {
        if (nT == 3) FD_SHARED_CASE(3, true)
        else if (nT == 6) FD_SHARED_CASE(6, true)
        else return hipErrorInvalidValue;
    } else {
        if (nT == 1) FD_SHARED_CASE(1, false)
        else if (nT == 2) FD_SHARED_CASE(2, false)
        else if (nT == 3) FD_SHARED_CASE(3, false)
        else return hipErrorInvalidValue;
    }
#undef FD_SHARED_CASE
    if (want_stamps && d_stamps) {
        unsigned long long h[64];
        if (hipMemcpy(h, d_stamps, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "[shared stamps, shader cycles per wave of workgroup 0: load+poly | K loop | transposes | per-vertex + frames]\n");
            for (int w = 0; w < 8; ++w)
                fprintf(stderr, "   wave %d: %8llu %8llu %8llu %8llu\n", w, h[w * 8], h[w * 8 + 1], h[w * 8 + 2], h[w * 8 + 3]);
        }
    }
    return hipGetLastError();
}

size_t shared_wtile_bytes(int Mpad, int nF)
{
    const int nkb = (Mpad / 16 + 1) / 2, nT = shared_tiles(nF);
    return (size_t)nkb * nT * 128 * 16 + (size_t)nT * 64 * 16;      // weight tiles + polynomial tiles
}
size_t shared_frame_bytes(int nF)
{
    return sizeof(SharedFrame) * (size_t)shared_slots(shared_tiles(nF), shared_dense(nF));
}

const char *deform_kernel_name(int kind, int precision, int variant)
{
    (void)kind;
    if (precision == FD_EVAL_FP64) return "k_deform64";
    (void)variant;
    return "k_deform32";
}

}  // namespace fd
