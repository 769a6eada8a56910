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

#include "fd_eval_common.h"

namespace fd {

namespace {

constexpr int kBlock = 256;
constexpr int kChunk = 64;  // centres per fp32 partial sum
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
    int delta;             // write the displacement, not P + displacement (fd_set_output); shares Mpad's 8 bytes
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

// ---- fp32 epilogue, same operation order as the reference (normalize3: fd_eval_common.h) ----

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
    // non-temporal: a frame's 16 MB of output need not displace what a build running beside the evaluation keeps in L2
    // (measured on the shared-rig launch: +6 % on the whole pipeline)
#ifdef FD_EVAL_TEMPORAL_STORES
    if (p.falloff_out) p.falloff_out[i] = falloff;
    const float b0 = p.delta ? 0.f : pos[0], b1 = p.delta ? 0.f : pos[1], b2 = p.delta ? 0.f : pos[2];
    p.P_out[3 * i] = b0 + disp[0] * falloff;
    p.P_out[3 * i + 1] = b1 + disp[1] * falloff;
    p.P_out[3 * i + 2] = b2 + disp[2] * falloff;
#else
    if (p.falloff_out) __builtin_nontemporal_store(falloff, &p.falloff_out[i]);
    // (FD_OUTPUT_DISPLACEMENT: the addend of :438 alone -- 0 + d f is d f exactly)
    const float b0 = p.delta ? 0.f : pos[0], b1 = p.delta ? 0.f : pos[1], b2 = p.delta ? 0.f : pos[2];
    __builtin_nontemporal_store(b0 + disp[0] * falloff, &p.P_out[3 * i]);
    __builtin_nontemporal_store(b1 + disp[1] * falloff, &p.P_out[3 * i + 1]);
    __builtin_nontemporal_store(b2 + disp[2] * falloff, &p.P_out[3 * i + 2]);
#endif
}

// ---- fp32 evaluation ----------------------------------------------------------
// LaneT = float: one vertex per register; LaneT = f32x2: two vertices per register
// pair, arithmetic on v_pk_{add,mul,fma}_f32 (measured 1.29x the issue rate of the
// scalar mix on gfx950, tools/ubench_valu.hip).  Wave-uniform operands are splat by
// the instruction's op_sel bits, so they cost no extra registers or moves.

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
            if (p.delta) {
                p.P_out[3 * i] = 0.f; p.P_out[3 * i + 1] = 0.f; p.P_out[3 * i + 2] = 0.f;      // a gated or unbuilt vertex does not move
            } else if (p.P_out != p.P_in) {
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

constexpr int kTileChunk = 24;   // centre tiles staged in LDS at a time (24 * 1280 B = 30 KiB)

// d2 * log2|d2| with the DX9 multiply: d2_log_d2 in fd_eval_common.h

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
                if (p.delta) {
                    p.P_out[3 * i] = 0.f; p.P_out[3 * i + 1] = 0.f; p.P_out[3 * i + 2] = 0.f;      // a gated or unbuilt vertex does not move
                } else if (p.P_out != p.P_in) {
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
            if (p.delta) {
                p.P_out[3 * i] = 0.f; p.P_out[3 * i + 1] = 0.f; p.P_out[3 * i + 2] = 0.f;      // a gated or unbuilt vertex does not move
            } else if (p.P_out != p.P_in) {
                p.P_out[3 * i] = pos[0]; p.P_out[3 * i + 1] = pos[1]; p.P_out[3 * i + 2] = pos[2];
            }
            continue;
        }
        float disp[3] = {(float)accx[v], (float)accy[v], (float)accz[v]};
        epilogue_store(p, i, pos, disp, d2v[v]);
    }
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
        static const bool bf16_tiles = tuning_env("FD_MFMA_BF16") != nullptr;
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
                const char *e = tuning_env("FD_MFMA_GRID");
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
            const unsigned ncu = device_cus(), share = (grid + ncu - 1) / ncu;
            const bool bal = tuning_env("FD_NO_BALANCE") == nullptr;
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
    const bool balance = tuning_env("FD_NO_BALANCE") == nullptr;
    const unsigned ncu = device_cus();
#define FD_LAUNCH(VV, LDS, LT)                                                                      \
    do {                                                                                             \
        const int64_t per = (int64_t)kBlock * (VV);                                                  \
        const unsigned grid = (unsigned)((a.N + per - 1) / per);                                     \
        size_t dyn = (LDS) ? lds_bytes : 0;                                                          \
        const unsigned share = (grid + ncu - 1) / ncu;                                               \
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
    p.delta = a.delta_out;
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
    p.delta = a.delta_out;
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
    static const bool bf16_tiles = tuning_env("FD_MFMA_BF16") != nullptr;
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

const char *deform_kernel_name(int kind, int precision, int variant)
{
    (void)kind;
    if (precision == FD_EVAL_FP64) return "k_deform64";
    (void)variant;
    return "k_deform32";
}

}  // namespace fd
