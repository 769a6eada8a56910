// fd_build_reg.hip -- the whole build of a rig of up to 256 control points in ONE launch of ONE workgroup per model, with the
// matrix in REGISTERS: control table, kernel-matrix assembly, null-space projection, blocked Cholesky, both substitutions,
// packing.
//
// Same system, same unknowns and the same mathematics as fd_nullspace.hip (replaces the control table of reference
// src/SOP_FaceDeform.cpp:268-287 and alglib::rbfsetpoints / rbfbuildmodel, :331-368, in north_star's dense formulation):
//
//     [ K   P ] [ w ]   [ f ]     P = Q [0; R],   B = Q^T K Q = K - V W^T - W V^T,   B11 y = (Q^T f)_1 (Cholesky, order
//     [ P^T 0 ] [ a ] = [ 0 ]     n1 = M - T),    R a = (Q^T f)_2 - B21 y,           w = Q [y; 0]
//
// Why another kernel.  Round 2's one-launch build (k_build_small) kept the matrix in L2: every phase was a handful of
// dependent ~1 us round trips that four waves could not overlap -- 0.37 ms at M = 256 against 0.25 ms for the launch chain.
// Here the lower triangle lives in the accumulator registers of a workgroup's waves and never leaves the CU:
//
//   * 512 threads = 8 waves, two per SIMD, 256 registers each.  Waves 0..6 (the WORKERS) hold the 16 x 16 tiles of the lower
//     triangle in the v_mfma_f64_16x16x4_f64 C/D layout (lane (c, g) = (lane & 15, lane >> 4) holds rows g, g + 4, g + 8, g + 12
//     of column c: 4 doubles = 8 registers per tile), up to 20 tiles = 160 registers per wave; order 256 = 136 tiles.  Wave 7
//     holds none: it runs what is sequential BESIDE the workers -- the reflectors of P with Q^T f folded in, the small 4 x 4
//     algebra of W, the factorisation of every diagonal block, the forward-substituted right-hand sides, the recovery of a
//     and w = Q [y; 0] -- with DPP reductions and no workgroup barrier of its own;
//   * tile (I, J) -> worker by tile_owner(): column J dealt round robin from a per-column offset, the diagonal tile with the
//     tile left of it (the look-ahead below needs them on one wave); slots in column-major order;
//   * the kernel matrix is assembled tile by tile in a ROLLED loop (one copy of the fp64 logarithm) and staged through the
//     context's otherwise unused matrix buffer -- 272 KB that come back from L2 by unconditional loads: with the resident tiles'
//     registers carried through that loop the compiler spilled them around every iteration;
//   * per 16 columns, with LOOK-AHEAD: wave 7 factorises diagonal block K and inverts the factor IN ITS REGISTERS, four columns
//     at a time -- the 4 x 4 pivot block goes to every lane by v_readlane and is factorised and inverted redundantly (no
//     cross-lane traffic inside it), then [T | I] <- row operations as matrix instructions: rows_b <- inv(L_bb) rows_b and the
//     rows below -= L_rb rows_b, whose multipliers ARE the transposed result of the first (the Schur complement is symmetric),
//     in exactly the operand layout; the workers turn the tiles below into L_IK = C_IK inv(L_KK)^T (four matrix instructions
//     each, through the tile's own slot of the LDS panel buffer), the owner of (K + 1, K) applies it to diagonal tile K + 1 at
//     once and hands that to wave 7, which factorises block K + 1 while the workers apply panel K to the rest
//     (C_IJ -= L_IK L_JK^T, four instructions per tile);
//   * the three right-hand sides live in LDS and ride along (z_K by wave 7, f_I -= z_K L_IK^T by the owners of the panel tiles);
//     back substitution in row form (y^T L = z^T), right-looking: a tile is its own B operand; the wave that owns tile
//     (I, I - 1) also solves block I - 1, so a step costs ONE barrier;
//   * Y = K V, the rotation to B = K - V W^T - W V^T and B21 work on the tiles in place; packing (fd_pack.h) is the last phase
//     and posts the status word itself.
//
// Roof: the fp64 matrix pipe (78.6 TFLOP/s); algorithmic work (1/3) n1^3 = 5.3 MFLOP at M = 256.  What bounds it is the
// factorisation's critical path -- 256 dependent pivots -- and the register budget (160 of a wave's 256 registers are tiles),
// not flops: DESIGN.md 4.2d has the phase stamps, the counters and what was tried.
// fd_set_deltas on a context built this way simply builds again (no factor is kept; the build is faster than the stored-
// factor path was), bit-identical by construction.
#include <cstdio>
#include <cstdlib>

#include "fd_internal.h"
#include "fd_pack.h"

namespace fd {

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr double kEps = 2.220446049250313e-16;
constexpr int kRegThreads = 512;        // 8 waves, two per SIMD: 256 registers each
constexpr int kRegWaves = kRegThreads / 64;
constexpr int kMaxBlocks = 16;          // 16 x 16 tiles per side: order <= 256
constexpr int kWorkers = 7;             // waves 0..6 hold the tiles; wave 7 factorises diagonal blocks beside their updates
constexpr int kSlots = 20;              // tiles per worker: ceil(136 / 7)
constexpr int kPitch = 17;              // doubles per row of a 16 x 16 tile in LDS
constexpr int kTileLds = 16 * kPitch;

__device__ __forceinline__ double readlane_f64(double v, int src_lane)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), src_lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), src_lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__device__ __forceinline__ int opaque_s(int v)
{
    asm volatile("" : "+s"(v));
    return v;
}

// sum over the 64 lanes of a wave, every lane gets it: rotate-reduce inside each row of 16 on the DPP network (row_ror
// 1, 2, 4, 8), then the four rows through v_readlane.  Fixed order: deterministic.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// (round 4: on the matrix pipe.  The DPP form -- row_ror 1, 2, 4, 8, then four v_readlane pairs -- is a chain of ~25 dependent
//  instructions with DPP and SGPR hazards between them: ~450 cycles a sum as measured in the reflector wave, whose seven sums per
//  reflector were most of its time.  v as the A operand against ones: D[i][j] = sum_g v(c = i, g) for every j, and a lane's four
//  result registers hold the row sums of c = g, g + 4, g + 8, g + 12; their sum t depends on g alone, and t against ones again gives
//  sum_g t_g -- the total -- in every lane.  Two matrix instructions and three adds; independent sums follow each other down the
//  pipe.  Called with the whole wave active (the matrix instructions read every lane whatever EXEC says).  Fixed order: deterministic.)
__device__ __forceinline__ double wave_sum(double v)
{
    const double4_t zero = {0.0, 0.0, 0.0, 0.0};
    const double4_t d = __builtin_amdgcn_mfma_f64_16x16x4f64(v, 1.0, zero, 0, 0, 0);
    const double t = (d[0] + d[1]) + (d[2] + d[3]);
    const double4_t e = __builtin_amdgcn_mfma_f64_16x16x4f64(t, 1.0, zero, 0, 0, 0);
    return e[0];
}
__device__ __forceinline__ double wave_max(double v)
{
    double o;
    o = dpp_f64<0x121>(v); v = o > v ? o : v;
    o = dpp_f64<0x122>(v); v = o > v ? o : v;
    o = dpp_f64<0x124>(v); v = o > v ? o : v;
    o = dpp_f64<0x128>(v); v = o > v ? o : v;
    const double a = readlane_f64(v, 0), b = readlane_f64(v, 16), c = readlane_f64(v, 32), d = readlane_f64(v, 48);
    const double ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

// lanes of one wave talking through LDS: a store by one lane and a load by another are unrelated to the compiler
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 1 / sqrt(d): the hardware estimate (v_rsq_f64: ~2^-23 relative) and ONE third-order correction,
// y = y0 (1 + e / 2 + 3 e^2 / 8), e = 1 - d y0^2 (the next term, 5 e^3 / 16, is below 2^-70) -- five dependent operations
// where two coupled Newton steps are eight: this sits 256 times on the factorisation's critical path
// (tools/ubench_f64.hip: 7 cycles per dependent v_fma_f64, 18 for the estimate).
__device__ __forceinline__ double rsqrt_nr(double d)
{
    const double y0 = __builtin_amdgcn_rsq(d);
    const double t = d * y0;
    const double e = fma(-t, y0, 1.0);
    double p = fma(e, 0.375, 0.5);
    p = p * e;
    return fma(y0, p, y0);
}

// ln of a positive, normal double to ~1.5 ulp (as fd_eval.hip's fast_log_pos: 2 atanh((m - 1) / (m + 1)), eleven odd terms)
__device__ __forceinline__ double log_pos(double x)
{
    double m = __builtin_amdgcn_frexp_mant(x);
    int e = __builtin_amdgcn_frexp_exp(x);
    if (m < 0.70710678118654752) { m *= 2.0; e -= 1; }
    const double num = m - 1.0, den = m + 1.0;
    double r = __builtin_amdgcn_rcp(den);
    r = fma(fma(-den, r, 1.0), r, r);
    r = fma(fma(-den, r, 1.0), r, r);
    double s0 = num * r;
    s0 = fma(fma(-den, s0, num), r, s0);
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
    const double lnm = fma(s0 * z, 2.0 * p, 2.0 * s0);
    return fma((double)e, 0.69314718055994530942, lnm);
}
// the kernel in fp64, as fd_build.hip's phi_d (the kinds this path takes)
__device__ __forceinline__ double phi_reg(int kind, double d2, double inv_r2)
{
    switch (kind) {
    case FD_KERNEL_GAUSSIAN: return exp(-d2 * inv_r2);
    case FD_KERNEL_THIN_PLATE: return d2 > 0.0 ? 0.5 * d2 * log_pos(d2) : 0.0;
    case FD_KERNEL_BIHARMONIC: return -sqrt(d2);
    default: return d2 * sqrt(d2);
    }
}

// LDS map (doubles).  The per-wave partial sums of Y = K V (8 x 4 M) exist only before the factorisation and lie over the
// inverse blocks and the panel buffer, which exist only from then on.
struct RegLds {
    double *cen, *V, *W, *F, *B21, *small, *minv, *P, *scr, *Z, *Y, *stat, *ypart, *D;
    int *tab, *flag;
};
constexpr int kSmallDoubles = 96;       // tau[4] R[16] Tm[16] g[12] G[16] misc
constexpr int kTau = 0, kR = 4, kTm = 20, kG = 36, kGm = 48;
constexpr int kRows = 16 * kMaxBlocks;   // the O(M) arrays are laid out for 256 rows whatever M is (rows beyond M read as zero)
__host__ __device__ inline size_t reg_lds_doubles(int)
{
    const size_t overlay = (size_t)kMaxBlocks * kTileLds + (size_t)(kMaxBlocks + 1) * kTileLds;      // minv + P
    const size_t ypart = (size_t)kRegWaves * 4 * kRows;
    return (size_t)3 * kRows + 4 * (size_t)kRows + 4 * (size_t)kRows + 3 * (size_t)kRows + 4 * (size_t)kRows + kSmallDoubles +
           (overlay > ypart ? overlay : ypart) + (size_t)kRegWaves * 2 * kTileLds + 3 * 256 + 3 * 256 + 8 + kTileLds + 160 /* tile table + flag, as ints */;
}
__device__ __forceinline__ RegLds carve(double *base, int)
{
    RegLds L;
    double *p = base;
    L.cen = p; p += 3 * kRows;
    L.V = p; p += 4 * kRows;
    L.W = p; p += 4 * kRows;
    L.F = p; p += 3 * kRows;
    L.B21 = p; p += 4 * kRows;
    L.small = p; p += kSmallDoubles;
    const size_t overlay = (size_t)kMaxBlocks * kTileLds + (size_t)(kMaxBlocks + 1) * kTileLds;
    const size_t ypart = (size_t)kRegWaves * 4 * kRows;
    L.minv = p; L.P = p + (size_t)kMaxBlocks * kTileLds; L.ypart = p;
    p += overlay > ypart ? overlay : ypart;
    L.scr = p; p += (size_t)kRegWaves * 2 * kTileLds;      // two transposition buffers per wave
    L.Z = p; p += 3 * 256;
    L.Y = p; p += 3 * 256;
    L.stat = p; p += 8;
    L.D = p; p += kTileLds;                                // the diagonal tile on its way to the factor wave
    L.tab = reinterpret_cast<int *>(p);
    L.flag = L.tab + 300;
    return L;
}

// ---- the diagonal tile, in registers --------------------------------------------------------------------------------
// T: the tile in the accumulator layout (symmetric; T[i] lane (c, g) = element (g + 4 i, c)).  Returns inv(L) of its Cholesky
// factor in the same layout.  Four columns at a time (b = 0..3):
//   1. the 4 x 4 pivot block S = T[4b.., 4b..] goes to every lane (ten v_readlane pairs), and every lane factorises it and
//      inverts the factor: the dependent chain of a block is four reciprocal square roots and nothing crosses lanes;
//   2. rows_b <- inv(L_bb) rows_b of [T | U] (U starts as the identity): one K = 4 matrix instruction each, the rows ARE the
//      B operand as they sit in register b;
//   3. rows below -= L_rb rows_b: the multipliers L_rb = (S_rb L_bb^-T) are the transpose of what step 2 left in the T part
//      (the Schur complement is symmetric) -- lane (m, k) of that result holds L[m][4 b + k]: the A operand as it sits.
// After four rounds U = inv(L).  Pivots at or below `tiny` (or NaN) are flagged and their rows zeroed, as everywhere else.
// `piv`: lane (c, .) comes back with the pivot of column c as it was met (the caller keeps the statistics per lane and
// reduces them once, after the last block).
__device__ __forceinline__ double4_t factor_invert_tile(double4_t Tt, int lane, double &piv)
{
    const int c = lane & 15, g = lane >> 4;
    piv = 0.0;
    double4_t U;
#pragma unroll
    for (int i = 0; i < 4; ++i) U[i] = (g + 4 * i == c) ? 1.0 : 0.0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        // element (4 b + p, 4 b + q) of the current tile: register b of lane (c = 4 b + q, g = p)
        double s[4][4];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q <= p; ++q) s[p][q] = readlane_f64(Tt[b], 4 * b + q + 16 * p);
        double l[4][4], inv[4], dk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double d = s[k][k];
#pragma unroll
            for (int e = 0; e < k; ++e) d = fma(-l[k][e], l[k][e], d);
            dk[k] = d;
            // (no test on the chain: a pivot at or below `tiny`, or NaN, is flagged below and the model reports -4 -- what its
            // numbers then are does not matter, and nothing of it reaches another model)
            inv[k] = rsqrt_nr(d);
#pragma unroll
            for (int p = k + 1; p < 4; ++p) {
                double v = s[p][k];
#pragma unroll
                for (int e = 0; e < k; ++e) v = fma(-l[p][e], l[k][e], v);
                l[p][k] = v * inv[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) piv = c == 4 * b + k ? dk[k] : piv;
        // inverse of the 4 x 4 factor (lower triangular): m[p][q]
        double m[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            m[q][q] = inv[q];
#pragma unroll
            for (int p = q + 1; p < 4; ++p) {
                double v = 0.0;
#pragma unroll
                for (int e = q; e < p; ++e) v = fma(l[p][e], m[e][q], v);
                m[p][q] = -v * inv[p];
            }
        }
        // A operand of step 2: lane (c = row p < 4, g = column q) holds m[p][q] (zero above the diagonal and for c >= 4)
        double aop = 0.0;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q <= p; ++q) aop = (c == p && g == q) ? m[p][q] : aop;
        const double4_t zero = {0.0, 0.0, 0.0, 0.0};
        const double4_t rt = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, Tt[b], zero, 0, 0, 0);      // register 0: lane (n, k) = new row 4 b + k, column n
        const double4_t ru = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, U[b], zero, 0, 0, 0);
        const double mult = c > 4 * b + 3 ? -rt[0] : 0.0;          // -L[c][4 b + g] for the rows below the block
        Tt = __builtin_amdgcn_mfma_f64_16x16x4f64(mult, rt[0], Tt, 0, 0, 0);
        U = __builtin_amdgcn_mfma_f64_16x16x4f64(mult, ru[0], U, 0, 0, 0);
        U[b] = ru[0];
        Tt[b] = rt[0];
    }
    return U;
}

// The packing's inputs out of this workgroup's LDS (fd_pack.h: PackGlobal reads the same doubles from global memory).  dlt == nullptr:
// the deltas from the context's global copy (the one-workgroup form has no LDS left for them).
constexpr int kAff = 64;                 // L.small[kAff + 3 k + e]: polynomial coefficient k of right-hand side e (recovery -> packing)
struct PackLds {
    const double *cen, *dlt, *sol, *aff;
    const float *dl_global;
    double R;
    int sing, dup;
    __device__ __forceinline__ double centre(int j, int q) const { return cen[3 * j + q]; }
    __device__ __forceinline__ bool has_delta() const { return true; }
    __device__ __forceinline__ double delta(int j, int q) const { return dlt ? dlt[3 * j + q] : (double)dl_global[3 * j + q]; }
    __device__ __forceinline__ double weight(int j, int c) const { return sol[c * 256 + j]; }
    __device__ __forceinline__ double affine(int cc, int k, int, int T) const { return k < T ? aff[3 * k + cc] : 0.0; }
    __device__ __forceinline__ double radius(int) const { return R; }
    __device__ __forceinline__ int sing_flag() const { return sing; }
    __device__ __forceinline__ int dup_flag() const { return dup; }
};

// An operand that only some lanes carry (three right-hand sides, four reflector columns): read UNCONDITIONALLY from a clamped
// address and selected afterwards.  Written as `cond ? lds[i] : 0.0` the read sits in an exec-masked block of its own, and the four
// operand reads of a tile are then issued one at a time, each behind a full LDS round trip (seen in the ISA: 4 x ~130 cycles in
// front of every tile's matrix instructions, in the right-hand sides' updates of both substitutions).
__device__ __forceinline__ double lds_where(bool cond, const double *p, int i_true)
{
    const double v = p[cond ? i_true : 0];
    return cond ? v : 0.0;
}

// ---- one 16 x 16 tile of K in the accumulator layout: e[i] of lane (c, g) = phi(|c_row - c_col|^2) (+ lambda on the diagonal) at
// row 16 I + g + 4 i, column 16 J + c; zero outside the M x M matrix.  Beside it the largest |element| and whether two different
// centres coincide (-> -5).
__device__ __forceinline__ void assemble_tile(const double *cen, int I, int J, int c, int g, int M, int kind, double lambda, double inv_r2,
                                              double (&e)[4], double &amax_w, bool &dup)
{
    const int col = 16 * J + c;
    const double cx = cen[3 * col], cy = cen[3 * col + 1], cz = cen[3 * col + 2];
    double d2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 16 * I + g + 4 * i;
        const double dx = cen[3 * row] - cx, dy = cen[3 * row + 1] - cy, dz = cen[3 * row + 2] - cz;
        d2[i] = dx * dx + dy * dy + dz * dz;
    }
    if (kind == FD_KERNEL_THIN_PLATE) {
        // (no branch around the logarithm: four independent chains the scheduler can interleave; d2 = 0 -- the
        // diagonal, padding, coincident centres -- goes through as 1 and is zeroed by the select)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool pos = d2[i] > 0.0;
            const double lg = log_pos(pos ? d2[i] : 1.0);
            e[i] = pos ? 0.5 * d2[i] * lg : 0.0;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) e[i] = phi_reg(kind, d2[i], inv_r2);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 16 * I + g + 4 * i;
        const bool real = row < M && col < M;
        if (row == col) e[i] += lambda;
        else if (d2[i] == 0.0 && real) dup = true;                 // coincident centres -> -5
        if (!real) e[i] = 0.0;
        const double ae = fabs(e[i]);
        amax_w = ae > amax_w ? ae : amax_w;
    }
}

// ---- the reflectors of P = [1 x y z] with f <- Q^T f folded in, the compact WY factor: ONE wave, four rows per lane, every sum a
// DPP reduction, no workgroup barrier.  In: L.cen (centres), L.F (the three right-hand sides), rows M .. 255 zero.  Out: L.V
// (reflectors), L.F (Q^T f with the pivot rows' shares aside in L.small[kG..]), L.small (tau, R, Tm), L.stat[3] (P rank-deficient).
__device__ __forceinline__ void reflect_wave(const RegLds &L, int M, int T, int lane)
{
    // ---- reflectors of P = [1 x y z] (dlarfg upside down: reflector k acts on rows 0 .. M-1-k, beta in row M-1-k) with
    // f <- Q^T f folded into the same sweep.  Each lane keeps its four rows (lane + 64 q) of [V | f] -- 28 doubles -- IN REGISTERS from
    // the first reflector to the last: a pivot row comes out by v_readlane, nothing else crosses lanes but the sums (DPP), and LDS
    // sees the rows once, at the end.  (Round 3 read and wrote them in LDS around every reflector, each `if (i < piv)` read an
    // exec-masked round trip: 24 k cycles -- hidden beside the assembly then, the critical path of k_reg_front1 now.  Same
    // operations in the same order: the same bits.)
    bool singular = false;
    double cn[4] = {0.0, 0.0, 0.0, 0.0};
    double v[4][4], f[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = lane + 64 * q;
        const bool in = i < M;
        const int ic = in ? i : 0;
        const double p[4] = {1.0, L.cen[3 * ic], L.cen[3 * ic + 1], L.cen[3 * ic + 2]};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            v[q][t] = (in && t < T) ? p[t] : 0.0;
            if (in) cn[t] = fma(v[q][t], v[q][t], cn[t]);
        }
#pragma unroll
        for (int e = 0; e < 3; ++e) { const double d = L.F[e * kRows + ic]; f[q][e] = in ? d : 0.0; }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) cn[t] = wave_sum(cn[t]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= T) break;
        const int piv = M - 1 - k;
        const int pl = __builtin_amdgcn_readfirstlane(piv & 63), pq = __builtin_amdgcn_readfirstlane(piv >> 6);
        // the pivot row as it stands: register set pq of lane pl
        // (a scalar branch on the register set, then plain v_readlanes: as a chain of selects in front of them the seven values
        //  cost 1 500 cycles of the reflector's 4 500)
        double rowv[4], rowf[3];
#define FD_PIVOT_ROW(Q) { _Pragma("unroll") for (int t = 0; t < 4; ++t) rowv[t] = readlane_f64(v[Q][t], pl); \
                          _Pragma("unroll") for (int e = 0; e < 3; ++e) rowf[e] = readlane_f64(f[Q][e], pl); }
        if (pq == 0) FD_PIVOT_ROW(0) else if (pq == 1) FD_PIVOT_ROW(1) else if (pq == 2) FD_PIVOT_ROW(2) else FD_PIVOT_ROW(3)
#undef FD_PIVOT_ROW
        // sigma, x . column c (c > k), x . f_c over the rows above the pivot
        double acc[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        double x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = lane + 64 * q;
            x[q] = i < piv ? v[q][k] : 0.0;
            if (i < piv) {
                acc[0] = fma(x[q], x[q], acc[0]);
#pragma unroll
                for (int cc = k + 1; cc < 4; ++cc) if (cc < T) acc[cc - k] = fma(x[q], v[q][cc], acc[cc - k]);
#pragma unroll
                for (int e = 0; e < 3; ++e) acc[4 + e] = fma(x[q], f[q][e], acc[4 + e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 7; ++e) acc[e] = wave_sum(acc[e]);
        const double xp = rowv[k];
        const double sigma = acc[0];
        const double norm = sqrt(fma(xp, xp, sigma));
        double beta = xp, tau = 0.0, scale = 0.0;
        if (sigma > 0.0) {
            beta = xp >= 0.0 ? -norm : norm;
            tau = (beta - xp) / beta;
            scale = 1.0 / (xp - beta);
        }
        if (!(norm > 64.0 * (double)M * kEps * sqrt(cn[k]))) singular = true;   // P has no full column rank (NaN too)
        double sc[4] = {0.0, 0.0, 0.0, 0.0}, prow[4] = {0.0, 0.0, 0.0, 0.0}, df[3];
#pragma unroll
        for (int cc = k + 1; cc < 4; ++cc) if (cc < T) { prow[cc] = rowv[cc]; sc[cc] = fma(scale, acc[cc - k], prow[cc]); }
#pragma unroll
        for (int e = 0; e < 3; ++e) df[e] = fma(scale, acc[4 + e], rowf[e]);       // v . f_e (v_piv = 1)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = lane + 64 * q;
            if (i < piv) {
                const double vv = x[q] * scale;
                v[q][k] = vv;
#pragma unroll
                for (int cc = k + 1; cc < 4; ++cc) if (cc < T) v[q][cc] = fma(-tau * sc[cc], vv, v[q][cc]);
#pragma unroll
                for (int e = 0; e < 3; ++e) f[q][e] = fma(-tau * df[e], vv, f[q][e]);
            } else if (i == piv) {
                // the pivot row: v = e_piv in column k, zero in the columns right of it; its share of Q^T f belongs to the
                // polynomial equations (aside, below) and is zero in the Cholesky's right-hand side
                v[q][k] = 1.0;
#pragma unroll
                for (int cc = k + 1; cc < 4; ++cc) if (cc < T) v[q][cc] = 0.0;
#pragma unroll
                for (int e = 0; e < 3; ++e) f[q][e] = 0.0;
            }
        }
        if (lane == 0) {
            L.small[kTau + k] = tau;
            L.small[kR + 4 * k + k] = beta;
#pragma unroll
            for (int cc = k + 1; cc < 4; ++cc) if (cc < T) L.small[kR + 4 * k + cc] = fma(-tau, sc[cc], prow[cc]);
#pragma unroll
            for (int e = 0; e < 3; ++e) L.small[kG + 3 * k + e] = fma(-tau, df[e], rowf[e]);
        }
    }
    // the rows go to LDS once (rows M .. 255 are zero in both arrays)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = lane + 64 * q;
#pragma unroll
        for (int t = 0; t < 4; ++t) L.V[4 * i + t] = v[q][t];
#pragma unroll
        for (int e = 0; e < 3; ++e) L.F[e * kRows + i] = f[q][e];
    }
    wave_lds_sync();
    // compact WY factor from the Gram matrix of V
    double gram[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = lane + 64 * q;
        if (i < M) {
            const double v0 = v[q][0], v1 = v[q][1], v2 = v[q][2], v3 = v[q][3];
            gram[0] = fma(v0, v1, gram[0]); gram[1] = fma(v0, v2, gram[1]); gram[2] = fma(v0, v3, gram[2]);
            gram[3] = fma(v1, v2, gram[3]); gram[4] = fma(v1, v3, gram[4]); gram[5] = fma(v2, v3, gram[5]);
        }
    }
#pragma unroll
    for (int e = 0; e < 6; ++e) gram[e] = wave_sum(gram[e]);
    {
        // (fully unrolled on purpose: a small matrix indexed by run-time loop counters lives in scratch memory)
        const double G[4][4] = {{0.0, gram[0], gram[1], gram[2]}, {0.0, 0.0, gram[3], gram[4]}, {0.0, 0.0, 0.0, gram[5]}, {0.0, 0.0, 0.0, 0.0}};
        double Tm[4][4] = {};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double tau = k < T ? L.small[kTau + k] : 0.0;
            Tm[k][k] = tau;
#pragma unroll
            for (int a = 0; a < k; ++a) {
                double v = 0.0;
#pragma unroll
                for (int b = a; b < k; ++b) v = fma(Tm[a][b], G[b][k], v);
                Tm[a][k] = -tau * v;
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) L.small[kTm + 4 * a + b] = Tm[a][b];
#pragma unroll
            for (int k = 0; k < 4; ++k) if (k >= T) L.small[kTau + k] = 0.0;
            L.stat[3] = singular ? 1.0 : 0.0;
        }
    }
}

// ---- G = Tm^T sym(V^T Y) Tm (4 x 4, symmetric) from L.V, L.W (= Y) and L.small[kTm..] into L.small[kGm..]: one wave
__device__ __forceinline__ void gram_wave(const RegLds &L, int M, int lane)
{
    double Sm[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) Sm[q] = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = lane + 64 * q;
        if (i < M) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) Sm[4 * a + b] = fma(L.V[4 * i + a], L.W[4 * i + b], Sm[4 * a + b]);
        }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) Sm[q] = wave_sum(Sm[q]);
    double Tm[16], ST[16], Gm[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) Tm[q] = L.small[kTm + q];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            double v = 0.0;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) v = fma(0.5 * (Sm[4 * a + cc] + Sm[4 * cc + a]), Tm[4 * cc + b], v);
            ST[4 * a + b] = v;
        }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            double v = 0.0;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) v = fma(Tm[4 * cc + a], ST[4 * cc + b], v);
            Gm[4 * a + b] = v;
        }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a + 1; b < 4; ++b) { const double m = 0.5 * (Gm[4 * a + b] + Gm[4 * b + a]); Gm[4 * a + b] = m; Gm[4 * b + a] = m; }
    if (lane < 16) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) v = lane == q ? Gm[q] : v;
        L.small[kGm + lane] = v;
    }
}

// ---- row i of W = Y Tm - (1/2) V G, in place over Y
__device__ __forceinline__ void w_row(const RegLds &L, int i)
{
    double y[4], v[4], w[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { y[t] = L.W[4 * i + t]; v[t] = L.V[4 * i + t]; }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        double z = 0.0, h = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a) { z = fma(y[a], L.small[kTm + 4 * a + b], z); h = fma(v[a], L.small[kGm + 4 * a + b], h); }
        w[b] = fma(-0.5, h, z);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) L.W[4 * i + t] = w[t];
}

// Which worker holds tile (I, J) of the lower triangle.  Column J is dealt round-robin from a per-column offset, and diagonal
// tile (J, J) sits with the tile left of it, (J, J - 1): in the factorisation that wave solves (J, J - 1) first, applies it
// to the diagonal tile at once and hands that to the factor wave, which then works beside everything else of the step.
// The offsets (three bits per column) balance the load: at 16 x 16 tiles the workers hold 19 or 20 each, and no more than
// kSlots for any smaller triangle.  Correctness does not rest on the co-location (see the hand-over in the step loop).
// Round 4: the wave w0(J) that owns (J + 1, J) -- and with it (J + 1, J + 1): the step's critical chain -- gets NO other tile of
// column J; the column's other tiles go round the six other waves.  In the panel phase that wave then has its two dependent
// chains and nothing behind them while the others solve their one or two tiles beside it (it had a second panel tile in eight of
// fifteen columns).  Offsets chosen by search (scored: tiles per wave <= 20; the trailing updates' per-SIMD maxima, waves w and
// w + 4 sharing a SIMD, summed over the steps: 196 against 203; the panel phases' longest wave: 32 against 39 tile-chains; the back
// substitution's longest wave per row: 39 against 43).
__device__ __forceinline__ int tile_owner(int I, int J)
{
    constexpr unsigned long long kOffsets = 0x8a3ad602e31bull;
    auto w0 = [](int K) { return (K + 1 + (int)((kOffsets >> (3 * K)) & 7ull)) % kWorkers; };
    if (I == J) return J == 0 ? 0 : w0(J - 1);
    const int j = I - J - 1;
    return j == 0 ? w0(J) : (w0(J) + 1 + (j - 1) % 6) % kWorkers;
}

// ---- one factorisation for a group of frames that share the rest rig (fd_batch_set_shared_factor; SURVEY 8e: "factor once and
// treat frames as extra right-hand sides").  The system matrix depends on the rest rig, the kernel and the term only
// (src/SOP_FaceDeform.cpp:331-363 rebuilds it every cook).  k_build_reg builds frame 0 as ever and, given `fac`, leaves behind what
// the other frames need: the Cholesky factor's tiles as they sit in the workers' registers ([column-major tile number][register]
// [lane]: 512 contiguous bytes per store), the inverted diagonal blocks, the reflectors, B21, the small matrices and the
// model's flags.  k_resolve_reg then runs ONE workgroup per remaining frame: Q^T f, both substitutions against the factor in L2
// (a column of tiles per step, the next one requested a step ahead), the polynomial, w = Q [y; 0], packing.
constexpr int kFacTiles = kMaxBlocks * (kMaxBlocks + 1) / 2;                         // 136
constexpr size_t kFacL = 0, kFacMinv = (size_t)kFacTiles * 256, kFacV = kFacMinv + (size_t)kMaxBlocks * kTileLds, kFacB21 = kFacV + 4 * kRows,
                 kFacSmall = kFacB21 + 4 * kRows, kFacMeta = kFacSmall + kSmallDoubles, kFacDoubles = kFacMeta + 8;
// meta: [0] amax, [1] coincident centres, [2] singular (pivot / rank of P / spin time-out), [3] smallest pivot, [4] largest pivot

// ---- the parallel front end (round 4).  Up to the projected matrix B = Q^T K Q nothing in the build is sequential: 136 tiles of
// phi, Y = K V and the rank-8 rotation are 70 of k_build_reg's 197 us on ONE CU -- chains of LDS round trips per tile -- while in
// the unpipelined cook 236 CUs idle.  Two short launches over (tiles, models) do that part; the factorisation, the substitutions, the
// recovery and the packing stay in the register-resident workgroup (k_build_reg<false>), which starts from the B tiles in L2.
//   k_reg_front1  (ceil(tiles / 7), 1, models) x 512 threads: control table; waves 0..6 ONE tile of K each -> the staging buffer
//                 (slot.A: [tile][register][lane], as ever) + the tile's largest |element| and coincidence flag; wave 7 the
//                 reflectors (every workgroup for itself: 4 us of latency, not of throughput); then the tile's two 16 x 4 shares
//                 of Y = K V on the matrix pipe -> front buffer.  Workgroup 0 also leaves V, Q^T f and the small matrices there.
//   k_reg_front2  (ceil(tiles / 8), 1, models) x 512 threads: Y summed in a fixed order (deterministic), G and W = Y Tm - V G / 2
//                 (every workgroup for itself), then a tile per wave: B = K - V W^T - W V^T in place, B21 aside, identity padding.
// The front buffer is the context's null-space scratch (slot.ns), which this path does not otherwise use.
constexpr int kFront1Tiles = kWorkers;     // tiles per workgroup of k_reg_front1 (wave 7 runs the reflectors)
constexpr int kFront2Tiles = kRegWaves;    // ... of k_reg_front2
constexpr size_t kFrV = 0, kFrF = 4 * kRows, kFrB21 = 7 * kRows, kFrSmall = 11 * kRows, kFrMeta = kFrSmall + kSmallDoubles,
                 kFrTile = kFrMeta + 8, kFrY = kFrTile + 2 * (size_t)(kMaxBlocks * (kMaxBlocks + 1) / 2);
// meta: [0] largest |element| of K, [1] coincident centres, [2] P rank-deficient
__host__ __device__ inline size_t reg_front_doubles(int M)
{
    const int nbk = (M + 15) / 16;
    return kFrY + (size_t)(nbk * (nbk + 1) / 2) * 128;
}
__device__ __forceinline__ int tile_number(int I, int J, int nbk) { return J * nbk - J * (J - 1) / 2 + (I - J); }     // column-major, lower triangle

constexpr int kFrontMaxTpw = 4;           // tiles per wave of the front-end launches: as many as keep the grid within the device's CUs
__global__ __launch_bounds__(kRegThreads) void k_reg_front1(const BatchSlot *tab, const PointSrc src, int use_src, int M, int T, int kind,
                                                             double lambda, double gauss_R, int tpw)
{
    const BatchSlot &slot = tab[blockIdx.z];
    __shared__ __attribute__((aligned(16))) double s_mem[10 * kRows + kSmallDoubles + 8 + kWorkers * kTileLds];
    RegLds L{};
    L.cen = s_mem; L.V = L.cen + 3 * kRows; L.F = L.V + 4 * kRows; L.small = L.F + 3 * kRows; L.stat = L.small + kSmallDoubles; L.scr = L.stat + 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int nbk = (M + 15) / 16, ntiles = nbk * (nbk + 1) / 2;
    const bool first = blockIdx.x == 0;
    {
        // control table (reference :268-287, widened to fp64); workgroup 0 leaves the context its copies
        const float *rest = use_src ? src.rest[blockIdx.z] : slot.rest;
        const float *delta = use_src ? src.delta[blockIdx.z] : slot.delta;
        for (int i = tid; i < M; i += kRegThreads) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float r = rest[3 * i + q], d = delta[3 * i + q];
                L.cen[3 * i + q] = (double)r;
                L.F[q * kRows + i] = (double)d;
                if (first) {
                    slot.centres[3 * i + q] = (double)r;
                    if (use_src) { slot.rest[3 * i + q] = r; slot.delta[3 * i + q] = d; }
                }
            }
            if (first) slot.radii[i] = gauss_R;
        }
        for (int e = tid; e < 4 * kRows; e += kRegThreads) {
            if (e >= 4 * M || T == 0) L.V[e] = 0.0;
            if (e < 3 * kRows && e % kRows >= M) L.F[e] = 0.0;
            if (e < 3 * kRows && e >= 3 * M) L.cen[e] = 0.0;
        }
        if (tid < kSmallDoubles) L.small[tid] = 0.0;
        if (tid == 0) L.stat[3] = 0.0;
    }
    __syncthreads();
    gdouble *fr = as_global(slot.ns);
    // worker wave w takes tiles (blockIdx.x * tpw + t) * 7 + w, t < tpw (the batch's launches keep their grid within the device's
    // CUs: a second workgroup on a CU shares its SIMDs with the first one's reflector wave, the launch's critical path)
    double e[kFrontMaxTpw][4];
    int tI[kFrontMaxTpw], tJ[kFrontMaxTpw];
    bool have[kFrontMaxTpw];
    if (wave < kWorkers) {
        gdouble *stage = as_global(slot.A);
        const double inv_r2 = 1.0 / (gauss_R * gauss_R);
#pragma unroll
        for (int t = 0; t < kFrontMaxTpw; ++t) {
            const int q = ((int)blockIdx.x * tpw + t) * kFront1Tiles + wave;
            have[t] = t < tpw && q < ntiles;
            tI[t] = tJ[t] = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) e[t][i] = 0.0;
            if (have[t]) {
                int r = q, J = 0;
                while (r >= nbk - J) { r -= nbk - J; ++J; }
                tI[t] = J + r; tJ[t] = J;
                double amax_w = 0.0;
                bool dup = false;
                assemble_tile(L.cen, tI[t], tJ[t], c, g, M, kind, lambda, inv_r2, e[t], amax_w, dup);
#pragma unroll
                for (int i = 0; i < 4; ++i) stage[((size_t)q * 4 + i) * 64 + lane] = e[t][i];
                amax_w = wave_max(amax_w);
                const bool any_dup = __any(dup);
                if (lane == 0) { fr[kFrTile + 2 * q] = amax_w; fr[kFrTile + 2 * q + 1] = any_dup ? 1.0 : 0.0; }
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < kFrontMaxTpw; ++t) have[t] = false;
        if (T > 0) reflect_wave(L, M, T, lane);
    }
    __threadfence_block();
    __syncthreads();
    if (T > 0) {
#pragma unroll
        for (int t = 0; t < kFrontMaxTpw; ++t) {
            if (!have[t]) continue;
            // the tile's shares of Y = K V.  In the accumulator layout it IS the A operand of K_IJ^T Z (slice s = rows 4 s .. 4 s + 3):
            // block J's share from block I; block I's share from block J needs the tile itself as the operand: through LDS.
            const int q = ((int)blockIdx.x * tpw + t) * kFront1Tiles + wave;
            const int I = tI[t], J = tJ[t];
            gdouble *yq = fr + kFrY + (size_t)q * 128;
            double *sb = L.scr + (size_t)wave * kTileLds;
            if (I != J) {
                wave_lds_sync();                     // (the previous tile's reads of the buffer)
#pragma unroll
                for (int i = 0; i < 4; ++i) sb[(g + 4 * i) * kPitch + c] = e[t][i];
            }
            {
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    const double b = lds_where(c < 4, L.V, 4 * (16 * I + 4 * s2 + g) + c);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(e[t][s2], b, acc, 0, 0, 0);
                }
                if (c < 4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) yq[(g + 4 * i) * 4 + c] = acc[i];
                }
            }
            if (I != J) {
                wave_lds_sync();
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    const double a = sb[c * kPitch + 4 * s2 + g];
                    const double b = lds_where(c < 4, L.V, 4 * (16 * J + 4 * s2 + g) + c);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                }
                if (c < 4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) yq[64 + (g + 4 * i) * 4 + c] = acc[i];
                }
            }
        }
    }
    if (first && wave == kWorkers) {
        for (int x = lane; x < 4 * kRows; x += 64) fr[kFrV + x] = L.V[x];
        for (int x = lane; x < 3 * kRows; x += 64) fr[kFrF + x] = L.F[x];
        for (int x = lane; x < kSmallDoubles; x += 64) fr[kFrSmall + x] = L.small[x];
        if (lane == 0) fr[kFrMeta + 2] = L.stat[3];
    }
}

__global__ __launch_bounds__(kRegThreads) void k_reg_front2(const BatchSlot *tab, int M, int T, int tpw)
{
    const BatchSlot &slot = tab[blockIdx.z];
    __shared__ __attribute__((aligned(16))) double s_mem[8 * kRows + kSmallDoubles];
    RegLds L{};
    L.V = s_mem; L.W = L.V + 4 * kRows; L.small = L.W + 4 * kRows;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int n1 = M - T;
    const int nbk = (M + 15) / 16, ntiles = nbk * (nbk + 1) / 2;
    gdouble *fr = as_global(slot.ns);
    for (int x = tid; x < 4 * kRows; x += kRegThreads) {
        L.V[x] = fr[kFrV + x];
        // Y, a fixed order of the tiles' shares per row: the tiles of tile row r left of the diagonal, the diagonal, those below it
        const int i = x >> 2, t = x & 3, r = i >> 4, row = i & 15;
        double y = 0.0;
        if (T > 0 && r < nbk) {
            // (all sixteen loads requested before the first add: one round trip to L2, not sixteen)
            gcdouble *yp = fr + kFrY + row * 4 + t;
            double v[kMaxBlocks];
#pragma unroll
            for (int k = 0; k < kMaxBlocks; ++k) {
                const int kk = k < nbk ? k : nbk - 1;
                const size_t at = kk < r ? (size_t)tile_number(r, kk, nbk) * 128 + 64 : (size_t)tile_number(kk, r, nbk) * 128;
                v[k] = yp[at];
            }
#pragma unroll
            for (int k = 0; k < kMaxBlocks; ++k) y += k < nbk ? v[k] : 0.0;
        }
        L.W[x] = y;
    }
    if (tid < kSmallDoubles) L.small[tid] = fr[kFrSmall + tid];
    if (blockIdx.x == 0 && wave == 1) {
        // the matrix' largest |element| and the coincidence flag from the tiles' (maxima: the order does not matter)
        double amax = 0.0, dup = 0.0;
        for (int q = lane; q < ntiles; q += 64) {
            const double a = fr[kFrTile + 2 * q], d = fr[kFrTile + 2 * q + 1];
            amax = a > amax ? a : amax; dup = d > dup ? d : dup;
        }
        amax = wave_max(amax); dup = wave_max(dup);
        if (lane == 0) { fr[kFrMeta + 0] = amax; fr[kFrMeta + 1] = dup; }
    }
    __syncthreads();
    if (T > 0) {
        if (wave == 0) gram_wave(L, M, lane);
        __syncthreads();
        if (tid < kRows) w_row(L, tid);
        __syncthreads();
    }
    gdouble *stage = as_global(slot.A);
    for (int t = 0; t < tpw; ++t) {
        const int q = ((int)blockIdx.x * tpw + t) * kFront2Tiles + wave;
        if (q >= ntiles) break;
        int I = 0, J = 0;
        {
            int r = q;
            while (r >= nbk - J) { r -= nbk - J; ++J; }
            I = J + r;
        }
        double4_t S;
#pragma unroll
        for (int i = 0; i < 4; ++i) S[i] = stage[((size_t)q * 4 + i) * 64 + lane];
        if (T > 0) {
            // B = K - V W^T - W V^T (two K = 4 matrix instructions)
            const int ri = 16 * I + c, rj = 16 * J + c;
            const double vi = L.V[4 * ri + g], wi = L.W[4 * ri + g];
            const double vj = L.V[4 * rj + g], wj = L.W[4 * rj + g];
            S = __builtin_amdgcn_mfma_f64_16x16x4f64(-vi, wj, S, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f64_16x16x4f64(-wi, vj, S, 0, 0, 0);
        }
        // B21 aside (pivot row M-1-k = equation of polynomial coefficient k), identity padding beyond n1
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 16 * I + g + 4 * i, col = 16 * J + c;
            if (row >= n1 && row < M && col < n1) fr[kFrB21 + 4 * col + (M - 1 - row)] = S[i];
            if (row >= n1 || col >= n1) S[i] = row == col ? 1.0 : 0.0;
            stage[((size_t)q * 4 + i) * 64 + lane] = S[i];
        }
    }
}

// FRONT: the whole build in this one workgroup (the round-3 form; the shared-factor group's first frame and hosts that ask for it);
// !FRONT: behind k_reg_front1 / k_reg_front2 -- the tiles of B, V, Q^T f, B21 and the small matrices come from L2.
template <bool FRONT>
__global__ __launch_bounds__(kRegThreads) void k_build_reg(const BatchSlot *tab, const PointSrc src, int use_src, int M, int T, int npad,
                                                            int kind, int Mpad, double lambda, double gauss_R, unsigned long long *stamps,
                                                            double *fac)
{
    const BatchSlot &slot = tab[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    const RegLds L = carve(dyn_lds, M);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool worker = wave < kWorkers;             // waves 0..6 hold the tiles; wave 7 runs the sequential pieces beside them
    const int c = lane & 15, g = lane >> 4;
    const int n1 = M - T;
    const int nbk = (M + 15) / 16;                   // tile rows / columns of K
    const int nb = (n1 + 15) / 16;                   // ... of the projected block that is factorised
    DevModel FD_GLOBAL *model = as_global(slot.model);
    unsigned long long st_prev = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    int st_k = 0;
#define FD_RSTAMP() if (stamps && blockIdx.z == 0 && tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamps[st_k] = t_ - st_prev; atomicAdd(&stamps[84 + st_k], t_ - st_prev); if (st_k == 0) atomicAdd(&stamps[83], 1ull); ++st_k; st_prev = t_; }
    __builtin_amdgcn_s_setprio(3);

    // ---- control table (reference :268-287, widened to fp64), status reset, tile table
    // Worker wave w lists its own tiles (tile_owner) in column-major order of the lower triangle: slot t of wave w is
    // L.tab[w * kSlots + t] = I | J << 8 | q << 16 (0xffff: none).  Ranks by ballot: three rounds of 64 tiles.
    const int ntiles = nbk * (nbk + 1) / 2;
    int nmine = 0;
    if (worker) {
        if (lane < kSlots) L.tab[wave * kSlots + lane] = 0xffff;
        wave_lds_sync();
        for (int q0 = 0; q0 < ntiles; q0 += 64) {
            int q = q0 + lane, J = 0;
            const bool in = q < ntiles;
            while (in && q >= nbk - J) { q -= nbk - J; ++J; }
            const int I = J + q;
            const bool mine = in && tile_owner(I, J) == wave;
            const unsigned long long mask = __ballot(mine);
            const int rank = nmine + __popcll(mask & ((1ull << lane) - 1ull));
            if (mine) L.tab[wave * kSlots + rank] = I | (J << 8) | ((q0 + lane) << 16);   // (+ its column-major number: staging index)
            nmine += __popcll(mask);
        }
        nmine = __builtin_amdgcn_readfirstlane(nmine);
    }
    if constexpr (!FRONT) {
        // V, Q^T f, B21 and the small matrices as the front end left them (k_reg_front1 / k_reg_front2)
        gcdouble *fr = as_global(slot.ns);
        // (a thread's two entries of each array requested together, stored afterwards: ONE round trip to L2 -- as a loop of
        //  load-and-store the second entry's loads waited behind the first's LDS stores)
        static_assert(4 * kRows == 2 * kRegThreads, "two entries per thread");
        double lv[2], lb[2], lf[2], ls[2], lc[2], ld[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int e = tid + kRegThreads * h;
            lv[h] = fr[kFrV + e];
            lb[h] = fr[kFrB21 + e];
            lf[h] = fr[kFrF + (e < 3 * kRows ? e : 0)];
            ls[h] = fr[kFrSmall + (e < kSmallDoubles ? e : 0)];
            // (for the packing, in the same round trip: the centres, and the deltas where the transposition buffers would be)
            lc[h] = slot.centres[e < 3 * M ? e : 0];
            ld[h] = (double)slot.delta[e < 3 * M ? e : 0];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int e = tid + kRegThreads * h;
            L.V[e] = lv[h];
            L.B21[e] = lb[h];
            if (e < 3 * kRows) L.F[e] = lf[h];
            if (e < kSmallDoubles) L.small[e] = ls[h];
            if (e < 3 * M) { L.cen[e] = lc[h]; L.scr[e] = ld[h]; }
        }
    } else {
        const float *rest = use_src ? src.rest[blockIdx.z] : slot.rest;
        const float *delta = use_src ? src.delta[blockIdx.z] : slot.delta;
        for (int i = tid; i < M; i += kRegThreads) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float r = rest[3 * i + q], d = delta[3 * i + q];
                L.cen[3 * i + q] = (double)r;
                L.F[q * kRows + i] = (double)d;
                slot.centres[3 * i + q] = (double)r;
                if (use_src) { slot.rest[3 * i + q] = r; slot.delta[3 * i + q] = d; }
            }
            slot.radii[i] = gauss_R;
        }
        // rows M .. 255 of the O(M) arrays read as zero: the tile phases address whole 16-row blocks without bounds checks
        for (int e = tid; e < 4 * kRows; e += kRegThreads) {
            if (e >= 4 * M) { L.V[e] = 0.0; L.W[e] = 0.0; L.B21[e] = 0.0; }
            if (e < 3 * kRows && e % kRows >= M) L.F[e] = 0.0;
            if (e < 3 * kRows && e >= 3 * M) L.cen[e] = 0.0;
        }
    }
    if (tid == 0) { L.stat[0] = INFINITY; L.stat[1] = 0.0; L.stat[2] = 0.0; L.stat[3] = FRONT ? 0.0 : as_global(slot.ns)[kFrMeta + 2]; L.flag[0] = 0; L.flag[1] = 0; L.flag[2] = 0; L.flag[3] = 0; }
    __syncthreads();
    // Lane t of `ijv` holds the coordinates of the wave's slot t (I | J << 8; 0xffff: no tile).  Which slots a phase touches
    // is ONE ballot over that register (a bit mask, then a bit test per slot); a slot's coordinates come out with one
    // v_readlane where its LDS addresses are formed.  (As scalar values they were spilled and re-read, compared and
    // branched on in every scan: five scans per factorisation step at ~500 cycles each.  Read through an opaque copy: with
    // the coordinates visibly loop-invariant the compiler hoists every tile's LDS addresses out of the step loop -- 40 more
    // live registers beside those of the tiles, and spills.)
    int ijv = 0xffff;
    if (worker && lane < kSlots) ijv = L.tab[wave * kSlots + lane];
    const int ivI = ijv & 0xff, ivJ = (ijv >> 8) & 0xff;
    auto slots_where = [&](bool cond) -> unsigned { return (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)__ballot(cond && lane < kSlots)); };
    const unsigned m_matrix = slots_where(ivI < kMaxBlocks);              // the wave's tiles
#define tI(t) (opaque_s(__builtin_amdgcn_readlane(ijv, t)) & 0xff)
#define tJ(t) ((opaque_s(__builtin_amdgcn_readlane(ijv, t)) >> 8) & 0xff)
#define FD_SLOT(m, t) (((m) >> (t)) & 1u)
    // (The scans of the step loop test the slots FOUR AT A TIME first.  A slot that is not in the mask costs a taken branch over its
    //  body -- ~35 cycles as measured: an instruction-fetch bubble each -- and a step runs eight scans of twenty: the waves with no
    //  tile in a panel spent 2 200 cycles of a 5 500-cycle panel phase skipping slots.  Masks are sparse -- a panel's tiles, a row's
    //  -- or a suffix of the column-major slot order, so most groups of four go in one branch.)
    FD_RSTAMP()

    if constexpr (FRONT) {
    if (worker) {
        // ---- K, tile by tile: element (16 I + g + 4 i, 16 J + c) = phi(|c_row - c_col|^2) (+ lambda on the diagonal).  A ROLLED
        // loop (one copy of the logarithm) that writes the tiles to a staging area -- the context's matrix buffer, which this
        // build does not otherwise use: [tile][register][lane], 512 contiguous bytes per store -- while no tile is live in
        // registers yet: with the registers of the resident tiles carried through this loop it spilled them around every
        // iteration.  The tiles come back from L2 by unconditional loads.
        gdouble *stage = as_global(slot.A);
        const double inv_r2 = 1.0 / (gauss_R * gauss_R);
        double amax_w = 0.0;
        bool dup = false;
#pragma nounroll
        for (int t = 0; t < nmine; ++t) {
            const int ij = __builtin_amdgcn_readfirstlane(L.tab[wave * kSlots + t]);
            const int I = ij & 0xff, J = (ij >> 8) & 0xff, q = ij >> 16;
            double e[4];
            assemble_tile(L.cen, I, J, c, g, M, kind, lambda, inv_r2, e, amax_w, dup);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                stage[((size_t)q * 4 + i) * 64 + lane] = e[i];
            }
        }
        amax_w = wave_max(amax_w);
        const bool any_dup = __any(dup);
        if (lane == 0) { L.ypart[wave] = amax_w; L.ypart[kRegWaves + wave] = any_dup ? 1.0 : 0.0; }      // (the overlay is free until Y = K V)
    } else if (T > 0) {
        reflect_wave(L, M, T, lane);
    }
    __threadfence_block();
    __syncthreads();
    }
    FD_RSTAMP()
    double amax = 0.0;
    bool dup_any = false;
    if constexpr (FRONT) {
#pragma unroll
        for (int w = 0; w < kWorkers; ++w) { amax = L.ypart[w] > amax ? L.ypart[w] : amax; dup_any = dup_any || L.ypart[kRegWaves + w] != 0.0; }
    } else {
        amax = as_global(slot.ns)[kFrMeta + 0];
        dup_any = as_global(slot.ns)[kFrMeta + 1] != 0.0;
    }
    double4_t S[kSlots];
    {
        gcdouble *stage = as_global(slot.A);
#pragma unroll
        for (int t = 0; t < kSlots; ++t) {
            const int q = __builtin_amdgcn_readlane(ijv, t) >> 16;
            S[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
            if (FD_SLOT(m_matrix, t)) {                      // (a wave reads back what it stored itself)
#pragma unroll
                for (int i = 0; i < 4; ++i) S[t][i] = stage[((size_t)q * 4 + i) * 64 + lane];
            }
        }
    }
    if constexpr (FRONT) {
    __syncthreads();                                 // (everybody has read the assembly's statistics out of the overlay)
    // per-wave partial sums of Y (every worker its own)
    double *yp = L.ypart + (size_t)wave * 4 * kRows;
    if (T > 0 && worker) for (int e = lane; e < 4 * kRows; e += 64) yp[e] = 0.0;
    wave_lds_sync();
    FD_RSTAMP()

    if (T > 0) {
        // ---- Y = K V from the tiles, into per-wave partial sums (LDS adds without return: fire and forget; one wave adds
        // to its own array in program order, and the arrays are summed in a fixed order: deterministic).  A tile in the
        // accumulator layout IS the A operand of K_IJ^T Z (slice i = rows 4 i .. 4 i + 3 of K_IJ): that gives block J of Y its
        // share from block I; the share of block I from block J needs K_IJ itself as the operand: through LDS, two buffers
        // in turn so that a tile's transposition does not wait for the previous one's reads.
        double *scr0 = L.scr + (size_t)wave * 2 * kTileLds;
#pragma unroll
        for (int t = 0; t < kSlots; ++t) {
            if (!FD_SLOT(m_matrix, t)) continue;
            const int I = tI(t), J = tJ(t);
            double *sb = scr0 + (t & 1) * kTileLds;
            if (I != J) {
#pragma unroll
                for (int i = 0; i < 4; ++i) sb[(g + 4 * i) * kPitch + c] = S[t][i];
            }
            {   // K_JI V_I -> rows of block J
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const double b = lds_where(c < 4, L.V, 4 * (16 * I + 4 * s + g) + c);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(S[t][s], b, acc, 0, 0, 0);
                }
                if (c < 4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        __hip_atomic_fetch_add(&yp[4 * (16 * J + g + 4 * i) + c], acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            if (I != J) {   // K_IJ V_J -> rows of block I
                wave_lds_sync();
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const double a = sb[c * kPitch + 4 * s + g];
                    const double b = lds_where(c < 4, L.V, 4 * (16 * J + 4 * s + g) + c);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                }
                if (c < 4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        __hip_atomic_fetch_add(&yp[4 * (16 * I + g + 4 * i) + c], acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        __syncthreads();
        for (int e = tid; e < 4 * kRows; e += kRegThreads) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < kWorkers; ++w) s += L.ypart[(size_t)w * 4 * kRows + e];
            L.W[e] = s;                             // Y for now
        }
        __syncthreads();
        FD_RSTAMP()

        // ---- G = Tm^T sym(V^T Y) Tm on wave 7; then W = Y Tm - (1/2) V G, a row per thread
        if (!worker) {
            gram_wave(L, M, lane);
        }
        __syncthreads();
        if (tid < kRows) w_row(L, tid);
        __syncthreads();

        // ---- B = K - V W^T - W V^T on the tiles in place (two K = 4 matrix instructions per tile)
#pragma unroll
        for (int t = 0; t < kSlots; ++t) {
            if (!FD_SLOT(m_matrix, t)) continue;
            const int ri = 16 * tI(t) + c, rj = 16 * tJ(t) + c;
            const double vi = L.V[4 * ri + g], wi = L.W[4 * ri + g];
            const double vj = L.V[4 * rj + g], wj = L.W[4 * rj + g];
            S[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-vi, wj, S[t], 0, 0, 0);
            S[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-wi, vj, S[t], 0, 0, 0);
        }
    }
    // ---- B21 aside (pivot row M-1-k = equation of polynomial coefficient k), identity padding beyond n1; the first diagonal
    //      tile to the factor wave's buffer
#pragma unroll
    for (int t = 0; t < kSlots; ++t) {
        if (FD_SLOT(m_matrix, t)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 16 * tI(t) + g + 4 * i, col = 16 * tJ(t) + c;
                if (row >= n1 && row < M && col < n1) L.B21[4 * col + (M - 1 - row)] = S[t][i];
                if (row >= n1 || col >= n1) S[t][i] = row == col ? 1.0 : 0.0;
            }
        }
    }
    }
    {
        const unsigned m_d0 = slots_where(ivI == 0 && ivJ == 0);
#pragma unroll
        for (int t = 0; t < kSlots; ++t) {
            if (FD_SLOT(m_d0, t)) {
#pragma unroll
                for (int i = 0; i < 4; ++i) L.D[(g + 4 * i) * kPitch + c] = S[t][i];
            }
        }
    }
    __syncthreads();                                 // the overlay (partial sums of Y) is dead: inverse blocks and panel buffer from here
    FD_RSTAMP()

    // ---- blocked Cholesky, 16 columns per step, with LOOK-AHEAD: wave 7 factorises and inverts diagonal block K while the
    // workers apply panel K - 1 to the rest of the matrix -- the owner of tile (K, K) updates that one first, hands it over
    // through LDS and raises a flag.  The three right-hand sides live in LDS (F, 3 x 256): wave 7 turns block K of them into
    // z_K = f_K inv(L_KK)^T once everything before has been applied, and a worker that holds panel tile (I, K) subtracts
    // z_K L_IK^T from block I when it applies that panel.
    const double tiny = (double)n1 * kEps * amax;
    double piv_min = INFINITY, piv_max = 0.0;         // (wave 7, per lane = per column of the block; reduced after the loop)
    bool piv_bad = false;
    // flag[0]: diagonal block handed to wave 7; [1]: z blocks written; [2]: panel tile (K+1, K) in LDS; [3]: workers past a step
#define FD_SPIN_UNTIL(cond) { int spin_ = 0; while (!(cond) && ++spin_ < (1 << 20)) __builtin_amdgcn_s_sleep(1); \
                              if (spin_ >= (1 << 20) && lane == 0) L.stat[2] = 1.0; __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
#define FD_FLAG(i) __hip_atomic_load(L.flag + (i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define FD_POST(i, v) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); if (lane == 0) __hip_atomic_store(L.flag + (i), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    for (int K = 0; K < nb; ++K) {
        if (worker) {
            if (K > 0) {
                const int Kp = K - 1;
                const unsigned m_upd = slots_where(ivJ > Kp && ivJ < nb && ivI < nb && !(ivI == K && ivJ == K));   // (the diagonal tile went ahead in the last step)
                const unsigned m_col = slots_where(ivJ == Kp && ivI > Kp && ivI < nb);       // my tiles of panel K - 1: their share of the right-hand sides
                // right-hand sides: f_I -= z_Kp L_I,Kp^T for my panel tiles (A = z_Kp as three rows, B = the panel tile from LDS)
                if (m_col) FD_SPIN_UNTIL(FD_FLAG(1) >= K)
#pragma unroll
                for (int tg_ = 0; tg_ < kSlots; tg_ += 4) if ((m_col >> tg_) & 0xfu)      // (four slots at a time: see FD_SLOT)
#pragma unroll
                for (int t = tg_; t < tg_ + 4; ++t) {
                    if (FD_SLOT(m_col, t)) {
                        const int I = tI(t);
                        const double *pb = L.P + (size_t)I * kTileLds;
                        double4_t acc = {0.0, 0.0, 0.0, 0.0};
                        double za[4], zb[4];
#pragma unroll
                        for (int s = 0; s < 4; ++s) { za[s] = lds_where(c < 3, L.Z, c * 256 + 16 * Kp + 4 * s + g); zb[s] = pb[c * kPitch + 4 * s + g]; }
                        // (every operand read of the tile in flight, ONE wait, then the matrix instructions: left to itself the scheduler
                        //  -- at the register limit here -- reads one operand, waits, issues, reads the next)
                        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#pragma unroll
                        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(za[s], zb[s], acc, 0, 0, 0);
                        if (g < 3) L.F[g * kRows + 16 * I + c] -= acc[0];
                    }
                }
                // the trailing matrix: C_IJ -= L_I,Kp L_J,Kp^T.  (Fetching the operands of slot t + 1 before the matrix instructions
                // of slot t -- two register sets -- was tried: 34 more spilled registers, and every step slower, 198k -> 225k cycles.)
#pragma unroll
                for (int t = 0; t < kSlots; ++t) {      // (a dense mask: every slot tested on its own)
                    if (FD_SLOT(m_upd, t)) {
                        const double *pa = L.P + (size_t)tI(t) * kTileLds, *pb = L.P + (size_t)tJ(t) * kTileLds;
                        double ua[4], ub[4];
#pragma unroll
                        for (int s = 0; s < 4; ++s) { ua[s] = -pa[c * kPitch + 4 * s + g]; ub[s] = pb[c * kPitch + 4 * s + g]; }
                        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#pragma unroll
                        for (int s = 0; s < 4; ++s) S[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[s], ub[s], S[t], 0, 0, 0);
                    }
                }
            }
        } else {
            if (K > 0) FD_SPIN_UNTIL(FD_FLAG(0) >= K)
            double4_t Td;
#pragma unroll
            for (int i = 0; i < 4; ++i) Td[i] = L.D[(g + 4 * i) * kPitch + c];
            double piv;
            const double4_t U = factor_invert_tile(Td, lane, piv);
            double *dst = L.minv + (size_t)K * kTileLds;
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[(g + 4 * i) * kPitch + c] = U[i];
            const bool live = 16 * K + c < n1;
            const double ad = fabs(piv);
            piv_bad = piv_bad || (live && !(piv > tiny));
            piv_min = (live && ad < piv_min) ? ad : piv_min;
            piv_max = (live && ad > piv_max) ? ad : piv_max;
        }
        __syncthreads();                              // inverse K is in LDS; panel K - 1 has been applied everywhere
        if (stamps && blockIdx.z == 0 && tid == 0) stamps[16 + 2 * K] = __builtin_amdgcn_s_memtime() - st_prev;
        if (worker) {
            // the tiles below the diagonal block: L_IK = C_IK inv(L_KK)^T, through the tile's own slot of the panel buffer.
            // Tile (K + 1, K) first and alone; its wave applies it to diagonal tile K + 1 straight away and hands that over:
            // wave 7 factorises block K + 1 beside the rest of this step and the updates of the next.
            const bool more = K + 1 < nb;
            const unsigned m_crit = slots_where(more && ivJ == K && ivI == K + 1);
            const unsigned m_next = slots_where(more && ivJ == K + 1 && ivI == K + 1);
            const unsigned m_panel = slots_where(ivJ == K && ivI > K + 1 && ivI < nb);
            const double *mk = L.minv + (size_t)K * kTileLds;
            double bop[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) bop[s] = mk[c * kPitch + 4 * s + g];         // B[k][n] = inv[n][4 s + k]
            if (m_crit) {
                double *pk = L.P + (size_t)(K + 1) * kTileLds;
#pragma unroll
                for (int tg_ = 0; tg_ < kSlots; tg_ += 4) if ((m_crit >> tg_) & 0xfu)      // (four slots at a time: see FD_SLOT)
#pragma unroll
                for (int t = tg_; t < tg_ + 4; ++t) {
                    if (FD_SLOT(m_crit, t)) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) pk[(g + 4 * i) * kPitch + c] = S[t][i];
                        wave_lds_sync();
                        double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pk[c * kPitch + 4 * s + g], bop[s], acc, 0, 0, 0);
                        S[t] = acc;
                        wave_lds_sync();
#pragma unroll
                        for (int i = 0; i < 4; ++i) pk[(g + 4 * i) * kPitch + c] = acc[i];
                    }
                }
                FD_POST(2, K + 1)
            }
            if (m_next) {
                // (the same wave by tile_owner's construction, and then the flag is already up; any other map waits here)
                FD_SPIN_UNTIL(FD_FLAG(2) >= K + 1)
                const double *pa = L.P + (size_t)(K + 1) * kTileLds;
#pragma unroll
                for (int tg_ = 0; tg_ < kSlots; tg_ += 4) if ((m_next >> tg_) & 0xfu)      // (four slots at a time: see FD_SLOT)
#pragma unroll
                for (int t = tg_; t < tg_ + 4; ++t) {
                    if (FD_SLOT(m_next, t)) {
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            const double a = pa[c * kPitch + 4 * s + g];
                            S[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a, a, S[t], 0, 0, 0);
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) L.D[(g + 4 * i) * kPitch + c] = S[t][i];
                    }
                }
                FD_POST(0, K + 1)
            }
#pragma unroll
            for (int tg_ = 0; tg_ < kSlots; tg_ += 4) if ((m_panel >> tg_) & 0xfu)      // (four slots at a time: see FD_SLOT)
#pragma unroll
            for (int t = tg_; t < tg_ + 4; ++t) {
                if (FD_SLOT(m_panel, t)) {
                    double *dst = L.P + (size_t)tI(t) * kTileLds;
#pragma unroll
                    for (int i = 0; i < 4; ++i) dst[(g + 4 * i) * kPitch + c] = S[t][i];
                }
            }
            wave_lds_sync();
#pragma unroll
            for (int tg_ = 0; tg_ < kSlots; tg_ += 4) if ((m_panel >> tg_) & 0xfu)      // (four slots at a time: see FD_SLOT)
#pragma unroll
            for (int t = tg_; t < tg_ + 4; ++t) {
                if (FD_SLOT(m_panel, t)) {
                    const double *sb = L.P + (size_t)tI(t) * kTileLds;
                    double4_t acc = {0.0, 0.0, 0.0, 0.0};
                    double sa[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) sa[s] = sb[c * kPitch + 4 * s + g];
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[s], bop[s], acc, 0, 0, 0);
                    S[t] = acc;
                }
            }
            wave_lds_sync();
#pragma unroll
            for (int tg_ = 0; tg_ < kSlots; tg_ += 4) if ((m_panel >> tg_) & 0xfu)      // (four slots at a time: see FD_SLOT)
#pragma unroll
            for (int t = tg_; t < tg_ + 4; ++t) {
                if (FD_SLOT(m_panel, t)) {
                    double *dst = L.P + (size_t)tI(t) * kTileLds;
#pragma unroll
                    for (int i = 0; i < 4; ++i) dst[(g + 4 * i) * kPitch + c] = S[t][i];
                }
            }
            // the workers meet (wave 7 is already on the next block): panel K is in LDS for everybody
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_fetch_add(L.flag + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            FD_SPIN_UNTIL(FD_FLAG(3) >= kWorkers * (K + 1))
        } else {
            // z_K = f_K inv(L_KK)^T: lanes = (right-hand side, column)
            if (lane < 48) {
                const int rhs = lane >> 4, n = lane & 15;
                const double *mk = L.minv + (size_t)K * kTileLds;
                double z = 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k) z = fma(L.F[rhs * kRows + 16 * K + k], mk[n * kPitch + k], z);
                L.Z[rhs * 256 + 16 * K + n] = z;
            }
            FD_POST(1, K + 1)
        }
        if (stamps && blockIdx.z == 0 && tid == 0) stamps[17 + 2 * K] = __builtin_amdgcn_s_memtime() - st_prev;
    }
    if (!worker) {
        piv_min = -wave_max(-piv_min); piv_max = wave_max(piv_max);
        const bool bad = __any(piv_bad);
        if (lane == 0) {
            L.stat[0] = piv_min; L.stat[1] = piv_max;
            if (bad) L.stat[2] = 1.0;
        }
    }
    FD_RSTAMP()

    // ---- y^T L = z^T, bottom up, right-looking: one row of tiles per step (a tile is its own B operand).  The wave that owns
    // tile (I, I - 1) subtracts its share from block I - 1 and solves that block right away -- every other contribution to it
    // came before an earlier barrier -- so a step costs one barrier.
    auto solve_block = [&](int I) {                   // Y_I = Z_I inv(L_II): lanes 0..47 = (right-hand side, column)
        if (lane < 48) {
            const int rhs = lane >> 4, n = lane & 15;
            const double *mi = L.minv + (size_t)I * kTileLds;
            double y = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) y = fma(L.Z[rhs * 256 + 16 * I + k], mi[k * kPitch + n], y);
            L.Y[rhs * 256 + 16 * I + n] = y;
        }
    };
    if (!worker && nb > 0) solve_block(nb - 1);
    __syncthreads();
    if (fac != nullptr && blockIdx.z == 0) {
        // the factor and its company for the group's other frames (k_resolve_reg); everything here is final: the tiles hold L,
        // minv the inverted diagonal blocks, stat the pivot statistics
        gdouble *fg = as_global(fac);
#pragma unroll
        for (int t = 0; t < kSlots; ++t) {
            if (FD_SLOT(m_matrix, t)) {
                const int q = __builtin_amdgcn_readlane(ijv, t) >> 16;
#pragma unroll
                for (int i = 0; i < 4; ++i) fg[kFacL + ((size_t)q * 4 + i) * 64 + lane] = S[t][i];
            }
        }
        for (int e = tid; e < kMaxBlocks * kTileLds; e += kRegThreads) fg[kFacMinv + e] = e < nb * kTileLds ? L.minv[e] : 0.0;
        for (int e = tid; e < 4 * kRows; e += kRegThreads) { fg[kFacV + e] = L.V[e]; fg[kFacB21 + e] = L.B21[e]; }
        if (tid < kSmallDoubles) fg[kFacSmall + tid] = L.small[tid];
        if (tid == 0) {
            fg[kFacMeta + 0] = amax; fg[kFacMeta + 1] = dup_any ? 1.0 : 0.0;
            fg[kFacMeta + 2] = (L.stat[2] != 0.0 || L.stat[3] != 0.0) ? 1.0 : 0.0;
            fg[kFacMeta + 3] = L.stat[0]; fg[kFacMeta + 4] = L.stat[1];
        }
    }
    for (int I = nb - 1; I >= 1; --I) {
        const unsigned m_row = slots_where(ivI == I && ivJ < I);
        const bool next_mine = slots_where(ivI == I && ivJ == I - 1) != 0u;
#pragma unroll
        for (int t = 0; t < kSlots; ++t) {      // (a dense mask: every slot tested on its own)
            if (FD_SLOT(m_row, t)) {
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
                double ya[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) ya[s] = lds_where(c < 3, L.Y, c * 256 + 16 * I + 4 * s + g);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[s], S[t][s], acc, 0, 0, 0);
                if (g < 3) L.Z[g * 256 + 16 * tJ(t) + c] -= acc[0];
            }
        }
        if (next_mine) { wave_lds_sync(); solve_block(I - 1); }
        __syncthreads();
    }
    FD_RSTAMP()

    // ---- R a = g - B21 y;  w = Q [y; 0] = H_0 .. H_{T-1} [y; 0]: wave 7 again
    if (!worker) {
        double q12[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) q12[e] = 0.0;
        double xr[4][3];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = lane + 64 * q;
            const bool in = i < n1;
            xr[q][0] = in ? L.Y[i] : 0.0; xr[q][1] = in ? L.Y[256 + i] : 0.0; xr[q][2] = in ? L.Y[512 + i] : 0.0;
            if (in && T > 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double b = k < T ? L.B21[4 * i + k] : 0.0;
                    q12[3 * k] = fma(b, xr[q][0], q12[3 * k]); q12[3 * k + 1] = fma(b, xr[q][1], q12[3 * k + 1]); q12[3 * k + 2] = fma(b, xr[q][2], q12[3 * k + 2]);
                }
            }
        }
        if (T > 0) {
#pragma unroll
            for (int e = 0; e < 12; ++e) q12[e] = wave_sum(q12[e]);
        }
        // R a = g - B21 y: upper triangular in (k, c); every lane the same values
        double a[4][3] = {};
#pragma unroll
        for (int k = 3; k >= 0; --k) {
            if (k >= T) continue;
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                double v = L.small[kG + 3 * k + e] - q12[3 * k + e];
#pragma unroll
                for (int c2 = k + 1; c2 < 4; ++c2) if (c2 < T) v = fma(-L.small[kR + 4 * k + c2], a[c2][e], v);
                a[k][e] = v / L.small[kR + 4 * k + k];
            }
        }
#pragma unroll
        for (int k = 3; k >= 0; --k) {
            if (k >= T) continue;
            const int piv = M - 1 - k;
            const double tau = L.small[kTau + k];
            double d[3] = {0.0, 0.0, 0.0}, v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = lane + 64 * q;
                v[q] = i <= piv ? L.V[4 * i + k] : 0.0;
#pragma unroll
                for (int e = 0; e < 3; ++e) d[e] = fma(v[q], xr[q][e], d[e]);
            }
#pragma unroll
            for (int e = 0; e < 3; ++e) d[e] = wave_sum(d[e]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 3; ++e) xr[q][e] = fma(-tau * d[e], v[q], xr[q][e]);
        }
        gdouble *X = as_global(slot.X);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = lane + 64 * q;
            if (i < M) { X[i] = xr[q][0]; X[(size_t)npad + i] = xr[q][1]; X[2 * (size_t)npad + i] = xr[q][2]; }
            // ... and where the packing reads it (y itself is used up: every lane has its rows in registers)
            if (i < M) { L.Y[i] = xr[q][0]; L.Y[256 + i] = xr[q][1]; L.Y[512 + i] = xr[q][2]; }
        }
        if (lane < 12) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 3; ++e) v = lane == 3 * k + e ? a[k][e] : v;
            L.small[kAff + lane] = v;
        }
        for (int r = M + lane; r < npad; r += 64) {
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                double v = 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k) v = (r - M == k && k < T) ? a[k][e] : v;
                X[(size_t)e * npad + r] = v;
            }
        }
        if (lane == 0) {
            model->terminationtype = 0;
            model->dup_flag = dup_any ? 1 : 0;
            model->sing_flag = (L.stat[2] != 0.0 || L.stat[3] != 0.0) ? 1 : 0;
            L.stat[4] = (L.stat[2] != 0.0 || L.stat[3] != 0.0) ? 1.0 : 0.0;
            L.stat[5] = dup_any ? 1.0 : 0.0;
            model->iterations = M + T;
            model->amax_bits = (unsigned long long)__double_as_longlong(amax);
            model->pivmin_bits = (unsigned long long)__double_as_longlong(L.stat[0]);
            model->pivmax_bits = (unsigned long long)__double_as_longlong(L.stat[1]);
        }
    }
    __threadfence_block();
    __syncthreads();
    FD_RSTAMP()
    // ---- evaluation records, centre tiles, status: the packing code every build shares (written for 256 threads; the other
    // four waves are done -- a barrier counts the waves that have not ended)
    if (tid >= 256) return;
    {
        const PackLds in{L.cen, FRONT ? nullptr : L.scr, L.Y, L.small + kAff, slot.delta, gauss_R, L.stat[4] != 0.0 ? 1 : 0, L.stat[5] != 0.0 ? 1 : 0};
        packing::pack_body_from(in, slot, npad, M, Mpad, T, kind, 0, 0);
    }
    FD_RSTAMP()
    if (kind == FD_KERNEL_THIN_PLATE) {
        __syncthreads();
        for (int tile = tid >> 6; tile < Mpad / 16; tile += 4) packing::pack_tiles_body(slot, Mpad, tile, lane);
    }
    FD_RSTAMP()
#undef FD_RSTAMP
#undef tI
#undef tJ
#undef FD_SLOT
}

// ---- frames 1 .. n - 1 of a shared-factor group: one workgroup of 256 threads per frame.  Thread = row (forward) or column
// (backward) of the system; the three right-hand sides of the frame ride in LDS.  tab + 1 + blockIdx.z is the frame's slot.
__global__ __launch_bounds__(256) void k_resolve_reg(const BatchSlot *tab, const PointSrc src, int use_src, int M, int T, int npad, int kind,
                                                      int Mpad, double gauss_R, const double *fac)
{
    const int fi = 1 + (int)blockIdx.z;
    const BatchSlot &slot = tab[fi];
    __shared__ double s_minv[kMaxBlocks * kTileLds];          // 34.8 KB
    __shared__ double s_V[4 * kRows], s_B21[4 * kRows], s_small[kSmallDoubles];
    __shared__ double s_f[3 * kRows];                         // f, then z, then y: in place
    __shared__ double s_blk[3 * 16];                          // the block just solved
    __shared__ double s_red[4 * 12];
    __shared__ double s_g[12], s_a[12];
    const int tid = threadIdx.x, lane = tid & 63;
    const int n1 = M - T, nb = (n1 + 15) / 16, nbk = (M + 15) / 16;
    gcdouble *fg = as_global(fac);
    DevModel FD_GLOBAL *model = as_global(slot.model);
    // column-major number of tile (I, J) of the lower triangle of an nbk x nbk tile grid (k_build_reg's staging index)
    auto tile_q = [&](int I, int J) { return J * nbk - J * (J - 1) / 2 + (I - J); };
    for (int e = tid; e < kMaxBlocks * kTileLds; e += 256) s_minv[e] = fg[kFacMinv + e];
    for (int e = tid; e < 4 * kRows; e += 256) { s_V[e] = fg[kFacV + e]; s_B21[e] = fg[kFacB21 + e]; }
    if (tid < kSmallDoubles) s_small[tid] = fg[kFacSmall + tid];
    {
        // control table of this frame (reference :268-287, widened to fp64); the centres are frame 0's (one rest rig)
        const float *rest = use_src ? src.rest[fi] : slot.rest;
        const float *delta = use_src ? src.delta[fi] : slot.delta;
        for (int i = tid; i < kRows; i += 256) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                double d = 0.0;
                if (i < M) {
                    const float r = rest[3 * i + q], dl = delta[3 * i + q];
                    d = (double)dl;
                    slot.centres[3 * i + q] = (double)r;
                    if (use_src) { slot.rest[3 * i + q] = r; slot.delta[3 * i + q] = dl; }
                }
                s_f[q * kRows + i] = d;
            }
            if (i < M) slot.radii[i] = gauss_R;
        }
    }
    __syncthreads();
    // ---- f <- Q^T f = H_{T-1} .. H_0 f; the pivot rows' shares g_k go to the polynomial equations (k_build_reg folds this into
    // the reflectors' own sweep; v_k is column k of V, 1 at its pivot row M-1-k and 0 below it)
    const int i0 = tid;                                        // this thread's row (kRows == 256)
    for (int k = 0; k < T; ++k) {
        const int piv = M - 1 - k;
        const double v = s_V[4 * i0 + k];
        double d[3] = {v * s_f[i0], v * s_f[kRows + i0], v * s_f[2 * kRows + i0]};
        packing::block_reduce_many<3, false>(d, s_red, tid);
        const double tau = s_small[kTau + k];
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const double fv = fma(-tau * d[e], v, s_f[e * kRows + i0]);
            if (i0 == piv) { s_g[3 * k + e] = fv; s_f[e * kRows + i0] = 0.0; }
            else if (i0 < piv) s_f[e * kRows + i0] = fv;
        }
        __syncthreads();
    }
    // ---- forward: L z = f, right-looking, a column of tiles per step.  Thread = row r: its 16 entries of column block K are 128
    // contiguous bytes of the tile image (register r % 16 / 4, lanes 16 (r % 4) .. + 15); the next column's are requested before
    // this column's are used.
    {
        const int r = tid, I = r >> 4, rr = r & 15;
        double cur[16], nxt[16];
        auto fetch = [&](int K, double (&dst)[16]) {
            if (I > K && I < nb) {
                gcdouble *p = fg + kFacL + ((size_t)tile_q(I, K) * 4 + (rr >> 2)) * 64 + 16 * (rr & 3);
#pragma unroll
                for (int k = 0; k < 16; ++k) dst[k] = p[k];
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) dst[k] = 0.0;
            }
        };
        fetch(0, cur);
        for (int K = 0; K < nb; ++K) {
            if (K + 1 < nb) fetch(K + 1, nxt);
            if (tid < 48) {                                    // z_K = inv(L_KK) f_K
                const int e = tid >> 4, n = tid & 15;
                const double *mk = s_minv + (size_t)K * kTileLds;
                double z = 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k) z = fma(s_f[e * kRows + 16 * K + k], mk[n * kPitch + k], z);
                s_blk[e * 16 + n] = z;
            }
            __syncthreads();
            if (tid < 48) s_f[(tid >> 4) * kRows + 16 * K + (tid & 15)] = s_blk[tid];
            if (I > K && I < nb) {
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    double acc = s_f[e * kRows + r];
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc = fma(-cur[k], s_blk[e * 16 + k], acc);
                    s_f[e * kRows + r] = acc;
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; ++k) cur[k] = nxt[k];
        }
    }
    // ---- backward: y^T L = z^T, bottom up.  Thread = column c of block J: for row block I > J it needs column c of tile (I, J).
    {
        const int cidx = tid, J = cidx >> 4, cc = cidx & 15;
        double cur[16], nxt[16];
        auto fetch = [&](int I, double (&dst)[16]) {
            if (I > J && I < nb) {
                gcdouble *p = fg + kFacL + (size_t)tile_q(I, J) * 256 + cc;
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) dst[rr] = p[(rr >> 2) * 64 + 16 * (rr & 3)];      // element (rr, cc)
            } else {
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) dst[rr] = 0.0;
            }
        };
        if (nb > 0) fetch(nb - 1, cur);
        for (int I = nb - 1; I >= 0; --I) {
            if (I > 0) fetch(I - 1, nxt);
            if (tid < 48) {                                    // y_I = inv(L_II)^T z_I
                const int e = tid >> 4, n = tid & 15;
                const double *mi = s_minv + (size_t)I * kTileLds;
                double y = 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k) y = fma(s_f[e * kRows + 16 * I + k], mi[k * kPitch + n], y);
                s_blk[e * 16 + n] = y;
            }
            __syncthreads();
            if (tid < 48) s_f[(tid >> 4) * kRows + 16 * I + (tid & 15)] = s_blk[tid];
            if (J < I) {
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    double acc = s_f[e * kRows + cidx];
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr) acc = fma(-cur[rr], s_blk[e * 16 + rr], acc);
                    s_f[e * kRows + cidx] = acc;
                }
            }
            __syncthreads();
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) cur[rr] = nxt[rr];
        }
    }
    // ---- R a = g - B21 y;  w = Q [y; 0] = H_0 .. H_{T-1} [y; 0]
    double xr[3];
    {
        const bool in = i0 < n1;
#pragma unroll
        for (int e = 0; e < 3; ++e) xr[e] = in ? s_f[e * kRows + i0] : 0.0;
        double q12[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double b = (in && k < T) ? s_B21[4 * i0 + k] : 0.0;
#pragma unroll
            for (int e = 0; e < 3; ++e) q12[3 * k + e] = b * xr[e];
        }
        if (T > 0) packing::block_reduce_many<12, false>(q12, s_red, tid);
        if (tid == 0) {
            double a[4][3] = {};
#pragma unroll
            for (int k = 3; k >= 0; --k) {
                if (k >= T) continue;
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    double v = s_g[3 * k + e] - q12[3 * k + e];
#pragma unroll
                    for (int c2 = k + 1; c2 < 4; ++c2) if (c2 < T) v = fma(-s_small[kR + 4 * k + c2], a[c2][e], v);
                    a[k][e] = v / s_small[kR + 4 * k + k];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 3; ++e) s_a[3 * k + e] = a[k][e];
        }
        for (int k = T - 1; k >= 0; --k) {
            const int piv = M - 1 - k;
            const double v = i0 <= piv ? s_V[4 * i0 + k] : 0.0;
            double d[3] = {v * xr[0], v * xr[1], v * xr[2]};
            packing::block_reduce_many<3, false>(d, s_red, tid);
            const double tau = s_small[kTau + k];
#pragma unroll
            for (int e = 0; e < 3; ++e) xr[e] = fma(-tau * d[e], v, xr[e]);
        }
        __syncthreads();
        gdouble *X = as_global(slot.X);
        if (i0 < M) { X[i0] = xr[0]; X[(size_t)npad + i0] = xr[1]; X[2 * (size_t)npad + i0] = xr[2]; }
        for (int rrow = M + tid; rrow < npad; rrow += 256) {
#pragma unroll
            for (int e = 0; e < 3; ++e) X[(size_t)e * npad + rrow] = (rrow - M < T) ? s_a[3 * (rrow - M) + e] : 0.0;
        }
        if (tid == 0) {
            model->terminationtype = 0;
            model->dup_flag = fg[kFacMeta + 1] != 0.0 ? 1 : 0;
            model->sing_flag = fg[kFacMeta + 2] != 0.0 ? 1 : 0;
            model->iterations = M + T;
            model->amax_bits = (unsigned long long)__double_as_longlong(fg[kFacMeta + 0]);
            model->pivmin_bits = (unsigned long long)__double_as_longlong(fg[kFacMeta + 3]);
            model->pivmax_bits = (unsigned long long)__double_as_longlong(fg[kFacMeta + 4]);
        }
    }
    __threadfence_block();
    __syncthreads();
    packing::pack_body(slot, npad, M, Mpad, T, kind, 0, 0);
    if (kind == FD_KERNEL_THIN_PLATE) {
        __syncthreads();
        for (int tile = tid >> 6; tile < Mpad / 16; tile += 4) packing::pack_tiles_body(slot, Mpad, tile, lane);
    }
}

}  // namespace

size_t reg_factor_doubles() { return kFacDoubles; }

bool reg_applicable(int kind, int term, double lambda, int M)
{
    return M <= 16 * kMaxBlocks && spd_applicable(kind, term, lambda, M);
}

// Once per device, OUTSIDE any stream capture (the C ABI calls it before it enqueues a build): the kernel's dynamic LDS limit
// is a per-device property.  (The packing code has a little static LDS of its own: the dynamic limit leaves room for it.)
hipError_t reg_build_init()
{
    static bool done[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && done[dev]) return hipSuccess;
    e = hipFuncSetAttribute((const void *)k_build_reg<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(sizeof(double) * reg_lds_doubles(16 * kMaxBlocks)));
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)k_build_reg<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(sizeof(double) * reg_lds_doubles(16 * kMaxBlocks)));
    if (e == hipSuccess && dev >= 0 && dev < 64) done[dev] = true;
    return e;
}

// The WHOLE build, control table included: one launch.  src == nullptr: the contexts' own copies of the control points.
// One factorisation for the whole batch (the contexts share rest rig, kernel and term: the caller has checked): frame 0 through
// k_build_reg, which leaves the factor in `fac` (reg_factor_doubles() doubles of device scratch), the others through k_resolve_reg.
hipError_t launch_build_reg_shared(const BuildBuffers &b, hipStream_t stream, const PointSrc *src, hipEvent_t ev_mid, double *fac)
{
    static const PointSrc none{};
    if (ev_mid) (void)hipEventRecord(ev_mid, stream);
    const size_t lds = sizeof(double) * reg_lds_doubles(b.M);
    hipLaunchKernelGGL(k_build_reg<true>, dim3(1, 1, 1), dim3(kRegThreads), lds, stream, b.d_slots, src ? *src : none, src ? 1 : 0, b.M, b.T,
                       b.npad, b.kind, b.Mpad, b.lambda, b.gauss_R, (unsigned long long *)nullptr, fac);
    if (b.nbatch > 1)
        hipLaunchKernelGGL(k_resolve_reg, dim3(1, 1, (unsigned)b.nbatch - 1), dim3(256), 0, stream, b.d_slots, src ? *src : none, src ? 1 : 0,
                           b.M, b.T, b.npad, b.kind, b.Mpad, b.gauss_R, (const double *)fac);
    return hipGetLastError();
}

hipError_t launch_build_reg(const BuildBuffers &b, hipStream_t stream, const PointSrc *src, hipEvent_t ev_mid)
{
    static const PointSrc none{};
    const unsigned nbatch = (unsigned)b.nbatch;
    if (ev_mid) (void)hipEventRecord(ev_mid, stream);
    const size_t lds = sizeof(double) * reg_lds_doubles(b.M);
    // (diagnostics: phase stamps, only outside stream capture -- FD_REG_STAMPS=1)
    static unsigned long long *d_stamps = nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    static const bool stamps_env = tuning_env("FD_REG_STAMPS") != nullptr;
    static const bool stamps_late = stamps_env && atoi(tuning_env("FD_REG_STAMPS")) == 2;      // 2: no read-back per launch (the pipeline stays a pipeline); the last launch's stamps at exit
    const bool want_stamps = stamps_env && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone;
    if (want_stamps && !d_stamps) { (void)hipMalloc((void **)&d_stamps, 96 * sizeof(unsigned long long)); (void)hipMemset(d_stamps, 0, 96 * sizeof(unsigned long long)); }
    // The front end over all CUs (k_reg_front1 / k_reg_front2), then the factorisation in one workgroup per model.  (Tuning builds:
    // FD_REG_SPLIT=0 keeps the whole build in the one workgroup, round 3's form.)
    static const bool split_env = [] { const char *e = tuning_env("FD_REG_SPLIT"); return e == nullptr || atoi(e) != 0; }();
    const bool split = split_env && b.reg_front != 0 && reg_front_doubles(b.M) <= ns_doubles(b.M);
    if (split) {
        const int nbk = (b.M + 15) / 16, ntiles = nbk * (nbk + 1) / 2;
        // tiles per wave: the fewest that keep a launch's workgroups within the CUs (batches: a workgroup alone on its CU runs its
        // serial wave -- reflectors, the 4 x 4 algebra -- at full speed; 20 models: 2 tiles per wave, 32: 3)
        const int cus = (int)device_cus();
        auto tiles_per_wave = [&](int per_wg) {
            int tpw = 1;
            while (tpw < kFrontMaxTpw && ((ntiles + per_wg * tpw - 1) / (per_wg * tpw)) * (int)nbatch > cus) ++tpw;
            return tpw;
        };
        const int tpw1 = tiles_per_wave(kFront1Tiles), tpw2 = tiles_per_wave(kFront2Tiles);
        hipLaunchKernelGGL(k_reg_front1, dim3((unsigned)((ntiles + kFront1Tiles * tpw1 - 1) / (kFront1Tiles * tpw1)), 1, nbatch), dim3(kRegThreads), 0, stream,
                           b.d_slots, src ? *src : none, src ? 1 : 0, b.M, b.T, b.kind, b.lambda, b.gauss_R, tpw1);
        hipLaunchKernelGGL(k_reg_front2, dim3((unsigned)((ntiles + kFront2Tiles * tpw2 - 1) / (kFront2Tiles * tpw2)), 1, nbatch), dim3(kRegThreads), 0, stream,
                           b.d_slots, b.M, b.T, tpw2);
        hipLaunchKernelGGL(k_build_reg<false>, dim3(1, 1, nbatch), dim3(kRegThreads), lds, stream, b.d_slots, none, 0, b.M, b.T,
                           b.npad, b.kind, b.Mpad, b.lambda, b.gauss_R, want_stamps ? d_stamps : nullptr, (double *)nullptr);
    } else
    hipLaunchKernelGGL(k_build_reg<true>, dim3(1, 1, nbatch), dim3(kRegThreads), lds, stream, b.d_slots, src ? *src : none, src ? 1 : 0, b.M, b.T,
                       b.npad, b.kind, b.Mpad, b.lambda, b.gauss_R, want_stamps ? d_stamps : nullptr, (double *)nullptr);
    if (want_stamps && d_stamps && stamps_late) {
        static bool registered = false;
        static unsigned long long *d_keep = nullptr;
        d_keep = d_stamps;
        if (!registered) {
            registered = true;
            atexit([] {
                unsigned long long h[96];
                if (d_keep && hipDeviceSynchronize() == hipSuccess && hipMemcpy(h, d_keep, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess && h[83]) {
                    fprintf(stderr, "[k_build_reg stamps, MEAN over %llu launches (no read-back per launch): table | assembly | tile loads | reflectors | KV | W + rotation | Cholesky | back substitution | recovery | pack]\n  ", h[83]);
                    for (int q = 0; q < 10; ++q) fprintf(stderr, " %llu", h[84 + q] / h[83]);
                    fprintf(stderr, "\n");
                }
            });
        }
    } else if (want_stamps && d_stamps) {
        unsigned long long h[80];
        (void)hipStreamSynchronize(stream);          // (a non-blocking stream: the copy below does not wait for it by itself)
        if (hipMemcpy(h, d_stamps, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "[k_build_reg stamps, shader cycles: table | assembly | tile loads | reflectors + Q^T f | Y = K V | W + rotation + B21 | Cholesky | back substitution | recovery | pack]\n  ");
            for (int q = 0; q < 10; ++q) fprintf(stderr, " %llu", h[q]);
            fprintf(stderr, "\n   Cholesky steps (cycles since its start: inverse K in LDS | panel K in LDS, as wave 0 sees them):");
            for (int q = 16; q < 48; ++q) fprintf(stderr, "%s%llu", (q - 16) % 2 ? " " : "\n     ", h[q]);
            fprintf(stderr, "\n");
        }
    }
    return hipGetLastError();
}

}  // namespace fd
