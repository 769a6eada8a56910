// fd_build_reg.hip -- the whole build of a rig of up to 256 control points in ONE launch of ONE workgroup per model, with the
// matrix in REGISTERS: null-space projection + blocked Cholesky + both substitutions + packing.
//
// Same system, same unknowns and the same mathematics as fd_nullspace.hip (replaces alglib::rbfbuildmodel, reference
// src/SOP_FaceDeform.cpp:363-368, in north_star's dense formulation):
//
//     [ K   P ] [ w ]   [ f ]     P = Q [0; R],   B = Q^T K Q = K - V W^T - W V^T,   B11 y = (Q^T f)_1 (Cholesky, order
//     [ P^T 0 ] [ a ] = [ 0 ]     n1 = M - T),    R a = (Q^T f)_2 - B21 y,           w = Q [y; 0]
//
// Why another kernel.  Round 2's one-launch build (k_build_small) kept the matrix in L2: every phase was a handful of
// dependent ~1 us round trips that four waves could not overlap -- 0.37 ms at M = 256 against 0.25 ms for the launch chain,
// whose nine step launches cost ~4 us of launch floor each.  Here the lower triangle lives in the accumulator registers of
// eight waves and never leaves the CU between steps:
//
//   * 16 x 16 tiles in the v_mfma_f64_16x16x4_f64 C/D layout (lane (c, g) = (lane & 15, lane >> 4) holds rows g, g + 4, g + 8,
//     g + 12 of column c: 4 doubles); order 256 = 136 lower tiles + 16 tiles that carry the three right-hand sides as extra
//     ROWS (so the forward substitution rides along) = 152 tiles = 19 per wave, dealt out round robin down the columns so that
//     every suffix of columns -- what is left at step K -- is spread evenly;
//   * per 16 columns: the wave that owns the diagonal tile factorises it AND inverts the factor in one pass (32 lanes: 16
//     rows of the tile + 16 rows of the identity under the same column operations: [A; I] L^-T = [L; L^-T]); the tiles below
//     become L_IK = C_IK inv(L_KK)^T as four matrix instructions each (no substitution chain per row), go through LDS once as
//     the operands of the trailing update, and C_IJ -= L_IK L_JK^T is four more matrix instructions per tile;
//   * back substitution in row form (y^T L = z^T), right-looking: a tile is its own B operand, one row of tiles per step;
//   * reflectors, Q^T f, Y = K V, W, the rotation, B21, the recovery of a and w: in LDS and on the tiles in place;
//   * the evaluation records / centre tiles / status are written by the same code as every other build (fd_pack.h).
//
// Roof: the fp64 matrix pipe (78.6 TFLOP/s); algorithmic work (1/3) n1^3 = 5.3 MFLOP at M = 256.  What bounds it is the
// factorisation's critical path -- 16 dependent diagonal blocks of 16 dependent columns -- not flops: see DESIGN.md 4.2d.
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
constexpr int kRhsRow = 16;             // tile-row index of the right-hand sides
constexpr int kSlots = 19;              // tiles per wave: (136 + 16) / 8
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

// N sums over the 512-thread workgroup at once; every thread gets all of them (fixed order: deterministic)
template <int N>
__device__ __forceinline__ void wg_sum(double (&v)[N], double *scratch /* [8][N] */, int tid)
{
#pragma unroll
    for (int q = 0; q < N; ++q)
        for (int off = 32; off >= 1; off >>= 1) v[q] += __shfl_xor(v[q], off);
    __syncthreads();
    if ((tid & 63) == 0) {
#pragma unroll
        for (int q = 0; q < N; ++q) scratch[(tid >> 6) * N + q] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < N; ++q) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kRegWaves; ++w) s += scratch[w * N + q];
        v[q] = s;
    }
}

// lanes of one wave talking through LDS: a store by one lane and a load by another are unrelated to the compiler
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 1 / sqrt(d): hardware estimate + two coupled Newton steps (as fd_nullspace.hip's sqrt_rsqrt)
__device__ __forceinline__ void sqrt_rsqrt(double d, double &root, double &inv)
{
    const double y0 = __builtin_amdgcn_rsq(d);
    double gg = d * y0, h = 0.5 * y0;
    double r = fma(-h, gg, 0.5);
    gg = fma(gg, r, gg); h = fma(h, r, h);
    r = fma(-h, gg, 0.5);
    gg = fma(gg, r, gg); h = fma(h, r, h);
    r = fma(-gg, gg, d);
    root = fma(r, h, gg);
    inv = 2.0 * h;
}

// LDS map (doubles).  The per-wave partial sums of Y = K V (8 x 4 M) exist only before the factorisation and lie over the
// inverse blocks and the panel buffer, which exist only from then on.
struct RegLds {
    double *cen, *V, *W, *F, *B21, *small, *red, *D, *line, *minv, *P, *scr, *Z, *Y, *stat, *ypart;
    int *tab;
};
constexpr int kSmallDoubles = 80;       // tau[4] R[16] Tm[16] g[12] G[16] misc
constexpr int kTau = 0, kR = 4, kTm = 20, kG = 36, kGm = 48, kAc = 64;
__host__ __device__ inline size_t reg_lds_doubles(int M)
{
    const size_t overlay = (size_t)kMaxBlocks * kTileLds + (size_t)(kMaxBlocks + 1) * kTileLds;      // minv + P
    const size_t ypart = (size_t)kRegWaves * 4 * (size_t)M;
    return (size_t)3 * M + 4 * (size_t)M + 4 * (size_t)M + 3 * (size_t)M + 4 * (size_t)M + kSmallDoubles + 8 * 16 + kTileLds + 48 +
           (overlay > ypart ? overlay : ypart) + (size_t)kRegWaves * kTileLds + 3 * 256 + 3 * 256 + 8 + 160 /* tile table, as ints */;
}
__device__ __forceinline__ RegLds carve(double *base, int M)
{
    RegLds L;
    double *p = base;
    L.cen = p; p += 3 * M;
    L.V = p; p += 4 * M;
    L.W = p; p += 4 * M;
    L.F = p; p += 3 * M;
    L.B21 = p; p += 4 * M;
    L.small = p; p += kSmallDoubles;
    L.red = p; p += 8 * 16;
    L.D = p; p += kTileLds;
    L.line = p; p += 48;
    const size_t overlay = (size_t)kMaxBlocks * kTileLds + (size_t)(kMaxBlocks + 1) * kTileLds;
    const size_t ypart = (size_t)kRegWaves * 4 * (size_t)M;
    L.minv = p; L.P = p + (size_t)kMaxBlocks * kTileLds; L.ypart = p;
    p += overlay > ypart ? overlay : ypart;
    L.scr = p; p += (size_t)kRegWaves * kTileLds;
    L.Z = p; p += 3 * 256;
    L.Y = p; p += 3 * 256;
    L.stat = p; p += 8;
    L.tab = reinterpret_cast<int *>(p);
    return L;
}

// The diagonal tile at sD (16 x 16, pitch 17, lower triangle read) -> inverse of its Cholesky factor at sMinv.  One wave.
// Row lanes 0..15 hold the rows of A, row lanes 16..31 the rows of the identity; both go through the same column operations
// (scale column k by 1 / l_kk, subtract l_mk times it from column m > k), which turn [A; I] into [L; L^-T]: row lane 16 + c
// ends with row c of L^-T = column c of L^-1.  A row is split over two lanes by column parity (lane = row lane + 32 h holds
// columns 2 j + h: 8 doubles -- with all 16 in one lane the routine did not fit beside the wave's 19 resident tiles: 260
// spills).  The scaled column travels through an LDS line ordered by row parity, so each half fetches the multipliers of
// ITS columns with broadcast reads; the NEXT pivot is formed ahead of that round trip from two v_readlanes, so its
// square-root chain runs meanwhile.
__device__ __forceinline__ void factor_invert_16(const double *sD, double *sLine /* [48] */, double *sMinv, double tiny, int live_cols,
                                                 double *sStat, int lane)
{
    const int rl = lane & 31, h = lane >> 5, row = rl & 15;
    const bool is_a = rl < 16;
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = is_a ? sD[row * kPitch + 2 * j + h] : ((2 * j + h == row) ? 1.0 : 0.0);
    const double *colp = sLine + 16 * h;                                   // l_{2 j + h, k} at colp[j]
    double *mine = is_a ? sLine + 16 * (row & 1) + (row >> 1) : sLine + 32 + row;      // this row's scaled element of column k
    double pmin = INFINITY, pmax = 0.0;
    bool singular = false;
    double d = readlane_f64(a[0], 0);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int hk = k & 1, jk = k >> 1;
        const bool ok = d > tiny;                    // false for NaN and for a lost definiteness
        if (k < live_cols) {
            if (!ok) singular = true;
            const double ad = fabs(d);
            pmin = ad < pmin ? ad : pmin;
            pmax = ad > pmax ? ad : pmax;
        }
        double root, inv;
        sqrt_rsqrt(ok ? d : 1.0, root, inv);
        if (!ok) inv = 0.0;
        const double own = a[jk] * inv;              // meaningful in the half that holds column k
        wave_lds_sync();                             // the previous column's line has been read by everyone
        if (h == hk) { a[jk] = own; *mine = own; }
        if (k + 1 < 16) {
            const int hn = (k + 1) & 1, jn = (k + 1) >> 1;
            const double lnext = readlane_f64(own, k + 1 + 32 * hk);        // l_{k+1,k}
            const double dold = readlane_f64(a[jn], k + 1 + 32 * hn);       // a_{k+1,k+1} before this column
            d = fma(-lnext, lnext, dold);
        }
        wave_lds_sync();
        const double lik = *mine;
        if (hk == 0 && h == 1) a[jk] = fma(-colp[jk], lik, a[jk]);          // column k + 1 sits in the other half
#pragma unroll
        for (int j = jk + 1; j < 8; ++j) a[j] = fma(-colp[j], lik, a[j]);
    }
    // inverse, plain [row m][col c]: row lane 16 + c holds x_m of column c, m = 2 j + h
    if (!is_a) {
#pragma unroll
        for (int j = 0; j < 8; ++j) sMinv[(2 * j + h) * kPitch + row] = a[j];
    }
    if (lane == 0) {
        if (singular) sStat[2] = 1.0;
        sStat[0] = pmin < sStat[0] ? pmin : sStat[0];
        sStat[1] = pmax > sStat[1] ? pmax : sStat[1];
    }
}

__global__ __launch_bounds__(kRegThreads) void k_build_reg(const BatchSlot *tab, int M, int T, int npad, int lda, int kind, int Mpad,
                                                            unsigned long long *stamps)
{
    const BatchSlot &slot = tab[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    const RegLds L = carve(dyn_lds, M);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int n1 = M - T;
    const int nbk = (M + 15) / 16;                   // tile rows / columns of K
    const int nb = (n1 + 15) / 16;                   // ... of the projected block that is factorised
    gcdouble *A = as_global(slot.A);
    DevModel FD_GLOBAL *model = as_global(slot.model);
    unsigned long long st_prev = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    int st_k = 0;
#define FD_RSTAMP() if (stamps && blockIdx.z == 0 && tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamps[st_k++] = t_ - st_prev; st_prev = t_; }
    __builtin_amdgcn_s_setprio(3);

    // ---- tile table: column J of the lower triangle top to bottom (I = J .. nbk - 1), then its right-hand-side tile;
    //      tile q belongs to wave q % 8, slot q / 8
    const int ntiles = nbk * (nbk + 1) / 2 + nbk;
    if (tid < ntiles) {
        int q = tid, J = 0;
        while (q >= nbk - J + 1) { q -= nbk - J + 1; ++J; }
        const int I = (q == nbk - J) ? kRhsRow : J + q;
        L.tab[tid] = I | (J << 8);
    }
    // centres, deltas, statistics
    for (int e = tid; e < 3 * M; e += kRegThreads) {
        L.cen[e] = as_global(slot.centres)[e];
        L.F[e] = A[(size_t)(npad + e / M) * lda + e % M];
    }
    if (tid == 0) { L.stat[0] = INFINITY; L.stat[1] = 0.0; L.stat[2] = 0.0; }
    __syncthreads();
    int tIJ[kSlots];                                 // I | J << 8, wave-uniform (0xffff: no tile)
#pragma unroll
    for (int t = 0; t < kSlots; ++t) {
        const int q = wave + kRegWaves * t;
        tIJ[t] = __builtin_amdgcn_readfirstlane(q < ntiles ? L.tab[q] : 0xffff);
    }
    // (read through an opaque copy: with the coordinates visibly loop-invariant the compiler hoists every tile's LDS
    // addresses out of the step loop -- 40 more live registers beside the 152 of the tiles, and spills)
#define tI(t) (opaque_s(tIJ[t]) & 0xff)
#define tJ(t) (opaque_s(tIJ[t]) >> 8)

    // ---- K tiles into registers.  S[t][i] = K[16 I + g + 4 i][16 J + c]; read as its transpose K[16 J + c][16 I + g + 4 i]
    // (the block is symmetric bit for bit for every kernel this path takes), so that 16 lanes read 128 contiguous bytes.
    double4_t S[kSlots];
#pragma unroll
    for (int t = 0; t < kSlots; ++t) {
        S[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
        if (tI(t) < kMaxBlocks) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 16 * tI(t) + g + 4 * i, col = 16 * tJ(t) + c;
                if (row < M && col < M) S[t][i] = A[(size_t)row * lda + col];
            }
        }
    }
    FD_RSTAMP()

    // ---- reflectors of P = [1 x y z] (dlarfg upside down: reflector k acts on rows 0 .. M-1-k, beta in row M-1-k)
    bool singular = false;
    if (T > 0) {
        const int i = tid;
        double cn[4] = {0.0, 0.0, 0.0, 0.0};
        if (i < M) {
            const double p[4] = {1.0, L.cen[3 * i], L.cen[3 * i + 1], L.cen[3 * i + 2]};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double v = t < T ? p[t] : 0.0;
                L.V[4 * i + t] = v;
                cn[t] = v * v;
            }
        }
        wg_sum<4>(cn, L.red, tid);              // its barriers also publish V
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k >= T) break;
            const int piv = M - 1 - k;
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            if (i < piv) {
                const double x = L.V[4 * i + k];
                acc[0] = x * x;
#pragma unroll
                for (int cc = k + 1; cc < 4; ++cc) if (cc < T) acc[cc - k] = x * L.V[4 * i + cc];
            }
            wg_sum<4>(acc, L.red, tid);
            const double xp = L.V[4 * piv + k];
            const double sigma = acc[0];
            const double norm = sqrt(fma(xp, xp, sigma));
            double beta = xp, tau = 0.0, scale = 0.0;
            if (sigma > 0.0) {
                beta = xp >= 0.0 ? -norm : norm;
                tau = (beta - xp) / beta;
                scale = 1.0 / (xp - beta);
            }
            if (!(norm > 64.0 * (double)M * kEps * sqrt(cn[k]))) singular = true;   // P has no full column rank (NaN too)
            double sc[4] = {0.0, 0.0, 0.0, 0.0}, prow[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int cc = k + 1; cc < 4; ++cc) if (cc < T) { prow[cc] = L.V[4 * piv + cc]; sc[cc] = fma(scale, acc[cc - k], prow[cc]); }
            __syncthreads();                     // everyone has read the pivot row
            if (i < piv) {
                const double v = L.V[4 * i + k] * scale;
                L.V[4 * i + k] = v;
#pragma unroll
                for (int cc = k + 1; cc < 4; ++cc) if (cc < T) L.V[4 * i + cc] = fma(-tau * sc[cc], v, L.V[4 * i + cc]);
            }
            if (tid == 0) {
                L.small[kTau + k] = tau;
                L.small[kR + 4 * k + k] = beta;
                L.V[4 * piv + k] = 1.0;
#pragma unroll
                for (int cc = k + 1; cc < 4; ++cc) if (cc < T) {
                    L.small[kR + 4 * k + cc] = fma(-tau, sc[cc], prow[cc]);
                    L.V[4 * piv + cc] = 0.0;
                }
            }
            __syncthreads();
        }
        // compact WY factor from the Gram matrix of V
        double gram[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (i < M) {
            const double v0 = L.V[4 * i], v1 = L.V[4 * i + 1], v2 = L.V[4 * i + 2], v3 = L.V[4 * i + 3];
            gram[0] = v0 * v1; gram[1] = v0 * v2; gram[2] = v0 * v3; gram[3] = v1 * v2; gram[4] = v1 * v3; gram[5] = v2 * v3;
        }
        wg_sum<6>(gram, L.red, tid);
        if (tid == 0) {
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
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) L.small[kTm + 4 * a + b] = Tm[a][b];
#pragma unroll
            for (int k = 0; k < 4; ++k) if (k >= T) L.small[kTau + k] = 0.0;
        }
        __syncthreads();
        // f <- Q^T f; the pivot rows' values go aside (they belong to the polynomial equations)
        for (int k = 0; k < T; ++k) {
            const int piv = M - 1 - k;
            const double tau = L.small[kTau + k];
            double d[3] = {0.0, 0.0, 0.0};
            double v = 0.0;
            if (i <= piv) {
                v = L.V[4 * i + k];
                d[0] = v * L.F[i]; d[1] = v * L.F[M + i]; d[2] = v * L.F[2 * M + i];
            }
            wg_sum<3>(d, L.red, tid);
            if (i <= piv) {
                L.F[i] = fma(-tau * d[0], v, L.F[i]); L.F[M + i] = fma(-tau * d[1], v, L.F[M + i]); L.F[2 * M + i] = fma(-tau * d[2], v, L.F[2 * M + i]);
            }
            __syncthreads();
        }
        if (tid < T) {
            const int piv = M - 1 - tid;
            L.small[kG + 3 * tid] = L.F[piv]; L.small[kG + 3 * tid + 1] = L.F[M + piv]; L.small[kG + 3 * tid + 2] = L.F[2 * M + piv];
        }
        FD_RSTAMP()

        // ---- Y = K V from the tiles: per-wave partial sums in LDS, added up in a fixed order.  A tile in the accumulator
        // layout IS the A operand of K_IJ^T Z (slice i = rows 4 i .. 4 i + 3 of K_IJ): that gives block J of Y its share from
        // block I; the share of block I from block J needs K_IJ itself as the operand: through the wave's LDS scratch.
        double *yp = L.ypart + (size_t)wave * 4 * M;
        for (int e = lane; e < 4 * M; e += 64) yp[e] = 0.0;
        double *scr = L.scr + (size_t)wave * kTileLds;
        wave_lds_sync();
#pragma unroll
        for (int t = 0; t < kSlots; ++t) {
            if (tI(t) >= kMaxBlocks) continue;
            const int I = tI(t), J = tJ(t);
            {   // K_JI V_I -> rows of block J
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int r = 16 * I + 4 * s + g;
                    const double b = (c < 4 && r < M) ? L.V[4 * r + c] : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(S[t][s], b, acc, 0, 0, 0);
                }
                if (c < 4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const int r = 16 * J + g + 4 * i; if (r < M) yp[4 * r + c] += acc[i]; }
                }
            }
            if (I != J) {   // K_IJ V_J -> rows of block I
#pragma unroll
                for (int i = 0; i < 4; ++i) scr[(g + 4 * i) * kPitch + c] = S[t][i];
                wave_lds_sync();
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const double a = scr[c * kPitch + 4 * s + g];
                    const int r = 16 * J + 4 * s + g;
                    const double b = (c < 4 && r < M) ? L.V[4 * r + c] : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                }
                wave_lds_sync();
                if (c < 4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const int r = 16 * I + g + 4 * i; if (r < M) yp[4 * r + c] += acc[i]; }
                }
            }
            wave_lds_sync();
        }
        __syncthreads();
        for (int e = tid; e < 4 * M; e += kRegThreads) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < kRegWaves; ++w) s += L.ypart[(size_t)w * 4 * M + e];
            L.W[e] = s;                             // Y for now
        }
        __syncthreads();
        FD_RSTAMP()

        // ---- W = Y Tm - (1/2) V G,  G = Tm^T sym(V^T Y) Tm
        {
            double Sm[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) Sm[q] = 0.0;
            if (i < M) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) Sm[4 * a + b] = L.V[4 * i + a] * L.W[4 * i + b];
            }
            wg_sum<16>(Sm, L.red, tid);
            if (tid == 0) {
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
#pragma unroll
                for (int q = 0; q < 16; ++q) L.small[kGm + q] = Gm[q];
            }
            __syncthreads();
            if (i < M) {
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
            __syncthreads();
        }

        // ---- B = K - V W^T - W V^T on the tiles in place (two K = 4 matrix instructions per tile)
#pragma unroll
        for (int t = 0; t < kSlots; ++t) {
            if (tI(t) >= kMaxBlocks) continue;
            const int ri = 16 * tI(t) + c, rj = 16 * tJ(t) + c;
            const double vi = ri < M ? L.V[4 * ri + g] : 0.0, wi = ri < M ? L.W[4 * ri + g] : 0.0;
            const double vj = rj < M ? L.V[4 * rj + g] : 0.0, wj = rj < M ? L.W[4 * rj + g] : 0.0;
            S[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-vi, wj, S[t], 0, 0, 0);
            S[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-wi, vj, S[t], 0, 0, 0);
        }
    }
    // ---- B21 aside (pivot row M-1-k = equation of polynomial coefficient k), identity padding beyond n1, right-hand-side tiles
#pragma unroll
    for (int t = 0; t < kSlots; ++t) {
        if (tI(t) < kMaxBlocks) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 16 * tI(t) + g + 4 * i, col = 16 * tJ(t) + c;
                if (row >= n1 && row < M && col < n1) L.B21[4 * col + (M - 1 - row)] = S[t][i];
                if (row >= n1 || col >= n1) S[t][i] = row == col ? 1.0 : 0.0;
            }
        } else if (tI(t) == kRhsRow) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rhs = g + 4 * i, col = 16 * tJ(t) + c;
                S[t][i] = (rhs < 3 && col < n1) ? L.F[rhs * M + col] : 0.0;
            }
        }
    }
    __syncthreads();                                 // the overlay (partial sums of Y) is dead: inverse blocks and panel buffer from here
    FD_RSTAMP()

    // ---- blocked Cholesky, 16 columns per step, right-hand sides as tile row 16
    const double amax = __longlong_as_double((long long)model->amax_bits);
    const double tiny = (double)n1 * kEps * amax;
    double *scr = L.scr + (size_t)wave * kTileLds;
    for (int K = 0; K < nb; ++K) {
        // (i) the owner of the diagonal tile: factor + inverse (its tile is up to date: it applied every earlier panel itself)
        bool mine = false;
#pragma unroll
        for (int t = 0; t < kSlots; ++t) {
            if (tI(t) == K && tJ(t) == K) {
#pragma unroll
                for (int i = 0; i < 4; ++i) L.D[(g + 4 * i) * kPitch + c] = S[t][i];
                mine = true;
            }
        }
        if (mine) {
            wave_lds_sync();
            const int live = n1 - 16 * K < 16 ? n1 - 16 * K : 16;
            factor_invert_16(L.D, L.line, L.minv + (size_t)K * kTileLds, tiny, live, L.stat, lane);
        }
        __syncthreads();
        // (ii) the tiles below it (and the right-hand sides): L_IK = C_IK inv(L_KK)^T; into the panel buffer
        const double *mk = L.minv + (size_t)K * kTileLds;
        double bop[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) bop[s] = mk[c * kPitch + 4 * s + g];         // B[k][n] = inv[n][4 s + k]
#pragma unroll
        for (int t = 0; t < kSlots; ++t) {
            if (tJ(t) == K && tI(t) > K && (tI(t) < nb || tI(t) == kRhsRow)) {
#pragma unroll
                for (int i = 0; i < 4; ++i) scr[(g + 4 * i) * kPitch + c] = S[t][i];
                wave_lds_sync();
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(scr[c * kPitch + 4 * s + g], bop[s], acc, 0, 0, 0);
                wave_lds_sync();
                S[t] = acc;
                double *dst = L.P + (size_t)tI(t) * kTileLds;
#pragma unroll
                for (int i = 0; i < 4; ++i) dst[(g + 4 * i) * kPitch + c] = acc[i];
            }
        }
        __syncthreads();
        // (iii) trailing update C_IJ -= L_IK L_JK^T for every tile right of the panel
#pragma unroll
        for (int t = 0; t < kSlots; ++t) {
            if (tJ(t) > K && tJ(t) < nb && (tI(t) < nb || tI(t) == kRhsRow)) {
                const double *pa = L.P + (size_t)tI(t) * kTileLds, *pb = L.P + (size_t)tJ(t) * kTileLds;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    S[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[c * kPitch + 4 * s + g], pb[c * kPitch + 4 * s + g], S[t], 0, 0, 0);
            }
        }
        // (no barrier: the next panel's tiles are written only after the barrier that follows the next diagonal block)
    }
    FD_RSTAMP()

    // ---- y^T L = z^T, bottom up, right-looking: one row of tiles per step (a tile is its own B operand)
#pragma unroll
    for (int t = 0; t < kSlots; ++t) {
        if (tI(t) == kRhsRow && tJ(t) < nb && g < 3) L.Z[g * 256 + 16 * tJ(t) + c] = S[t][0];
    }
    __syncthreads();
    for (int I = nb - 1; I >= 0; --I) {
        if (wave == 0 && lane < 48) {
            const int rhs = lane >> 4, n = lane & 15;
            const double *mi = L.minv + (size_t)I * kTileLds;
            double y = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) y = fma(L.Z[rhs * 256 + 16 * I + k], mi[k * kPitch + n], y);
            L.Y[rhs * 256 + 16 * I + n] = y;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < kSlots; ++t) {
            if (tI(t) == I && tJ(t) < I) {
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const double a = c < 3 ? L.Y[c * 256 + 16 * I + 4 * s + g] : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, S[t][s], acc, 0, 0, 0);
                }
                if (g < 3) L.Z[g * 256 + 16 * tJ(t) + c] -= acc[0];
            }
        }
        __syncthreads();
    }
    FD_RSTAMP()

    // ---- R a = g - B21 y;  w = Q [y; 0] = H_0 .. H_{T-1} [y; 0]  (x lives where f did)
    {
        const int i = tid;
        gdouble *X = as_global(slot.X);
        double q[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) q[e] = 0.0;
        if (i < n1 && T > 0) {
            const double y0 = L.Y[i], y1 = L.Y[256 + i], y2 = L.Y[512 + i];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double b = k < T ? L.B21[4 * i + k] : 0.0;
                q[3 * k] = b * y0; q[3 * k + 1] = b * y1; q[3 * k + 2] = b * y2;
            }
        }
        wg_sum<12>(q, L.red, tid);
        if (tid < 3) {                               // R a = g - B21 y: output tid, upper triangular in (k, c)
            double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 3; k >= 0; --k) {
                if (k >= T) continue;
                double v = L.small[kG + 3 * k + tid] - q[3 * k + tid];
#pragma unroll
                for (int c2 = k + 1; c2 < 4; ++c2) if (c2 < T) v = fma(-L.small[kR + 4 * k + c2], a[c2], v);
                a[k] = v / L.small[kR + 4 * k + k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) L.small[kAc + 3 * k + tid] = a[k];
        }
        if (i < M) {
            const bool in = i < n1;
            L.F[i] = in ? L.Y[i] : 0.0; L.F[M + i] = in ? L.Y[256 + i] : 0.0; L.F[2 * M + i] = in ? L.Y[512 + i] : 0.0;
        }
        __syncthreads();
        for (int k = T - 1; k >= 0; --k) {
            const int piv = M - 1 - k;
            const double tau = L.small[kTau + k];
            double d[3] = {0.0, 0.0, 0.0};
            double v = 0.0;
            if (i <= piv) {
                v = L.V[4 * i + k];
                d[0] = v * L.F[i]; d[1] = v * L.F[M + i]; d[2] = v * L.F[2 * M + i];
            }
            wg_sum<3>(d, L.red, tid);
            if (i <= piv) {
                L.F[i] = fma(-tau * d[0], v, L.F[i]); L.F[M + i] = fma(-tau * d[1], v, L.F[M + i]); L.F[2 * M + i] = fma(-tau * d[2], v, L.F[2 * M + i]);
            }
            __syncthreads();
        }
        for (int e = tid; e < 3 * npad; e += kRegThreads) {
            const int cc = e / npad, r = e % npad;
            double v = 0.0;
            if (r < M) v = L.F[cc * M + r];
            else if (r < M + T) v = L.small[kAc + 3 * (r - M) + cc];
            X[e] = v;
        }
        if (tid == 0) {
            const bool sing = singular || L.stat[2] != 0.0;
            if (sing) model->sing_flag = 1;
            model->iterations = M + T;
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
    packing::pack_body(slot, npad, M, Mpad, T, kind, 0, 0);
    if (kind == FD_KERNEL_THIN_PLATE) {
        __syncthreads();
        for (int tile = tid >> 6; tile < Mpad / 16; tile += 4) packing::pack_tiles_body(slot, Mpad, tile, lane);
    }
    FD_RSTAMP()
#undef FD_RSTAMP
#undef tI
#undef tJ
}

}  // namespace

bool reg_applicable(int kind, int term, double lambda, int M)
{
    return M <= 16 * kMaxBlocks && spd_applicable(kind, term, lambda, M);
}

// Once per device, OUTSIDE any stream capture (the C ABI calls it before it captures a build): the kernel's dynamic LDS limit
// is a per-device property.  (The packing code has a little static LDS of its own: the dynamic limit leaves room for it.)
hipError_t reg_build_init()
{
    static bool done[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && done[dev]) return hipSuccess;
    e = hipFuncSetAttribute((const void *)k_build_reg, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(sizeof(double) * reg_lds_doubles(16 * kMaxBlocks)));
    if (e == hipSuccess && dev >= 0 && dev < 64) done[dev] = true;
    return e;
}

// everything after k_prepare
hipError_t launch_build_reg(const BuildBuffers &b, hipStream_t stream, hipEvent_t ev_mid)
{
    const unsigned nbatch = (unsigned)b.nbatch;
    const int M = b.M, T = b.T;
    hipError_t e = launch_assemble_block(b, stream, round_up(M, 32));
    if (e != hipSuccess) return e;
    if (ev_mid) (void)hipEventRecord(ev_mid, stream);
    const size_t lds = sizeof(double) * reg_lds_doubles(M);
    // (diagnostics: phase stamps, only outside stream capture -- FD_NO_GRAPH=1 FD_REG_STAMPS=1)
    static unsigned long long *d_stamps = nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    static const bool stamps_env = getenv("FD_REG_STAMPS") != nullptr;
    const bool want_stamps = stamps_env && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone;
    if (want_stamps && !d_stamps) { (void)hipMalloc((void **)&d_stamps, 32 * sizeof(unsigned long long)); (void)hipMemset(d_stamps, 0, 32 * sizeof(unsigned long long)); }
    hipLaunchKernelGGL(k_build_reg, dim3(1, 1, nbatch), dim3(kRegThreads), lds, stream, b.d_slots, M, T, b.npad, b.lda, b.kind, b.Mpad,
                       want_stamps ? d_stamps : nullptr);
    if (want_stamps && d_stamps) {
        unsigned long long h[32];
        if (hipMemcpy(h, d_stamps, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "[k_build_reg stamps, shader cycles: load | reflectors + Q^T f | Y = K V | W + rotation + B21 | Cholesky | back substitution | recovery | pack]\n  ");
            for (int q = 0; q < 8; ++q) fprintf(stderr, " %llu", h[q]);
            fprintf(stderr, "\n");
        }
    }
    return hipGetLastError();
}

}  // namespace fd
