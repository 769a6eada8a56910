// fd_nullspace.hip -- the symmetric definite build: null-space projection + blocked Cholesky.
//
// Same system as fd_build.hip (replaces alglib::rbfbuildmodel, reference
// src/SOP_FaceDeform.cpp:363-368, in north_star's dense formulation)
//
//     [ K   P ] [ w ]   [ f ]        K = Phi + lambda*I  (M x M, symmetric)
//     [ P^T 0 ] [ a ] = [ 0 ]        P = [1 x y z]       (M x T)
//
// but solved without pivoting.  For the kernels that are conditionally positive definite of the
// order their polynomial term covers (thin-plate and cubic with the linear term, biharmonic with
// a constant or linear term, the fixed-radius Gaussian with any term) K is positive definite on
// the null space of P^T.  With P = Q [0; R] (T Householder reflectors, R in the LAST T rows):
//
//     w = Q [y; 0],   B = Q^T K Q,   B11 y = (Q^T f)_1,   R a = (Q^T f)_2 - B21 y
//
// and B11 (order n1 = M - T) has a Cholesky factorisation.  Partial pivoting made the LU panel a
// serial chain on ONE workgroup (fd_build.hip: ~0.8-1.6 us per column, 3.3 of 6.6 ms at order
// 2052); here only the 32 x 32 diagonal block is sequential, the rows below it are independent.
//
// Pipeline (every kernel indexes the batch table with blockIdx.z, like the LU kernels):
//   k_ns_reflectors   one workgroup: Householder vectors V (M x T), tau, R, the compact-WY factor;
//                     then f <- Q^T f in the right-hand-side columns of A (k_ns_rhs on its own
//                     for fd_set_deltas)
//   k_assemble        (fd_build.hip) K into A, full square
//   k_ns_kv           Y = K V, then (its last workgroup) W = Y Tm - (1/2) V (Tm^T V^T Y Tm),
//                     so that B = K - V W^T - W V^T
//   k_ns_rotate       lower triangle of B11 in place, B21 aside, identity padding to npc
//   k_chol_step       ONE launch per 32 columns.  Panel workgroups: the next diagonal block with
//                     the current panel applied (MFMA, into LDS), factorised by one wave in every
//                     workgroup for itself, while the other waves apply the current panel to
//                     their rows of the next panel; then X L11^T = A21, one row per thread.  The
//                     other workgroups: A22 -= L21 L21^T on the lower triangle in 32 x 32 macro
//                     tiles of v_mfma_f64_16x16x4_f64 -- above order 512 only every fourth step,
//                     at rank 128, with the steps in between keeping one column pair current.
//                     The right-hand sides ride along as three extra rows, so the forward solve
//                     costs nothing; L^T is mirrored into the upper triangle as it is produced.
//                     (k_chol_first / k_chol_solve / k_chol_trail: the unfused pieces, used by
//                     fd_set_deltas and by FD_CHOL_UNFUSED.)
//   k_backsub_inv     L^T y = z with the inverted diagonal blocks (a by-product of an otherwise
//                     idle wave of the right-hand-side workgroup), then -- one-range systems --
//                     a from R and w = Q [y; 0] into X in the layout k_pack expects
//                     (k_ns_recover on its own above order 512, after the chained ranges and
//                     fd_build.hip's k_backsub_update).
//
// fd_set_deltas (launch_resolve_spd) sends new right-hand sides through the same kernels with the
// matrix work switched off: identical operands in identical order, bit-identical weights.
#include <cstdio>
#include <cstdlib>

#include "fd_internal.h"
#include "fd_pack.h"

namespace fd {

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

// The evaluation kernel runs four 128-VGPR waves per SIMD (fd_eval.hip) and, in a pipeline of
// frames, fills the device while the next frames' builds run beside it.  A build wave that needs
// more than 128 VGPRs only fits where TWO evaluation waves have retired from one SIMD at once,
// and the dispatcher refills a single freed slot first -- such kernels were measured waiting
// hundreds of microseconds under a running evaluation.  Keep them within one slot.
#define FD_FIT_BESIDE_EVAL __attribute__((amdgpu_waves_per_eu(4, 8)))

constexpr double kEps = 2.220446049250313e-16;
constexpr int kNB = 32;                 // Cholesky block width
__device__ __forceinline__ int round_up_dev(int v, int m) { return (v + m - 1) / m * m; }
constexpr int kSlab = 256;              // rows per workgroup of the panel's triangular solve
constexpr int kBulk = 4;                // fused step, large systems: the bulk of the trailing matrix is updated every kBulk steps
constexpr size_t kStepPanelLds = sizeof(double) * (2 * 32 * 34 + 2 * 32);   // diagonal block + its transpose + two 32-vectors
constexpr int kStepSlab = 192;          // ... in the fused step kernel: three waves of rows, the fourth factorises

// small block of the solver state, after V, W and B21 (each M x 4)
constexpr int kTau = 0;                 // tau[4]
constexpr int kR = 4;                   // R[k][c], row k = pivot row M-1-k
constexpr int kTm = 20;                 // compact WY: Q = I - V Tm V^T, upper triangular
constexpr int kG = 36;                  // (Q^T f) in the pivot rows: g[k][c]
constexpr int kAff = 48;                // multilayer model: least-squares polynomial a[k][c]
constexpr int kTicket = 60;             // k_ns_kv's workgroups take a number here; the last one finishes W (an unsigned, zeroed by k_ns_reflectors)
constexpr int kSmall = 64;
// after the small block: the factorised diagonal blocks, one per 32 columns -- L11 column-major
// (32 x 32, zeros above the diagonal) followed by the reciprocals of its diagonal: what the
// triangular solves stage into LDS, in the full build and in fd_set_deltas alike; then the
// inverse of L11, which turns the back-substitution's 32-step triangle into a matrix product.
constexpr int kLdStride = 66 * 32;
constexpr int kLdInv = 34 * 32;         // offset of inverse(L11), row-major [k][j], inside a block
__device__ __forceinline__ gdouble *ld_block(double *ns, int M, int k0) { return as_global(ns) + (size_t)12 * M + kSmall + (size_t)(k0 / kNB) * kLdStride; }

__device__ __forceinline__ double readlane_f64(double v, int src_lane)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), src_lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), src_lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// N sums over a 256-thread workgroup at once; every thread gets all of them
template <int N>
__device__ __forceinline__ void block_sum_n(double (&v)[N], double *scratch /* [4][N] */, int tid)
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
    for (int q = 0; q < N; ++q) v[q] = (scratch[q] + scratch[N + q]) + (scratch[2 * N + q] + scratch[3 * N + q]);
}

__device__ __forceinline__ void ns_rhs_body(const BatchSlot &s, int M, int T, int npad, int lda, double *s_red);

// ---- reflectors ----------------------------------------------------------------------
// LAPACK dlarfg turned upside down: reflector k acts on rows 0 .. M-1-k and leaves its beta in
// row M-1-k, so the null-space block comes FIRST in the rotated system and the Cholesky starts
// at row 0 on tile boundaries.
__device__ __forceinline__ void ns_reflectors_body(const BatchSlot &s, int M, int T, int npad, int lda, int with_rhs)
{
    gcdouble *centres = as_global(s.centres);
    gdouble *V = as_global(s.ns), *small = V + (size_t)12 * M;
    __shared__ double s_red[4 * 8];
    const int tid = threadIdx.x;

    double cn[4] = {0.0, 0.0, 0.0, 0.0};     // squared norms of the columns of P
    for (int i = tid; i < M; i += 256) {
        const double p[4] = {1.0, centres[3 * i], centres[3 * i + 1], centres[3 * i + 2]};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double v = t < T ? p[t] : 0.0;
            V[4 * (size_t)i + t] = v;
            cn[t] = fma(v, v, cn[t]);
        }
    }
    block_sum_n<4>(cn, s_red, tid);          // its barriers also publish V
    bool singular = false;
    for (int k = 0; k < T; ++k) {
        const int piv = M - 1 - k;
        double acc[4] = {0.0, 0.0, 0.0, 0.0}; // sigma, then x . column c for c = k+1 ..
        for (int i = tid; i < piv; i += 256) {
            const double x = V[4 * (size_t)i + k];
            acc[0] = fma(x, x, acc[0]);
            for (int c = k + 1; c < T; ++c) acc[c - k] = fma(x, V[4 * (size_t)i + c], acc[c - k]);
        }
        block_sum_n<4>(acc, s_red, tid);
        const double xp = V[4 * (size_t)piv + k];
        const double sigma = acc[0];
        const double norm = sqrt(fma(xp, xp, sigma));
        double beta = xp, tau = 0.0, scale = 0.0;
        if (sigma > 0.0) {
            beta = xp >= 0.0 ? -norm : norm;
            tau = (beta - xp) / beta;
            scale = 1.0 / (xp - beta);
        }
        if (!(norm > 64.0 * (double)M * kEps * sqrt(cn[k]))) singular = true;   // P has no full column rank (NaN too)
        double sc[4] = {0.0, 0.0, 0.0, 0.0};  // v . column c
        double prow[4] = {0.0, 0.0, 0.0, 0.0};
        for (int c = k + 1; c < T; ++c) { prow[c] = V[4 * (size_t)piv + c]; sc[c] = fma(scale, acc[c - k], prow[c]); }
        __syncthreads();                      // everyone has read the pivot row
        for (int i = tid; i < piv; i += 256) {
            const double v = V[4 * (size_t)i + k] * scale;
            V[4 * (size_t)i + k] = v;
            for (int c = k + 1; c < T; ++c) V[4 * (size_t)i + c] = fma(-tau * sc[c], v, V[4 * (size_t)i + c]);
        }
        if (tid == 0) {
            small[kTau + k] = tau;
            small[kR + 4 * k + k] = beta;
            V[4 * (size_t)piv + k] = 1.0;
            for (int c = k + 1; c < T; ++c) {
                small[kR + 4 * k + c] = fma(-tau, sc[c], prow[c]);
                V[4 * (size_t)piv + c] = 0.0;   // the later reflectors end above this row
            }
        }
        __syncthreads();
    }
    // compact WY factor (dlarft, forward columnwise) from the Gram matrix of V
    double gram[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
    for (int i = tid; i < M; i += 256) {
        const double v0 = V[4 * (size_t)i], v1 = V[4 * (size_t)i + 1], v2 = V[4 * (size_t)i + 2], v3 = V[4 * (size_t)i + 3];
        gram[0] = fma(v0, v1, gram[0]); gram[1] = fma(v0, v2, gram[1]); gram[2] = fma(v0, v3, gram[2]);
        gram[3] = fma(v1, v2, gram[3]); gram[4] = fma(v1, v3, gram[4]); gram[5] = fma(v2, v3, gram[5]);
    }
    block_sum_n<8>(gram, s_red, tid);
    if (tid == 0) {
        double G[4][4] = {};
        G[0][1] = gram[0]; G[0][2] = gram[1]; G[0][3] = gram[2]; G[1][2] = gram[3]; G[1][3] = gram[4]; G[2][3] = gram[5];
        double Tm[4][4] = {};
        for (int k = 0; k < T; ++k) {
            const double tau = small[kTau + k];
            Tm[k][k] = tau;
            for (int a = 0; a < k; ++a) {
                double v = 0.0;
                for (int b = a; b < k; ++b) v = fma(Tm[a][b], G[b][k], v);
                Tm[a][k] = -tau * v;
            }
        }
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) small[kTm + 4 * a + b] = Tm[a][b];
        for (int k = T; k < 4; ++k) small[kTau + k] = 0.0;
        *(unsigned FD_GLOBAL *)(small + kTicket) = 0u;
        if (singular) s.model->sing_flag = 1;
    }
    if (with_rhs) {
        __threadfence_block();
        __syncthreads();                      // tau is in memory
        ns_rhs_body(s, M, T, npad, lda, s_red);
    }
}

__global__ __launch_bounds__(256) void k_ns_reflectors(const BatchSlot *tab, int M, int T, int npad, int lda, int with_rhs)
{
    ns_reflectors_body(tab[blockIdx.z], M, T, npad, lda, with_rhs);
}

// f <- Q^T f = H_{T-1} .. H_0 f in the right-hand-side columns; the pivot rows' values go aside
// (they belong to the polynomial equations) and read as zero padding afterwards.
__device__ __forceinline__ void ns_rhs_body(const BatchSlot &s, int M, int T, int npad, int lda, double *s_red /* [12] */)
{
    gcdouble *V = as_global(s.ns);
    gdouble *small = as_global(s.ns) + (size_t)12 * M;
    gdouble *f0 = as_global(s.A) + (size_t)npad * lda, *f1 = f0 + lda, *f2 = f1 + lda;
    const int tid = threadIdx.x;
    for (int k = 0; k < T; ++k) {
        const int piv = M - 1 - k;
        const double tau = small[kTau + k];
        double d[3] = {0.0, 0.0, 0.0};
        for (int i = tid; i <= piv; i += 256) {
            const double v = V[4 * (size_t)i + k];
            d[0] = fma(v, f0[i], d[0]); d[1] = fma(v, f1[i], d[1]); d[2] = fma(v, f2[i], d[2]);
        }
        block_sum_n<3>(d, s_red, tid);
        for (int i = tid; i <= piv; i += 256) {
            const double v = V[4 * (size_t)i + k];
            f0[i] = fma(-tau * d[0], v, f0[i]); f1[i] = fma(-tau * d[1], v, f1[i]); f2[i] = fma(-tau * d[2], v, f2[i]);
        }
        __syncthreads();
    }
    if (tid < T) {
        const int piv = M - 1 - tid;
        small[kG + 3 * tid] = f0[piv]; small[kG + 3 * tid + 1] = f1[piv]; small[kG + 3 * tid + 2] = f2[piv];
        f0[piv] = 0.0; f1[piv] = 0.0; f2[piv] = 0.0;
    }
}

// on its own for fd_set_deltas (the reflectors are there already); the full build runs it at the
// end of k_ns_reflectors -- same code, same bits
__global__ __launch_bounds__(256) void k_ns_rhs(const BatchSlot *tab, int M, int T, int npad, int lda)
{
    __shared__ double s_red[4 * 3];
    ns_rhs_body(tab[blockIdx.z], M, T, npad, lda, s_red);
}

// ---- W = Y Tm - (1/2) V G,  G = Tm^T (V^T Y) Tm ---------------------------------------------
// (256 threads; runs in the last workgroup of k_ns_kv to finish)
__device__ __forceinline__ void ns_w_body(const BatchSlot &s, int M)
{
    gcdouble *V = as_global(s.ns), *small = V + (size_t)12 * M;
    gdouble *Y = as_global(s.ns) + (size_t)4 * M;          // becomes W
    __shared__ double s_red[4 * 16];
    const int tid = threadIdx.x;
    double S[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) S[q] = 0.0;
    for (int i = tid; i < M; i += 256) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) S[4 * a + b] = fma(V[4 * (size_t)i + a], Y[4 * (size_t)i + b], S[4 * a + b]);
    }
    block_sum_n<16>(S, s_red, tid);
    __shared__ double s_Tm[16], s_G[16];
    if (tid == 0) {
        double Tm[16], ST[16];
        for (int q = 0; q < 16; ++q) { Tm[q] = small[kTm + q]; s_Tm[q] = Tm[q]; }
        // V^T K V is symmetric in exact arithmetic: use the mean of the two roundings so that G is
        // exactly symmetric and B = K - V W^T - W V^T stays a symmetric update
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) {       // ST = Ssym Tm
                double v = 0.0;
                for (int c = 0; c < 4; ++c) v = fma(0.5 * (S[4 * a + c] + S[4 * c + a]), Tm[4 * c + b], v);
                ST[4 * a + b] = v;
            }
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) {       // G = Tm^T ST
                double v = 0.0;
                for (int c = 0; c < 4; ++c) v = fma(Tm[4 * c + a], ST[4 * c + b], v);
                s_G[4 * a + b] = v;
            }
        for (int a = 0; a < 4; ++a)
            for (int b = a + 1; b < 4; ++b) { const double m = 0.5 * (s_G[4 * a + b] + s_G[4 * b + a]); s_G[4 * a + b] = m; s_G[4 * b + a] = m; }
    }
    __syncthreads();
    for (int i = tid; i < M; i += 256) {
        double y[4], v[4], w[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { y[t] = Y[4 * (size_t)i + t]; v[t] = V[4 * (size_t)i + t]; }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            double z = 0.0, h = 0.0;
#pragma unroll
            for (int a = 0; a < 4; ++a) { z = fma(y[a], s_Tm[4 * a + b], z); h = fma(v[a], s_G[4 * a + b], h); }
            w[b] = fma(-0.5, h, z);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) Y[4 * (size_t)i + t] = w[t];
    }
}

// ---- Y = K V ---------------------------------------------------------------------------
// 64 rows per workgroup, one row per lane (consecutive lanes walk a column of A: coalesced); the
// four waves split the columns, each with its slice of V in LDS, and their partial sums are added
// in a fixed order
// rows 64 vb .. 64 vb + 63 of Y = K V (one workgroup of the k_ns_kv grid, or one turn of the one-launch build's loop)
__device__ __forceinline__ void ns_kv_body(const BatchSlot &s, int M, int lda, int vb)
{
    gcdouble *A = as_global(s.A), *V = as_global(s.ns);
    gdouble *Y = as_global(s.ns) + (size_t)4 * M;
    __shared__ __attribute__((aligned(16))) double s_v[4][64][4];
    __shared__ double s_part[4][64][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = vb * 64 + lane;
    const int ic = i < M ? i : M - 1;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const int per = ((M + 3) / 4 + 63) & ~63;          // columns per wave, a multiple of the LDS slice
    const int jlo = wave * per, jhi = jlo + per < M ? jlo + per : M;
    for (int j0 = jlo; j0 < jhi; j0 += 64) {
        {
            const int j = j0 + lane;
#pragma unroll
            for (int t = 0; t < 4; ++t) s_v[wave][lane][t] = j < jhi ? V[4 * (size_t)j + t] : 0.0;   // my wave's slice only
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 2
        for (int q0 = 0; q0 < 64; q0 += 8) {
            double a[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { const int j = j0 + q0 + q < M ? j0 + q0 + q : M - 1; a[q] = A[(size_t)j * lda + ic]; }
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = fma(a[q], s_v[wave][q0 + q][t], acc[t]);     // zero beyond jhi
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) s_part[wave][lane][t] = acc[t];
    __syncthreads();
    if (wave == 0 && i < M) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            Y[4 * (size_t)i + t] = (s_part[0][lane][t] + s_part[1][lane][t]) + (s_part[2][lane][t] + s_part[3][lane][t]);
    }
}

__global__ __launch_bounds__(256) void k_ns_kv(const BatchSlot *tab, int M, int lda)
{
    const BatchSlot &s = tab[blockIdx.z];
    ns_kv_body(s, M, lda, blockIdx.x);
    // the last workgroup to get here turns Y into W (it needs all of Y: V^T Y is a sum over every
    // row).  Which workgroup that is does not matter to the result.
    __shared__ unsigned s_ticket;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = atomicAdd((unsigned *)(s.ns + (size_t)12 * M + kTicket), 1u);
    __syncthreads();
    if (s_ticket != gridDim.x - 1) return;
    __threadfence();
    ns_w_body(s, M);
}

// ---- B = K - V W^T - W V^T, lower triangle of the leading n1 x n1 block in place ----------------
// Rows n1 .. M-1 (pivot row M-1-k = equation of polynomial coefficient k) go to B21 and read as
// padding afterwards; columns n1 .. npc-1 become identity padding.
__device__ __forceinline__ void ns_rotate_body(const BatchSlot &s, int M, int T, int npc, int lda, int ti, int tj)
{
    if (ti < tj) return;
    gdouble *A = as_global(s.A);
    gcdouble *V = as_global(s.ns), *W = V + (size_t)4 * M;
    gdouble *B21 = as_global(s.ns) + (size_t)8 * M;
    const int n1 = M - T;
    const int i = ti * 32 + (threadIdx.x & 31);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = tj * 32 + (threadIdx.x >> 5) + 8 * q;
        if (j >= npc || i < j) continue;
        if (j >= n1) {
            if (i < npc) A[(size_t)j * lda + i] = i == j ? 1.0 : 0.0;
            continue;
        }
        if (i >= M) continue;                 // zero padding rows, written by the assembly
        double b = A[(size_t)j * lda + i];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            b = fma(-V[4 * (size_t)i + t], W[4 * (size_t)j + t], b);
            b = fma(-W[4 * (size_t)i + t], V[4 * (size_t)j + t], b);
        }
        if (i < n1) {
            A[(size_t)j * lda + i] = b;
        } else {
            B21[4 * (size_t)j + (M - 1 - i)] = b;
            if (i < npc) A[(size_t)j * lda + i] = 0.0;
        }
    }
}

__global__ __launch_bounds__(256) void k_ns_rotate(const BatchSlot *tab, int M, int T, int npc, int lda)
{
    ns_rotate_body(tab[blockIdx.z], M, T, npc, lda, blockIdx.x, blockIdx.y);
}

// ---- Cholesky: diagonal block --------------------------------------------------------------------
// sqrt(d) and 1/sqrt(d) together: hardware estimate + two coupled Newton steps (Goldschmidt)
__device__ __forceinline__ void sqrt_rsqrt(double d, double &root, double &inv)
{
    const double y0 = __builtin_amdgcn_rsq(d);
    double g = d * y0, h = 0.5 * y0;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-g, g, d);                        // one correction of the root itself
    root = fma(r, h, g);
    inv = 2.0 * h;
}

constexpr int kLdsRow = kNB + 2;          // even: rows stay 16-byte aligned for ds_read_b128

// The 32 x 32 diagonal block at (kb, kb) goes through three stages, all on LDS copies:
//   block_load_update  all 256 threads: the block and, after step kprev, its share of that step's
//                      trailing update (C -= Lr Lr^T with Lr = rows kb .. kb+31 of that panel);
//   factor_wave        one wave: rows in registers, columns right-looking, multipliers broadcast
//                      through an LDS line;
//   block_store        L11 to the side store (what the triangular solves and fd_set_deltas read),
//                      and into A mirrored above the diagonal.
__device__ __forceinline__ void block_load_update(gcdouble *A, int lda, int kb, int kprev, double (*sC)[kLdsRow],
                                                  double (*sR)[kLdsRow])
{
    const int tid = threadIdx.x;
    for (int e = tid; e < kNB * kNB; e += 256) {
        const int r = e & 31, c = e >> 5;
        sC[r][c] = A[(size_t)(kb + c) * lda + kb + r];
        if (kprev >= 0) sR[r][c] = A[(size_t)(kprev + c) * lda + kb + r];
    }
    __syncthreads();
    if (kprev >= 0) {
        double v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q;
            const int r = e & 31, c = e >> 5;
            double acc = sC[r][c];
#pragma unroll
            for (int k = 0; k < kNB; ++k) acc = fma(-sR[r][k], sR[c][k], acc);
            v[q] = acc;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q;
            sC[e & 31][e >> 5] = v[q];
        }
        __syncthreads();
    }
}

// wave 0 only (tid < 64); `stats`: this workgroup reports pivots, progress and failure
__device__ __forceinline__ void factor_wave(DevModel FD_GLOBAL *model, int n1, int kb, double (*sC)[kLdsRow], double *sInv,
                                            double *sCol, bool stats)
{
    const int tid = threadIdx.x;
    // Row i of the block lives in lanes i and i + 32: the lower half keeps the even columns,
    // the upper half the odd ones (16 doubles each).  Column j, once scaled, goes through a
    // 32-double LDS line ordered by row parity, so each half fetches the multipliers of ITS
    // columns with 16-byte broadcast reads; one v_readlane pair per column (the diagonal) is
    // all that is left on the scalar path.  (Broadcasting every multiplier with v_readlane
    // cost three instructions and a hazard stall per fma: 4000 instructions per block.)
    const int i = tid & 31, h = tid >> 5;
    double a[kNB / 2];
#pragma unroll
    for (int kk = 0; kk < kNB / 2; ++kk) a[kk] = sC[i][2 * kk + h];
    const double *colp = sCol + 16 * h;                   // L[2 kk + h][j] at colp[kk]
    double *mine = sCol + 16 * (i & 1) + (i >> 1);        // L[i][j]
    const double amax = __longlong_as_double((long long)model->amax_bits);
    const double tiny = (double)n1 * kEps * amax;
    double pmin = INFINITY, pmax = 0.0, myinv = 0.0;
    bool singular = false;
    // The diagonal of the NEXT column is formed ahead of the LDS round trip from two
    // v_readlanes (its own old value and the multiplier l_{j+1,j}): the square-root chain of
    // column j+1 then runs while column j's multipliers travel through LDS.  It is the same
    // fma the owning lane performs on its register copy, so both hold the same bits.
    double d = readlane_f64(a[0], 0);
#pragma unroll
    for (int j = 0; j < kNB; ++j) {
        const int hj = j & 1, jj = j >> 1;
        const bool ok = d > tiny;            // false for NaN and for a lost definiteness
        if (kb + j < n1) {
            if (!ok) singular = true;
            const double ad = fabs(d);
            pmin = ad < pmin ? ad : pmin;
            pmax = ad > pmax ? ad : pmax;
        }
        double root, inv;
        sqrt_rsqrt(ok ? d : 1.0, root, inv);
        if (!ok) inv = 0.0;
        // lanes talk through the LDS line: to the compiler a store by one lane and a load by
        // another are unrelated, so the order is pinned on both sides (no instructions: the
        // wave executes its LDS operations in program order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (h == hj) {
            const double l = i == j ? root : a[jj] * inv;
            a[jj] = l;
            *mine = l;
            if (i == j) myinv = inv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (j + 1 < kNB) {
            const int hn = (j + 1) & 1, jn = (j + 1) >> 1;
            const double lnext = readlane_f64(a[jj], j + 1 + 32 * hj);       // l_{j+1,j}
            const double dold = readlane_f64(a[jn], j + 1 + 32 * hn);         // a_{j+1,j+1} before this column
            d = fma(-lnext, lnext, dold);
        }
        const double lij = *mine;
        if (hj == 0 && h == 1) a[jj] = fma(-lij, colp[jj], a[jj]);   // column j + 1 sits in the other half
#pragma unroll
        for (int kk = jj + 1; kk < kNB / 2; ++kk) a[kk] = fma(-lij, colp[kk], a[kk]);
    }
#pragma unroll
    for (int kk = 0; kk < kNB / 2; ++kk) sC[i][2 * kk + h] = a[kk];
    if (h == (i & 1)) sInv[i] = myinv;
    if (stats && tid == 0) {
        model->iterations = kb + kNB < n1 ? kb + kNB : n1;
        if (singular) model->sing_flag = 1;
        if (pmax > 0.0 || pmin < INFINITY) {
            atomicMin((unsigned long long *)&model->pivmin_bits, (unsigned long long)__double_as_longlong(pmin));
            atomicMax((unsigned long long *)&model->pivmax_bits, (unsigned long long)__double_as_longlong(pmax));
        }
    }
}

__device__ __forceinline__ void block_store(const BatchSlot &s, int M, int lda, int kb, const double (*sC)[kLdsRow],
                                            const double *sInv, bool into_A)
{
    gdouble *A = as_global(s.A);
    gdouble *Ld = ld_block(s.ns, M, kb);
    const int tid = threadIdx.x;
    for (int e = tid; e < kNB * kNB; e += 256) {
        const int r = e & 31, c = e >> 5;
        const double l = r >= c ? sC[r][c] : sC[c][r];
        Ld[e] = r >= c ? l : 0.0;
        if (into_A) A[(size_t)(kb + c) * lda + kb + r] = l;
    }
    if (tid < kNB) Ld[kNB * kNB + tid] = sInv[tid];
}

__device__ __forceinline__ void factor_block(const BatchSlot &s, int M, int lda, int n1, int kb, int kprev,
                                             double (*sC)[kLdsRow], double (*sR)[kLdsRow], double *sInv)
{
    __shared__ __attribute__((aligned(16))) double sCol[kNB];
    block_load_update(as_global(s.A), lda, kb, kprev, sC, sR);
    if (threadIdx.x < 64) factor_wave(as_global(s.model), n1, kb, sC, sInv, sCol, true);
    __syncthreads();
    block_store(s, M, lda, kb, sC, sInv, true);
}

// the first diagonal block has no trailing update before it
__global__ __launch_bounds__(256) FD_FIT_BESIDE_EVAL void k_chol_first(const BatchSlot *tab, int M, int lda, int n1)
{
    __shared__ __attribute__((aligned(16))) double sC[kNB][kLdsRow];
    __shared__ double sInv[kNB];
    __builtin_amdgcn_s_setprio(3);
    factor_block(tab[blockIdx.z], M, lda, n1, 0, -1, sC, sC, sInv);
}

// ---- Cholesky: rows below the diagonal block ---------------------------------------------------
// Element `byte_off` of a wave-uniform column: the pointer is pinned into SGPRs so that the access
// is scalar base + one 32-bit lane offset.  Left alone, the compiler folds 32 such columns into 32
// per-lane 64-bit addresses and keeps them all alive between the loads and the stores.
__device__ __forceinline__ gdouble *pinned_column(gdouble *col, unsigned byte_off)
{
    const unsigned long long p = (unsigned long long)col;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)p);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));
    return (gdouble *)((char FD_GLOBAL *)(((unsigned long long)hi << 32) | lo) + byte_off);
}

// x <- x L11^-T in registers.  Right-looking: x_k is final once columns 0 .. k-1 have been applied;
// its update of the columns to the right is 31-k independent fmas (a dot-product form would be one
// dependent chain per element, and there is one wave per SIMD to hide it).  sL[k][c] = L11[c][k].
__device__ __forceinline__ void solve_row(double (&x)[kNB], const double (*sL)[kLdsRow], const double *sInv)
{
#pragma unroll
    for (int k = 0; k < kNB; ++k) {
        x[k] *= sInv[k];
        // eight multipliers per burst of 16-byte LDS reads: the whole column at once would not fit
        // beside x[] in one 128-VGPR slot
#pragma unroll
        for (int c0 = (k + 1) & ~7; c0 < kNB; c0 += 8) {
            double lk[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) lk[q] = sL[k][c0 + q];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (c0 + q > k) x[c0 + q] = fma(-x[k], lk[q], x[c0 + q]);
            __builtin_amdgcn_sched_barrier(0);      // keep the bursts apart: hoisted together they spill x[]
        }
    }
}

// X L11^T = A21, one row per thread, L11 and the reciprocals of its diagonal from the side store.
// Workgroups 0 .. nslab-1 take 256 matrix rows each (and mirror their result into the upper
// triangle); workgroup nslab takes the three right-hand-side rows, which live transposed in the
// RHS columns of A, and inverts L11 with its second wave.  fd_set_deltas launches that last
// workgroup alone.
__global__ __launch_bounds__(256) FD_FIT_BESIDE_EVAL void k_chol_solve(const BatchSlot *tab, int M, int lda, int npad, int npc, int k0, int nslab,
                                                                       int with_inverse)
{
    const BatchSlot &s = tab[blockIdx.z];
    gdouble *A = as_global(s.A);
    __shared__ __attribute__((aligned(16))) double sL[kNB][kLdsRow];
    __shared__ double sInv[kNB];
    const int tid = threadIdx.x;
    const bool rhs = (int)blockIdx.x == nslab;
    __builtin_amdgcn_s_setprio(3);
    gdouble *Ld = ld_block(s.ns, M, k0);

    if (!rhs) {
        // 256 matrix rows: column c of the panel is a wave-uniform pointer, the row a 32-bit lane
        // offset -- one address register for all 64 accesses.  The loads are issued before L11 is
        // staged so that they fly meanwhile.
        const int grow = k0 + kNB + (int)blockIdx.x * kSlab + tid;
        const bool active = grow < npc;
        const unsigned rowb = 8u * (unsigned)(active ? grow : k0 + kNB);      // BYTE offset: zext(u32) is what the saddr form takes
        gdouble *col0 = A + (size_t)k0 * lda;
        auto at = [&](int c) -> gdouble & {
            // the column pointer pinned into SGPRs: left alone, the compiler folds it into 32
            // per-lane 64-bit addresses and keeps them all alive for the stores
            const unsigned long long p = (unsigned long long)(col0 + (size_t)c * lda);
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)p);
            const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));
            return *(gdouble *)((char FD_GLOBAL *)(((unsigned long long)hi << 32) | lo) + rowb);
        };
        double x[kNB];
#pragma unroll
        for (int c = 0; c < kNB; ++c) x[c] = at(c);
        for (int e = tid; e < kNB * kNB; e += 256) sL[e >> 5][e & 31] = Ld[e];     // column k of L11 contiguous
        if (tid < kNB) sInv[tid] = Ld[kNB * kNB + tid];
        __syncthreads();
        if (!active) return;
        solve_row(x, sL, sInv);
#pragma unroll
        for (int c = 0; c < kNB; ++c) at(c) = x[c];
        gdouble *up = A + (size_t)grow * lda + k0;     // row k0+c of column grow: U[k0+c][grow] = L[grow][k0+c]
#pragma unroll
        for (int c = 0; c < kNB; ++c) up[c] = x[c];
        return;
    }

    // right-hand-side workgroup: threads 0..2 take the three RHS rows (contiguous in the RHS
    // columns of A); the second wave takes the rows of the identity, i.e. inverts L11
    const int unit = tid - 64;
    const bool active = tid < 3;
    const bool invert = with_inverse && unit >= 0 && unit < kNB;
    gdouble *rowp = A + (size_t)(npad + (active ? tid : 0)) * lda + k0;
    double x[kNB];
#pragma unroll
    for (int c = 0; c < kNB; ++c) x[c] = rowp[c];
    if (invert) {
#pragma unroll
        for (int c = 0; c < kNB; ++c) x[c] = c == unit ? 1.0 : 0.0;
    }
    for (int e = tid; e < kNB * kNB; e += 256) sL[e >> 5][e & 31] = Ld[e];
    if (tid < kNB) sInv[tid] = Ld[kNB * kNB + tid];
    __syncthreads();
    if (!active && !invert) return;
    solve_row(x, sL, sInv);
    if (invert) {
        // e_j L11^-T = row j of L11^-T = column j of inverse(L11): stored as [k][j]
        gdouble *inv = Ld + kLdInv;
#pragma unroll
        for (int c = 0; c < kNB; ++c) inv[c * kNB + unit] = x[c];
        return;
    }
#pragma unroll
    for (int c = 0; c < kNB; ++c) rowp[c] = x[c];
}

// ---- recover a and w ------------------------------------------------------------------------------
__device__ __forceinline__ void ns_recover_body(const BatchSlot &s, int M, int T, int npad, double *s_red /* [48] */)
{
    gcdouble *V = as_global(s.ns), *B21 = V + (size_t)8 * M, *small = V + (size_t)12 * M;
    gdouble *X = as_global(s.X);
    const int tid = threadIdx.x;
    const int n1 = M - T;

    // (B21 y)[k][c]
    double q[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) q[e] = 0.0;
    for (int j = tid; j < n1; j += 256) {
        const double y0 = X[j], y1 = X[(size_t)npad + j], y2 = X[2 * (size_t)npad + j];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double b = B21[4 * (size_t)j + k];
            q[3 * k] = fma(b, y0, q[3 * k]); q[3 * k + 1] = fma(b, y1, q[3 * k + 1]); q[3 * k + 2] = fma(b, y2, q[3 * k + 2]);
        }
    }
    block_sum_n<12>(q, s_red, tid);
    // R a = g - B21 y, row k of R sits in pivot row M-1-k and is upper triangular in (k, c)
    double a[4][3];
    for (int k = T - 1; k >= 0; --k)
        for (int c = 0; c < 3; ++c) {
            double v = small[kG + 3 * k + c] - q[3 * k + c];
            for (int cc = k + 1; cc < T; ++cc) v = fma(-small[kR + 4 * k + cc], a[cc][c], v);
            a[k][c] = v / small[kR + 4 * k + k];
        }
    // w = Q [y; 0] = H_0 .. H_{T-1} [y; 0]
    for (int i = n1 + tid; i < npad; i += 256) { X[i] = 0.0; X[(size_t)npad + i] = 0.0; X[2 * (size_t)npad + i] = 0.0; }
    __syncthreads();
    for (int k = T - 1; k >= 0; --k) {
        const int piv = M - 1 - k;
        const double tau = small[kTau + k];
        double d[3] = {0.0, 0.0, 0.0};
        for (int i = tid; i <= piv; i += 256) {
            const double v = V[4 * (size_t)i + k];
            d[0] = fma(v, X[i], d[0]); d[1] = fma(v, X[(size_t)npad + i], d[1]); d[2] = fma(v, X[2 * (size_t)npad + i], d[2]);
        }
        block_sum_n<3>(d, s_red, tid);
        for (int i = tid; i <= piv; i += 256) {
            const double v = V[4 * (size_t)i + k];
            X[i] = fma(-tau * d[0], v, X[i]);
            X[(size_t)npad + i] = fma(-tau * d[1], v, X[(size_t)npad + i]);
            X[2 * (size_t)npad + i] = fma(-tau * d[2], v, X[2 * (size_t)npad + i]);
        }
        __syncthreads();
    }
    if (tid < 3 * T) {
        const int k = tid / 3, c = tid % 3;
        X[(size_t)c * npad + M + k] = a[k][c];
    }
    // the report counts eliminated unknowns: n1 by the Cholesky, T constraints by the reflectors,
    // T coefficients through R
    if (tid == 0 && s.model->iterations >= n1) s.model->iterations = M + T;
}

__global__ __launch_bounds__(256) void k_ns_recover(const BatchSlot *tab, int M, int T, int npad)
{
    __shared__ double s_red[4 * 12];
    ns_recover_body(tab[blockIdx.z], M, T, npad, s_red);
}

// ---- back-substitution with the inverted diagonal blocks ----------------------------------------------
// L^T y = z over the rows [row_lo, row_hi), bottom up, one workgroup: z of the range lives in LDS;
// per 32-row block y_b = inverse(L_bb)^T z_b is 96 dot products instead of a 32-step dependent
// chain, then the rows above in the range take the block's contribution (the mirrored U = L^T, one
// row per thread, coalesced).  Ranges above 512 rows are chained with k_backsub_update as in the
// LU path.  Same role and data layout as fd_build.hip's k_backsub_all.
__device__ __forceinline__ void backsub_inv_body(const BatchSlot &s, int M, int lda, int npad, int row_lo, int row_hi, int recover_T,
                                                 double *s_y /* dynamic LDS, [3][w] */)
{
    gcdouble *A = as_global(s.A);
    gdouble *X = as_global(s.X);
    __shared__ double s_li[2][kNB][kNB + 1];
    __shared__ double s_x[kNB][3];
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x;
    const int w = row_hi - row_lo;

    for (int e = tid; e < 3 * w; e += 256) s_y[e] = A[(size_t)(npad + e / w) * lda + row_lo + e % w];
    double nxt[4];
    {
        gcdouble *li = ld_block(s.ns, M, row_hi - kNB) + kLdInv;
#pragma unroll
        for (int q = 0; q < 4; ++q) nxt[q] = li[tid + 256 * q];
    }
    int buf = 0;
    for (int b0 = row_hi - kNB; b0 >= row_lo; b0 -= kNB, buf ^= 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int e = tid + 256 * q; s_li[buf][e >> 5][e & 31] = nxt[q]; }
        if (b0 - kNB >= row_lo) {
            gcdouble *li = ld_block(s.ns, M, b0 - kNB) + kLdInv;
#pragma unroll
            for (int q = 0; q < 4; ++q) nxt[q] = li[tid + 256 * q];
        }
        // my first row's segment of U for the update below (does not depend on y)
        const int i0 = row_lo + tid;
        double uik[kNB];
        gcdouble *ub = A + (size_t)b0 * lda;       // wave-uniform base, 32-bit offsets below
        if (i0 < b0) {
#pragma unroll
            for (int k = 0; k < kNB; ++k) uik[k] = (ub + (size_t)k * lda)[(unsigned)i0];
        }
        __syncthreads();                 // the inverse block and the z rows of this block are in LDS
        const int l0 = b0 - row_lo;
        if (tid < 96) {
            const int i = tid & 31, c = tid >> 5;
            double acc = 0.0;
#pragma unroll 8
            for (int k = 0; k < kNB; ++k) acc = fma(s_li[buf][k][i], s_y[c * w + l0 + k], acc);   // zeros above the diagonal
            s_x[i][c] = acc;
        }
        __syncthreads();
        if (tid < 96) { const int i = tid & 31, c = tid >> 5; s_y[c * w + l0 + i] = s_x[i][c]; }
        for (int i = i0; i < b0; i += 256) {
            if (i != i0) {
#pragma unroll
                for (int k = 0; k < kNB; ++k) uik[k] = (ub + (size_t)k * lda)[(unsigned)i];
            }
            double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
            for (int k = 0; k < kNB; ++k) {
                const double x0 = s_x[k][0], x1 = s_x[k][1], x2 = s_x[k][2];
                a0 = fma(uik[k], x0, a0);
                a1 = fma(uik[k], x1, a1);
                a2 = fma(uik[k], x2, a2);
                // the scheduler would otherwise hoist all 96 LDS reads and spill the U row
                if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
            const int li = i - row_lo;
            s_y[li] -= a0; s_y[w + li] -= a1; s_y[2 * w + li] -= a2;
        }
        // the next iteration's first barrier orders these writes before its products; s_x is not
        // rewritten before that barrier, and s_li alternates
    }
    __syncthreads();
    for (int e = tid; e < 3 * w; e += 256) X[(size_t)(e / w) * npad + row_lo + e % w] = s_y[e];
    if (recover_T >= 0) {                     // whole system in one call: the polynomial and w = Q [y; 0] right away
        __threadfence_block();
        __syncthreads();
        ns_recover_body(s, M, recover_T, npad, &s_li[0][0][0]);
    }
}

__global__ __launch_bounds__(256) FD_FIT_BESIDE_EVAL void k_backsub_inv(const BatchSlot *tab, int M, int lda, int npad, int row_lo, int row_hi,
                                                                        int recover_T)
{
    extern __shared__ __attribute__((aligned(16))) double s_y_dyn[];   // [3][w]
    backsub_inv_body(tab[blockIdx.z], M, lda, npad, row_lo, row_hi, recover_T, s_y_dyn);
}

// ---- Cholesky: trailing update ---------------------------------------------------------------
// A22 -= L21 L21^T on the 16 x 16 tiles at and below the diagonal, plus the right-hand-side tile
// of every 16-column block (three live rows, stored transposed in the RHS columns), with
// v_mfma_f64_16x16x4_f64.  Workgroup (cb, chunk) takes every (4 * nchunk)-th tile of column block
// cb, two in flight per wave.  The LAST workgroup is different: it updates the next diagonal block
// by itself and factorises it while the others are busy -- the only sequential part of the
// factorisation runs beside the trailing update instead of after it.  (The tiles of that block
// are skipped by everybody else.)  rhs_only (fd_set_deltas): the right-hand-side tiles alone.
__global__ __launch_bounds__(256) FD_FIT_BESIDE_EVAL void k_chol_trail(const BatchSlot *tab, int M, int lda, int npad, int npc, int n1, int k0,
                                                    int nchunk, int rhs_only)
{
    const BatchSlot &slot = tab[blockIdx.z];
    gdouble *A = as_global(slot.A);
    const int tid = threadIdx.x;
    __builtin_amdgcn_s_setprio(3);
    if (!rhs_only && blockIdx.x == gridDim.x - 1) {
        __shared__ __attribute__((aligned(16))) double sC[kNB][kLdsRow];
        __shared__ __attribute__((aligned(16))) double sR[kNB][kLdsRow];
        __shared__ double sInv[kNB];
        factor_block(slot, M, lda, n1, k0 + kNB, k0, sC, sR, sInv);
        return;
    }
    const int lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int cb = (int)blockIdx.x / nchunk, chunk = (int)blockIdx.x % nchunk;
    const int c0 = k0 + kNB + cb * 16;
    constexpr int S = kNB / 4;

    double u[S];
#pragma unroll
    for (int s = 0; s < S; ++s) u[s] = -A[(size_t)(k0 + g + 4 * s) * lda + c0 + c];

    const int mtiles = (npc - c0) / 16;               // matrix tiles of this column block; index mtiles = RHS tile
    // tiles of the next diagonal block belong to the factorising workgroup
    const int skip = rhs_only ? mtiles : (cb == 0 ? 2 : (cb == 1 ? 1 : 0));
    const int stride = 4 * nchunk;
    for (int t0 = skip + chunk * 4 + wave; t0 <= mtiles; t0 += 2 * stride) {
        const int t1 = t0 + stride;
        const bool two = t1 <= mtiles;
        gdouble *cp[2];
        size_t rs[2];
        double av[2][S];
        double4_t acc[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int t = q == 0 ? t0 : (two ? t1 : t0);
            if (t < mtiles) {
                const int r0 = c0 + t * 16;
                cp[q] = A + (size_t)(c0 + g) * lda + r0 + c;      // transposed accumulator: 16 consecutive rows per load
                rs[q] = (size_t)4 * lda;
                gcdouble *aptr = A + (size_t)(k0 + g) * lda + r0 + c;
#pragma unroll
                for (int s = 0; s < S; ++s) av[q][s] = aptr[(size_t)(4 * s) * lda];
            } else {
                cp[q] = A + (size_t)(npad + c) * lda + c0 + g;
                rs[q] = 4;
                gcdouble *aptr = A + (size_t)(npad + c) * lda + k0 + g;
#pragma unroll
                for (int s = 0; s < S; ++s) av[q][s] = aptr[4 * s];
            }
            acc[q][0] = cp[q][0]; acc[q][1] = cp[q][rs[q]]; acc[q][2] = cp[q][2 * rs[q]]; acc[q][3] = cp[q][3 * rs[q]];
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(u[s], av[0][s], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(u[s], av[1][s], acc[1], 0, 0, 0);
        }
        cp[0][0] = acc[0][0]; cp[0][rs[0]] = acc[0][1]; cp[0][2 * rs[0]] = acc[0][2]; cp[0][3 * rs[0]] = acc[0][3];
        if (two) { cp[1][0] = acc[1][0]; cp[1][rs[1]] = acc[1][1]; cp[1][2 * rs[1]] = acc[1][2]; cp[1][3 * rs[1]] = acc[1][3]; }
    }
}

// ---- Cholesky: one fused step --------------------------------------------------------------------
// Panel k0 is solved; this kernel does everything up to and including the solve of panel
// kb = k0 + 32 -- one launch per 32 columns instead of two, and the factor never leaves LDS:
//   panel workgroups (blockIdx.x < npanel; 192 rows each, the last one the right-hand sides):
//     - the diagonal block (kb, kb) with panel k0's update applied, in LDS, EVERY workgroup for
//       itself: the redundant factorisations run side by side, nobody waits for anybody;
//     - wave 0 factorises it while waves 1..3 apply panel k0 to the workgroup's own rows of
//       columns kb .. kb+31 (one A operand serves both 16-column blocks);
//     - then X L11^T = A21 on those rows, L11 read from LDS.  The right-hand-side workgroup also
//       files L11 in the side store and inverts it (for fd_set_deltas and the back-substitution);
//   the other workgroups: the trailing update of the columns from kb + 32 on, as k_chol_trail.
// Nothing here waits on a flag: every dependency is inside one workgroup or across the launch.
__global__ __launch_bounds__(256) FD_FIT_BESIDE_EVAL void k_chol_step(const BatchSlot *tab, int M, int lda, int npad, int npc, int n1,
                                                                      int k0, int nchunk, int npanel, int bulk_every)
{
    const BatchSlot &slot = tab[blockIdx.z];
    gdouble *A = as_global(slot.A);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int kb = k0 + kNB;
    constexpr int S = kNB / 4;
    __builtin_amdgcn_s_setprio(3);
    // One dynamic LDS block, used either way round: the panel workgroups keep the diagonal block in
    // it, the others the B operands of their pending panels (8 KB per panel).  Small systems
    // (bulk_every = 1) then need no more LDS than the panel work -- a build that runs beside an
    // evaluation has to find its LDS between that kernel's workgroups as well as its registers.
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];

    if ((int)blockIdx.x >= npanel) {
        // Trailing update of a PAIR of 16-column blocks (32 columns from kb + 32 on) in 32 x 32
        // macro tiles: two A operands (16 rows each) against two B operands give four MFMA chains,
        // so every operand byte loaded is used twice and a wave has 16 KB in flight.  The last
        // "tile" of a column pair is the right-hand sides (16 rows, of which three live).
        //
        // Deep updates: only every kBulk-th step touches the bulk of the trailing matrix, with all
        // kBulk panels since the last time in one pass (rank 128: each C tile is read and written
        // once per 128 columns instead of once per 32); the steps in between bring just the column
        // pair that becomes the next panel but one up to date, with the panels pending for it.
        // (The next panel's own columns get every panel on time from the panel workgroups.)  The
        // B operands of all pending panels wait in LDS.
        const int cbid = (int)blockIdx.x - npanel;
        const int cpair = 1 + cbid / nchunk, chunk = cbid % nchunk;
        const int c0 = kb + 32 * cpair;
        // panels pending for these columns: all since the last bulk step (bulk_every steps apart)
        const int depth = ((k0 >> 5) % bulk_every) + 1;
        double (*sU)[2][S][64] = reinterpret_cast<double (*)[2][S][64]>(dyn_lds);
        for (int q = wave; q < 2 * depth; q += 4) {
            const int pp = q >> 1, hh = q & 1;
            const int kp = k0 - 32 * (depth - 1 - pp);
#pragma unroll
            for (int s = 0; s < S; ++s) sU[pp][hh][s][lane] = -A[(size_t)(kp + g + 4 * s) * lda + c0 + 16 * hh + c];
        }
        __syncthreads();
        const int mt = (npc - c0) / 32;                   // macro tiles of this column pair; index mt = RHS
        const size_t cs = (size_t)4 * lda;
        for (int t = chunk * 4 + wave; t <= mt; t += 4 * nchunk) {
            if (t < mt) {
                const int r0 = c0 + 32 * t;
                gdouble *p00 = A + (size_t)(c0 + g) * lda + r0 + c;      // rows r0.., columns c0.. (accumulator transposed)
                gdouble *p01 = p00 + (size_t)16 * lda;                    // columns c0 + 16..
                gdouble *p10 = p00 + 16, *p11 = p01 + 16;                 // rows r0 + 16..
                double4_t a00, a01, a10, a11;
#pragma unroll
                for (int r = 0; r < 4; ++r) { a00[r] = p00[r * cs]; a01[r] = p01[r * cs]; a10[r] = p10[r * cs]; a11[r] = p11[r * cs]; }
                for (int pp = 0; pp < depth; ++pp) {
                    const int kp = k0 - 32 * (depth - 1 - pp);
                    gcdouble *aptr = A + (size_t)(kp + g) * lda + r0 + c;
                    double av0[S], av1[S];
#pragma unroll
                    for (int s = 0; s < S; ++s) { av0[s] = aptr[(size_t)(4 * s) * lda]; av1[s] = aptr[(size_t)(4 * s) * lda + 16]; }
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const double ua = sU[pp][0][s][lane], ub = sU[pp][1][s][lane];
                        a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(ua, av0[s], a00, 0, 0, 0);
                        a01 = __builtin_amdgcn_mfma_f64_16x16x4f64(ub, av0[s], a01, 0, 0, 0);
                        a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(ua, av1[s], a10, 0, 0, 0);
                        a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(ub, av1[s], a11, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) { p00[r * cs] = a00[r]; p01[r * cs] = a01[r]; p10[r * cs] = a10[r]; p11[r * cs] = a11[r]; }
            } else {
                gdouble *p0 = A + (size_t)(npad + c) * lda + c0 + g, *p1 = p0 + 16;
                double4_t a0, a1;
#pragma unroll
                for (int r = 0; r < 4; ++r) { a0[r] = p0[4 * r]; a1[r] = p1[4 * r]; }
                for (int pp = 0; pp < depth; ++pp) {
                    const int kp = k0 - 32 * (depth - 1 - pp);
                    gcdouble *aptr = A + (size_t)(npad + c) * lda + kp + g;
                    double av[S];
#pragma unroll
                    for (int s = 0; s < S; ++s) av[s] = aptr[4 * s];
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(sU[pp][0][s][lane], av[s], a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(sU[pp][1][s][lane], av[s], a1, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) { p0[4 * r] = a0[r]; p1[4 * r] = a1[r]; }
            }
        }
        return;
    }

    // ---- panel workgroup: wave 0 factorises; waves 1..3 own 64 rows each -- they apply panel k0 to
    // them, fetch them back into registers (no workgroup barrier in between: a wave reads what it
    // wrote) and solve them once L11 is there
    double (*sC)[kLdsRow] = reinterpret_cast<double (*)[kLdsRow]>(dyn_lds);
    double (*sR)[kLdsRow] = reinterpret_cast<double (*)[kLdsRow]>(dyn_lds + kNB * kLdsRow);
    double *sCol = dyn_lds + 2 * kNB * kLdsRow;
    double *sInv = sCol + kNB;
    const bool rhs = (int)blockIdx.x == npanel - 1;
    const int slab0 = kb + kNB + (int)blockIdx.x * kStepSlab;   // first row of a matrix slab
    const bool first = k0 < 0;                // the first block: no panel before it, nothing to apply
    // operands of the next panel's two 16-column blocks: rows kb.. / kb+16.. of panel k0, negated
    double u0[S], u1[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        u0[s] = first ? 0.0 : -A[(size_t)(k0 + g + 4 * s) * lda + kb + c];
        u1[s] = first ? 0.0 : -A[(size_t)(k0 + g + 4 * s) * lda + kb + 16 + c];
    }
    // the diagonal block (kb, kb) with panel k0 applied, into LDS: its three lower 16 x 16 tiles by
    // waves 1..3, one MFMA chain each -- the row operand of a tile is the (negated) column operand
    // of its row half, so nothing else has to be loaded; wave 0 clears the tile above the diagonal
    if (wave == 0) {
        for (int e = lane; e < 256; e += 64) sC[e & 15][16 + (e >> 4)] = 0.0;
    } else {
        const int ta = wave == 1 ? 0 : 1, tb = wave == 3 ? 1 : 0;          // tile (row half, column half)
        gcdouble *cp = A + (size_t)(kb + 16 * tb + g) * lda + kb + 16 * ta + c;   // transposed accumulator
        const size_t cs = (size_t)4 * lda;
        double4_t acc;
        acc[0] = cp[0]; acc[1] = cp[cs]; acc[2] = cp[2 * cs]; acc[3] = cp[3 * cs];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const double colop = tb ? u1[s] : u0[s];
            const double rowop = -(ta ? u1[s] : u0[s]);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(colop, rowop, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sC[16 * ta + c][16 * tb + g + 4 * r] = acc[r];
    }
    __syncthreads();

    double x[kNB];
    bool active = false, invert = false;
    const int unit = tid - 128;               // RHS workgroup, third wave: row `unit` of the identity
    const int grow = slab0 + (tid - 64);      // matrix slab: waves 1..3 <-> rows slab0 .. slab0 + 191
    if (wave == 0) {
        factor_wave(as_global(slot.model), n1, kb, sC, sInv, sCol, rhs);
        // L11^T for the solves: column k of L11 contiguous, zeros above the diagonal
        const int i = tid & 31, h = tid >> 5;
#pragma unroll
        for (int kk = 0; kk < kNB / 2; ++kk) { const int k = 2 * kk + h; sR[k][i] = i >= k ? sC[i][k] : 0.0; }
    } else {
        const int wrow0 = slab0 + (wave - 1) * 64;             // my wave's first row
        const int ntr = first ? 0 : (rhs ? (wave == 1 ? 1 : 0) : ((npc - wrow0 < 64 ? (npc - wrow0 > 0 ? npc - wrow0 : 0) : 64) / 16));
        for (int rt = 0; rt < ntr; ++rt) {
            gdouble *cp0, *cp1;
            size_t rs;
            double av[S];
            if (!rhs) {
                const int r0 = wrow0 + rt * 16;
                cp0 = A + (size_t)(kb + g) * lda + r0 + c;
                cp1 = cp0 + (size_t)16 * lda;
                rs = (size_t)4 * lda;
                gcdouble *aptr = A + (size_t)(k0 + g) * lda + r0 + c;
#pragma unroll
                for (int s = 0; s < S; ++s) av[s] = aptr[(size_t)(4 * s) * lda];
            } else {
                cp0 = A + (size_t)(npad + c) * lda + kb + g;
                cp1 = cp0 + 16;
                rs = 4;
                gcdouble *aptr = A + (size_t)(npad + c) * lda + k0 + g;
#pragma unroll
                for (int s = 0; s < S; ++s) av[s] = aptr[4 * s];
            }
            double4_t acc0, acc1;
            acc0[0] = cp0[0]; acc0[1] = cp0[rs]; acc0[2] = cp0[2 * rs]; acc0[3] = cp0[3 * rs];
            acc1[0] = cp1[0]; acc1[1] = cp1[rs]; acc1[2] = cp1[2 * rs]; acc1[3] = cp1[3 * rs];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(u0[s], av[s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(u1[s], av[s], acc1, 0, 0, 0);
            }
            cp0[0] = acc0[0]; cp0[rs] = acc0[1]; cp0[2 * rs] = acc0[2]; cp0[3 * rs] = acc0[3];
            cp1[0] = acc1[0]; cp1[rs] = acc1[1]; cp1[2 * rs] = acc1[2]; cp1[3 * rs] = acc1[3];
        }
        // my wave's rows are complete in memory: other lanes of this wave wrote them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (!rhs) {
            active = grow < npc;
            if (active) {
#pragma unroll
                for (int cc = 0; cc < kNB; ++cc) x[cc] = *pinned_column(A + (size_t)(kb + cc) * lda, 8u * (unsigned)grow);
            }
        } else if (wave == 1) {
            active = lane < 3;
            gcdouble *rowp = A + (size_t)(npad + (active ? lane : 0)) * lda + kb;
#pragma unroll
            for (int cc = 0; cc < kNB; ++cc) x[cc] = rowp[cc];
        } else if (wave == 2) {
            invert = unit >= 0 && unit < kNB;
#pragma unroll
            for (int cc = 0; cc < kNB; ++cc) x[cc] = cc == unit ? 1.0 : 0.0;
        }
    }
    __syncthreads();                          // L11 (sC), L11^T (sR) and 1 / diag (sInv) are in LDS
    if (rhs) block_store(slot, M, lda, kb, sC, sInv, false);
    if (!active && !invert) return;
    solve_row(x, sR, sInv);
    if (invert) {
        gdouble *inv = ld_block(slot.ns, M, kb) + kLdInv;
#pragma unroll
        for (int cc = 0; cc < kNB; ++cc) inv[cc * kNB + unit] = x[cc];
        return;
    }
    if (rhs) {
        gdouble *rowp = A + (size_t)(npad + lane) * lda + kb;
#pragma unroll
        for (int cc = 0; cc < kNB; ++cc) rowp[cc] = x[cc];
        return;
    }
#pragma unroll
    for (int cc = 0; cc < kNB; ++cc) *pinned_column(A + (size_t)(kb + cc) * lda, 8u * (unsigned)grow) = x[cc];
    gdouble *up = A + (size_t)grow * lda + kb;             // mirrored: U[kb+cc][grow] = L[grow][kb+cc]
#pragma unroll
    for (int cc = 0; cc < kNB; ++cc) up[cc] = x[cc];
}

// ---- the whole build of a small system in ONE launch of ONE workgroup (FD_SOLVER_ONE_WORKGROUP; FD_SMALL_BUILD=1) ----
// Slower than the chain for a lone build, the better citizen for batches solved beside a running evaluation: see
// launch_build_spd for the measurements.
// A lone order-256 system is 5.6 MFLOP of fp64 work that the chain above spreads over 18 dependent
// launches: 0.24 ms, most of it launch floor and dispatch.  Up to order 512
// everything after the assembly of K -- reflectors, Y = K V, W, the rotation, every Cholesky step,
// back-substitution, recovery of the polynomial, packing of the evaluation records -- runs here in
// one workgroup of four waves (one per SIMD: the fp64 matrix pipe of a CU is 128 flop / clock, and
// 1/3 n^3 flop of an order-256 system are 20 us of it), with workgroup barriers where the chain had
// kernel boundaries.  (The assembly stays a grid of its own: 65 536 fp64 logarithms are ~70 us on
// one CU and 8 us on sixty-four.)  A batch is a grid of such workgroups, one per model: the same
// code, so single and batched builds agree bit for bit, and 32 models cost what one does.
//
// Per 32 columns (panel k0 solved, block kb = k0 + 32 next):
//   (i)   all four waves apply panel k0 to the columns kb .. kb+31 (rows >= kb, right-hand sides),
//   (ii)  wave 0 factorises the diagonal block (kb, kb) -- the only sequential piece, 3.5 us --
//         WHILE waves 1..3 apply panel k0 to everything right of those columns (look-ahead),
//   (iii) all waves solve the rows below the block, one row per thread; the right-hand sides ride
//         along as three more rows, the block's inverse is a by-product (back-substitution).
// The right-hand sides see exactly the operations, operands and order of k_chol_solve /
// k_chol_trail (fd_set_deltas), so new deltas through the stored factor stay bit-identical to a rebuild.
constexpr int kSmallMaxNpc = 512;

// C -= L_r L_c^T on one 32 x 32 macro tile with panel kp, by one wave: rows r0 .., columns c0 .. of
// the trailing matrix (lower triangle, accumulator transposed as in k_chol_step), or -- rhs_tile --
// the three right-hand-side rows against the columns c0 .. c0+31
__device__ __forceinline__ void small_macro_tile(gdouble *A, int lda, int npad, int kp, int c0, int r0, bool rhs_tile, int lane)
{
    constexpr int S = kNB / 4;
    const int c = lane & 15, g = lane >> 4;
    double ua[S], ub[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        ua[s] = -A[(size_t)(kp + g + 4 * s) * lda + c0 + c];
        ub[s] = -A[(size_t)(kp + g + 4 * s) * lda + c0 + 16 + c];
    }
    if (!rhs_tile) {
        const size_t cs = (size_t)4 * lda;
        gdouble *p00 = A + (size_t)(c0 + g) * lda + r0 + c;
        gdouble *p01 = p00 + (size_t)16 * lda;
        gdouble *p10 = p00 + 16, *p11 = p01 + 16;
        double4_t a00, a01, a10, a11;
#pragma unroll
        for (int r = 0; r < 4; ++r) { a00[r] = p00[r * cs]; a01[r] = p01[r * cs]; a10[r] = p10[r * cs]; a11[r] = p11[r * cs]; }
        gcdouble *aptr = A + (size_t)(kp + g) * lda + r0 + c;
        double av0[S], av1[S];
#pragma unroll
        for (int s = 0; s < S; ++s) { av0[s] = aptr[(size_t)(4 * s) * lda]; av1[s] = aptr[(size_t)(4 * s) * lda + 16]; }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[s], av0[s], a00, 0, 0, 0);
            a01 = __builtin_amdgcn_mfma_f64_16x16x4f64(ub[s], av0[s], a01, 0, 0, 0);
            a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[s], av1[s], a10, 0, 0, 0);
            a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(ub[s], av1[s], a11, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { p00[r * cs] = a00[r]; p01[r * cs] = a01[r]; p10[r * cs] = a10[r]; p11[r * cs] = a11[r]; }
    } else {
        gdouble *p0 = A + (size_t)(npad + c) * lda + c0 + g, *p1 = p0 + 16;
        double4_t a0, a1;
#pragma unroll
        for (int r = 0; r < 4; ++r) { a0[r] = p0[4 * r]; a1[r] = p1[4 * r]; }
        gcdouble *aptr = A + (size_t)(npad + c) * lda + kp + g;
        double av[S];
#pragma unroll
        for (int s = 0; s < S; ++s) av[s] = aptr[4 * s];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[s], av[s], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ub[s], av[s], a1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { p0[4 * r] = a0[r]; p1[4 * r] = a1[r]; }
    }
}

// (one wave per SIMD and no register cap: phases with 32-double rows plus MFMA tiles in flight spilled 237 registers under the 128 of FD_FIT_BESIDE_EVAL)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_build_small(const BatchSlot *tab, int M, int T, int npad, int lda, int kind, int Mpad,
                                                                        unsigned long long *stamps)
{
    const BatchSlot &slot = tab[blockIdx.z];
    // diagnostics (FD_SMALL_STAMPS): shader-clock stamps of the phases, workgroup 0, thread 0
    unsigned long long st_prev = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    int st_k = 0;
#define FD_BSTAMP() if (stamps && blockIdx.z == 0 && threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamps[st_k++] = t_ - st_prev; st_prev = t_; }
    gdouble *A = as_global(slot.A);
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n1 = M - T, npc = round_up_dev(n1, kNB);
    __builtin_amdgcn_s_setprio(3);

    // ---- projection: reflectors of P (and Q^T f), Y = K V, W, B = K - V W^T - W V^T
    if (T > 0) {
        ns_reflectors_body(slot, M, T, npad, lda, 1);
        __syncthreads();
        FD_BSTAMP()
        for (int vb = 0; vb < (M + 63) / 64; ++vb) { ns_kv_body(slot, M, lda, vb); __syncthreads(); }
        FD_BSTAMP()
        ns_w_body(slot, M);
        __syncthreads();
        FD_BSTAMP()
        const int g32 = ((npc > M ? npc : M) + 31) / 32;
        for (int tj = 0; tj < g32; ++tj)
            for (int ti = tj; ti < g32; ++ti) ns_rotate_body(slot, M, T, npc, lda, ti, tj);
        __syncthreads();
        FD_BSTAMP()
    }

    // ---- blocked Cholesky with the right-hand sides as three more rows
    double (*sC)[kLdsRow] = reinterpret_cast<double (*)[kLdsRow]>(dyn_lds);
    double (*sR)[kLdsRow] = reinterpret_cast<double (*)[kLdsRow]>(dyn_lds + kNB * kLdsRow);
    double *sCol = dyn_lds + 2 * kNB * kLdsRow;
    double *sInv = sCol + kNB;
    for (int k0 = -kNB; k0 + kNB < npc; k0 += kNB) {
        const int kb = k0 + kNB;
        const int nbelow = (npc - kb) / kNB;          // 32-row blocks from kb down (the diagonal block included)
        if (k0 >= 0) {
            // (i) panel k0 onto the next panel's columns: macro tiles (kb + 32 t, kb), t = 0 .. nbelow - 1, + the RHS tile
            for (int t = wave; t <= nbelow; t += 4) small_macro_tile(A, lda, npad, k0, kb, kb + kNB * t, t == nbelow, lane);
            __syncthreads();
        }
        if (k0 <= 0) FD_BSTAMP()
        // (ii) wave 0: the diagonal block; waves 1..3: panel k0 onto the columns right of kb + 31
        if (wave == 0) {
            for (int e = lane; e < kNB * kNB; e += 64) {
                const int r = e & 31, c = e >> 5;
                sC[r][c] = A[(size_t)(kb + c) * lda + kb + r];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            factor_wave(as_global(slot.model), n1, kb, sC, sInv, sCol, true);
            const int i = tid & 31, h = tid >> 5;      // L11^T for the solves: column k of L11 contiguous, zeros above the diagonal
#pragma unroll
            for (int kk = 0; kk < kNB / 2; ++kk) { const int k = 2 * kk + h; sR[k][i] = i >= k ? sC[i][k] : 0.0; }
        } else if (k0 >= 0) {
            // column pairs c0 = kb + 32 p, p = 1 .. nbelow - 1; pair p has nbelow - p macro tiles + the RHS tile
            int idx = wave - 1;
            for (int pcol = 1; pcol < nbelow; ++pcol) {
                const int c0 = kb + kNB * pcol, ntile = nbelow - pcol;
                for (int t = 0; t <= ntile; ++t, ++idx) {
                    if (idx % 3 != 0) continue;
                    small_macro_tile(A, lda, npad, k0, c0, c0 + kNB * t, t == ntile, lane);
                }
                idx %= 3;
            }
        }
        __syncthreads();
        if (k0 <= 0) FD_BSTAMP()
        // (iii) L11 to the side store and into A, its inverse, and X L11^T = A21 on the rows below
        block_store(slot, M, lda, kb, sC, sInv, true);
        // Every thread takes ONE vector through the 32 dependent columns of L11 (solve_row: ~20 k cycles whatever the
        // vector): a row of A21, or -- threads that have no row left at this step -- a row of the identity (-> the
        // block's inverse, for the back-substitution) or one of the three right-hand sides.  Side by side, not one
        // after the other: the same operations on the same operands, so the same bits, in half the time of a step.
        const int nrows = npc - kb - kNB;
        const bool spare = 256 - (nrows < 256 ? nrows : 256) >= kNB + 3;        // 35 idle threads in the (only) pass over the rows
        for (int row0 = kb + kNB, pass = 0; row0 < npc || (pass == 0 && spare); row0 += 256, ++pass) {
            const int row = row0 + tid;
            const int extra = tid - nrows;                                      // >= 0: no row in the first pass
            const bool is_row = row < npc;
            const bool is_inv = pass == 0 && spare && extra >= 0 && extra < kNB;
            const bool is_rhs = pass == 0 && spare && extra >= kNB && extra < kNB + 3;
            if (!(is_row || is_inv || is_rhs)) continue;
            gdouble *rowp = A + (size_t)(npad + (is_rhs ? extra - kNB : 0)) * lda + kb;
            double x[kNB];
#pragma unroll
            for (int cc = 0; cc < kNB; ++cc)
                x[cc] = is_row ? *pinned_column(A + (size_t)(kb + cc) * lda, 8u * (unsigned)row) : (is_inv ? (cc == extra ? 1.0 : 0.0) : rowp[cc]);
            solve_row(x, sR, sInv);
            if (is_row) {
#pragma unroll
                for (int cc = 0; cc < kNB; ++cc) *pinned_column(A + (size_t)(kb + cc) * lda, 8u * (unsigned)row) = x[cc];
                gdouble *up = A + (size_t)row * lda + kb;          // mirrored: U[kb + cc][row] = L[row][kb + cc]
#pragma unroll
                for (int cc = 0; cc < kNB; ++cc) up[cc] = x[cc];
            } else if (is_inv) {
                gdouble *inv = ld_block(slot.ns, M, kb) + kLdInv;
#pragma unroll
                for (int cc = 0; cc < kNB; ++cc) inv[cc * kNB + extra] = x[cc];
            } else {
#pragma unroll
                for (int cc = 0; cc < kNB; ++cc) rowp[cc] = x[cc];
            }
        }
        if (!spare) {
            const int unit = tid - 64;                         // second wave: row `unit` of the identity -> inverse(L11)
            const bool rhsrow = tid < 3, invert = unit >= 0 && unit < kNB;
            if (rhsrow || invert) {
                gdouble *rowp = A + (size_t)(npad + (rhsrow ? tid : 0)) * lda + kb;
                double x[kNB];
#pragma unroll
                for (int cc = 0; cc < kNB; ++cc) x[cc] = invert ? (cc == unit ? 1.0 : 0.0) : rowp[cc];
                solve_row(x, sR, sInv);
                if (invert) {
                    gdouble *inv = ld_block(slot.ns, M, kb) + kLdInv;
#pragma unroll
                    for (int cc = 0; cc < kNB; ++cc) inv[cc * kNB + unit] = x[cc];
                } else {
#pragma unroll
                    for (int cc = 0; cc < kNB; ++cc) rowp[cc] = x[cc];
                }
            }
        }
        __syncthreads();
        if (k0 <= 0) FD_BSTAMP()
    }
    FD_BSTAMP()

    // ---- L^T y = z, the polynomial from R, w = Q [y; 0]; then the evaluation records
    backsub_inv_body(slot, M, lda, npad, 0, npc, T, dyn_lds);
    __syncthreads();
    FD_BSTAMP()
    packing::pack_body(slot, npad, M, Mpad, T, kind, 0, 0);
    if (kind == FD_KERNEL_THIN_PLATE) {
        __syncthreads();
        for (int tile = wave; tile < Mpad / 16; tile += 4) packing::pack_tiles_body(slot, Mpad, tile, lane);
    }
    FD_BSTAMP()
#undef FD_BSTAMP
}

// ---- multilayer Gaussian model (FD_KERNEL_GAUSSIAN_ML) ---------------------------------------------
// The SOP's model = 1, alglib::rbfsetalgomultilayer(model, radius, layers, lambda)
// (reference src/SOP_FaceDeform.cpp:346-348), in dense form: the term's polynomial is fitted to
// the deltas FIRST, by least squares, and removed (ALGLIB's order; the other kinds solve it
// together with the weights); then layer l = 0 .. L-1 fits what is left with Gaussians of radius
// R / 2^l on every centre,
//     (Phi_l + lambda I) w_l = r_l,      r_{l+1} = r_l - Phi_l w_l  (= lambda w_l),
// each a symmetric positive definite system: the Cholesky kernels above with nothing to project.
// The solved model is M * L Gaussian records with their own radii plus the polynomial -- exactly
// what the per-centre-radius evaluation kernel takes.

// Least-squares polynomial through the reflectors of P (k_ns_reflectors): a = R^-1 (Q^T f)_pivot rows;
// f <- f - P a in the right-hand-side columns.  X is scratch for Q^T f.
__global__ __launch_bounds__(256) void k_ml_affine(const BatchSlot *tab, int M, int T, int npad, int lda)
{
    const BatchSlot &s = tab[blockIdx.z];
    gcdouble *V = as_global(s.ns), *centres = as_global(s.centres);
    gdouble *small = as_global(s.ns) + (size_t)12 * M;
    gdouble *f0 = as_global(s.A) + (size_t)npad * lda, *f1 = f0 + lda, *f2 = f1 + lda;
    gdouble *x0 = as_global(s.X), *x1 = x0 + npad, *x2 = x1 + npad;
    __shared__ double s_red[4 * 3];
    const int tid = threadIdx.x;
    for (int i = tid; i < M; i += 256) { x0[i] = f0[i]; x1[i] = f1[i]; x2[i] = f2[i]; }
    __syncthreads();
    for (int k = 0; k < T; ++k) {
        const int piv = M - 1 - k;
        const double tau = small[kTau + k];
        double d[3] = {0.0, 0.0, 0.0};
        for (int i = tid; i <= piv; i += 256) {
            const double v = V[4 * (size_t)i + k];
            d[0] = fma(v, x0[i], d[0]); d[1] = fma(v, x1[i], d[1]); d[2] = fma(v, x2[i], d[2]);
        }
        block_sum_n<3>(d, s_red, tid);
        for (int i = tid; i <= piv; i += 256) {
            const double v = V[4 * (size_t)i + k];
            x0[i] = fma(-tau * d[0], v, x0[i]); x1[i] = fma(-tau * d[1], v, x1[i]); x2[i] = fma(-tau * d[2], v, x2[i]);
        }
        __syncthreads();
    }
    double a[4][3] = {};
    for (int k = T - 1; k >= 0; --k)
        for (int c = 0; c < 3; ++c) {
            double v = (c == 0 ? x0 : (c == 1 ? x1 : x2))[M - 1 - k];
            for (int cc = k + 1; cc < T; ++cc) v = fma(-small[kR + 4 * k + cc], a[cc][c], v);
            a[k][c] = v / small[kR + 4 * k + k];
        }
    for (int i = tid; i < M; i += 256) {
        const double px = centres[3 * (size_t)i], py = centres[3 * (size_t)i + 1], pz = centres[3 * (size_t)i + 2];
        f0[i] -= fma(a[3][0], pz, fma(a[2][0], py, fma(a[1][0], px, a[0][0])));
        f1[i] -= fma(a[3][1], pz, fma(a[2][1], py, fma(a[1][1], px, a[0][1])));
        f2[i] -= fma(a[3][2], pz, fma(a[2][2], py, fma(a[1][2], px, a[0][2])));
    }
    if (tid < 12) small[kAff + tid] = a[tid / 3][tid % 3];
}

// the records of every layer: centre j again, radius R / 2^l
__global__ void k_ml_setup(const BatchSlot *tab, int M, int L, double R)
{
    const BatchSlot &s = tab[blockIdx.z];
    gdouble *centres = as_global(s.centres), *radii = as_global(s.radii);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * L) return;
    const int l = idx / M, j = idx - l * M;
    if (l > 0)
        for (int c = 0; c < 3; ++c) centres[3 * (size_t)idx + c] = centres[3 * (size_t)j + c];
    radii[idx] = ldexp(R, -l);
}

// layer l solved (X): its weights into the model, lambda w_l as the next layer's right-hand sides
__global__ void k_ml_layer(const BatchSlot *tab, int M, int npad, int lda, int l, int last, double lambda)
{
    const BatchSlot &s = tab[blockIdx.z];
    gcdouble *X = as_global(s.X);
    gdouble *W = as_global(s.W), *A = as_global(s.A);
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    for (int c = 0; c < 3; ++c) {
        const double w = X[(size_t)c * npad + j];
        W[3 * ((size_t)l * M + j) + c] = w;
        if (!last) A[(size_t)(npad + c) * lda + j] = lambda * w;
    }
}

__global__ void k_ml_finish(const BatchSlot *tab, int M, int T, int L)
{
    const BatchSlot &s = tab[blockIdx.z];
    gcdouble *small = as_global(s.ns) + (size_t)12 * M;
    gdouble *W = as_global(s.W);
    const int t = threadIdx.x;
    if (t < 12) { const int k = t / 3, c = t % 3; W[3 * ((size_t)M * L + k) + c] = k < T ? small[kAff + 3 * k + c] : 0.0; }
    if (t == 0 && s.model->iterations >= M) s.model->iterations = M + T;      // all unknowns eliminated
}

// QNN model: the polynomial (k_ml_affine left it in the solver state) into the rows of X that
// k_pack reads it from; a rig with fewer points than polynomial terms has no fit at all
__global__ void k_qnn_finish(const BatchSlot *tab, int M, int T, int npad, int fit_ok)
{
    const BatchSlot &s = tab[blockIdx.z];
    gcdouble *small = as_global(s.ns) + (size_t)12 * M;
    gdouble *X = as_global(s.X);
    const int t = threadIdx.x;
    if (t < 3 * T) { const int k = t / 3, c = t % 3; X[(size_t)c * npad + M + k] = fit_ok ? small[kAff + 3 * k + c] : 0.0; }
    if (t == 0) {
        if (!fit_ok) s.model->sing_flag = 1;
        if (s.model->iterations >= M) s.model->iterations = M + T;      // all unknowns eliminated
    }
}

// the factorisation loop; with rhs_only the matrix is left alone and only the right-hand-side rows
// travel through the stored factor
void launch_factor(const BuildBuffers &b, hipStream_t stream, int npc, int n1, int rhs_only)
{
    const unsigned nb = (unsigned)b.nbatch;
    static const bool unfused = tuning_env("FD_CHOL_UNFUSED") != nullptr;      // A/B: two launches per step
    if (!rhs_only && unfused) hipLaunchKernelGGL(k_chol_first, dim3(1, 1, nb), dim3(256), 0, stream, b.d_slots, b.M, b.lda, n1);
    if (!rhs_only && !unfused) {
        // block 0: the step kernel with no panel before it (factorise, solve the rows below, nothing else)
        const int npanel = (npc - kNB + kStepSlab - 1) / kStepSlab + 1;
        hipLaunchKernelGGL(k_chol_step, dim3(npanel, 1, nb), dim3(256), kStepPanelLds, stream, b.d_slots, b.M, b.lda, b.npad,
                           npc, n1, -kNB, 1, npanel, 1);
    }
    for (int k0 = 0; k0 < npc; k0 += kNB) {
        const int below = npc - k0 - kNB;
        if (rhs_only || unfused) {
            const int nslab = rhs_only ? 0 : (below + kSlab - 1) / kSlab;
            hipLaunchKernelGGL(k_chol_solve, dim3(nslab + 1, 1, nb), dim3(256), 0, stream, b.d_slots, b.M, b.lda, b.npad, npc,
                               k0, nslab, rhs_only ? 0 : 1);
        }
        if (below <= 0) break;
        const int ncb = below / 16;
        if (rhs_only || unfused) {
            // enough workgroups to cover the device while the trailing matrix is large
            int nchunk = rhs_only ? 1 : (ncb + 15) / 16;
            nchunk = nchunk < 1 ? 1 : (nchunk > 8 ? 8 : nchunk);
            hipLaunchKernelGGL(k_chol_trail, dim3(ncb * nchunk + (rhs_only ? 0 : 1), 1, nb), dim3(256), 0, stream, b.d_slots,
                               b.M, b.lda, b.npad, npc, n1, k0, nchunk, rhs_only);
        } else {
            // fused: this launch also factorises block k0 + 32 and solves its panel
            const int below_next = below - kNB;
            const int npanel = (below_next + kStepSlab - 1) / kStepSlab + 1;
            // macro tiles per column pair: up to (ncb / 2 - 1); four waves per workgroup, about two tiles each
            int nchunk = (ncb / 2 + 7) / 8;
            nchunk = nchunk < 1 ? 1 : (nchunk > 8 ? 8 : nchunk);
            // bulk steps update everything beyond the next panel, the others one column pair
            // deferring the bulk pays where the trailing matrix is large; a small system updates
            // everything every step and keeps its LDS footprint at the panel work's
            // (a batch defers twice as long: its updates are bound by the traffic of 32 trailing matrices, a lone system by the chain
            //  of launches -- C3: batched +4 % at 8, single build 4 % slower)
            const int bulk_every = npc > 512 ? (nb >= 4 ? 2 * kBulk : kBulk) : 1;
            const bool bulk = ((k0 >> 5) % bulk_every) == bulk_every - 1;
            const int nreg = ncb > 2 ? (bulk ? (ncb / 2 - 1) * nchunk : nchunk) : 0;
            const size_t lds = (size_t)bulk_every * 8192 > kStepPanelLds ? (size_t)bulk_every * 8192 : kStepPanelLds;
            hipLaunchKernelGGL(k_chol_step, dim3(npanel + nreg, 1, nb), dim3(256), lds, stream, b.d_slots, b.M, b.lda, b.npad, npc,
                               n1, k0, nchunk, npanel, bulk_every);
        }
    }
}

// L^T y = z over rows [0, rows): one call up to 512 rows, 256-row ranges chained above that
// recover_T >= 0: also recover the polynomial and w (k_ns_recover's work) -- inside the same launch
// when the system is one range, as a launch of its own otherwise; -1: back-substitution only
hipError_t launch_backsub_spd(const BuildBuffers &b, hipStream_t stream, int rows, int recover_T)
{
    const unsigned nb = (unsigned)b.nbatch;
    if (rows <= 512) {
        hipLaunchKernelGGL(k_backsub_inv, dim3(1, 1, nb), dim3(256), sizeof(double) * 3 * (size_t)rows, stream, b.d_slots,
                           b.M, b.lda, b.npad, 0, rows, recover_T);
        return hipGetLastError();
    }
    constexpr int W = 256;
    for (int hi = rows; hi > 0; hi -= W) {
        const int lo = hi > W ? hi - W : 0;
        hipLaunchKernelGGL(k_backsub_inv, dim3(1, 1, nb), dim3(256), sizeof(double) * 3 * (size_t)(hi - lo), stream,
                           b.d_slots, b.M, b.lda, b.npad, lo, hi, -1);
        if (lo > 0) {
            hipError_t e = launch_backsub_update(b, stream, lo, hi - lo);
            if (e != hipSuccess) return e;
        }
    }
    if (recover_T >= 0) hipLaunchKernelGGL(k_ns_recover, dim3(1, 1, nb), dim3(256), 0, stream, b.d_slots, b.M, recover_T, b.npad);
    return hipGetLastError();
}

}  // namespace

bool spd_applicable(int kind, int term, double lambda, int M)
{
    if (M < 16 || !(lambda >= 0.0)) return false;
    switch (kind) {
    case FD_KERNEL_GAUSSIAN: return true;                                   // positive definite by itself
    case FD_KERNEL_THIN_PLATE: return term == FD_TERM_LINEAR;               // conditionally p.d. of order 2
    case FD_KERNEL_CUBIC: return term == FD_TERM_LINEAR;                    // order 2
    case FD_KERNEL_BIHARMONIC: return term != FD_TERM_ZERO;                 // -r: order 1
    default: return false;                                                  // QNN radii: Phi is not symmetric
    }
}

// everything after k_prepare
hipError_t launch_build_spd(const BuildBuffers &b, hipStream_t stream, hipEvent_t ev_mid)
{
    const unsigned nb = (unsigned)b.nbatch;
    const int M = b.M, T = b.T;
    const int n1 = M - T, npc = round_up(n1, kNB), npa = round_up(M, 32);
    // fd_config.solver = FD_SOLVER_ONE_WORKGROUP (or FD_SMALL_BUILD=1 for every context): the one-workgroup build
    // (k_build_small) up to order 512.  Not what AUTO takes -- measured
    // on MI355X it LOSES to the chain: 0.37 vs 0.25 ms at M = 256, 1.19 vs 0.43 ms at M = 512, and no gain
    // for batches of 32 either (profiles/r02_build_small.txt).  With the matrix in L2 every phase is a
    // handful of dependent ~1 us round trips that four waves cannot hide, where the chain's kernels spread
    // them over many CUs; what would win is the matrix in registers / LDS.  In a pipeline that evaluates
    // 32 frames on most of the device while the next 32 models are solved it is the other way round: 32 workgroups
    // on 32 CUs for 0.65 ms disturb the evaluation less than 18 launches with grids all over the device (bench.py:
    // 120-122k against 112k Mverts/s).
    static const bool use_small = [] { const char *e = tuning_env("FD_SMALL_BUILD"); return e && atoi(e) == 1; }();
    if (npc <= kSmallMaxNpc && (use_small || b.small)) {
        // one workgroup per model does everything after the assembly
        hipError_t e0 = launch_assemble_block(b, stream, npa);
        if (e0 != hipSuccess) return e0;
        if (ev_mid) (void)hipEventRecord(ev_mid, stream);
        const size_t lds = kStepPanelLds > sizeof(double) * 3 * (size_t)npc ? kStepPanelLds : sizeof(double) * 3 * (size_t)npc;
        static unsigned long long *d_stamps = nullptr;
        static const bool want_stamps = tuning_env("FD_SMALL_STAMPS") != nullptr;
        if (want_stamps && !d_stamps) { (void)hipMalloc((void **)&d_stamps, 32 * sizeof(unsigned long long)); (void)hipMemset(d_stamps, 0, 32 * sizeof(unsigned long long)); }
        hipLaunchKernelGGL(k_build_small, dim3(1, 1, nb), dim3(256), lds, stream, b.d_slots, M, T, b.npad, b.lda, b.kind, b.Mpad,
                           want_stamps ? d_stamps : nullptr);
        if (want_stamps && d_stamps) {
            unsigned long long h[32];
            if (hipMemcpy(h, d_stamps, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
                fprintf(stderr, "[k_build_small stamps, shader cycles: reflectors | KV | W | rotate | first: factor, solve | step 0: (i), (ii), (iii) | rest of the steps | backsub | pack]\n  ");
                for (int q = 0; q < 14; ++q) fprintf(stderr, " %llu", h[q]);
                fprintf(stderr, "\n");
            }
        }
        return hipGetLastError();
    }
    if (T > 0) {
        hipLaunchKernelGGL(k_ns_reflectors, dim3(1, 1, nb), dim3(256), 0, stream, b.d_slots, M, T, b.npad, b.lda, 1);
    }
    hipError_t e = launch_assemble_block(b, stream, npa);
    if (e != hipSuccess) return e;
    if (T > 0) {
        hipLaunchKernelGGL(k_ns_kv, dim3((M + 63) / 64, 1, nb), dim3(256), 0, stream, b.d_slots, M, b.lda);
        const unsigned g = (unsigned)((npc > M ? npc : M) + 31) / 32;
        hipLaunchKernelGGL(k_ns_rotate, dim3(g, g, nb), dim3(256), 0, stream, b.d_slots, M, T, npc, b.lda);
    }
    if (ev_mid) (void)hipEventRecord(ev_mid, stream);
    launch_factor(b, stream, npc, n1, 0);
    e = launch_backsub_spd(b, stream, npc, T);
    if (e != hipSuccess) return e;
    return launch_pack(b, stream);
}

// everything after k_prepare for the multilayer model
hipError_t launch_build_ml(const BuildBuffers &b, hipStream_t stream, hipEvent_t ev_mid)
{
    const unsigned nb = (unsigned)b.nbatch;
    const int M = b.M, T = b.T, L = b.ml_layers;
    const int npc = round_up(M, kNB);
    if (T > 0) {
        hipLaunchKernelGGL(k_ns_reflectors, dim3(1, 1, nb), dim3(256), 0, stream, b.d_slots, M, T, b.npad, b.lda, 0);
        hipLaunchKernelGGL(k_ml_affine, dim3(1, 1, nb), dim3(256), 0, stream, b.d_slots, M, T, b.npad, b.lda);
    }
    hipLaunchKernelGGL(k_ml_setup, dim3((M * L + 255) / 256, 1, nb), dim3(256), 0, stream, b.d_slots, M, L, b.gauss_R);
    if (ev_mid) (void)hipEventRecord(ev_mid, stream);
    for (int l = 0; l < L; ++l) {
        hipError_t e = launch_assemble_block(b, stream, npc, l * M);
        if (e != hipSuccess) return e;
        launch_factor(b, stream, npc, M, 0);
        e = launch_backsub_spd(b, stream, npc, -1);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_ml_layer, dim3((M + 255) / 256, 1, nb), dim3(256), 0, stream, b.d_slots, M, b.npad, b.lda, l,
                           l + 1 == L ? 1 : 0, b.lambda);
    }
    hipLaunchKernelGGL(k_ml_finish, dim3(1, 1, nb), dim3(64), 0, stream, b.d_slots, M, T, L);
    return launch_pack_records(b, stream, M * L, FD_KERNEL_GAUSSIAN_QNN, 2, L);
}

// The SOP's default model (model = 0, rbfsetalgoqnn, reference src/SOP_FaceDeform.cpp:343-345) in
// ALGLIB's order: the term's polynomial is a least-squares fit to the deltas
// -- through the QR of P the reflectors give, no normal equations -- and is removed; the Gaussians
// with their per-centre radii then fit what is left, (Phi + lambda I) w = f - P a.  Phi is not
// symmetric, so this is the pivoted LU (fd_build.hip) of the kernel block alone: order M, the
// rest of the padded system an identity block with zero right-hand sides.
static BuildBuffers kernel_block_only(const BuildBuffers &b)
{
    BuildBuffers k = b;
    k.T = 0;
    k.n = b.M;
    k.term = FD_TERM_ZERO;
    return k;
}

hipError_t launch_build_qnn(const BuildBuffers &b, hipStream_t stream, hipEvent_t ev_mid)
{
    const unsigned nb = (unsigned)b.nbatch;
    const int M = b.M, T = b.T;
    const bool fit = M >= T;
    hipError_t e = launch_qnn_radii(b, stream);
    if (e != hipSuccess) return e;
    if (T > 0 && fit) {
        hipLaunchKernelGGL(k_ns_reflectors, dim3(1, 1, nb), dim3(256), 0, stream, b.d_slots, M, T, b.npad, b.lda, 0);
        hipLaunchKernelGGL(k_ml_affine, dim3(1, 1, nb), dim3(256), 0, stream, b.d_slots, M, T, b.npad, b.lda);
    }
    const BuildBuffers k = kernel_block_only(b);
    e = launch_assemble_block(k, stream, b.npad);
    if (e != hipSuccess) return e;
    if (ev_mid) (void)hipEventRecord(ev_mid, stream);
    e = launch_lu_factor_solve(k, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_qnn_finish, dim3(1, 1, nb), dim3(64), 0, stream, b.d_slots, M, T, b.npad, fit ? 1 : 0);
    return launch_pack(b, stream);
}

// fd_set_deltas for the QNN model: the polynomial of the new deltas, then their remainder through the stored LU
hipError_t launch_resolve_qnn(const BuildBuffers &b, hipStream_t stream, const PointSrc *src)
{
    const unsigned nb = (unsigned)b.nbatch;
    const int M = b.M, T = b.T;
    const bool fit = M >= T;
    hipError_t e = launch_prepare_rhs(b, stream, src);
    if (e != hipSuccess) return e;
    if (T > 0 && fit) hipLaunchKernelGGL(k_ml_affine, dim3(1, 1, nb), dim3(256), 0, stream, b.d_slots, M, T, b.npad, b.lda);
    e = launch_lu_resolve_core(kernel_block_only(b), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_qnn_finish, dim3(1, 1, nb), dim3(64), 0, stream, b.d_slots, M, T, b.npad, fit ? 1 : 0);
    return launch_pack(b, stream);
}

// fd_set_deltas: new right-hand sides through the stored reflectors and Cholesky factor
hipError_t launch_resolve_spd(const BuildBuffers &b, hipStream_t stream, const PointSrc *src)
{
    const unsigned nb = (unsigned)b.nbatch;
    const int M = b.M, T = b.T;
    const int n1 = M - T, npc = round_up(n1, kNB);
    hipError_t e = launch_prepare_rhs(b, stream, src);
    if (e != hipSuccess) return e;
    if (T > 0) hipLaunchKernelGGL(k_ns_rhs, dim3(1, 1, nb), dim3(256), 0, stream, b.d_slots, M, T, b.npad, b.lda);
    launch_factor(b, stream, npc, n1, 1);
    e = launch_backsub_spd(b, stream, npc, T);
    if (e != hipSuccess) return e;
    return launch_pack(b, stream);
}

}  // namespace fd
