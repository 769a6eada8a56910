// fd_morph.hip -- next row N1: morph-space reprojection on the device.
//
// Replaces DirectBSEdit (reference src/dbse.hpp:7-33, src/dbse.cpp:9-87) and the loop that
// applies it (src/SOP_FaceDeform.cpp:444-473):
//   init            shapes matrix A (3N x S, fp64, column-major) of fp32 deltas shape - rest, and
//                   its Householder QR in Eigen's packed form (dbse.cpp:9-37)
//   compute weights w_s = sum_i float(P_i - rest_i) * QR[i][s]        (dbse.cpp:39-60)
//   displace        P = rest + sum_s float(A[.][s]) * clamp(float(3 w_s)) [+ (P - rest) * falloffradius]
//                                                                      (dbse.cpp:62-77, SOP :458-473)
// Everything is HBM-bound streaming over the 3N x S matrix: rows run along lanes (coalesced),
// columns are walked in order.  The per-cook passes read the matrix once each: 8 B per entry
// for the weights (the packed QR, fp64 as the reference keeps it) and 4 B per entry for the
// displacement (the unfactored deltas are fp32 values by construction; an fp32 copy is kept).
// The QR itself runs once per change of the blendshape set.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "facedeform_hip.h"
#include "fd_tuning.h"

namespace {

constexpr int kT = 256;             // threads per workgroup everywhere in this file
constexpr int kRowsPerWg = 4096;    // rows of the 3N x S matrix one workgroup walks (16 per thread)
constexpr int kR = kRowsPerWg / kT;

__device__ __forceinline__ double wg_sum(double v, double *scratch)
{
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

// dbse.cpp:16-31: column s of the shapes matrix, fp32 delta widened (and its fp32 twin)
__global__ __launch_bounds__(kT) void k_shape_column(const float *rest, const float *shape, int64_t rows,
                                                      double *col64, float *col32)
{
    const int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (e >= rows) return;
    const float d = shape[e] - rest[e];
    col64[e] = (double)d;
    col32[e] = d;
}

// ---- Householder QR, one column at a time (Eigen's unblocked order = LAPACK dgeqr2) ----------
// hh[0] = beta, hh[1] = tau, hh[2] = c0 - beta (0 when the reflector is the identity)

// partial sums of squares of x[k+1:]
__global__ __launch_bounds__(kT) void k_qr_tail_norm(const double *x, int64_t rows, int k, double *partial)
{
    __shared__ double scratch[4];
    const int64_t base = (int64_t)blockIdx.x * kRowsPerWg;
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < kR; ++r) {
        const int64_t i = base + r * kT + threadIdx.x;
        if (i > k && i < rows) { const double v = x[i]; acc = fma(v, v, acc); }
    }
    acc = wg_sum(acc, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// makeHouseholderInPlace (Eigen/src/Householder/Householder.h): beta, tau, the divisor of the tail
__global__ __launch_bounds__(kT) void k_qr_reflector(double *x, int k, const double *partial, int npartial,
                                                      double *hh, double *tau_out)
{
    __shared__ double scratch[4];
    double acc = 0.0;
    for (int q = threadIdx.x; q < npartial; q += kT) acc += partial[q];
    const double tail2 = wg_sum(acc, scratch);
    if (threadIdx.x == 0) {
        const double c0 = x[k];
        double beta, tau, denom;
        if (tail2 <= DBL_MIN) { tau = 0.0; beta = c0; denom = 0.0; }
        else {
            beta = sqrt(c0 * c0 + tail2);
            if (c0 >= 0.0) beta = -beta;
            denom = c0 - beta;
            tau = (beta - c0) / beta;
        }
        x[k] = beta;
        hh[0] = beta; hh[1] = tau; hh[2] = denom;
        tau_out[k] = tau;
    }
}

// scale the tail into the essential part v (in place) and form the partial dots v . a_j of every
// trailing column j: partial[wg * ld + (j - k - 1)].  Waves reduce their own column sums and park
// them in LDS; the workgroup synchronises once at the end.
__global__ __launch_bounds__(kT) void k_qr_dots(double *A, int64_t rows, int S, int k, const double *hh,
                                                 double *partial, int ld)
{
    extern __shared__ double s_part[];       // [4][S - k - 1]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nt = S - k - 1;
    const double denom = hh[2];
    double *x = A + (size_t)k * rows;
    const int64_t base = (int64_t)blockIdx.x * kRowsPerWg;
    double v[kR];
#pragma unroll
    for (int r = 0; r < kR; ++r) {
        const int64_t i = base + r * kT + threadIdx.x;
        v[r] = 0.0;
        if (i > k && i < rows) {
            v[r] = denom != 0.0 ? x[i] / denom : 0.0;
            x[i] = v[r];
        }
    }
    for (int j = k + 1; j < S; ++j) {
        const double *a = A + (size_t)j * rows;
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            const int64_t i = base + r * kT + threadIdx.x;
            if (i > k && i < rows) acc = fma(v[r], a[i], acc);
        }
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) s_part[wave * nt + (j - k - 1)] = acc;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < nt; q += kT)
        partial[(size_t)blockIdx.x * ld + q] = (s_part[q] + s_part[nt + q]) + (s_part[2 * nt + q] + s_part[3 * nt + q]);
}

// tmp_j = essential^T a_j + a_j[k]; the row-k update of applyHouseholderOnTheLeft happens here too
__global__ __launch_bounds__(kT) void k_qr_dots_reduce(double *A, int64_t rows, int k, const double *hh,
                                                        const double *partial, int npartial, int ld, double *tmp)
{
    __shared__ double scratch[4];
    const int j = k + 1 + blockIdx.x;
    double acc = 0.0;
    for (int q = threadIdx.x; q < npartial; q += kT) acc += partial[(size_t)q * ld + blockIdx.x];
    acc = wg_sum(acc, scratch);
    if (threadIdx.x == 0) {
        double *a = A + (size_t)j * rows;
        const double t = acc + a[k];
        tmp[blockIdx.x] = t;
        a[k] -= hh[1] * t;
    }
}

// bottom -= tau * essential * tmp
__global__ __launch_bounds__(kT) void k_qr_apply(double *A, int64_t rows, int S, int k, const double *hh,
                                                  const double *tmp)
{
    const double tau = hh[1];
    const double *x = A + (size_t)k * rows;
    const int64_t base = (int64_t)blockIdx.x * kRowsPerWg;
    double v[kR];
#pragma unroll
    for (int r = 0; r < kR; ++r) {
        const int64_t i = base + r * kT + threadIdx.x;
        v[r] = (i > k && i < rows) ? x[i] : 0.0;
    }
    for (int j = k + 1; j < S; ++j) {
        double *a = A + (size_t)j * rows;
        const double t = tmp[j - k - 1];
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            const int64_t i = base + r * kT + threadIdx.x;
            if (i > k && i < rows) a[i] -= tau * v[r] * t;
        }
    }
}

// ---- Householder QR, blocked (default) --------------------------------------------------------
// Same reflectors as the column-by-column form above (Eigen's HouseholderQR is itself a panel
// factorisation followed by a block-reflector update: Eigen/src/QR/HouseholderQR.h,
// householder_qr_inplace_blocked), far fewer passes over the 3N x S matrix:
//   inside a panel of kNB columns ONE pass per column: it applies reflector k to the panel's later
//     columns and, while their updated values are in registers, forms the tail norm of column k+1
//     and its dot products with the columns after it -- everything the next reflector needs;
//   behind a panel the kNB reflectors go through the trailing columns together (compact WY:
//     C -= V T^T (V^T C)): one read-only pass for V^T C and V^T V, one read-write pass that also
//     forms the first column's sums for the next panel.
// Passes over a column: ~(kNB + 1) S inside panels + 3 S^2 / (2 kNB) behind them, instead of 1.5 S^2.
constexpr int kNB = 8;              // panel width
constexpr int kBR = 4;              // rows per lane
constexpr int kBRows = kBR * kT;    // rows per workgroup
constexpr int kSlices = 16;         // first-stage slices of the reduction over workgroups

__device__ __forceinline__ double wave_sum(double v)
{
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// acc[0..nq) summed over the workgroup -> out[0..nq)
__device__ __forceinline__ void wg_reduce_store(const double (&acc)[kNB], int nq, double *s_red /* [4][kNB] */, double *out)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < kNB; ++q) {
        if (q < nq) {
            const double t = wave_sum(acc[q]);
            if (lane == 0) s_red[wave * kNB + q] = t;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < nq)
        out[threadIdx.x] = (s_red[threadIdx.x] + s_red[kNB + threadIdx.x]) + (s_red[2 * kNB + threadIdx.x] + s_red[3 * kNB + threadIdx.x]);
}

// sums of column k against itself and the nq - 1 columns after it over the rows below k (first panel only)
__global__ __launch_bounds__(kT) void k_qr_gram(const double *A, int64_t rows, int k, int nq, double *partial)
{
    __shared__ double s_red[4 * kNB];
    const int64_t base = (int64_t)blockIdx.x * kBRows;
    double acc[kNB];
#pragma unroll
    for (int q = 0; q < kNB; ++q) acc[q] = 0.0;
#pragma unroll
    for (int r = 0; r < kBR; ++r) {
        const int64_t i = base + r * kT + threadIdx.x;
        if (i > k && i < rows) {
            const double u = A[(size_t)k * rows + i];
#pragma unroll
            for (int q = 0; q < kNB; ++q)
                if (q < nq) acc[q] = fma(u, q == 0 ? u : A[(size_t)(k + q) * rows + i], acc[q]);
        }
    }
    wg_reduce_store(acc, nq, s_red, partial + (size_t)blockIdx.x * kNB);
}

// One workgroup: the sums of the nwg workgroups -> reflector k (makeHouseholderInPlace), the
// products tmp_l = essential . a_l + a_l[k] of the panel's later columns, and their row-k update.
constexpr int kSetupT = 1024;
__global__ __launch_bounds__(kSetupT) void k_qr_col_setup(double *A, int64_t rows, int k, int nq, const double *partial, int nwg,
                                                           double *hh, double *tau_out, double *tmp)
{
    __shared__ double s_acc[kSetupT];
    __shared__ double tot[kNB];
    __shared__ double sh[2];
    const int q = threadIdx.x & (kNB - 1), w0 = threadIdx.x / kNB;       // 128 workgroups' sums per sweep
    constexpr int kSweep = kSetupT / kNB;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;                       // four loads in flight per lane
    int w = w0;
    for (; w + 3 * kSweep < nwg; w += 4 * kSweep) {
        a0 += partial[(size_t)w * kNB + q];
        a1 += partial[(size_t)(w + kSweep) * kNB + q];
        a2 += partial[(size_t)(w + 2 * kSweep) * kNB + q];
        a3 += partial[(size_t)(w + 3 * kSweep) * kNB + q];
    }
    for (; w < nwg; w += kSweep) a0 += partial[(size_t)w * kNB + q];
    s_acc[threadIdx.x] = q < nq ? (a0 + a1) + (a2 + a3) : 0.0;
    __syncthreads();
    if (threadIdx.x < 64) {                                              // 8 lanes per quantity, 16 sums each
        const int qq = threadIdx.x & (kNB - 1), part = threadIdx.x / kNB;
        double t = 0.0;
        for (int j = part; j < kSweep; j += 8) t += s_acc[j * kNB + qq];
        t += __shfl_xor(t, 8); t += __shfl_xor(t, 16); t += __shfl_xor(t, 32);
        if (threadIdx.x < kNB) tot[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double *x = A + (size_t)k * rows;
        const double c0 = x[k], tail2 = tot[0];
        double beta, tau, denom;
        if (tail2 <= DBL_MIN) { tau = 0.0; beta = c0; denom = 0.0; }
        else {
            beta = sqrt(c0 * c0 + tail2);
            if (c0 >= 0.0) beta = -beta;
            denom = c0 - beta;
            tau = (beta - c0) / beta;
        }
        x[k] = beta;
        hh[0] = beta; hh[1] = tau; hh[2] = denom;
        tau_out[k] = tau;
        sh[0] = tau; sh[1] = denom;
    }
    __syncthreads();
    if (threadIdx.x >= 1 && (int)threadIdx.x < nq) {
        double *a = A + (size_t)(k + threadIdx.x) * rows + k;
        const double t = (sh[1] != 0.0 ? tot[threadIdx.x] / sh[1] : 0.0) + *a;
        tmp[threadIdx.x - 1] = t;
        *a -= sh[0] * t;
    }
}

// The pass of column k inside its panel [.., pend): scale the tail into the essential part, apply
// the reflector to the panel's later columns, and form column k+1's sums from the updated values.
__global__ __launch_bounds__(kT) void k_qr_panel_apply(double *A, int64_t rows, int k, int pend, const double *hh,
                                                        const double *tmp, double *partial)
{
    __shared__ double s_red[4 * kNB];
    const double tau = hh[1], denom = hh[2];
    const int nl = pend - k - 1;                         // later columns of the panel: 0 .. kNB-1
    double t[kNB - 1];
#pragma unroll
    for (int l = 0; l < kNB - 1; ++l) t[l] = l < nl ? tmp[l] : 0.0;
    const int64_t base = (int64_t)blockIdx.x * kBRows;
    double acc[kNB];
#pragma unroll
    for (int q = 0; q < kNB; ++q) acc[q] = 0.0;
    double *x = A + (size_t)k * rows;
#pragma unroll
    for (int r = 0; r < kBR; ++r) {
        const int64_t i = base + r * kT + threadIdx.x;
        if (i > k && i < rows) {
            const double v = denom != 0.0 ? x[i] / denom : 0.0;
            x[i] = v;
            double val[kNB - 1];
#pragma unroll
            for (int l = 0; l < kNB - 1; ++l) {
                if (l < nl) {
                    double *a = A + (size_t)(k + 1 + l) * rows + i;
                    val[l] = *a - tau * v * t[l];
                    *a = val[l];
                } else val[l] = 0.0;
            }
            if (i > k + 1) {
#pragma unroll
                for (int l = 0; l < kNB - 1; ++l)
                    if (l < nl) acc[l] = fma(val[0], val[l], acc[l]);
            }
        }
    }
    if (nl > 0) wg_reduce_store(acc, nl, s_red, partial + (size_t)blockIdx.x * kNB);
}

// V of the panel [k0, k0 + nb) as the block reflector sees it: unit diagonal, zeros above it
__device__ __forceinline__ double wy_v(const double *A, int64_t rows, int col, int64_t i)
{
    return i > col && i < rows ? A[(size_t)col * rows + i] : (i == col ? 1.0 : 0.0);
}

// 8 per-lane values, summed over the 64 lanes, for 8 different quantities at once: halve the set at
// each of the first three exchange steps (10 exchanges instead of 48).  The sum of quantity
// p = 4 b5 + 2 b4 + b3 ends in every lane whose bits 5, 4, 3 are (b5, b4, b3).
__device__ __forceinline__ double wave_sum8(const double (&a)[kNB])
{
    const int lane = threadIdx.x & 63;
    const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8;
    double b[4], c[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double keep = h5 ? a[4 + i] : a[i], send = h5 ? a[i] : a[4 + i];
        b[i] = keep + __shfl_xor(send, 32);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double keep = h4 ? b[2 + i] : b[i], send = h4 ? b[i] : b[2 + i];
        c[i] = keep + __shfl_xor(send, 16);
    }
    double d = (h3 ? c[1] : c[0]) + __shfl_xor(h3 ? c[0] : c[1], 8);
    d += __shfl_xor(d, 4);
    d += __shfl_xor(d, 2);
    d += __shfl_xor(d, 1);
    return d;
}

// partial[wg][c * kNB + p] = sum over the workgroup's rows of V[.][p] C[.][c]   (c < ntrail), then
// partial[wg][ntrail * kNB + q (q - 1) / 2 + p] = V[.][p] . V[.][q]            (p < q < nb)
__global__ __launch_bounds__(kT) void k_qr_wy_dots(const double *A, int64_t rows, int k0, int nb, int S, double *partial, int ld)
{
    extern __shared__ double s_part[];       // [4][ld]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pend = k0 + nb, ntrail = S - pend;
    const int64_t base = (int64_t)blockIdx.x * kBRows;
    double V[kBR][kNB];
#pragma unroll
    for (int r = 0; r < kBR; ++r)
#pragma unroll
        for (int p = 0; p < kNB; ++p) V[r][p] = p < nb ? wy_v(A, rows, k0 + p, base + r * kT + threadIdx.x) : 0.0;
    const int myp = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
    double cur[kBR], nxt[kBR];
    auto load = [&](int c, double (&dst)[kBR]) {
        const double *col = A + (size_t)(pend + c) * rows;
#pragma unroll
        for (int r = 0; r < kBR; ++r) {
            const int64_t i = base + r * kT + threadIdx.x;
            dst[r] = i >= k0 && i < rows ? col[i] : 0.0;
        }
    };
    if (ntrail > 0) load(0, cur);
    for (int c = 0; c < ntrail; ++c) {
        if (c + 1 < ntrail) load(c + 1, nxt);
        double acc[kNB];
#pragma unroll
        for (int p = 0; p < kNB; ++p) {
            acc[p] = 0.0;
#pragma unroll
            for (int r = 0; r < kBR; ++r) acc[p] = fma(V[r][p], cur[r], acc[p]);
        }
        const double d = wave_sum8(acc);
        if ((lane & 7) == 0) s_part[wave * ld + c * kNB + myp] = d;
#pragma unroll
        for (int r = 0; r < kBR; ++r) cur[r] = nxt[r];
    }
    // V^T V, strictly upper part, in groups of 8 quantities
    {
        double acc[kNB];
        int filled = 0, first = 0;
#pragma unroll
        for (int q = 1; q < kNB; ++q)
#pragma unroll
            for (int p = 0; p < q; ++p) {
                double a = 0.0;
#pragma unroll
                for (int r = 0; r < kBR; ++r) a = fma(V[r][p], V[r][q], a);
                acc[filled++] = a;
                if (filled == kNB || (q == kNB - 1 && p == q - 1)) {
                    for (int z = filled; z < kNB; ++z) acc[z] = 0.0;
                    const double d = wave_sum8(acc);
                    if ((lane & 7) == 0 && myp < filled) s_part[wave * ld + ntrail * kNB + first + myp] = d;
                    first += filled;
                    filled = 0;
                }
            }
    }
    __syncthreads();
    const int nvals = ntrail * kNB + kNB * (kNB - 1) / 2;
    for (int q = threadIdx.x; q < nvals; q += kT)
        partial[(size_t)blockIdx.x * ld + q] = (s_part[q] + s_part[ld + q]) + (s_part[2 * ld + q] + s_part[3 * ld + q]);
}

// first stage of the sum over workgroups: slice y of them, 32 consecutive values per workgroup
__global__ __launch_bounds__(kT) void k_qr_wy_reduce(const double *partial, int nwg, int ld, int nvals, double *partial2)
{
    __shared__ double s[kT];
    const int idx = blockIdx.x * 32 + (threadIdx.x & 31), wl = threadIdx.x >> 5;
    const int per = (nwg + kSlices - 1) / kSlices;
    const int w_lo = blockIdx.y * per, w_hi = min(nwg, w_lo + per);
    double acc = 0.0;
    if (idx < nvals)
        for (int w = w_lo + wl; w < w_hi; w += kT / 32) acc += partial[(size_t)w * ld + idx];
    s[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < 32 && idx < nvals) {
        double t = 0.0;
        for (int j = 0; j < kT / 32; ++j) t += s[j * 32 + threadIdx.x];
        partial2[(size_t)blockIdx.y * ld + idx] = t;
    }
}

// One workgroup: W = V^T C and V^T V from the slices, the triangular factor T of the block
// reflector H_0 .. H_{nb-1} = I - V T V^T (T_jj = tau_j, T[0:j][j] = -tau_j T[0:j][0:j] V[:, 0:j]^T v_j),
// and Y = T^T W for the update C -= V Y.
__global__ __launch_bounds__(kT) void k_qr_wy_T(const double *partial2, int ld, int k0, int nb, int ntrail, const double *tau,
                                                 double *Y)
{
    extern __shared__ double s_w[];          // [ld] sums, then [kNB * kNB] T
    double *T = s_w + ld;
    const int nvals = ntrail * kNB + kNB * (kNB - 1) / 2;
    for (int q = threadIdx.x; q < nvals; q += kT) {
        double t = 0.0;
        for (int y = 0; y < kSlices; ++y) t += partial2[(size_t)y * ld + q];
        s_w[q] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double *vtv = s_w + ntrail * kNB;           // (p < q) at q (q - 1) / 2 + p
        for (int j = 0; j < kNB; ++j)
            for (int i = 0; i < kNB; ++i) T[i * kNB + j] = 0.0;
        for (int j = 0; j < nb; ++j) {
            const double tj = tau[k0 + j];
            for (int i = 0; i < j; ++i) {
                double acc = 0.0;
                for (int z = i; z < j; ++z) acc = fma(T[i * kNB + z], vtv[j * (j - 1) / 2 + z], acc);
                T[i * kNB + j] = -tj * acc;
            }
            T[j * kNB + j] = tj;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nb * ntrail; e += kT) {
        const int p = e / ntrail, c = e - p * ntrail;
        double acc = 0.0;
        for (int q = 0; q <= p; ++q) acc = fma(T[q * kNB + p], s_w[c * kNB + q], acc);
        Y[(size_t)p * ntrail + c] = acc;
    }
}

// C -= V Y over the trailing columns; the next panel's first column leaves its sums behind
__global__ __launch_bounds__(kT) void k_qr_wy_apply(double *A, int64_t rows, int k0, int nb, int S, const double *Y,
                                                     int next_nq, double *partial)
{
    extern __shared__ double s_y[];          // [nb][ntrail], then 4 * kNB of reduction scratch
    const int pend = k0 + nb, ntrail = S - pend;
    for (int e = threadIdx.x; e < nb * ntrail; e += kT) s_y[e] = Y[e];
    double *s_red = s_y + kNB * ntrail;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kBRows;
    double V[kBR][kNB];
#pragma unroll
    for (int r = 0; r < kBR; ++r)
#pragma unroll
        for (int p = 0; p < kNB; ++p) V[r][p] = p < nb ? wy_v(A, rows, k0 + p, base + r * kT + threadIdx.x) : 0.0;
    double acc[kNB], u[kBR];
#pragma unroll
    for (int q = 0; q < kNB; ++q) acc[q] = 0.0;
    double cur[kBR], nxt[kBR];
    auto load = [&](int c, double (&dst)[kBR]) {
        const double *col = A + (size_t)(pend + c) * rows;
#pragma unroll
        for (int r = 0; r < kBR; ++r) {
            const int64_t i = base + r * kT + threadIdx.x;
            dst[r] = i >= k0 && i < rows ? col[i] : 0.0;
        }
    };
    load(0, cur);
    for (int c = 0; c < ntrail; ++c) {
        if (c + 1 < ntrail) load(c + 1, nxt);
        double *col = A + (size_t)(pend + c) * rows;
        double y[kNB];
#pragma unroll
        for (int p = 0; p < kNB; ++p) y[p] = p < nb ? s_y[p * ntrail + c] : 0.0;
        double dot = 0.0;
#pragma unroll
        for (int r = 0; r < kBR; ++r) {
            const int64_t i = base + r * kT + threadIdx.x;
            double d = 0.0;
#pragma unroll
            for (int p = 0; p < kNB; ++p) d = fma(V[r][p], y[p], d);
            const double x = cur[r] - d;
            if (i >= k0 && i < rows) col[i] = x;
            if (c == 0) u[r] = i > pend && i < rows ? x : 0.0;
            dot = fma(u[r], x, dot);
        }
#pragma unroll
        for (int q = 0; q < kNB; ++q)
            if (q == c && c < next_nq) acc[q] = dot;
#pragma unroll
        for (int r = 0; r < kBR; ++r) cur[r] = nxt[r];
    }
    wg_reduce_store(acc, next_nq, s_red, partial + (size_t)blockIdx.x * kNB);
}

// ---- per cook --------------------------------------------------------------------------------
// dbse.cpp:39-60: partial[wg * S + s] = sum over the workgroup's rows of float(P - rest) * QR[.][s].
// A lane keeps its kWR deltas in registers and walks the columns; a wave reduces each column by
// itself and parks the result in LDS, so nothing but the last step synchronises the workgroup
// (with a workgroup-wide reduction per column this pass ran at 2.9 TB/s).
constexpr int kWR = 8;                       // rows per lane
constexpr int kWRows = kWR * kT;             // rows per workgroup
__global__ __launch_bounds__(kT) void k_morph_weights(const double *QR, int64_t rows, int S, const float *P,
                                                       const float *rest, double *partial)
{
    extern __shared__ double s_part[];       // [4][S]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t base = (int64_t)blockIdx.x * kWRows;
    double d[kWR];
#pragma unroll
    for (int r = 0; r < kWR; ++r) {
        const int64_t i = base + r * kT + threadIdx.x;
        d[r] = i < rows ? (double)(P[i] - rest[i]) : 0.0;          // fp32 subtraction, then widened (:49-51)
    }
    const bool full = base + kWRows <= rows;
    for (int s = 0; s < S; ++s) {
        const double *q = QR + (size_t)s * rows + base + threadIdx.x;
        double v[kWR];
        if (full) {
#pragma unroll
            for (int r = 0; r < kWR; ++r) v[r] = __builtin_nontemporal_load(q + r * kT);
        } else {
#pragma unroll
            for (int r = 0; r < kWR; ++r) v[r] = base + r * kT + threadIdx.x < rows ? q[r * kT] : 0.0;
        }
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < kWR; ++r) acc = fma(d[r], v[r], acc);
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) s_part[wave * S + s] = acc;
    }
    __syncthreads();
    for (int s = threadIdx.x; s < S; s += kT)
        partial[(size_t)blockIdx.x * S + s] = (s_part[s] + s_part[S + s]) + (s_part[2 * S + s] + s_part[3 * S + s]);
}

__global__ __launch_bounds__(kT) void k_morph_weights_reduce(const double *partial, int npartial, int S, double *w)
{
    __shared__ double scratch[4];
    const int s = blockIdx.x;
    double acc = 0.0;
    for (int q = threadIdx.x; q < npartial; q += kT) acc += partial[(size_t)q * S + s];
    acc = wg_sum(acc, scratch);
    if (threadIdx.x == 0) w[s] = acc;
}

// dbse.cpp:62-77 and SOP_FaceDeform.cpp:458-473, fp32 in the reference's order (this file is
// built with -ffp-contract=off: multiply and add stay separate roundings, as in unfused CPU code)
__global__ __launch_bounds__(kT) void k_morph_displace(const float *S32, int64_t N, int S, const double *w,
                                                        float clamp_lo, float clamp_hi, int do_clamp, int add_delta,
                                                        float falloffradius, const float *rest, float *P)
{
    extern __shared__ float s_cw[];
    for (int s = threadIdx.x; s < S; s += kT) {
        const float ws = (float)(w[s] * 3);                        // :70
        s_cw[s] = do_clamp ? (ws < clamp_lo ? clamp_lo : (ws > clamp_hi ? clamp_hi : ws)) : ws;
    }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (i >= N) return;
    const int64_t rows = 3 * N;
    float dx = 0.f, dy = 0.f, dz = 0.f;
    for (int s = 0; s < S; ++s) {
        const float *c = S32 + (size_t)s * rows + 3 * i;
        const float cw = s_cw[s];
        dx = dx + c[0] * cw; dy = dy + c[1] * cw; dz = dz + c[2] * cw;
    }
    const float rx = rest[3 * i], ry = rest[3 * i + 1], rz = rest[3 * i + 2];
    if (add_delta) {
        dx = dx + (P[3 * i] - rx) * falloffradius;
        dy = dy + (P[3 * i + 1] - ry) * falloffradius;
        dz = dz + (P[3 * i + 2] - rz) * falloffradius;
    }
    P[3 * i] = rx + dx; P[3 * i + 1] = ry + dy; P[3 * i + 2] = rz + dz;
}

thread_local char g_merr[512] = {0};

}  // namespace

struct fd_morph {
    int device = 0;
    int64_t N = 0;
    int S = 0;
    bool initialised = false, computed = false;
    hipStream_t stream = nullptr;
    float *d_rest = nullptr, *d_S32 = nullptr, *d_stage = nullptr, *d_P = nullptr;
    // the `rest` point attribute the cook passes use (SOP_FaceDeform.cpp:178-184,445); NULL: the
    // init rest pose (they differ only when input 0 carries its own rest attribute)
    float *d_rest_attr = nullptr;
    bool use_rest_attr = false;
    double *d_QR = nullptr, *d_tau = nullptr, *d_w = nullptr, *d_partial = nullptr, *d_hh = nullptr, *d_tmp = nullptr;
    double *d_bpartial = nullptr, *d_bpartial2 = nullptr, *d_Y = nullptr;   // blocked factorisation
    size_t cap_bpartial = 0;
    size_t cap_entries = 0;       // 3N * S capacity of d_QR / d_S32
    int64_t cap_N = 0;
    int cap_S = 0;
    size_t cap_partial = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_init_ms = 0.f;
    char err[512] = {0};
};

static void merr(fd_morph *m, const char *fmt, ...)
{
    char *dst = m ? m->err : g_merr;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
}

#define FDM_HIP(m, call)                                                                     \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            merr(m, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return FD_E_DEVICE;                                                              \
        }                                                                                    \
    } while (0)

template <typename T>
static int mrealloc(fd_morph *m, T **p, size_t count)
{
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (hipMalloc((void **)p, (count ? count : 1) * sizeof(T)) != hipSuccess) {
        *p = nullptr;
        merr(m, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(hipGetLastError()));
        return FD_E_NOMEM;
    }
    return FD_OK;
}

// row length of the per-workgroup sums the block-reflector pass leaves: kNB per trailing column + V^T V
static size_t wy_ld(int S) { return (size_t)(S > 0 ? S : 1) * kNB + 32; }

static int morph_reserve(fd_morph *m, int64_t N, int S)
{
    int rc;
    const size_t rows = 3 * (size_t)N;
    const size_t nwg = (rows + kWRows - 1) / kWRows;        // the finer of the two row partitions
    if (N > m->cap_N) {
        if ((rc = mrealloc(m, &m->d_rest, rows)) || (rc = mrealloc(m, &m->d_stage, rows)) ||
            (rc = mrealloc(m, &m->d_P, rows))) return rc;
        if (m->d_rest_attr) { (void)hipFree(m->d_rest_attr); m->d_rest_attr = nullptr; }
        m->cap_N = N;
    }
    if (rows * (size_t)S > m->cap_entries) {
        if ((rc = mrealloc(m, &m->d_QR, rows * (size_t)S)) || (rc = mrealloc(m, &m->d_S32, rows * (size_t)S))) return rc;
        m->cap_entries = rows * (size_t)S;
    }
    if (S > m->cap_S) {
        if ((rc = mrealloc(m, &m->d_tau, (size_t)S)) || (rc = mrealloc(m, &m->d_w, (size_t)S)) ||
            (rc = mrealloc(m, &m->d_tmp, (size_t)S)) || (rc = mrealloc(m, &m->d_Y, (size_t)kNB * (size_t)S)) ||
            (rc = mrealloc(m, &m->d_bpartial2, (size_t)kSlices * wy_ld(S)))) return rc;
        m->cap_S = S;
    }
    {   // blocked factorisation: per-workgroup sums of V^T C and V^T V, their slices, Y
        const size_t nwg_b = (rows + kBRows - 1) / kBRows, ld = wy_ld(S);
        if (nwg_b * ld > m->cap_bpartial) {
            if ((rc = mrealloc(m, &m->d_bpartial, nwg_b * ld))) return rc;
            m->cap_bpartial = nwg_b * ld;
        }
    }
    if (nwg * (size_t)(S ? S : 1) > m->cap_partial) {
        if ((rc = mrealloc(m, &m->d_partial, nwg * (size_t)(S ? S : 1)))) return rc;
        m->cap_partial = nwg * (size_t)(S ? S : 1);
    }
    return FD_OK;
}

static int morph_factor_blocked(fd_morph *m)
{
    const int64_t rows = 3 * m->N;
    const int S = m->S;
    const unsigned nwg = (unsigned)((rows + kBRows - 1) / kBRows);
    hipStream_t st = m->stream;
    double *A = m->d_QR;
    for (int k0 = 0; k0 < S; k0 += kNB) {
        const int nb = S - k0 < kNB ? S - k0 : kNB, pend = k0 + nb, ntrail = S - pend;
        if (k0 == 0) hipLaunchKernelGGL(k_qr_gram, dim3(nwg), dim3(kT), 0, st, A, rows, 0, nb, m->d_bpartial);
        for (int k = k0; k < pend; ++k) {
            hipLaunchKernelGGL(k_qr_col_setup, dim3(1), dim3(kSetupT), 0, st, A, rows, k, pend - k, m->d_bpartial, (int)nwg, m->d_hh,
                               m->d_tau, m->d_tmp);
            hipLaunchKernelGGL(k_qr_panel_apply, dim3(nwg), dim3(kT), 0, st, A, rows, k, pend, m->d_hh, m->d_tmp, m->d_bpartial);
        }
        if (ntrail > 0) {
            const int ld = ntrail * kNB + 32, nvals = ntrail * kNB + kNB * (kNB - 1) / 2;
            hipLaunchKernelGGL(k_qr_wy_dots, dim3(nwg), dim3(kT), sizeof(double) * 4 * (size_t)ld, st, A, rows, k0, nb, S,
                               m->d_bpartial, ld);
            hipLaunchKernelGGL(k_qr_wy_reduce, dim3((nvals + 31) / 32, kSlices), dim3(kT), 0, st, m->d_bpartial, (int)nwg, ld,
                               nvals, m->d_bpartial2);
            hipLaunchKernelGGL(k_qr_wy_T, dim3(1), dim3(kT), sizeof(double) * ((size_t)ld + kNB * kNB), st, m->d_bpartial2, ld, k0,
                               nb, ntrail, m->d_tau, m->d_Y);
            hipLaunchKernelGGL(k_qr_wy_apply, dim3(nwg), dim3(kT), sizeof(double) * ((size_t)kNB * ntrail + 4 * kNB), st, A, rows,
                               k0, nb, S, m->d_Y, ntrail < kNB ? ntrail : kNB, m->d_bpartial);
        }
    }
    FDM_HIP(m, hipGetLastError());
    return FD_OK;
}

static int morph_factor(fd_morph *m)
{
    // FD_MORPH_UNBLOCKED: the column-by-column form (A/B measurements; also what very wide shape sets take,
    // whose per-workgroup sums would not fit the LDS of the block-reflector pass)
    static const bool unblocked = tuning_env("FD_MORPH_UNBLOCKED") != nullptr;
    if (!unblocked && m->S <= 256) return morph_factor_blocked(m);
    const int64_t rows = 3 * m->N;
    const int S = m->S;
    const unsigned nwg = (unsigned)((rows + kRowsPerWg - 1) / kRowsPerWg);
    hipStream_t st = m->stream;
    for (int k = 0; k < S; ++k) {
        double *x = m->d_QR + (size_t)k * rows;
        hipLaunchKernelGGL(k_qr_tail_norm, dim3(nwg), dim3(kT), 0, st, x, rows, k, m->d_partial);
        hipLaunchKernelGGL(k_qr_reflector, dim3(1), dim3(kT), 0, st, x, k, m->d_partial, (int)nwg, m->d_hh, m->d_tau);
        if (k + 1 < S) {
            const int nt = S - k - 1;
            hipLaunchKernelGGL(k_qr_dots, dim3(nwg), dim3(kT), sizeof(double) * 4 * (size_t)nt, st, m->d_QR, rows, S, k,
                               m->d_hh, m->d_partial, nt);
            hipLaunchKernelGGL(k_qr_dots_reduce, dim3(nt), dim3(kT), 0, st, m->d_QR, rows, k, m->d_hh, m->d_partial,
                               (int)nwg, nt, m->d_tmp);
            hipLaunchKernelGGL(k_qr_apply, dim3(nwg), dim3(kT), 0, st, m->d_QR, rows, S, k, m->d_hh, m->d_tmp);
        } else {
            // last column: only its own tail is scaled
            hipLaunchKernelGGL(k_qr_dots, dim3(nwg), dim3(kT), sizeof(double) * 4, st, m->d_QR, rows, S, k, m->d_hh,
                               m->d_partial, 1);
        }
    }
    FDM_HIP(m, hipGetLastError());
    return FD_OK;
}

extern "C" {

const char *fd_morph_last_error(const fd_morph *m) { return m ? m->err : g_merr; }

fd_morph *fd_morph_create(const fd_config *cfg)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        merr(nullptr, "fd_morph_create: no HIP device visible; this engine has no CPU path");
        return nullptr;
    }
    fd_morph *m = new (std::nothrow) fd_morph();
    if (!m) { merr(nullptr, "fd_morph_create: out of host memory"); return nullptr; }
    int dev = cfg ? cfg->device : -1;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    hipDeviceProp_t prop;
    if (dev >= ndev || hipSetDevice(dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        merr(nullptr, "fd_morph_create: cannot use device %d", dev);
        delete m;
        return nullptr;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        merr(nullptr, "fd_morph_create: device %d is %s; kernels are built for gfx950 (MI355X) only", dev, prop.gcnArchName);
        delete m;
        return nullptr;
    }
    m->device = dev;
    bool ok = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreate(&m->ev0) == hipSuccess && hipEventCreate(&m->ev1) == hipSuccess;
    ok = ok && hipMalloc((void **)&m->d_hh, 4 * sizeof(double)) == hipSuccess;
    if (!ok) {
        merr(nullptr, "fd_morph_create: device resource allocation failed: %s", hipGetErrorString(hipGetLastError()));
        fd_morph_destroy(m);
        return nullptr;
    }
    return m;
}

void fd_morph_destroy(fd_morph *m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    void *bufs[] = {m->d_rest_attr, m->d_rest, m->d_S32, m->d_stage, m->d_P, m->d_QR, m->d_tau, m->d_w, m->d_partial, m->d_hh, m->d_tmp,
                    m->d_bpartial, m->d_bpartial2, m->d_Y};
    for (void *p : bufs) if (p) (void)hipFree(p);
    if (m->ev0) (void)hipEventDestroy(m->ev0);
    if (m->ev1) (void)hipEventDestroy(m->ev1);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

static int morph_init_common(fd_morph *m, int64_t N, int S, const float *rest, const float *const *shapes, bool on_device)
{
    if (!m) return FD_E_INVALID;
    if (N <= 0 || S < 0 || !rest || (S > 0 && !shapes)) { merr(m, "fd_morph_init: need N > 0, S >= 0 and the arrays"); return FD_E_INVALID; }
    if ((int64_t)S > 3 * N) { merr(m, "fd_morph_init: more blendshapes (%d) than rows (%lld)", S, (long long)(3 * N)); return FD_E_INVALID; }
    for (int s = 0; s < S; ++s)
        if (!shapes[s]) { merr(m, "fd_morph_init: shape %d is NULL", s); return FD_E_INVALID; }
    FDM_HIP(m, hipSetDevice(m->device));
    int rc = morph_reserve(m, N, S);
    if (rc) return rc;
    m->N = N; m->S = S;
    m->initialised = false; m->computed = false;
    m->use_rest_attr = false;
    const size_t rows = 3 * (size_t)N, bytes = rows * sizeof(float);
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    hipStream_t st = m->stream;
    FDM_HIP(m, hipMemcpyAsync(m->d_rest, rest, bytes, kind, st));
    FDM_HIP(m, hipEventRecord(m->ev0, st));
    const unsigned g = (unsigned)((rows + kT - 1) / kT);
    for (int s = 0; s < S; ++s) {
        const float *src = shapes[s];
        if (!on_device) {
            FDM_HIP(m, hipMemcpyAsync(m->d_stage, shapes[s], bytes, hipMemcpyHostToDevice, st));
            src = m->d_stage;
        }
        hipLaunchKernelGGL(k_shape_column, dim3(g), dim3(kT), 0, st, m->d_rest, src, (int64_t)rows,
                           m->d_QR + (size_t)s * rows, m->d_S32 + (size_t)s * rows);
    }
    if ((rc = morph_factor(m))) return rc;
    FDM_HIP(m, hipEventRecord(m->ev1, st));
    FDM_HIP(m, hipStreamSynchronize(st));
    (void)hipEventElapsedTime(&m->last_init_ms, m->ev0, m->ev1);
    m->initialised = true;
    return FD_OK;
}

int fd_morph_init(fd_morph *m, int64_t N, int S, const float *rest_xyz, const float *const *shapes_xyz)
{
    return morph_init_common(m, N, S, rest_xyz, shapes_xyz, false);
}

int fd_morph_init_dev(fd_morph *m, int64_t N, int S, const float *d_rest_xyz, const float *const *d_shapes_xyz)
{
    return morph_init_common(m, N, S, d_rest_xyz, d_shapes_xyz, true);
}

int fd_morph_set_rest(fd_morph *m, const float *rest_xyz, int on_device)
{
    if (!m) return FD_E_INVALID;
    if (!m->initialised) { merr(m, "fd_morph_set_rest: fd_morph_init has not succeeded"); return FD_E_NOT_BUILT; }
    if (!rest_xyz) { m->use_rest_attr = false; return FD_OK; }
    FDM_HIP(m, hipSetDevice(m->device));
    const size_t rows = 3 * (size_t)m->N;
    if (!m->d_rest_attr) {
        int rc = mrealloc(m, &m->d_rest_attr, 3 * (size_t)m->cap_N);
        if (rc) return rc;
    }
    FDM_HIP(m, hipMemcpyAsync(m->d_rest_attr, rest_xyz, rows * sizeof(float),
                              on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, m->stream));
    if (!on_device) FDM_HIP(m, hipStreamSynchronize(m->stream));
    m->use_rest_attr = true;
    return FD_OK;
}

int fd_morph_is_initialised(const fd_morph *m) { return m && m->initialised && m->N > 0; }   // dbse.hpp:16 isInitialized
int fd_morph_is_computed(const fd_morph *m) { return m && m->computed; }                      // dbse.hpp:18 isComputed
int fd_morph_shape_count(const fd_morph *m) { return m ? m->S : 0; }
float fd_morph_last_init_ms(const fd_morph *m) { return m ? m->last_init_ms : 0.f; }

int fd_morph_compute_weights_dev(fd_morph *m, const float *d_P_xyz, void *hip_stream)
{
    if (!m || !d_P_xyz) return FD_E_INVALID;
    if (!m->initialised) { merr(m, "fd_morph_compute_weights: fd_morph_init has not succeeded"); return FD_E_NOT_BUILT; }
    FDM_HIP(m, hipSetDevice(m->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : m->stream;
    const int64_t rows = 3 * m->N;
    const unsigned nwg = (unsigned)((rows + kWRows - 1) / kWRows);
    if (m->S > 0) {
        hipLaunchKernelGGL(k_morph_weights, dim3(nwg), dim3(kT), sizeof(double) * 4 * (size_t)m->S, st, m->d_QR, rows, m->S,
                           d_P_xyz, m->use_rest_attr ? m->d_rest_attr : m->d_rest, m->d_partial);
        hipLaunchKernelGGL(k_morph_weights_reduce, dim3(m->S), dim3(kT), 0, st, m->d_partial, (int)nwg, m->S, m->d_w);
    }
    FDM_HIP(m, hipGetLastError());
    m->computed = true;
    return FD_OK;
}

int fd_morph_displace_dev(fd_morph *m, float *d_P_xyz, const float *clamp_lo_hi, int add_delta, float falloffradius,
                          void *hip_stream)
{
    if (!m || !d_P_xyz) return FD_E_INVALID;
    if (!m->initialised || !m->computed) { merr(m, "fd_morph_displace: weights have not been computed"); return FD_E_NOT_BUILT; }
    FDM_HIP(m, hipSetDevice(m->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : m->stream;
    const unsigned g = (unsigned)((m->N + kT - 1) / kT);
    hipLaunchKernelGGL(k_morph_displace, dim3(g), dim3(kT), sizeof(float) * (size_t)(m->S ? m->S : 1), st, m->d_S32, m->N,
                       m->S, m->d_w, clamp_lo_hi ? clamp_lo_hi[0] : 0.f, clamp_lo_hi ? clamp_lo_hi[1] : 0.f,
                       clamp_lo_hi ? 1 : 0, add_delta ? 1 : 0, falloffradius, m->use_rest_attr ? m->d_rest_attr : m->d_rest,
                       d_P_xyz);
    FDM_HIP(m, hipGetLastError());
    return FD_OK;
}

int fd_morph_get_weights(fd_morph *m, double *w)
{
    if (!m || !w) return FD_E_INVALID;
    if (!m->computed) { merr(m, "fd_morph_get_weights: weights have not been computed"); return FD_E_NOT_BUILT; }
    FDM_HIP(m, hipSetDevice(m->device));
    FDM_HIP(m, hipMemcpyAsync(w, m->d_w, sizeof(double) * (size_t)m->S, hipMemcpyDeviceToHost, m->stream));
    FDM_HIP(m, hipStreamSynchronize(m->stream));
    return FD_OK;
}

int fd_morph_get_qr(fd_morph *m, double *qr, double *tau)
{
    if (!m || !qr) return FD_E_INVALID;
    if (!m->initialised) { merr(m, "fd_morph_get_qr: fd_morph_init has not succeeded"); return FD_E_NOT_BUILT; }
    FDM_HIP(m, hipSetDevice(m->device));
    FDM_HIP(m, hipMemcpyAsync(qr, m->d_QR, sizeof(double) * 3 * (size_t)m->N * (size_t)m->S, hipMemcpyDeviceToHost, m->stream));
    if (tau) FDM_HIP(m, hipMemcpyAsync(tau, m->d_tau, sizeof(double) * (size_t)m->S, hipMemcpyDeviceToHost, m->stream));
    FDM_HIP(m, hipStreamSynchronize(m->stream));
    return FD_OK;
}

int fd_morph_apply(fd_morph *m, float *P_xyz, const float *clamp_lo_hi, int add_delta, float falloffradius, double *w_out)
{
    if (!m || !P_xyz) return FD_E_INVALID;
    if (!m->initialised) { merr(m, "fd_morph_apply: fd_morph_init has not succeeded"); return FD_E_NOT_BUILT; }
    FDM_HIP(m, hipSetDevice(m->device));
    const size_t bytes = sizeof(float) * 3 * (size_t)m->N;
    FDM_HIP(m, hipMemcpyAsync(m->d_P, P_xyz, bytes, hipMemcpyHostToDevice, m->stream));
    int rc = fd_morph_compute_weights_dev(m, m->d_P, m->stream);
    if (!rc) rc = fd_morph_displace_dev(m, m->d_P, clamp_lo_hi, add_delta, falloffradius, m->stream);
    if (rc) return rc;
    FDM_HIP(m, hipMemcpyAsync(P_xyz, m->d_P, bytes, hipMemcpyDeviceToHost, m->stream));
    if (w_out && m->S > 0) FDM_HIP(m, hipMemcpyAsync(w_out, m->d_w, sizeof(double) * (size_t)m->S, hipMemcpyDeviceToHost, m->stream));
    FDM_HIP(m, hipStreamSynchronize(m->stream));
    return FD_OK;
}

}  // extern "C"
