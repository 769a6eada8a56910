// fd_build.hip -- kernel-matrix assembly + dense fp64 solve for the RBF weights.
//
// Replaces alglib::rbfsetpoints / rbfsetalgo* / rbfset*term / rbfbuildmodel,
// reference src/SOP_FaceDeform.cpp:331-368, with the dense formulation
//
//     [ Phi + lambda*I   P ] [ w ]   [ f ]        Phi_ij = phi_j(|c_i - c_j|^2)
//     [ P^T              0 ] [ v ] = [ 0 ]        P = [1 x y z] (linear term)
//
// Device data layout: A is column-major, lda x ncols fp64; the order n = M + T
// is padded with an identity block to npad (a multiple of 32) so that no kernel
// has edge tiles; columns npad..npad+15 carry the three right-hand sides (one
// MFMA tile wide), so forward elimination is applied to them by the same
// trailing-update kernel and L is never revisited.
//
// LU with partial pivoting, right-looking, two launches per block step:
//   k_lu_panel  one workgroup keeps the whole panel (rows x NB) in registers --
//               32 doubles per lane; NB shrinks 32/16/8/4 as the panel gets
//               taller, since 1024 lanes x 64 VGPRs is what one CU can hold.
//               Pivot search = wave shuffles + one LDS exchange, one barrier
//               per column.
//   k_lu_trail  one workgroup per 16-column block of the trailing matrix: row
//               interchanges (as a gather/scatter of the <= 2*NB moved rows),
//               the NB x 16 triangular solve in registers in MFMA B-fragment
//               layout, then C -= L21 * U12 with v_mfma_f64_16x16x4_f64.
// Back substitution: one launch per 32-row block, bottom up.
#include "fd_internal.h"

namespace fd {

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr double kEps = 2.220446049250313e-16;
constexpr int kColBlock = 16;

// ---- phi in fp64 (assembly) ---------------------------------------------------
__device__ __forceinline__ double phi_d(int kind, double d2, double inv_r2)
{
    switch (kind) {
    case FD_KERNEL_GAUSSIAN:
    case FD_KERNEL_GAUSSIAN_QNN: return exp(-d2 * inv_r2);
    case FD_KERNEL_THIN_PLATE: return d2 > 0.0 ? 0.5 * d2 * log(d2) : 0.0;
    case FD_KERNEL_BIHARMONIC: return -sqrt(d2);
    default: return d2 * sqrt(d2);
    }
}

// ---- prepare: centres, RHS columns, status reset ------------------------------
// reference src/SOP_FaceDeform.cpp:268-287 (table), widened to fp64
__global__ void k_prepare(const float *rest, const float *delta, int M, int npad, int lda,
                          double *centres, double *radii, double gauss_R, double *A,
                          DevModel *model)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        model->terminationtype = 0;
        model->dup_flag = 0;
        model->sing_flag = 0;
        model->iterations = 0;
        model->amax_bits = 0ull;
        model->pivmin_bits = 0x7FF0000000000000ull;  // +inf
        model->pivmax_bits = 0ull;
    }
    if (i < npad) {
        for (int c = 0; c < kRhsCols; ++c) {
            double v = 0.0;
            if (i < M && c < 3) v = (double)delta[3 * i + c];
            A[(size_t)(npad + c) * lda + i] = v;
        }
    }
    if (i < M) {
        centres[3 * i] = (double)rest[3 * i];
        centres[3 * i + 1] = (double)rest[3 * i + 1];
        centres[3 * i + 2] = (double)rest[3 * i + 2];
        radii[i] = gauss_R;
    }
}

// QNN radii: R_i = q * distance to the nearest other centre
__global__ void k_qnn_nearest(const double *centres, int M, double q, double *radii)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const double x = centres[3 * i], y = centres[3 * i + 1], z = centres[3 * i + 2];
    double best = INFINITY;
    for (int j = 0; j < M; ++j) {
        if (j == i) continue;
        const double dx = x - centres[3 * j], dy = y - centres[3 * j + 1], dz = z - centres[3 * j + 2];
        const double d2 = dx * dx + dy * dy + dz * dz;
        best = d2 < best ? d2 : best;
    }
    radii[i] = M > 1 ? q * sqrt(best) : q;
}

// lower median by rank selection, then R_i = min(R_i, z * median)
__global__ void k_qnn_median(const double *radii, int M, double *median_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const double r = radii[i];
    int rank = 0;
    for (int j = 0; j < M; ++j) {
        const double o = radii[j];
        rank += (o < r) || (o == r && j < i);
    }
    if (rank == (M - 1) / 2) *median_out = r;
}
__global__ void k_qnn_cap(double *radii, int M, double z, const double *median)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const double cap = z * (*median);
    if (radii[i] > cap) radii[i] = cap;
}

// ---- assembly -----------------------------------------------------------------
__global__ __launch_bounds__(256) void k_assemble(const double *centres, const double *radii, int M,
                                                   int n, int npad, int lda, int kind, int T,
                                                   double lambda, double *A, DevModel *model)
{
    // 16 x 16 element tile per workgroup; consecutive threads walk a column (coalesced)
    const int i = blockIdx.x * 16 + (threadIdx.x & 15);
    const int j = blockIdx.y * 16 + (threadIdx.x >> 4);
    double v = 0.0;
    bool real = false;
    if (i < npad && j < npad) {
        if (i < M && j < M) {
            const double dx = centres[3 * i] - centres[3 * j];
            const double dy = centres[3 * i + 1] - centres[3 * j + 1];
            const double dz = centres[3 * i + 2] - centres[3 * j + 2];
            const double d2 = dx * dx + dy * dy + dz * dz;
            const double r = radii[j];
            v = phi_d(kind, d2, 1.0 / (r * r));
            if (i == j) v += lambda;
            else if (d2 == 0.0) model->dup_flag = 1;  // coincident centres -> -5
            real = true;
        } else if (i < n && j < n) {
            // polynomial block: column M is 1, columns M+1..M+3 are x, y, z
            const int row = i < M ? i : j;      // the centre index
            const int col = (i < M ? j : i) - M;
            if (i < M || j < M) v = col == 0 ? 1.0 : centres[3 * row + col - 1];
            real = true;
            (void)T;
        } else {
            v = (i == j) ? 1.0 : 0.0;           // identity padding
        }
        A[(size_t)j * lda + i] = v;
    }
    // max |A_ij| over the real system, for the singularity threshold
    double m = real ? fabs(v) : 0.0;
    for (int off = 32; off >= 1; off >>= 1) {
        const double o = __shfl_xor(m, off);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0 && m > 0.0)
        atomicMax(&model->amax_bits, (unsigned long long)__double_as_longlong(m));
}

// ---- LU panel -------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(kPanelThreads) void k_lu_panel(double *A, int lda, int npad, int n_real,
                                                            int k0, int *ipiv, int *moves,
                                                            DevModel *model)
{
    constexpr int R = 32 / NB;          // rows per lane: R * NB = 32 doubles in registers
    constexpr int kDiag = 16;           // LDS slot of the current diagonal row
    __shared__ double s_row[2][17][NB];
    __shared__ int s_orig[2][17];
    __shared__ double s_val[2][16];
    __shared__ int s_idx[2][16];
    __shared__ int s_count;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nthreads = blockDim.x;
    const int nwaves = nthreads >> 6;
    const int nrem = npad - k0;

    double a[R][NB];
    int orig[R];
#pragma clang loop unroll(full)
    for (int t = 0; t < R; ++t) {
        const int slot = tid + t * nthreads;
        orig[t] = k0 + slot;
#pragma clang loop unroll(full)
        for (int c = 0; c < NB; ++c)
            a[t][c] = slot < nrem ? A[(size_t)(k0 + c) * lda + k0 + slot] : 0.0;
    }
    if (tid == 0) s_count = 0;

    const double amax = __longlong_as_double((long long)model->amax_bits);
    const double tiny = (double)n_real * kEps * amax;
    double pmin = INFINITY, pmax = 0.0;
    bool singular = false;

#pragma clang loop unroll(full)
    for (int j = 0; j < NB; ++j) {
        const int buf = j & 1;
        // candidate among my rows at or below the diagonal
        double best = -1.0;
        int bslot = 0x7fffffff;
#pragma clang loop unroll(full)
        for (int t = 0; t < R; ++t) {
            const int slot = tid + t * nthreads;
            if (slot >= j && slot < nrem) {
                const double v = fabs(a[t][j]);
                if (v > best || (v == best && slot < bslot)) { best = v; bslot = slot; }
            }
        }
        // wave argmax: larger |a| wins, ties go to the smaller row (first maximum)
#pragma clang loop unroll(full)
        for (int off = 32; off >= 1; off >>= 1) {
            const double ov = __shfl_xor(best, off);
            const int os = __shfl_xor(bslot, off);
            if (ov > best || (ov == best && os < bslot)) { best = ov; bslot = os; }
        }
#pragma clang loop unroll(full)
        for (int t = 0; t < R; ++t) {
            const int slot = tid + t * nthreads;
            if (slot == bslot && best >= 0.0) {
#pragma clang loop unroll(full)
                for (int c = 0; c < NB; ++c) s_row[buf][wave][c] = a[t][c];
                s_orig[buf][wave] = orig[t];
            }
            if (slot == j) {
#pragma clang loop unroll(full)
                for (int c = 0; c < NB; ++c) s_row[buf][kDiag][c] = a[t][c];
                s_orig[buf][kDiag] = orig[t];
            }
        }
        if (lane == 0) { s_val[buf][wave] = best; s_idx[buf][wave] = bslot; }
        __syncthreads();

        double gbest = -1.0;
        int gslot = 0x7fffffff, gw = 0;
        for (int w = 0; w < nwaves; ++w) {
            const double v = s_val[buf][w];
            const int s = s_idx[buf][w];
            if (v > gbest || (v == gbest && s < gslot)) { gbest = v; gslot = s; gw = w; }
        }
        // interchange rows j and gslot (contents travel, slots stay)
        if (gslot != j) {
#pragma clang loop unroll(full)
            for (int t = 0; t < R; ++t) {
                const int slot = tid + t * nthreads;
                if (slot == j) {
#pragma clang loop unroll(full)
                    for (int c = 0; c < NB; ++c) a[t][c] = s_row[buf][gw][c];
                    orig[t] = s_orig[buf][gw];
                } else if (slot == gslot) {
#pragma clang loop unroll(full)
                    for (int c = 0; c < NB; ++c) a[t][c] = s_row[buf][kDiag][c];
                    orig[t] = s_orig[buf][kDiag];
                }
            }
        }
        const double piv = s_row[buf][gw][j];
        const bool ok = gbest > tiny;   // false for NaN as well
        if (k0 + j < n_real) {
            if (!ok) singular = true;
            pmin = gbest < pmin ? gbest : pmin;
            pmax = gbest > pmax ? gbest : pmax;
        }
        if (tid == 0) ipiv[k0 + j] = k0 + gslot;
        const double inv = ok ? 1.0 / piv : 0.0;
#pragma clang loop unroll(full)
        for (int t = 0; t < R; ++t) {
            const int slot = tid + t * nthreads;
            if (slot > j && slot < nrem) {
                const double l = a[t][j] * inv;
                a[t][j] = l;
#pragma clang loop unroll(full)
                for (int c = j + 1; c < NB; ++c) a[t][c] = fma(-l, s_row[buf][gw][c], a[t][c]);
            }
        }
    }

    // write the factored panel back, and the list of rows that moved
#pragma clang loop unroll(full)
    for (int t = 0; t < R; ++t) {
        const int slot = tid + t * nthreads;
        if (slot < nrem) {
#pragma clang loop unroll(full)
            for (int c = 0; c < NB; ++c) A[(size_t)(k0 + c) * lda + k0 + slot] = a[t][c];
            if (orig[t] != k0 + slot) {
                const int q = atomicAdd(&s_count, 1);
                moves[1 + 2 * q] = k0 + slot;   // destination row
                moves[2 + 2 * q] = orig[t];     // row (as of panel start) whose content lands there
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        moves[0] = s_count;
        const int done = (k0 + NB < n_real ? k0 + NB : n_real);
        model->iterations = done;
        if (singular) model->sing_flag = 1;
        if (pmax > 0.0 || pmin < INFINITY) {
            atomicMin(&model->pivmin_bits, (unsigned long long)__double_as_longlong(pmin));
            atomicMax(&model->pivmax_bits, (unsigned long long)__double_as_longlong(pmax));
        }
    }
}

// ---- LU trailing update ---------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(256) void k_lu_trail(double *A, int lda, int npad, int k0,
                                                  const int *moves)
{
    constexpr int S = NB / 4;           // k-steps of the f64 16x16x4 MFMA
    __shared__ double sL[NB][NB + 1];   // L11 (unit lower), +1 pad: column reads conflict-free

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int c0 = k0 + NB + blockIdx.x * kColBlock;
    const int c = lane & 15;            // column inside the block
    const int g = lane >> 4;            // row group (MFMA k index)

    // 1. row interchanges of this panel, restricted to my 16 columns:
    //    read every moved row first, then write (the two sets overlap)
    const int nmov = moves[0];
    double tmp[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = tid + q * 256;
        tmp[q] = 0.0;
        if (e < nmov * kColBlock) {
            const int m = e >> 4;
            tmp[q] = A[(size_t)(c0 + (e & 15)) * lda + moves[2 + 2 * m]];
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = tid + q * 256;
        if (e < nmov * kColBlock) {
            const int m = e >> 4;
            A[(size_t)(c0 + (e & 15)) * lda + moves[1 + 2 * m]] = tmp[q];
        }
    }
    for (int e = tid; e < NB * NB; e += 256) {
        const int r = e % NB, cc = e / NB;
        sL[r][cc] = A[(size_t)(k0 + cc) * lda + k0 + r];
    }
    __syncthreads();

    // 2. U12 = L11^-1 * A12 for my columns, in registers, already in the layout of
    //    the MFMA B operand: lane (c, g) holds rows g + 4s.  Every wave does it.
    double u[S];
#pragma unroll
    for (int s = 0; s < S; ++s) u[s] = A[(size_t)(c0 + c) * lda + k0 + g + 4 * s];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const double uj = __shfl(u[j >> 2], c + 16 * (j & 3));
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int i = g + 4 * s;
            if (i > j) u[s] = fma(-sL[i][j], uj, u[s]);
        }
    }
    if (wave == 0) {
#pragma unroll
        for (int s = 0; s < S; ++s) A[(size_t)(c0 + c) * lda + k0 + g + 4 * s] = u[s];
    }
#pragma unroll
    for (int s = 0; s < S; ++s) u[s] = -u[s];

    // 3. A22 -= L21 * U12, 16 x 16 tiles, one wave per tile.  Tiles are aligned to 16
    //    rows; when NB < 16 the first tile starts inside the panel rows, whose lanes
    //    feed zeros to the MFMA and are not stored.
    const int rfirst = k0 + NB;
    const int rbase = rfirst & ~15;
    const int ntiles = (npad - rbase) / 16;
    for (int tile = wave; tile < ntiles; tile += 4) {
        const int r0 = rbase + tile * 16;
        double *cptr = A + (size_t)(c0 + c) * lda + r0 + g;
        double4_t acc;
        acc[0] = cptr[0]; acc[1] = cptr[4]; acc[2] = cptr[8]; acc[3] = cptr[12];
        const double *aptr = A + (size_t)(k0 + g) * lda + r0 + c;
        const bool arow_ok = r0 + c >= rfirst;
        double av[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const double v = aptr[(size_t)(4 * s) * lda];
            av[s] = arow_ok ? v : 0.0;
        }
#pragma unroll
        for (int s = 0; s < S; ++s)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], u[s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r0 + g + 4 * r >= rfirst) cptr[4 * r] = acc[r];
    }
}

// ---- back substitution, one 32-row block per launch ------------------------------
// Solves U * X = Y bottom-up.  Y lives in columns npad.. of A; X goes to its own
// buffer so that workgroups re-solving the diagonal block never race with a writer.
__global__ __launch_bounds__(256) void k_backsub(double *A, int lda, int npad, int b0, double *X)
{
    __shared__ double sU[32][33];
    __shared__ double sx[32][3];
    const int tid = threadIdx.x;
    for (int e = tid; e < 32 * 32; e += 256) {
        const int r = e & 31, cc = e >> 5;
        sU[r][cc] = A[(size_t)(b0 + cc) * lda + b0 + r];
    }
    if (tid < 96) sx[tid & 31][tid >> 5] = A[(size_t)(npad + (tid >> 5)) * lda + b0 + (tid & 31)];
    __syncthreads();
    for (int k = 31; k >= 0; --k) {
        if (tid < 3) sx[k][tid] = sx[k][tid] / sU[k][k];
        __syncthreads();
        if (tid < 96) {
            const int i = tid & 31, rc = tid >> 5;
            if (i < k) sx[i][rc] = fma(-sU[i][k], sx[k][rc], sx[i][rc]);
        }
        __syncthreads();
    }
    if (blockIdx.x == 0 && tid < 96) X[(size_t)(tid >> 5) * npad + b0 + (tid & 31)] = sx[tid & 31][tid >> 5];
    // rows above the block: y_i -= U[i, b0:b0+32] * x_b
    const int i = blockIdx.x * 256 + tid;
    if (i < b0) {
        double y0 = 0.0, y1 = 0.0, y2 = 0.0;
        for (int k = 0; k < 32; ++k) {
            const double uik = A[(size_t)(b0 + k) * lda + i];
            y0 = fma(uik, sx[k][0], y0);
            y1 = fma(uik, sx[k][1], y1);
            y2 = fma(uik, sx[k][2], y2);
        }
        A[(size_t)(npad + 0) * lda + i] -= y0;
        A[(size_t)(npad + 1) * lda + i] -= y1;
        A[(size_t)(npad + 2) * lda + i] -= y2;
    }
}

// ---- pack: solution -> weights, evaluation records, status -----------------------
// from_w != 0: the weights are already in W (fd_import_model); only the records,
// the affine part and the status are produced.
__global__ __launch_bounds__(256) void k_pack(const double *X, int npad, const double *centres,
                                              const double *radii, int M, int Mpad, int T, int kind,
                                              double *W, Rec32 *rec32, Rec64 *rec64, DevModel *model,
                                              int from_w)
{
    __shared__ int s_bad;
    const int tid = threadIdx.x;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    const double kLn2 = 0.6931471805599453;
    const double kLog2e = 1.4426950408889634;
    double wscale32 = 1.0, wscale64 = 1.0;
    if (kind == FD_KERNEL_THIN_PLATE) { wscale32 = 0.5 * kLn2; wscale64 = 0.5; }
    if (kind == FD_KERNEL_BIHARMONIC) { wscale32 = -1.0; wscale64 = -1.0; }
    const bool gauss = kind == FD_KERNEL_GAUSSIAN || kind == FD_KERNEL_GAUSSIAN_QNN;
    bool bad = false;
    for (int j = tid; j < Mpad; j += 256) {
        Rec32 r32 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        Rec64 r64 = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (j < M) {
            const double wx = from_w ? W[3 * j] : X[j];
            const double wy = from_w ? W[3 * j + 1] : X[npad + j];
            const double wz = from_w ? W[3 * j + 2] : X[2 * (size_t)npad + j];
            bad |= !(isfinite(wx) && isfinite(wy) && isfinite(wz));
            if (!from_w) { W[3 * j] = wx; W[3 * j + 1] = wy; W[3 * j + 2] = wz; }
            const double R = radii[j];
            r64.cx = centres[3 * j]; r64.cy = centres[3 * j + 1]; r64.cz = centres[3 * j + 2];
            r64.s = gauss ? -1.0 / (R * R) : 0.0;
            r64.wx = wx * wscale64; r64.wy = wy * wscale64; r64.wz = wz * wscale64;
            r32.cx = (float)r64.cx; r32.cy = (float)r64.cy; r32.cz = (float)r64.cz;
            r32.s = gauss ? (float)(-kLog2e / (R * R)) : 0.f;
            r32.wx = (float)(wx * wscale32); r32.wy = (float)(wy * wscale32); r32.wz = (float)(wz * wscale32);
        }
        rec32[j] = r32;
        rec64[j] = r64;
    }
    if (tid < 12) {
        // W rows M..M+3 = const, x, y, z;  affine[c*4 + k] = coefficient k of output c
        const int cc = tid / 4, k = tid % 4;
        double v = 0.0;
        if (from_w) v = W[3 * (M + k) + cc];
        else if (k < T) v = X[(size_t)cc * npad + M + k];
        bad |= !isfinite(v);
        if (!from_w) W[3 * (M + k) + cc] = v;
        model->affine64[tid] = v;
        model->affine32[tid] = (float)v;
    }
    if (bad) s_bad = 1;
    __syncthreads();
    if (tid == 0) {
        int tt = 1;
        if (from_w) {
            if (s_bad) tt = -4;
        } else {
            if (model->sing_flag || s_bad) tt = -4;
            if (model->dup_flag) tt = -5;
        }
        model->terminationtype = tt;
    }
}

template <int NB>
void lu_step(const BuildBuffers &b, int k0, hipStream_t stream)
{
    constexpr int R = 32 / NB;
    const int nrem = b.npad - k0;
    int threads = round_up((nrem + R - 1) / R, 64);
    if (threads < 64) threads = 64;
    hipLaunchKernelGGL((k_lu_panel<NB>), dim3(1), dim3(threads), 0, stream, b.d_A, b.lda, b.npad,
                       b.n, k0, b.d_ipiv, b.d_moves, b.d_model);
    // the last block may run into the 16 zero columns allocated past ncols
    const int ncb = (b.ncols - (k0 + NB) + kColBlock - 1) / kColBlock;
    if (ncb > 0)
        hipLaunchKernelGGL((k_lu_trail<NB>), dim3(ncb), dim3(256), 0, stream, b.d_A, b.lda, b.npad,
                           k0, b.d_moves);
}

}  // namespace

hipError_t launch_build(const BuildBuffers &b, hipStream_t stream, hipEvent_t ev_mid)
{
    const int M = b.M;
    {
        const int threads = 256;
        const int blocks = (b.npad + threads - 1) / threads;
        hipLaunchKernelGGL(k_prepare, dim3(blocks), dim3(threads), 0, stream, b.d_rest, b.d_delta, M,
                           b.npad, b.lda, b.d_centres, b.d_radii, b.gauss_R, b.d_A, b.d_model);
        if (b.kind == FD_KERNEL_GAUSSIAN_QNN) {
            const int mb = (M + threads - 1) / threads;
            double *median = b.d_W;  // scratch: W is rewritten by the pack kernel
            hipLaunchKernelGGL(k_qnn_nearest, dim3(mb), dim3(threads), 0, stream, b.d_centres, M,
                               b.qnn_q, b.d_radii);
            hipLaunchKernelGGL(k_qnn_median, dim3(mb), dim3(threads), 0, stream, b.d_radii, M, median);
            hipLaunchKernelGGL(k_qnn_cap, dim3(mb), dim3(threads), 0, stream, b.d_radii, M, b.qnn_z,
                               median);
        }
        const dim3 grid(b.npad / 16, b.npad / 16);
        hipLaunchKernelGGL(k_assemble, grid, dim3(256), 0, stream, b.d_centres, b.d_radii, M, b.n,
                           b.npad, b.lda, b.kind, b.T, b.lambda, b.d_A, b.d_model);
    }
    if (ev_mid) (void)hipEventRecord(ev_mid, stream);

    int k0 = 0;
    while (k0 < b.npad) {
        const int nrem = b.npad - k0;
        if (nrem <= 1024) { lu_step<32>(b, k0, stream); k0 += 32; }
        else if (nrem <= 2048) { lu_step<16>(b, k0, stream); k0 += 16; }
        else if (nrem <= 4096) { lu_step<8>(b, k0, stream); k0 += 8; }
        else { lu_step<4>(b, k0, stream); k0 += 4; }
    }
    double *X = b.d_X;
    for (int b0 = b.npad - 32; b0 >= 0; b0 -= 32) {
        const int blocks = b0 > 0 ? (b0 + 255) / 256 : 1;
        hipLaunchKernelGGL(k_backsub, dim3(blocks), dim3(256), 0, stream, b.d_A, b.lda, b.npad, b0, X);
    }
    hipError_t e = launch_pack(b, stream);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

hipError_t launch_pack(const BuildBuffers &b, hipStream_t stream)
{
    const double *X = b.d_X;
    hipLaunchKernelGGL(k_pack, dim3(1), dim3(256), 0, stream, X, b.npad, b.d_centres, b.d_radii, b.M,
                       b.Mpad, b.T, b.kind, b.d_W, b.d_rec32, b.d_rec64, b.d_model, 0);
    return hipGetLastError();
}

hipError_t launch_pack_from_weights(const BuildBuffers &b, hipStream_t stream)
{
    hipLaunchKernelGGL(k_pack, dim3(1), dim3(256), 0, stream, (const double *)nullptr, b.npad,
                       b.d_centres, b.d_radii, b.M, b.Mpad, b.T, b.kind, b.d_W, b.d_rec32, b.d_rec64,
                       b.d_model, 1);
    return hipGetLastError();
}

}  // namespace fd
