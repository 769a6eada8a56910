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
// Back substitution: one workgroup, bottom up over 32-row blocks.
#include "fd_internal.h"
#include "fd_pack.h"

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
// Every build kernel finds its buffers in tab[blockIdx.z]: one launch chain assembles and
// factorises all the models of a batch (a single build is a batch of one).  With use_src the
// control points come straight from the caller's device arrays (and are mirrored into the
// context's own copy so that a later single rebuild still finds them).
__global__ void k_prepare(const BatchSlot *tab, const PointSrc src, int use_src, int M, int npad, int lda,
                          int ncols, double gauss_R)
{
    const BatchSlot &s = tab[blockIdx.z];
    const float *rest = use_src ? src.rest[blockIdx.z] : s.rest;
    const float *delta = use_src ? src.delta[blockIdx.z] : s.delta;
    double *centres = s.centres, *radii = s.radii, *A = s.A;
    DevModel *model = s.model;
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
        // the 16 columns past the RHS block absorb the trailing update's block overrun
        for (int c = 0; c < 16; ++c) A[(size_t)(ncols + c) * lda + i] = 0.0;
    }
    if (i < M) {
        if (use_src) {
            for (int c = 0; c < 3; ++c) { s.rest[3 * i + c] = rest[3 * i + c]; s.delta[3 * i + c] = delta[3 * i + c]; }
        }
        centres[3 * i] = (double)rest[3 * i];
        centres[3 * i + 1] = (double)rest[3 * i + 1];
        centres[3 * i + 2] = (double)rest[3 * i + 2];
        radii[i] = gauss_R;
    }
}

// New right-hand sides for an existing factorisation (fd_set_deltas): only the RHS columns are
// rewritten; centres, matrix and the singular / duplicate flags of the factorisation stay.
__global__ void k_prepare_rhs(const BatchSlot *tab, const PointSrc src, int use_src, int M, int npad, int lda)
{
    const BatchSlot &s = tab[blockIdx.z];
    const float *delta = use_src ? src.delta[blockIdx.z] : s.delta;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) s.model->terminationtype = 0;
    if (i < npad) {
        for (int c = 0; c < kRhsCols; ++c) {
            double v = 0.0;
            if (i < M && c < 3) v = (double)delta[3 * i + c];
            s.A[(size_t)(npad + c) * lda + i] = v;
        }
    }
    if (use_src && i < M)
        for (int c = 0; c < 3; ++c) s.delta[3 * i + c] = delta[3 * i + c];
}

// QNN radii: R_i = q * distance to the nearest other centre
__global__ void k_qnn_nearest(const BatchSlot *tab, int M, double q)
{
    const double *centres = tab[blockIdx.z].centres;
    double *radii = tab[blockIdx.z].radii;
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

// median by rank selection -- element M / 2 of the sorted radii (the upper one for even M: ALGLIB's
// tmp[n/2] as recalled) -- then R_i = min(R_i, z * median)
__global__ void k_qnn_median(const BatchSlot *tab, int M)
{
    const double *radii = tab[blockIdx.z].radii;
    double *median_out = tab[blockIdx.z].W;      // scratch: W is rewritten by the pack kernel
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const double r = radii[i];
    int rank = 0;
    for (int j = 0; j < M; ++j) {
        const double o = radii[j];
        rank += (o < r) || (o == r && j < i);
    }
    if (rank == M / 2) *median_out = r;
}
__global__ void k_qnn_cap(const BatchSlot *tab, int M, double z)
{
    double *radii = tab[blockIdx.z].radii;
    const double *median = tab[blockIdx.z].W;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const double cap = z * (*median);
    if (radii[i] > cap) radii[i] = cap;
}

// The three steps above in ONE launch of one workgroup per model, up to 1024 centres: centres and radii in LDS (the separate
// kernels walk global memory M times per thread: 37 + 22 + 5 us at M = 256, a sixth of the QNN build).  Same arithmetic.
__global__ __launch_bounds__(1024) void k_qnn_radii_small(const BatchSlot *tab, int M, double q, double z)
{
    const double *centres = tab[blockIdx.z].centres;
    double *radii = tab[blockIdx.z].radii;
    __shared__ double s_c[3 * 1024];
    __shared__ double s_r[1024];
    __shared__ double s_median;
    const int i = threadIdx.x;
    for (int e = i; e < 3 * M; e += blockDim.x) s_c[e] = centres[e];
    __syncthreads();
    double r = 0.0;
    if (i < M) {
        const double x = s_c[3 * i], y = s_c[3 * i + 1], zz = s_c[3 * i + 2];
        double best = INFINITY;
        for (int j = 0; j < M; ++j) {
            if (j == i) continue;
            const double dx = x - s_c[3 * j], dy = y - s_c[3 * j + 1], dz = zz - s_c[3 * j + 2];
            const double d2 = dx * dx + dy * dy + dz * dz;
            best = d2 < best ? d2 : best;
        }
        r = M > 1 ? q * sqrt(best) : q;
        s_r[i] = r;
    }
    __syncthreads();
    if (i < M) {
        int rank = 0;
        for (int j = 0; j < M; ++j) {
            const double o = s_r[j];
            rank += (o < r) || (o == r && j < i);
        }
        if (rank == M / 2) s_median = r;
    }
    __syncthreads();
    if (i < M) {
        const double cap = z * s_median;
        radii[i] = r > cap ? cap : r;
    }
}

// ---- assembly -----------------------------------------------------------------
// A workgroup fills a (16 TS) x (16 TS) tile, 16 x 16 elements at a time (consecutive threads
// walk a column: coalesced), and contributes ONE atomicMax to max|A|: with a 16 x 16 tile per
// workgroup the 67 600 same-address atomics of an order-2080 system took 0.7 ms by themselves.
// sym: the block is symmetric bit for bit (a kernel without per-column radii, no polynomial columns: (c_i - c_j)^2 and
// (c_j - c_i)^2 are the same doubles), so the tiles above the diagonal are not computed -- the tile across the diagonal
// stores them too.  Half the fp64 logarithms of a thin-plate build.
template <int TS>
__global__ __launch_bounds__(256) void k_assemble(const BatchSlot *tab, int M,
                                                   int n, int npad, int lda, int kind, int T,
                                                   double lambda, int radii_off, int sym)
{
    if (sym && blockIdx.x > blockIdx.y) return;
    const bool mirror = sym && blockIdx.x < blockIdx.y;
    const double *centres = tab[blockIdx.z].centres, *radii = tab[blockIdx.z].radii + radii_off;
    double *A = tab[blockIdx.z].A;
    DevModel *model = tab[blockIdx.z].model;
    __shared__ double s_max[4];
    __shared__ double s_t[16][17];       // a 16 x 16 piece on its way to the other side of the diagonal
    double m = 0.0;                      // max |A_ij| over the real system, for the singularity threshold
    bool dup = false;
#pragma unroll
    for (int tj = 0; tj < TS; ++tj) {
        const int j = (blockIdx.y * TS + tj) * 16 + (threadIdx.x >> 4);
        double cjx = 0.0, cjy = 0.0, cjz = 0.0, inv_r2 = 1.0;
        if (j < M) {
            cjx = centres[3 * j]; cjy = centres[3 * j + 1]; cjz = centres[3 * j + 2];
            const double r = radii[j];
            inv_r2 = 1.0 / (r * r);
        }
#pragma unroll
        for (int ti = 0; ti < TS; ++ti) {
            const int i = (blockIdx.x * TS + ti) * 16 + (threadIdx.x & 15);
            double v = 0.0;
            if (i < npad && j < npad) {
                bool real = false;
                if (i < M && j < M) {
                    const double dx = centres[3 * i] - cjx;
                    const double dy = centres[3 * i + 1] - cjy;
                    const double dz = centres[3 * i + 2] - cjz;
                    const double d2 = dx * dx + dy * dy + dz * dz;
                    v = phi_d(kind, d2, inv_r2);
                    if (i == j) v += lambda;
                    else if (d2 == 0.0) dup = true;     // coincident centres -> -5
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
                if (real) m = fmax(m, fabs(v));
            }
            if (mirror) {
                // the piece across the diagonal, through LDS so that consecutive threads again write consecutive addresses
                // (straight from the registers it was sixteen 32-byte pieces per wave: slower than computing the half anew)
                __syncthreads();
                s_t[threadIdx.x >> 4][threadIdx.x & 15] = v;
                __syncthreads();
                const int ii = (blockIdx.x * TS + ti) * 16 + (threadIdx.x >> 4), jj = (blockIdx.y * TS + tj) * 16 + (threadIdx.x & 15);
                if (ii < npad && jj < npad) A[(size_t)ii * lda + jj] = s_t[threadIdx.x & 15][threadIdx.x >> 4];
            }
        }
    }
    if (dup) model->dup_flag = 1;
    for (int off = 32; off >= 1; off >>= 1) {
        const double o = __shfl_xor(m, off);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmax(fmax(s_max[0], s_max[1]), fmax(s_max[2], s_max[3]));
        if (m > 0.0) atomicMax(&model->amax_bits, (unsigned long long)__double_as_longlong(m));
    }
}

// ---- LU panel -------------------------------------------------------------------
// wave-wide maximum of a 32-bit key without touching LDS: rotate-reduce inside each
// 16-lane row on the DPP network, then combine the four rows through SGPRs
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
#define FD_DPP_MAX(CTRL)                                                                         \
    {                                                                                            \
        const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false); \
        v = o > v ? o : v;                                                                       \
    }
    FD_DPP_MAX(0x121)   // row_ror:1
    FD_DPP_MAX(0x122)   // row_ror:2
    FD_DPP_MAX(0x124)   // row_ror:4
    FD_DPP_MAX(0x128)   // row_ror:8
#undef FD_DPP_MAX
    const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)v, 0);
    const unsigned r1 = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned r2 = (unsigned)__builtin_amdgcn_readlane((int)v, 32);
    const unsigned r3 = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    const unsigned m01 = r0 > r1 ? r0 : r1, m23 = r2 > r3 ? r2 : r3;
    return m01 > m23 ? m01 : m23;
}

// 1/x to within an ulp or two: hardware estimate + two Newton steps (the IEEE divide
// sequence is ~3x longer and sits on the per-column critical path of the panel)
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

__device__ __forceinline__ double readlane_f64(double v, int src_lane)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), src_lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), src_lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Diagnostic build only (tools/panel_bench.hip defines FD_PANEL_STAMPS): shader-clock shares of
// the phases of one column step, summed over the panel by thread 0.  Never in the product.
#ifdef FD_PANEL_B64
#define FD_WRITE_FENCE asm volatile("" ::: "memory");
#else
#define FD_WRITE_FENCE
#endif
#ifdef FD_PANEL_STAMPS
__device__ unsigned long long g_panel_stamps[16];
#define FD_STAMP_DECL unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_prev = __builtin_amdgcn_s_memtime();
#define FD_STAMP(K) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[K] += t_ - st_prev; st_prev = t_; __builtin_amdgcn_sched_barrier(0); }
#define FD_STAMP_FLUSH if (threadIdx.x == 0) { for (int q_ = 0; q_ < 10; ++q_) g_panel_stamps[q_] = st_acc[q_]; }
#else
#define FD_STAMP_DECL
#define FD_STAMP(K)
#define FD_STAMP_FLUSH
#endif

// Pivot rule: the row with the largest |a| as seen through an fp32 key (so rows whose
// magnitudes agree to 2^-24 tie), ties to the smaller row index across waves and to the
// lower lane inside a wave.  Any such pivot is as good as the exact maximum for partial
// pivoting; the choice is deterministic.
//
// The column loop is fully unrolled so that every register index is static and each step
// touches only the live columns j..NB-1 (a rolled loop with a rotating register row was
// measured slower: it has to mask and move all NB columns every step).
template <int NB, int TMAX>
__global__ __launch_bounds__(TMAX) void k_lu_panel(const BatchSlot *tab, int lda, int npad, int n_real,
                                                   int k0, int step)
{
    double *A = tab[blockIdx.z].A;
    // one move list per elimination step: with look-ahead the next panel writes its list while
    // the rest of this step's trailing update still reads the current one
    int *ipiv = tab[blockIdx.z].ipiv, *moves = tab[blockIdx.z].moves + (size_t)step * kMovesStride;
    DevModel *model = tab[blockIdx.z].model;
    constexpr int R = 32 / NB;          // rows per lane: R * NB = 32 doubles in registers
    // live part of the pivot row of the current column, one slot per wave's candidate
    __shared__ __attribute__((aligned(16))) double s_row[2][16][NB];
    __shared__ unsigned long long s_key[3];
    __shared__ int s_count;
    (void)ipiv;

    // latency chain on one CU: when it shares SIMDs with another stream's evaluation waves,
    // let it issue first
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nthreads = blockDim.x;
    const int nrem = npad - k0;

    // Rows never travel between threads.  pos[t] is the row's current LOGICAL position in the
    // panel: an interchange swaps two integers, and the write-back scatters each row to
    // k0 + pos[t].  (Moving 32-double rows through LDS costs 450-850 cycles per row: one
    // active lane per 16-byte ds_write; measured in tools/lds_write_bench.hip.)
    double a[R][NB];
    int pos[R];
#pragma clang loop unroll(full)
    for (int t = 0; t < R; ++t) {
        const int slot = tid + t * nthreads;
        pos[t] = slot;
#pragma clang loop unroll(full)
        for (int c = 0; c < NB; ++c)
            a[t][c] = slot < nrem ? A[(size_t)(k0 + c) * lda + k0 + slot] : 0.0;
    }
    if (tid == 0) { s_count = 0; s_key[0] = 0ull; s_key[1] = 0ull; s_key[2] = 0ull; }
    __syncthreads();

    const double amax = __longlong_as_double((long long)model->amax_bits);
    const double tiny = (double)n_real * kEps * amax;
    double pmin = INFINITY, pmax = 0.0;
    bool singular = false;
    FD_STAMP_DECL

#pragma clang loop unroll(full)
    for (int j = 0; j < NB; ++j) {
        const int buf = j & 1;
        // candidate among my rows at or below the diagonal
        double best = -1.0;
        int bpos = 0x7fffffff;
        int bt = 0;
#pragma clang loop unroll(full)
        for (int t = 0; t < R; ++t) {
            if (pos[t] >= j && tid + t * nthreads < nrem) {
                const double v = fabs(a[t][j]);
                if (v > best) { best = v; bpos = pos[t]; bt = t; }
            }
        }
        // fp32 key, +1 so that a real candidate of magnitude 0 still beats "no candidate"
        const bool has = best >= 0.0;
        const unsigned key = has ? __float_as_uint((float)best) + 1u : 0u;
        FD_STAMP(0)
        const unsigned wmax = wave_max_u32(key);
        const unsigned long long cand = __ballot(has && key == wmax);
        FD_STAMP(1)
        if (cand != 0ull) {
            const int src = __ffsll((unsigned long long)cand) - 1;
            const int wpos = __builtin_amdgcn_readlane(bpos, src);
            if (lane == src) {
#pragma clang loop unroll(full)
                for (int t = 0; t < R; ++t) {
                    if (t == bt) {
#pragma clang loop unroll(full)
                        for (int c = j & ~1; c < NB; ++c) s_row[buf][wave][c] = a[t][c];
                    }
                }
            }
            // low word: the smaller logical row wins a key tie; the owning wave rides along
            if (lane == 0)
                atomicMax(&s_key[j % 3], ((unsigned long long)wmax << 32) |
                                             (unsigned)(((0xffff - wpos) << 8) | wave));
        }
        // the next cell was last read two columns ago, i.e. before a barrier everyone has passed
        if (tid == 0) s_key[(j + 1) % 3] = 0ull;
        FD_STAMP(2)
        __syncthreads();
        FD_STAMP(3)

        const unsigned long long gkey = s_key[j % 3];
        // no key at all: every candidate of this column was NaN (a matrix poisoned upstream, e.g.
        // coincident centres under the QNN radius rule).  Keep the diagonal row where it is -- the
        // zero key would decode to logical position 65535 and the write-back would leave the matrix
        const bool none = gkey == 0ull;
        const unsigned glow = (unsigned)(gkey & 0xffffffffull);
        const int gpos = none ? j : 0xffff - (int)(glow >> 8);
        const int gw = none ? 0 : (int)(glow & 0xffu);
        // the interchange: the pivot row takes logical position j, the row that was there takes gpos
#pragma clang loop unroll(full)
        for (int t = 0; t < R; ++t) {
            const int p = pos[t];
            pos[t] = p == gpos ? j : (p == j ? gpos : p);
        }
        FD_STAMP(4)
        // the live part of the pivot row into registers in one burst of LDS reads -- read one
        // element at a time next to its fma, hipcc waits for every read separately
        double prow[NB];
#pragma clang loop unroll(full)
        for (int c = j & ~1; c < NB; ++c) prow[c] = s_row[buf][gw][c];
        asm volatile("" ::: "memory");
        const double piv = prow[j];
        const double gbest = fabs(piv);
        const bool ok = !none && gbest > tiny;   // false for NaN as well
        if (k0 + j < n_real) {
            if (!ok) singular = true;
            pmin = gbest < pmin ? gbest : pmin;
            pmax = gbest > pmax ? gbest : pmax;
        }
        const double inv = ok ? fast_rcp(piv) : 0.0;
        FD_STAMP(5)
#pragma clang loop unroll(full)
        for (int t = 0; t < R; ++t) {
            if (pos[t] > j && tid + t * nthreads < nrem) {
                const double l = a[t][j] * inv;
                a[t][j] = l;
#pragma clang loop unroll(full)
                for (int c = j + 1; c < NB; ++c) a[t][c] = fma(-l, prow[c], a[t][c]);
            }
        }
        FD_STAMP(6)
        FD_STAMP(7)
    }
    FD_STAMP(8)

    // write the factored panel back, every row at its logical position, and list the rows
    // that moved for the trailing update
#pragma clang loop unroll(full)
    for (int t = 0; t < R; ++t) {
        const int slot = tid + t * nthreads;
        if (slot < nrem) {
#pragma clang loop unroll(full)
            for (int c = 0; c < NB; ++c) A[(size_t)(k0 + c) * lda + k0 + pos[t]] = a[t][c];
            if (pos[t] != slot) {
                const int q = atomicAdd(&s_count, 1);
                moves[1 + 2 * q] = k0 + pos[t];   // destination row
                moves[2 + 2 * q] = k0 + slot;     // row (as of panel start) whose content lands there
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
    FD_STAMP(9)
    FD_STAMP_FLUSH
}

// ---- the same panel WITHOUT pivot search (QNN model at the SOP's defaults) ---------------------------------
// With R_j = q * (distance to the nearest neighbour), q <= 1, every off-diagonal entry of column j of the QNN kernel block is
// at most e^-1 of its diagonal one and the elimination keeps it so: partial pivoting never interchanges, on any rig measured
// (profiles/r02_qnn_pivot_stats.txt) -- and then the 32 barriers, wave maxima and LDS atomics of the search buy nothing.
// Here wave 0 factorises the 32 x 32 diagonal block in registers (row j comes from lane j by v_readlane; its lanes 32..63 --
// rows below the block -- ride along), puts U and 1 / u_jj in LDS, and after ONE barrier every other row runs down its own
// forward substitution.  Operation for operation the arithmetic is k_lu_panel's with the diagonal row as the pivot: the factors
// are bit-identical to the pivoted ones whenever that kernel would not interchange.  The largest |multiplier| is recorded; above
// kMaxMultiplier (or with a pivot at or below the threshold) the model is flagged, reports -4, and the host repeats the build
// with k_lu_panel (fd_capi.hip: prefer_lu).  One row per thread: order <= 1024.
constexpr double kMaxMultiplier = 4.0;
template <int TMAX>
__global__ __launch_bounds__(TMAX) void k_lu_panel_np(const BatchSlot *tab, int lda, int npad, int n_real, int k0, int step)
{
    constexpr int NB = 32;
    double *A = tab[blockIdx.z].A;
    int *moves = tab[blockIdx.z].moves + (size_t)step * kMovesStride;
    DevModel *model = tab[blockIdx.z].model;
    __shared__ __attribute__((aligned(16))) double s_U[NB][NB];
    __shared__ double s_inv[NB];
    __shared__ double s_mult[TMAX / 64];
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nrem = npad - k0;
    const bool in = tid < nrem;
    double a[NB];
#pragma clang loop unroll(full)
    for (int c = 0; c < NB; ++c) a[c] = in ? A[(size_t)(k0 + c) * lda + k0 + tid] : 0.0;
    const double amax = __longlong_as_double((long long)model->amax_bits);
    const double tiny = (double)n_real * kEps * amax;
    double mmax = 0.0;
    if (wave == 0) {
        double pmin = INFINITY, pmax = 0.0;
        bool singular = false;
#pragma clang loop unroll(full)
        for (int j = 0; j < NB; ++j) {
            double prow[NB];
#pragma clang loop unroll(full)
            for (int c = j; c < NB; ++c) prow[c] = readlane_f64(a[c], j);
            const double piv = prow[j];
            const double gbest = fabs(piv);
            const bool ok = gbest > tiny;            // false for NaN as well
            if (k0 + j < n_real) {
                if (!ok) singular = true;
                pmin = gbest < pmin ? gbest : pmin;
                pmax = gbest > pmax ? gbest : pmax;
            }
            const double inv = ok ? fast_rcp(piv) : 0.0;
            if (lane == j) s_inv[j] = inv;
            if (lane > j && in) {
                const double l = a[j] * inv;
                a[j] = l;
                mmax = fabs(l) > mmax ? fabs(l) : mmax;
#pragma clang loop unroll(full)
                for (int c = j + 1; c < NB; ++c) a[c] = fma(-l, prow[c], a[c]);
            }
        }
        if (lane < NB) {
#pragma clang loop unroll(full)
            for (int c = 0; c < NB; ++c) s_U[lane][c] = a[c];
        }
        if (lane == 0) {
            const int done = (k0 + NB < n_real ? k0 + NB : n_real);
            model->iterations = done;
            if (singular) model->sing_flag = 1;
            if (pmax > 0.0 || pmin < INFINITY) {
                atomicMin(&model->pivmin_bits, (unsigned long long)__double_as_longlong(pmin));
                atomicMax(&model->pivmax_bits, (unsigned long long)__double_as_longlong(pmax));
            }
            moves[0] = 0;
        }
    }
    __syncthreads();
    if (wave != 0 && in) {
#pragma clang loop unroll(full)
        for (int j = 0; j < NB; ++j) {
            const double l = a[j] * s_inv[j];
            a[j] = l;
            mmax = fabs(l) > mmax ? fabs(l) : mmax;
#pragma clang loop unroll(full)
            for (int c = j + 1; c < NB; ++c) a[c] = fma(-l, s_U[j][c], a[c]);
        }
    }
    if (in) {
#pragma clang loop unroll(full)
        for (int c = 0; c < NB; ++c) A[(size_t)(k0 + c) * lda + k0 + tid] = a[c];
    }
    for (int off = 32; off >= 1; off >>= 1) { const double o = __shfl_xor(mmax, off); mmax = o > mmax ? o : mmax; }
    if (lane == 0) s_mult[wave] = mmax;
    __syncthreads();
    if (tid == 0) {
        double m = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) m = s_mult[w] > m ? s_mult[w] : m;
        if (m > kMaxMultiplier) model->sing_flag = 2;      // partial pivoting would have interchanged: the pivoted LU has to do this one
    }
}

// Row interchanges of elimination step `step` applied to the 16 columns starting at c0 and
// nothing else: brings the L columns of the first panel of a pair into the row order the
// second panel chose (LAPACK's laswp on the left columns).
__global__ __launch_bounds__(256) void k_lu_apply_moves(const BatchSlot *tab, int lda, int step, int c0_first)
{
    const int c0 = c0_first + (int)blockIdx.x * kColBlock;
    double *A = tab[blockIdx.z].A;
    const int *moves = tab[blockIdx.z].moves + (size_t)step * kMovesStride;
    __shared__ int s_moves[kMovesStride];
    const int tid = threadIdx.x;
    if (tid < kMovesStride) s_moves[tid] = moves[tid];
    __syncthreads();
    const int nmov = s_moves[0];
    double tmp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + q * 256;
        tmp[q] = 0.0;
        if (e < nmov * kColBlock) tmp[q] = A[(size_t)(c0 + (e & 15)) * lda + s_moves[2 + 2 * (e >> 4)]];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + q * 256;
        if (e < nmov * kColBlock) A[(size_t)(c0 + (e & 15)) * lda + s_moves[1 + 2 * (e >> 4)]] = tmp[q];
    }
}

// ---- LU trailing update ---------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(256) void k_lu_trail(const BatchSlot *tab, int lda, int npad, int k0, int step, int cb0,
                                                  int nlists)
{
    double *A = tab[blockIdx.z].A;
    const int *moves = tab[blockIdx.z].moves + (size_t)step * kMovesStride;
    constexpr int S = NB / 4;           // k-steps of the f64 16x16x4 MFMA
    __shared__ double sL[NB][NB + 1];   // L11 (unit lower), +1 pad: column reads conflict-free
    __builtin_amdgcn_s_setprio(3);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int c0 = k0 + NB + (cb0 + (int)blockIdx.x) * kColBlock;
    const int c = lane & 15;            // column inside the block
    const int g = lane >> 4;            // row group (MFMA k index)

    // 1. row interchanges of this panel, restricted to my 16 columns: the move list comes
    //    in with one coalesced load; every moved row is read before any is written
    //    (sources and destinations overlap).  A 32-wide update over a PAIR of 16-wide panels
    //    (nlists = 2) applies the first panel's list, then the second's.
    __shared__ int s_moves[2 * 2 * NB + 1];
    for (int l = 0; l < nlists; ++l) {
        __syncthreads();
        if (tid < 2 * 2 * NB + 1) s_moves[tid] = moves[(size_t)l * kMovesStride + tid];
        __syncthreads();
        const int nmov = s_moves[0];
        double tmp[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + q * 256;
            tmp[q] = 0.0;
            if (e < nmov * kColBlock) tmp[q] = A[(size_t)(c0 + (e & 15)) * lda + s_moves[2 + 2 * (e >> 4)]];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + q * 256;
            if (e < nmov * kColBlock) A[(size_t)(c0 + (e & 15)) * lda + s_moves[1 + 2 * (e >> 4)]] = tmp[q];
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
    // two tiles per iteration: all loads of both tiles are issued before the first MFMA
    for (int tile = wave; tile < ntiles; tile += 8) {
        const int r0 = rbase + tile * 16;
        const bool two = tile + 4 < ntiles;
        const int r1 = two ? r0 + 64 : r0;
        // accumulator transposed (the update U12 as the MFMA's A operand, L21 as its B operand):
        // a lane then owns ONE row and four columns of the tile, and every load / store covers 16
        // consecutive rows of a column (128 B) instead of 4 rows of 16 columns
        const size_t cs = (size_t)4 * lda;
        double *cptr0 = A + (size_t)(c0 + g) * lda + r0 + c;
        double *cptr1 = A + (size_t)(c0 + g) * lda + r1 + c;
        const double *aptr0 = A + (size_t)(k0 + g) * lda + r0 + c;
        const double *aptr1 = A + (size_t)(k0 + g) * lda + r1 + c;
        double4_t acc0, acc1;
        acc0[0] = cptr0[0]; acc0[1] = cptr0[cs]; acc0[2] = cptr0[2 * cs]; acc0[3] = cptr0[3 * cs];
        acc1[0] = cptr1[0]; acc1[1] = cptr1[cs]; acc1[2] = cptr1[2 * cs]; acc1[3] = cptr1[3 * cs];
        const bool ok0 = r0 + c >= rfirst, ok1 = r1 + c >= rfirst;
        double av0[S], av1[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const double v0 = aptr0[(size_t)(4 * s) * lda];
            const double v1 = aptr1[(size_t)(4 * s) * lda];
            av0[s] = ok0 ? v0 : 0.0;
            av1[s] = ok1 ? v1 : 0.0;
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(u[s], av0[s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(u[s], av1[s], acc1, 0, 0, 0);
        }
        if (ok0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) cptr0[r * cs] = acc0[r];
        }
        if (two && ok1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) cptr1[r * cs] = acc1[r];
        }
    }
}

// ---- back substitution, one workgroup, bottom-up over 32-row blocks ----------------
// Solves U * X = Y (Y = columns npad.. of A after forward elimination) for the rows
// [row_lo, row_hi): Y of that range lives in LDS for the whole kernel; the diagonal block of the
// NEXT step is fetched while the current one is solved; every thread issues the loads of its U
// row segment before wave 0 starts the 32 x 32 triangle (rows in registers, x_k travels by
// v_readlane: no LDS round trip, no barrier inside), so global latency hides behind the solve.
// Small systems (npad <= 512) are one call over [0, npad).  Larger ones go top-down in 256-row
// ranges: this kernel on the range, then k_backsub_update pushes the solved block into the Y
// rows above it with as many workgroups as there are rows (one workgroup walking all 2080 rows
// per 32-column step took 2.4 ms at M = 2048).
template <int T>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_backsub_all(const BatchSlot *tab, int lda, int npad, int row_lo, int row_hi)
{
    double *A = tab[blockIdx.z].A;
    double *X = tab[blockIdx.z].X;
    extern __shared__ __attribute__((aligned(16))) double s_y[];   // [3][w]
    __shared__ double s_u[2][32][33];
    __shared__ double s_x[32][3];
    constexpr int E = 1024 / T;          // diagonal-block elements per thread
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = row_hi - row_lo;

    for (int e = tid; e < 3 * w; e += T) s_y[e] = A[(size_t)(npad + e / w) * lda + row_lo + e % w];
    double dnext[E];
#pragma unroll
    for (int q = 0; q < E; ++q) {
        const int e = tid + q * T;
        dnext[q] = A[(size_t)(row_hi - 32 + (e >> 5)) * lda + row_hi - 32 + (e & 31)];
    }
    int buf = 0;
    for (int b0 = row_hi - 32; b0 >= row_lo; b0 -= 32, buf ^= 1) {
#pragma unroll
        for (int q = 0; q < E; ++q) {
            const int e = tid + q * T;
            s_u[buf][e & 31][e >> 5] = dnext[q];
        }
        if (b0 - 32 >= row_lo) {
#pragma unroll
            for (int q = 0; q < E; ++q) {
                const int e = tid + q * T;
                dnext[q] = A[(size_t)(b0 - 32 + (e >> 5)) * lda + b0 - 32 + (e & 31)];
            }
        }
        // my first row's segment of U for the update below (does not depend on x); only the
        // small-system variant has the registers to hold it across the solve
        // (no prefetch of the U row across the triangle any more: together with the triangle's own
        // row it needs more than the 128 VGPRs of one evaluation-wave slot, and a kernel that does
        // waits for two of them to retire when it runs beside an evaluation -- fd_nullspace.hip)
        constexpr bool kPrefetch = false;
        const int i0 = row_lo + tid;
        double uik[32];
        if (kPrefetch && i0 < b0) {
#pragma unroll
            for (int k = 0; k < 32; ++k) uik[k] = A[(size_t)(b0 + k) * lda + i0];
        }
        __syncthreads();                 // diagonal block and the Y rows of this block are in LDS

        const int l0 = b0 - row_lo;      // this block's offset inside s_y
        if (tid < 64) {
            const int i = lane & 31;
            double ub[32];
#pragma unroll
            for (int cc = 0; cc < 32; ++cc) ub[cc] = s_u[buf][i][cc];
            double y0 = s_y[l0 + i], y1 = s_y[w + l0 + i], y2 = s_y[2 * w + l0 + i];
            const double dinv = 1.0 / s_u[buf][i][i];
#pragma unroll
            for (int k = 31; k >= 0; --k) {
                const double inv = readlane_f64(dinv, k);
                const double x0 = readlane_f64(y0, k) * inv;
                const double x1 = readlane_f64(y1, k) * inv;
                const double x2 = readlane_f64(y2, k) * inv;
                if (i < k) {
                    y0 = fma(-ub[k], x0, y0);
                    y1 = fma(-ub[k], x1, y1);
                    y2 = fma(-ub[k], x2, y2);
                } else if (i == k) {
                    y0 = x0; y1 = x1; y2 = x2;
                }
            }
            if (lane < 32) {
                s_x[i][0] = y0; s_x[i][1] = y1; s_x[i][2] = y2;
                s_y[l0 + i] = y0; s_y[w + l0 + i] = y1; s_y[2 * w + l0 + i] = y2;
            }
        }
        __syncthreads();
        for (int i = i0; i < b0; i += T) {
            if (!kPrefetch || i != i0) {
#pragma unroll
                for (int k = 0; k < 32; ++k) uik[k] = A[(size_t)(b0 + k) * lda + i];
            }
            double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                a0 = fma(uik[k], s_x[k][0], a0);
                a1 = fma(uik[k], s_x[k][1], a1);
                a2 = fma(uik[k], s_x[k][2], a2);
                if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // or all 96 LDS reads are hoisted and the U row spills
            }
            const int li = i - row_lo;
            s_y[li] -= a0; s_y[w + li] -= a1; s_y[2 * w + li] -= a2;
        }
        // no barrier here: the next iteration's barrier orders these writes before its solve,
        // and s_x / s_u[buf] are not rewritten before that barrier either (s_u alternates)
    }
    __syncthreads();
    for (int e = tid; e < 3 * w; e += T) X[(size_t)(e / w) * npad + row_lo + e % w] = s_y[e];
}

// Y[i] -= U[i, row_lo .. row_lo + w) * X[row_lo .. row_lo + w) for every row i < row_lo: one row
// per thread (consecutive threads walk a column of U: coalesced), the solved block in LDS.
__global__ __launch_bounds__(256) void k_backsub_update(const BatchSlot *tab, int lda, int npad, int row_lo, int w)
{
    double *A = tab[blockIdx.z].A;
    const double *X = tab[blockIdx.z].X;
    __shared__ double s_x[3][256];
    const int tid = threadIdx.x;
    for (int e = tid; e < 3 * w; e += 256) s_x[e / w][e % w] = X[(size_t)(e / w) * npad + row_lo + e % w];
    __syncthreads();
    const int i = blockIdx.x * 256 + tid;
    if (i >= row_lo) return;
    const double *u = A + (size_t)row_lo * lda + i;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int k0 = 0; k0 < w; k0 += 16) {
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = u[(size_t)(k0 + k) * lda];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            a0 = fma(v[k], s_x[0][k0 + k], a0);
            a1 = fma(v[k], s_x[1][k0 + k], a1);
            a2 = fma(v[k], s_x[2][k0 + k], a2);
        }
    }
    A[(size_t)npad * lda + i] -= a0;
    A[(size_t)(npad + 1) * lda + i] -= a1;
    A[(size_t)(npad + 2) * lda + i] -= a2;
}

// ---- pack: solution -> weights, evaluation records, status (bodies in fd_pack.h) ------------------
// one 128-VGPR slot: the kernel runs beside the evaluation of earlier frames (fd_nullspace.hip, FD_FIT_BESIDE_EVAL)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_pack(const BatchSlot *tab, int npad, int M, int Mpad, int T, int kind,
                                              int from_w, int layers)
{
    packing::pack_body(tab[blockIdx.z], npad, M, Mpad, T, kind, from_w, layers);
}

__global__ __launch_bounds__(64) void k_pack_tiles(const BatchSlot *tab, int Mpad)
{
    packing::pack_tiles_body(tab[blockIdx.z], Mpad, blockIdx.x, threadIdx.x);
}

// NB of the panel that starts at column k0 (the register budget of one workgroup decides)
int panel_width(int npad, int k0)
{
    const int nrem = npad - k0;
    return nrem <= 1024 ? 32 : (nrem <= 2048 ? 16 : (nrem <= 4096 ? 8 : 4));
}

struct LuStreams {
    hipStream_t main, aux;      // aux == nullptr: no look-ahead, everything on main
    hipEvent_t ev_panel[2], ev_rest[2];
    bool rest_pending = false;  // a part-B update has been enqueued on aux and not yet waited for
    int last_rest = 0;
};

// One elimination step.  Without look-ahead (the default, see make_lookahead in fd_capi.hip):
// panel, then the whole trailing update, on one stream.  With it the update is split: part A = the column blocks the NEXT panel will read,
// on the main stream right after the panel; part B = everything to the right of them, on the
// aux stream.  The next panel therefore overlaps part B, and the critical path per step is
// max(panel, part B) + part A instead of panel + full update.  Part A of step k+1 lies inside
// part B of step k, hence the wait on ev_rest before it.
template <int NB>
void lu_step(const BuildBuffers &b, int k0, int step, LuStreams &st)
{
    constexpr int R = 32 / NB;
    const int nrem = b.npad - k0;
    const unsigned nb = (unsigned)b.nbatch;
    int threads = round_up((nrem + R - 1) / R, 64);
    if (threads < 64) threads = 64;
    if (b.nopivot && NB == 32 && threads <= 512)
        hipLaunchKernelGGL((k_lu_panel_np<512>), dim3(1, 1, nb), dim3(threads), 0, st.main, b.d_slots, b.lda, b.npad, b.n, k0, step);
    else if (b.nopivot && NB == 32)
        hipLaunchKernelGGL((k_lu_panel_np<1024>), dim3(1, 1, nb), dim3(threads), 0, st.main, b.d_slots, b.lda, b.npad, b.n, k0, step);
    else if (threads <= 512)
        hipLaunchKernelGGL((k_lu_panel<NB, 512>), dim3(1, 1, nb), dim3(threads), 0, st.main, b.d_slots, b.lda,
                           b.npad, b.n, k0, step);
    else
        hipLaunchKernelGGL((k_lu_panel<NB, 1024>), dim3(1, 1, nb), dim3(threads), 0, st.main, b.d_slots, b.lda,
                           b.npad, b.n, k0, step);
    // the last block may run into the 16 zero columns allocated past ncols
    const int ncb = (b.ncols - (k0 + NB) + kColBlock - 1) / kColBlock;
    if (ncb <= 0) return;
    if (!st.aux) {
        hipLaunchKernelGGL((k_lu_trail<NB>), dim3(ncb, 1, nb), dim3(256), 0, st.main, b.d_slots, b.lda, b.npad, k0,
                           step, 0, 1);
        return;
    }
    const int next_k0 = k0 + NB;
    int na = next_k0 < b.npad ? (panel_width(b.npad, next_k0) + kColBlock - 1) / kColBlock : 0;
    if (na > ncb) na = ncb;
    const int par = step & 1;
    if (ncb > na) {
        (void)hipEventRecord(st.ev_panel[par], st.main);
        (void)hipStreamWaitEvent(st.aux, st.ev_panel[par], 0);
    }
    if (na > 0) {
        if (st.rest_pending) { (void)hipStreamWaitEvent(st.main, st.ev_rest[st.last_rest], 0); st.rest_pending = false; }
        hipLaunchKernelGGL((k_lu_trail<NB>), dim3(na, 1, nb), dim3(256), 0, st.main, b.d_slots, b.lda, b.npad, k0,
                           step, 0, 1);
    }
    if (ncb > na) {
        hipLaunchKernelGGL((k_lu_trail<NB>), dim3(ncb - na, 1, nb), dim3(256), 0, st.aux, b.d_slots, b.lda, b.npad, k0,
                           step, na, 1);
        (void)hipEventRecord(st.ev_rest[par], st.aux);
        st.rest_pending = true;
        st.last_rest = par;
    }
}

// Consecutive panels can share ONE deep trailing update: np panels of width w, then a single
// update of depth np * w over everything to their right.  That divides the passes over the
// trailing matrix by np where they are bound by HBM traffic (37 MB read + written per step at
// order 2080).  Inside the group every panel is followed by a narrow update of the group's own
// remaining columns, and its row interchanges are applied to the L columns of the panels before
// it (LAPACK's laswp on the left), so that the group ends up as one proper np*w-wide LU panel.
// Returns the number of panels to group at k0 (1 = ordinary step).
int group_at(int npad, int k0, bool enabled)
{
    if (!enabled) return 1;
    const int w = panel_width(npad, k0);
    const int nrem = npad - k0;
    int want = 1;
    if (w == 16) want = 4;                       // 1024 < rows <= 2048: depth 64
    else if (w == 32 && nrem > 512) want = 2;    // large 32-wide panels: depth 64; small systems stay launch-lean
    int np = 1;
    while (np < want && k0 + np * w < npad && panel_width(npad, k0 + np * w) == w) ++np;
    if (np == 3) np = 2;                         // depths 32 and 64 are instantiated
    return np;
}

template <int W>
void launch_panel(const BuildBuffers &b, int kk, int step, hipStream_t stream)
{
    constexpr int R = 32 / W;
    const unsigned nb = (unsigned)b.nbatch;
    const int nrem = b.npad - kk;
    int threads = round_up((nrem + R - 1) / R, 64);
    if (threads < 64) threads = 64;
    if (b.nopivot && W == 32 && threads <= 512)
        hipLaunchKernelGGL((k_lu_panel_np<512>), dim3(1, 1, nb), dim3(threads), 0, stream, b.d_slots, b.lda, b.npad, b.n, kk, step);
    else if (b.nopivot && W == 32)
        hipLaunchKernelGGL((k_lu_panel_np<1024>), dim3(1, 1, nb), dim3(threads), 0, stream, b.d_slots, b.lda, b.npad, b.n, kk, step);
    else if (threads <= 512)
        hipLaunchKernelGGL((k_lu_panel<W, 512>), dim3(1, 1, nb), dim3(threads), 0, stream, b.d_slots, b.lda, b.npad, b.n, kk, step);
    else
        hipLaunchKernelGGL((k_lu_panel<W, 1024>), dim3(1, 1, nb), dim3(threads), 0, stream, b.d_slots, b.lda, b.npad, b.n, kk, step);
}

template <int W>
void lu_group_step(const BuildBuffers &b, int k0, int step, int np, hipStream_t stream)
{
    const unsigned nb = (unsigned)b.nbatch;
    constexpr int WB = W / kColBlock;                 // 16-column blocks per panel
    for (int i = 0; i < np; ++i) {
        const int kk = k0 + W * i;
        launch_panel<W>(b, kk, step + i, stream);
        if (i > 0)                                    // my interchanges on the L columns of panels 0 .. i-1
            hipLaunchKernelGGL(k_lu_apply_moves, dim3(i * WB, 1, nb), dim3(256), 0, stream, b.d_slots, b.lda, step + i, k0);
        if (i + 1 < np)                               // my update of the group's remaining columns
            hipLaunchKernelGGL((k_lu_trail<W>), dim3((np - 1 - i) * WB, 1, nb), dim3(256), 0, stream, b.d_slots, b.lda, b.npad,
                               kk, step + i, 0, 1);
    }
    const int depth = W * np;
    const int ncb = (b.ncols - (k0 + depth) + kColBlock - 1) / kColBlock;
    if (ncb <= 0) return;
    if (depth == 32)
        hipLaunchKernelGGL((k_lu_trail<32>), dim3(ncb, 1, nb), dim3(256), 0, stream, b.d_slots, b.lda, b.npad, k0, step, 0, np);
    else
        hipLaunchKernelGGL((k_lu_trail<64>), dim3(ncb, 1, nb), dim3(256), 0, stream, b.d_slots, b.lda, b.npad, k0, step, 0, np);
}

}  // namespace

static void launch_backsub(const BuildBuffers &b, hipStream_t stream, int rows)
{
    const unsigned nb = (unsigned)b.nbatch;
    if (rows <= 512) {
        const size_t ybytes = sizeof(double) * 3 * (size_t)rows;
        hipLaunchKernelGGL((k_backsub_all<256>), dim3(1, 1, nb), dim3(256), ybytes, stream, b.d_slots, b.lda,
                           b.npad, 0, rows);
    } else {
        constexpr int W = 256;      // rows per diagonal range (a multiple of 32, like npad)
        for (int hi = rows; hi > 0; hi -= W) {
            const int lo = hi > W ? hi - W : 0;
            hipLaunchKernelGGL((k_backsub_all<256>), dim3(1, 1, nb), dim3(256), sizeof(double) * 3 * (size_t)(hi - lo),
                               stream, b.d_slots, b.lda, b.npad, lo, hi);
            if (lo > 0)
                hipLaunchKernelGGL(k_backsub_update, dim3((lo + 255) / 256, 1, nb), dim3(256), 0, stream,
                                   b.d_slots, b.lda, b.npad, lo, hi - lo);
        }
    }
}

// push the solved rows [row_lo, row_lo + w) into the right-hand sides of the rows above them
hipError_t launch_backsub_update(const BuildBuffers &b, hipStream_t stream, int row_lo, int w)
{
    hipLaunchKernelGGL(k_backsub_update, dim3((row_lo + 255) / 256, 1, (unsigned)b.nbatch), dim3(256), 0, stream, b.d_slots,
                       b.lda, b.npad, row_lo, w);
    return hipGetLastError();
}

hipError_t launch_backsub_rows(const BuildBuffers &b, hipStream_t stream, int rows)
{
    launch_backsub(b, stream, rows);
    return hipGetLastError();
}

// the kernel block alone: phi + lambda on the diagonal for i, j < M, identity padding up to npad_a
hipError_t launch_assemble_block(const BuildBuffers &b, hipStream_t stream, int npad_a, int radii_off)
{
    const unsigned nb = (unsigned)b.nbatch;
    // kernels without per-column radii give a block that is symmetric bit for bit: half of it is computed
    static const bool no_sym = tuning_env("FD_ASSEMBLE_FULL") != nullptr;
    const int sym = (!no_sym && b.kind != FD_KERNEL_GAUSSIAN && b.kind != FD_KERNEL_GAUSSIAN_QNN && b.kind != FD_KERNEL_GAUSSIAN_ML) ? 1 : 0;
    if (npad_a <= 512) {
        const unsigned g = (unsigned)(npad_a + 31) / 32;
        hipLaunchKernelGGL((k_assemble<2>), dim3(g, g, nb), dim3(256), 0, stream, b.d_slots, b.M, b.M, npad_a,
                           b.lda, b.kind, 0, b.lambda, radii_off, sym);
    } else {
        const unsigned g = (unsigned)(npad_a + 63) / 64;
        hipLaunchKernelGGL((k_assemble<4>), dim3(g, g, nb), dim3(256), 0, stream, b.d_slots, b.M, b.M, npad_a,
                           b.lda, b.kind, 0, b.lambda, radii_off, sym);
    }
    return hipGetLastError();
}

hipError_t launch_prepare_rhs(const BuildBuffers &b, hipStream_t stream, const PointSrc *src)
{
    static const PointSrc none{};
    const int threads = 256;
    const int blocks = (b.npad + threads - 1) / threads;
    hipLaunchKernelGGL(k_prepare_rhs, dim3(blocks, 1, (unsigned)b.nbatch), dim3(threads), 0, stream, b.d_slots,
                       src ? *src : none, src ? 1 : 0, b.M, b.npad, b.lda);
    return hipGetLastError();
}

// centres, right-hand sides, status reset.  src == nullptr: read the contexts' own copies of
// the control points; otherwise straight from the caller's arrays (one pointer pair per model).
hipError_t launch_prepare(const BuildBuffers &b, hipStream_t stream, const PointSrc *src)
{
    static const PointSrc none{};
    const int threads = 256;
    const int blocks = (b.npad + threads - 1) / threads;
    hipLaunchKernelGGL(k_prepare, dim3(blocks, 1, b.nbatch), dim3(threads), 0, stream, b.d_slots,
                       src ? *src : none, src ? 1 : 0, b.M, b.npad, b.lda, b.ncols, b.gauss_R);
    return hipGetLastError();
}

// QNN radii of every model of the batch: q * nearest neighbour, capped at z * median
hipError_t launch_qnn_radii(const BuildBuffers &b, hipStream_t stream)
{
    const unsigned nb = (unsigned)b.nbatch;
    if (b.M <= 1024) {
        hipLaunchKernelGGL(k_qnn_radii_small, dim3(1, 1, nb), dim3(round_up(b.M, 64)), 0, stream, b.d_slots, b.M, b.qnn_q, b.qnn_z);
        return hipGetLastError();
    }
    const int threads = 256, mb = (b.M + threads - 1) / threads;
    hipLaunchKernelGGL(k_qnn_nearest, dim3(mb, 1, nb), dim3(threads), 0, stream, b.d_slots, b.M, b.qnn_q);
    hipLaunchKernelGGL(k_qnn_median, dim3(mb, 1, nb), dim3(threads), 0, stream, b.d_slots, b.M);
    hipLaunchKernelGGL(k_qnn_cap, dim3(mb, 1, nb), dim3(threads), 0, stream, b.d_slots, b.M, b.qnn_z);
    return hipGetLastError();
}

// the pivoted LU of the assembled order-npad system (right-hand sides in the columns npad ..) and
// its back-substitution into X; b.n real unknowns, the rest identity padding
hipError_t launch_lu_factor_solve(const BuildBuffers &b, hipStream_t stream)
{
    LuStreams st{};
    st.main = stream;
    st.aux = b.aux_stream;
    for (int q = 0; q < 2; ++q) { st.ev_panel[q] = b.aux_events[q]; st.ev_rest[q] = b.aux_events[2 + q]; }
    int k0 = 0, step = 0;
    while (k0 < b.npad) {
        const int np = st.aux ? 1 : group_at(b.npad, k0, b.group_panels != 0);
        if (np > 1) {
            const int wg = panel_width(b.npad, k0);
            if (wg == 16) lu_group_step<16>(b, k0, step, np, stream);
            else lu_group_step<32>(b, k0, step, np, stream);
            k0 += wg * np;
            step += np;
            continue;
        }
        const int w = panel_width(b.npad, k0);
        if (w == 32) lu_step<32>(b, k0, step, st);
        else if (w == 16) lu_step<16>(b, k0, step, st);
        else if (w == 8) lu_step<8>(b, k0, step, st);
        else lu_step<4>(b, k0, step, st);
        k0 += w;
        ++step;
    }
    // rejoin: whatever part-B update is still running on the aux stream
    if (st.rest_pending) (void)hipStreamWaitEvent(st.main, st.ev_rest[st.last_rest], 0);
    launch_backsub(b, stream, b.npad);
    return hipGetLastError();
}

// everything after k_prepare: radii, assembly, LU, back-substitution, packing
hipError_t launch_build(const BuildBuffers &b, hipStream_t stream, hipEvent_t ev_mid)
{
    if (b.ml_layers) return launch_build_ml(b, stream, ev_mid);
    if (b.spd && b.reg) return launch_build_reg(b, stream, nullptr, ev_mid);      // (the C ABI calls it directly, without k_prepare)
    if (b.spd) return launch_build_spd(b, stream, ev_mid);
    if (b.kind == FD_KERNEL_GAUSSIAN_QNN) return launch_build_qnn(b, stream, ev_mid);   // polynomial first (fd_nullspace.hip)
    const int M = b.M;
    const unsigned nb = (unsigned)b.nbatch;
    if (b.npad <= 512) {
        const unsigned g = (unsigned)(b.npad + 31) / 32;
        hipLaunchKernelGGL((k_assemble<2>), dim3(g, g, nb), dim3(256), 0, stream, b.d_slots, M, b.n, b.npad,
                           b.lda, b.kind, b.T, b.lambda, 0, 0);
    } else {
        const unsigned g = (unsigned)(b.npad + 63) / 64;
        hipLaunchKernelGGL((k_assemble<4>), dim3(g, g, nb), dim3(256), 0, stream, b.d_slots, M, b.n, b.npad,
                           b.lda, b.kind, b.T, b.lambda, 0, 0);
    }
    if (ev_mid) (void)hipEventRecord(ev_mid, stream);
    hipError_t e = launch_lu_factor_solve(b, stream);
    if (e != hipSuccess) return e;
    e = launch_pack(b, stream);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

hipError_t launch_pack(const BuildBuffers &b, hipStream_t stream)
{
    const unsigned nb = (unsigned)b.nbatch;
    hipLaunchKernelGGL(k_pack, dim3(1, 1, nb), dim3(256), 0, stream, b.d_slots, b.npad, b.M, b.Mpad, b.T, b.kind, 0, 0);
    if (b.kind == FD_KERNEL_THIN_PLATE)
        hipLaunchKernelGGL(k_pack_tiles, dim3(b.Mpad / 16, 1, nb), dim3(64), 0, stream, b.d_slots, b.Mpad);
    return hipGetLastError();
}

hipError_t launch_pack_records(const BuildBuffers &b, hipStream_t stream, int records, int kind, int mode, int layers)
{
    hipLaunchKernelGGL(k_pack, dim3(1, 1, (unsigned)b.nbatch), dim3(256), 0, stream, b.d_slots, b.npad, records,
                       round_up(records, kRecPad), b.T, kind, mode, layers);
    return hipGetLastError();
}

hipError_t launch_pack_from_weights(const BuildBuffers &b, hipStream_t stream, int layers)
{
    const unsigned nb = (unsigned)b.nbatch;
    hipLaunchKernelGGL(k_pack, dim3(1, 1, nb), dim3(256), 0, stream, b.d_slots, b.npad, b.M, b.Mpad, b.T, b.kind, 1, layers);
    if (b.kind == FD_KERNEL_THIN_PLATE)
        hipLaunchKernelGGL(k_pack_tiles, dim3(b.Mpad / 16, 1, nb), dim3(64), 0, stream, b.d_slots, b.Mpad);
    return hipGetLastError();
}

// fd_set_deltas: the factorisation in A (L below the diagonal as the panels left it, U above,
// one move list per step) is reused; the 16-column RHS block alone goes through every step's
// row interchanges, triangular solve and update -- the same k_lu_trail, one workgroup per step --
// then back-substitution and packing as usual.  Same kernels, same operand values, same order
// as in a full build: the weights are bit-identical to rebuilding from scratch.
// Requires every panel width to be a multiple of the 16-column block (order <= 2048), so that
// the RHS block is exactly one of the trailing update's column blocks.
// the right-hand-side block alone through every step of the stored LU, then back-substitution
hipError_t launch_lu_resolve_core(const BuildBuffers &b, hipStream_t stream)
{
    const unsigned nb = (unsigned)b.nbatch;
    if (b.npad > 2048) return hipErrorInvalidValue;
    int k0 = 0, step = 0;
    while (k0 < b.npad) {
        const int np = group_at(b.npad, k0, b.group_panels != 0);
        if (np > 1) {   // as in launch_build: one deep update for the group of panels
            const int depth = panel_width(b.npad, k0) * np;
            const int cb = (b.npad - (k0 + depth)) / kColBlock;
            if (depth == 32)
                hipLaunchKernelGGL((k_lu_trail<32>), dim3(1, 1, nb), dim3(256), 0, stream, b.d_slots, b.lda, b.npad, k0, step, cb, np);
            else
                hipLaunchKernelGGL((k_lu_trail<64>), dim3(1, 1, nb), dim3(256), 0, stream, b.d_slots, b.lda, b.npad, k0, step, cb, np);
            k0 += depth;
            step += np;
            continue;
        }
        const int w = panel_width(b.npad, k0);
        const int cb = (b.npad - (k0 + w)) / kColBlock;      // the RHS block among this step's column blocks
        if (w == 32)
            hipLaunchKernelGGL((k_lu_trail<32>), dim3(1, 1, nb), dim3(256), 0, stream, b.d_slots, b.lda, b.npad, k0, step, cb, 1);
        else
            hipLaunchKernelGGL((k_lu_trail<16>), dim3(1, 1, nb), dim3(256), 0, stream, b.d_slots, b.lda, b.npad, k0, step, cb, 1);
        k0 += w;
        ++step;
    }
    launch_backsub(b, stream, b.npad);
    return hipGetLastError();
}

hipError_t launch_resolve(const BuildBuffers &b, hipStream_t stream, const PointSrc *src)
{
    if (b.spd) return launch_resolve_spd(b, stream, src);
    if (b.npad > 2048) return hipErrorInvalidValue;
    if (b.kind == FD_KERNEL_GAUSSIAN_QNN && b.ml_layers == 0) return launch_resolve_qnn(b, stream, src);
    hipError_t e = launch_prepare_rhs(b, stream, src);
    if (e != hipSuccess) return e;
    e = launch_lu_resolve_core(b, stream);
    if (e != hipSuccess) return e;
    e = launch_pack(b, stream);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

}  // namespace fd
