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
            (rc = mrealloc(m, &m->d_tmp, (size_t)S))) return rc;
        m->cap_S = S;
    }
    if (nwg * (size_t)(S ? S : 1) > m->cap_partial) {
        if ((rc = mrealloc(m, &m->d_partial, nwg * (size_t)(S ? S : 1)))) return rc;
        m->cap_partial = nwg * (size_t)(S ? S : 1);
    }
    return FD_OK;
}

static int morph_factor(fd_morph *m)
{
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
    void *bufs[] = {m->d_rest_attr, m->d_rest, m->d_S32, m->d_stage, m->d_P, m->d_QR, m->d_tau, m->d_w, m->d_partial, m->d_hh, m->d_tmp};
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
