/*
 * fd_oracle.c -- see fd_oracle.h.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED
 * against the reference (no reference tests exist; ALGLIB is absent), pinned
 * instead by SciPy-generated golden vectors and closed-form known answers.
 *
 * Plain C99 + pthreads, fp64 arithmetic for the RBF, fp32 for the epilogue
 * exactly where the reference uses UT_Vector3 / float.
 */
#include "fd_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define FDO_CLONES __attribute__((target_clones("avx2", "default")))
#else
#define FDO_CLONES
#endif

/* ---- A2: control table (src/SOP_FaceDeform.cpp:268-287) ------------------ */
void fdo_control_table(const float *rest_xyz, const float *deform_xyz, int M, double *table)
{
    for (int i = 0; i < M; ++i) {
        for (int c = 0; c < 3; ++c) {
            /* :278 subtracts in fp32 (UT_Vector3), :279 widens to double */
            const float delta = deform_xyz[3 * i + c] - rest_xyz[3 * i + c];
            table[6 * i + c] = (double)rest_xyz[3 * i + c];
            table[6 * i + 3 + c] = (double)delta;
        }
    }
}

static int term_cols(int term)
{
    return term == FDO_TERM_LINEAR ? 4 : (term == FDO_TERM_CONST ? 1 : 0);
}

static double param_lambda(int kind, const double *params, int nparams)
{
    int idx;
    switch (kind) {
    case FDO_KERNEL_GAUSSIAN: idx = 1; break;
    case FDO_KERNEL_GAUSSIAN_QNN: idx = 2; break;
    default: idx = 0; break;
    }
    return (params && nparams > idx) ? params[idx] : 0.0;
}

static int cmp_double(const void *a, const void *b)
{
    const double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* ---- A4: radii (src/SOP_FaceDeform.cpp:342-349 parameter meaning) -------- */
int fdo_radii(const double *table, int M, int kind, const double *params, int nparams,
              double *radii)
{
    if (kind == FDO_KERNEL_GAUSSIAN) {
        const double R = (params && nparams > 0) ? params[0] : 1.0;
        if (!(R > 0.0)) return -1;
        for (int i = 0; i < M; ++i) radii[i] = R;
        return 0;
    }
    if (kind != FDO_KERNEL_GAUSSIAN_QNN) {
        for (int i = 0; i < M; ++i) radii[i] = 1.0;
        return 0;
    }
    const double q = (params && nparams > 0) ? params[0] : 1.0;
    const double z = (params && nparams > 1) ? params[1] : 5.0;
    if (!(q > 0.0) || !(z > 0.0)) return -1;
    double *tmp = (double *)malloc(sizeof(double) * (size_t)(M > 0 ? M : 1));
    if (!tmp) return -2;
    for (int i = 0; i < M; ++i) {
        double best = INFINITY;
        for (int j = 0; j < M; ++j) {
            if (j == i) continue;
            const double dx = table[6 * i] - table[6 * j];
            const double dy = table[6 * i + 1] - table[6 * j + 1];
            const double dz = table[6 * i + 2] - table[6 * j + 2];
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (d2 < best) best = d2;
        }
        radii[i] = (M > 1) ? q * sqrt(best) : q;
        tmp[i] = radii[i];
    }
    qsort(tmp, (size_t)M, sizeof(double), cmp_double);
    /* the median is the sorted array's element M / 2 (the UPPER one for even M): ALGLIB's
     * tmp[n/2] as two independent recollections of its rbf unit have it; ALGLIB is
     * absent, so this stays a recollection -- DESIGN.md 6d.  Odd M: the one median either way. */
    const double med = M > 0 ? tmp[M / 2] : 1.0;
    for (int i = 0; i < M; ++i)
        if (radii[i] > z * med) radii[i] = z * med;
    free(tmp);
    return 0;
}

static inline double phi(int kind, double d2, double inv_r2)
{
    switch (kind) {
    case FDO_KERNEL_GAUSSIAN:
    case FDO_KERNEL_GAUSSIAN_QNN: return exp(-d2 * inv_r2);
    case FDO_KERNEL_THIN_PLATE: return d2 > 0.0 ? 0.5 * d2 * log(d2) : 0.0;
    case FDO_KERNEL_BIHARMONIC: return -sqrt(d2);
    case FDO_KERNEL_CUBIC: return d2 * sqrt(d2);
    default: return 0.0;
    }
}

/* ---- dense LU with partial pivoting, n x n row-major + nrhs columns ------- */
FDO_CLONES
static void lu_row_update(double *restrict ai, const double *restrict ak, double l, int from,
                          int to)
{
    for (int j = from; j < to; ++j) ai[j] -= l * ak[j];
}

static int lu_solve(double *A, int n, double *B, int nrhs)
{
    double amax = 0.0;
    for (size_t i = 0; i < (size_t)n * n; ++i) {
        const double v = fabs(A[i]);
        if (v > amax) amax = v;
    }
    if (!(amax > 0.0) || !isfinite(amax)) return n > 0 ? -4 : 0;
    const double tiny = (double)n * 2.220446049250313e-16 * amax;
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = fabs(A[(size_t)k * n + k]);
        for (int i = k + 1; i < n; ++i) {
            const double v = fabs(A[(size_t)i * n + k]);
            if (v > best) { best = v; p = i; }
        }
        if (!(best > tiny) || !isfinite(best)) return -4;
        if (p != k) {
            for (int j = 0; j < n; ++j) {
                const double t = A[(size_t)k * n + j];
                A[(size_t)k * n + j] = A[(size_t)p * n + j];
                A[(size_t)p * n + j] = t;
            }
            for (int j = 0; j < nrhs; ++j) {
                const double t = B[(size_t)k * nrhs + j];
                B[(size_t)k * nrhs + j] = B[(size_t)p * nrhs + j];
                B[(size_t)p * nrhs + j] = t;
            }
        }
        const double inv = 1.0 / A[(size_t)k * n + k];
        for (int i = k + 1; i < n; ++i) {
            const double l = A[(size_t)i * n + k] * inv;
            if (l != 0.0) {
                lu_row_update(A + (size_t)i * n, A + (size_t)k * n, l, k + 1, n);
                for (int j = 0; j < nrhs; ++j) B[(size_t)i * nrhs + j] -= l * B[(size_t)k * nrhs + j];
            }
        }
    }
    for (int k = n - 1; k >= 0; --k) {
        for (int j = 0; j < nrhs; ++j) {
            double s = B[(size_t)k * nrhs + j];
            for (int c = k + 1; c < n; ++c) s -= A[(size_t)k * n + c] * B[(size_t)c * nrhs + j];
            B[(size_t)k * nrhs + j] = s / A[(size_t)k * n + k];
        }
    }
    return 0;
}

/* Least-squares polynomial a (T x 3, rows: constant, x, y, z) of the deltas over the rest points:
 * min |f - P a|_2 by Householder QR of P = [1 x y z] (M x T), then f <- f - P a in place.
 * This is the order ALGLIB's Gaussian models work in ("Term": the linear /
 * constant term is fitted first and removed, the RBF fits the remainder).  -4 when P has no full
 * column rank (fewer points than terms, or all of them in one plane / on one line). */
static int ls_polynomial(const double *table, int M, int T, double *f /* M x 3 */, double a[4][3])
{
    for (int k = 0; k < 4; ++k) a[k][0] = a[k][1] = a[k][2] = 0.0;
    if (T == 0) return 0;
    if (M < T) return -4;
    double *Q = (double *)malloc(sizeof(double) * (size_t)M * (size_t)(T + 3));
    if (!Q) return -2;
    const int C = T + 3;                      /* [P | f] goes through the reflectors together */
    double cn[4] = {0, 0, 0, 0};
    for (int i = 0; i < M; ++i) {
        Q[(size_t)i * C] = 1.0;
        for (int t = 1; t < T; ++t) Q[(size_t)i * C + t] = table[6 * i + t - 1];
        for (int c = 0; c < 3; ++c) Q[(size_t)i * C + T + c] = f[3 * i + c];
        for (int t = 0; t < T; ++t) cn[t] += Q[(size_t)i * C + t] * Q[(size_t)i * C + t];
    }
    int rc = 0;
    for (int k = 0; k < T && rc == 0; ++k) {
        double sigma = 0.0;
        for (int i = k + 1; i < M; ++i) sigma += Q[(size_t)i * C + k] * Q[(size_t)i * C + k];
        const double xk = Q[(size_t)k * C + k];
        const double norm = sqrt(xk * xk + sigma);
        if (!(norm > 64.0 * (double)M * 2.220446049250313e-16 * sqrt(cn[k]))) { rc = -4; break; }
        if (sigma > 0.0) {
            const double beta = xk >= 0.0 ? -norm : norm;
            const double tau = (beta - xk) / beta, scale = 1.0 / (xk - beta);
            for (int c = k + 1; c < C; ++c) {
                double d = Q[(size_t)k * C + c];
                for (int i = k + 1; i < M; ++i) d += scale * Q[(size_t)i * C + k] * Q[(size_t)i * C + c];
                Q[(size_t)k * C + c] -= tau * d;
                for (int i = k + 1; i < M; ++i) Q[(size_t)i * C + c] -= tau * d * scale * Q[(size_t)i * C + k];
            }
            Q[(size_t)k * C + k] = beta;
        }
    }
    if (rc == 0) {
        for (int c = 0; c < 3; ++c)
            for (int k = T - 1; k >= 0; --k) {
                double v = Q[(size_t)k * C + T + c];
                for (int t = k + 1; t < T; ++t) v -= Q[(size_t)k * C + t] * a[t][c];
                a[k][c] = v / Q[(size_t)k * C + k];
            }
        for (int i = 0; i < M; ++i)
            for (int c = 0; c < 3; ++c) {
                double p = a[0][c];
                for (int t = 1; t < T; ++t) p += a[t][c] * table[6 * i + t - 1];
                f[3 * i + c] -= p;
            }
    }
    free(Q);
    return rc;
}

/* ---- A3-A6 (src/SOP_FaceDeform.cpp:331-368) ------------------------------ */
int fdo_build(const double *table, int M, int kind, const double *params, int nparams,
              int term, double *W, double *radii_out, int *terminationtype)
{
    int tt = 1;
    const int T = term_cols(term);
    const int n = M + T;
    memset(W, 0, sizeof(double) * (size_t)(M + 4) * 3);
    if (M <= 0 || kind < 0 || kind > FDO_KERNEL_CUBIC || term < 0 || term > 2) {
        if (terminationtype) *terminationtype = -4;
        return -1;
    }
    double *radii = radii_out;
    double *radii_own = NULL;
    if (!radii) {
        radii_own = (double *)malloc(sizeof(double) * (size_t)M);
        radii = radii_own;
    }
    if (fdo_radii(table, M, kind, params, nparams, radii) != 0) {
        free(radii_own);
        if (terminationtype) *terminationtype = -4;
        return -1;
    }
    /* coincident centres: ALGLIB reports -5 */
    for (int i = 0; i < M && tt == 1; ++i)
        for (int j = i + 1; j < M; ++j) {
            if (table[6 * i] == table[6 * j] && table[6 * i + 1] == table[6 * j + 1] &&
                table[6 * i + 2] == table[6 * j + 2]) {
                tt = -5;
                break;
            }
        }
    if (tt != 1) {
        free(radii_own);
        if (terminationtype) *terminationtype = tt;
        return -5;
    }
    const double lambda = param_lambda(kind, params, nparams);
    if (kind == FDO_KERNEL_GAUSSIAN_QNN) {
        /* The SOP's model = 0, rbfsetalgoqnn (src/SOP_FaceDeform.cpp:343-345), in ALGLIB's order:
         * the term's polynomial by least squares first, then the Gaussians on what is left,
         *     (Phi + lambda I) w = f - P a,     Phi_ij = exp(-|c_i - c_j|^2 / R_j^2)
         * (Phi is not symmetric with per-centre radii: pivoted LU).  Round 1 solved the polynomial
         * together with the weights (the saddle-point system below); both interpolate at the rig
         * points but differ everywhere else. */
        double *A = (double *)calloc((size_t)M * M, sizeof(double));
        double *B = (double *)calloc((size_t)M * 3, sizeof(double));
        double a[4][3];
        if (!A || !B) {
            free(A); free(B); free(radii_own);
            if (terminationtype) *terminationtype = -4;
            return -2;
        }
        for (int i = 0; i < M; ++i)
            for (int c = 0; c < 3; ++c) B[(size_t)i * 3 + c] = table[6 * i + 3 + c];
        int rc = ls_polynomial(table, M, T, B, a);
        if (rc == 0) {
            for (int i = 0; i < M; ++i) {
                for (int j = 0; j < M; ++j) {
                    const double dx = table[6 * i] - table[6 * j];
                    const double dy = table[6 * i + 1] - table[6 * j + 1];
                    const double dz = table[6 * i + 2] - table[6 * j + 2];
                    A[(size_t)i * M + j] = phi(kind, dx * dx + dy * dy + dz * dz, 1.0 / (radii[j] * radii[j]));
                }
                A[(size_t)i * M + i] += lambda;
            }
            rc = lu_solve(A, M, B, 3);
        }
        if (rc != 0) tt = -4;
        else {
            for (int i = 0; i < M; ++i)
                for (int c = 0; c < 3; ++c) {
                    if (!isfinite(B[(size_t)i * 3 + c])) tt = -4;
                    W[(size_t)i * 3 + c] = B[(size_t)i * 3 + c];
                }
            for (int k = 0; k < T; ++k)
                for (int c = 0; c < 3; ++c) {
                    if (!isfinite(a[k][c])) tt = -4;
                    W[(size_t)(M + k) * 3 + c] = a[k][c];
                }
            if (tt != 1) memset(W, 0, sizeof(double) * (size_t)(M + 4) * 3);
        }
        free(A); free(B); free(radii_own);
        if (terminationtype) *terminationtype = tt;
        return tt == 1 ? 0 : -4;
    }
    double *A = (double *)calloc((size_t)n * n, sizeof(double));
    double *B = (double *)calloc((size_t)n * 3, sizeof(double));
    if (!A || !B) {
        free(A); free(B); free(radii_own);
        if (terminationtype) *terminationtype = -4;
        return -2;
    }
    for (int i = 0; i < M; ++i) {
        for (int j = 0; j < M; ++j) {
            const double dx = table[6 * i] - table[6 * j];
            const double dy = table[6 * i + 1] - table[6 * j + 1];
            const double dz = table[6 * i + 2] - table[6 * j + 2];
            const double d2 = dx * dx + dy * dy + dz * dz;
            A[(size_t)i * n + j] = phi(kind, d2, 1.0 / (radii[j] * radii[j]));
        }
        A[(size_t)i * n + i] += lambda;
        if (T >= 1) {
            A[(size_t)i * n + M] = 1.0;
            A[(size_t)M * n + i] = 1.0;
        }
        if (T == 4) {
            for (int c = 0; c < 3; ++c) {
                A[(size_t)i * n + M + 1 + c] = table[6 * i + c];
                A[(size_t)(M + 1 + c) * n + i] = table[6 * i + c];
            }
        }
        for (int c = 0; c < 3; ++c) B[(size_t)i * 3 + c] = table[6 * i + 3 + c];
    }
    const int rc = lu_solve(A, n, B, 3);
    if (rc != 0) {
        tt = -4;
    } else {
        for (int i = 0; i < n; ++i)
            for (int c = 0; c < 3; ++c) {
                if (!isfinite(B[(size_t)i * 3 + c])) tt = -4;
                W[(size_t)i * 3 + c] = B[(size_t)i * 3 + c];
            }
        if (tt != 1) memset(W, 0, sizeof(double) * (size_t)(M + 4) * 3);
    }
    free(A); free(B); free(radii_own);
    if (terminationtype) *terminationtype = tt;
    return tt == 1 ? 0 : -4;
}

/* ---- A8: rbfcalc equivalent (src/SOP_FaceDeform.cpp:412-415) ------------- */
static inline void eval_one(const double *table, int M, int kind, const double *inv_r2,
                            const double *W, const double x[3], double out[3])
{
    /* affine part first: rows M (const), M+1..M+3 (linear) */
    double ax = W[3 * M + 0], ay = W[3 * M + 1], az = W[3 * M + 2];
    for (int c = 0; c < 3; ++c) {
        ax += W[3 * (M + 1 + c) + 0] * x[c];
        ay += W[3 * (M + 1 + c) + 1] * x[c];
        az += W[3 * (M + 1 + c) + 2] * x[c];
    }
    for (int j = 0; j < M; ++j) {
        const double dx = x[0] - table[6 * j];
        const double dy = x[1] - table[6 * j + 1];
        const double dz = x[2] - table[6 * j + 2];
        const double d2 = dx * dx + dy * dy + dz * dz;
        const double p = phi(kind, d2, inv_r2[j]);
        ax += p * W[3 * j];
        ay += p * W[3 * j + 1];
        az += p * W[3 * j + 2];
    }
    out[0] = ax; out[1] = ay; out[2] = az;
}

static double *make_inv_r2(int M, const double *radii)
{
    double *inv = (double *)malloc(sizeof(double) * (size_t)(M > 0 ? M : 1));
    if (!inv) return NULL;
    for (int j = 0; j < M; ++j) inv[j] = radii ? 1.0 / (radii[j] * radii[j]) : 1.0;
    return inv;
}

void fdo_eval(const double *table, int M, int kind, const double *radii, const double *W,
              int64_t N, const double *x_xyz, double *delta_out)
{
    double *inv = make_inv_r2(M, radii);
    if (!inv) return;
    for (int64_t i = 0; i < N; ++i) eval_one(table, M, kind, inv, W, x_xyz + 3 * i, delta_out + 3 * i);
    free(inv);
}

/* ---- A9 (src/SOP_FaceDeform.hpp:28-41), all fp32 ------------------------- */
static inline void normalize3f(float v[3])
{
    const float l2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    if (l2 > 0.f) {
        const float inv = 1.f / sqrtf(l2);
        v[0] *= inv; v[1] *= inv; v[2] *= inv;
    }
}

void fdo_project_to_tangents(const float u[3], const float v[3], const float n[3], float disp[3])
{
    const float b[3][3] = {{u[0], u[1], u[2]}, {v[0], v[1], v[2]}, {n[0], n[1], n[2]}};
    float g[3][3]; /* b^T * b  (hpp:35) */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            g[i][j] = b[0][i] * b[0][j] + b[1][i] * b[1][j] + b[2][i] * b[2][j];
    float a1[3], a2[3]; /* row-vector * matrix (hpp:36-37) */
    for (int j = 0; j < 3; ++j) {
        a1[j] = u[0] * g[0][j] + u[1] * g[1][j] + u[2] * g[2][j];
        a2[j] = v[0] * g[0][j] + v[1] * g[1][j] + v[2] * g[2][j];
    }
    normalize3f(a1);
    normalize3f(a2);
    const float da1 = disp[0] * a1[0] + disp[1] * a1[1] + disp[2] * a1[2];
    const float da2 = disp[0] * a2[0] + disp[1] * a2[1] + disp[2] * a2[2];
    for (int c = 0; c < 3; ++c) disp[c] = a1[c] * da1 + a2[c] * da2;
}

/* ---- A7-A10: the loop body (src/SOP_FaceDeform.cpp:404-439) -------------- */
typedef struct {
    const double *table; int M; int kind; const double *inv_r2; const double *W;
    int64_t begin, end;
    const float *P_in; float *P_out; const float *dist2; float *falloff_out;
    const float *tu, *tv, *nrm; float radius2, falloffrate;
} deform_job;

static void deform_range(const deform_job *jb)
{
    const int do_tangent = jb->tu && jb->tv && jb->nrm;
    for (int64_t i = jb->begin; i < jb->end; ++i) {
        float distance_sqrt = 0.f;                       /* :405 */
        if (jb->dist2) distance_sqrt = jb->dist2[i];     /* :406-407 */
        const float pos[3] = {jb->P_in[3 * i], jb->P_in[3 * i + 1], jb->P_in[3 * i + 2]};
        if (distance_sqrt > jb->radius2) {               /* :408-410, gate on squares */
            if (jb->P_out != jb->P_in) {
                jb->P_out[3 * i] = pos[0]; jb->P_out[3 * i + 1] = pos[1]; jb->P_out[3 * i + 2] = pos[2];
            }
            continue;                                    /* no fd_falloff write (B2) */
        }
        const double dp[3] = {pos[0], pos[1], pos[2]};   /* :412 widen */
        double result[3];
        eval_one(jb->table, jb->M, jb->kind, jb->inv_r2, jb->W, dp, result); /* :414 */
        float displace[3] = {(float)result[0], (float)result[1], (float)result[2]}; /* :415 */
        if (do_tangent) {                                /* :416-422 */
            float u[3] = {jb->tu[3 * i], jb->tu[3 * i + 1], jb->tu[3 * i + 2]};
            float v[3] = {jb->tv[3 * i], jb->tv[3 * i + 1], jb->tv[3 * i + 2]};
            float n[3] = {jb->nrm[3 * i], jb->nrm[3 * i + 1], jb->nrm[3 * i + 2]};
            normalize3f(u); normalize3f(v); normalize3f(n);
            fdo_project_to_tangents(u, v, n, displace);
        }
        float falloff = fminf(distance_sqrt / jb->radius2, 1.f); /* :423 */
        falloff = powf(1.f - falloff, jb->falloffrate);          /* :424 */
        if (jb->falloff_out) jb->falloff_out[i] = falloff;       /* :425 */
        for (int c = 0; c < 3; ++c) jb->P_out[3 * i + c] = pos[c] + displace[c] * falloff; /* :437-438 */
    }
}

static void *deform_thread(void *arg)
{
    deform_range((const deform_job *)arg);
    return NULL;
}

int fdo_deform(const double *table, int M, int kind, const double *radii, const double *W,
               int64_t N, const float *P_in, float *P_out, const float *dist2,
               float *falloff_out, const float *tu, const float *tv, const float *nrm,
               float radius2, float falloffrate, int nthreads)
{
    double *inv = make_inv_r2(M, radii);
    if (!inv) return -2;
    deform_job base = {table, M, kind, inv, W, 0, N, P_in, P_out, dist2, falloff_out,
                       tu, tv, nrm, radius2, falloffrate};
    if (nthreads <= 1 || N < 2 * (int64_t)nthreads) {
        deform_range(&base);
        free(inv);
        return 0;
    }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    deform_job *jobs = (deform_job *)malloc(sizeof(deform_job) * (size_t)nthreads);
    if (!th || !jobs) { free(th); free(jobs); free(inv); return -2; }
    for (int t = 0; t < nthreads; ++t) {
        jobs[t] = base;
        jobs[t].begin = N * t / nthreads;
        jobs[t].end = N * (t + 1) / nthreads;
        if (pthread_create(&th[t], NULL, deform_thread, &jobs[t]) != 0) {
            deform_range(&jobs[t]);
            th[t] = 0;
        }
    }
    for (int t = 0; t < nthreads; ++t)
        if (th[t]) pthread_join(th[t], NULL);
    free(th); free(jobs); free(inv);
    return 0;
}

/* ---- next row N1: morph-space reprojection (reference src/dbse.cpp) ---------------------- */

/* dbse.cpp:16-31: one column per blendshape, rows 3i..3i+2 = fp32 (shape - rest) widened */
void fdo_morph_shapes_matrix(const float *rest_xyz, const float *const *shapes_xyz, int64_t N, int S, double *A)
{
    const int64_t rows = 3 * N;
    for (int s = 0; s < S; ++s)
        for (int64_t e = 0; e < rows; ++e) {
            const float d = shapes_xyz[s][e] - rest_xyz[e];      /* UT_Vector3 subtraction, :25 */
            A[(size_t)s * rows + e] = (double)d;
        }
}

/* dbse.cpp:33 `new QRMatrix(myShapesMatrix)`: Eigen::HouseholderQR<MatrixXd>.  Unblocked form
 * (Eigen/src/QR/HouseholderQR.h householder_qr_inplace_unblocked; the blocked variant Eigen picks
 * for wide matrices applies the same reflectors): per column k, makeHouseholderInPlace on
 * A[k:, k] then applyHouseholderOnTheLeft to A[k:, k+1:].  Identical to LAPACK dgeqr2. */
void fdo_morph_qr(double *A, int64_t rows, int S, double *tau)
{
    for (int k = 0; k < S && k < rows; ++k) {
        double *x = A + (size_t)k * rows;
        const int64_t nt = rows - k - 1;
        double tail2 = 0.0;
        for (int64_t i = k + 1; i < rows; ++i) tail2 += x[i] * x[i];
        const double c0 = x[k];
        double beta, t;
        if (nt == 0 || tail2 <= 2.2250738585072014e-308) {       /* numeric_limits<double>::min() */
            t = 0.0; beta = c0;
            for (int64_t i = k + 1; i < rows; ++i) x[i] = 0.0;
        } else {
            beta = sqrt(c0 * c0 + tail2);
            if (c0 >= 0.0) beta = -beta;
            for (int64_t i = k + 1; i < rows; ++i) x[i] = x[i] / (c0 - beta);   /* Eigen divides, dlarfg scales by the reciprocal */
            t = (beta - c0) / beta;
        }
        x[k] = beta;
        tau[k] = t;
        /* applyHouseholderOnTheLeft: tmp = essential^T * bottom + row0; row0 -= tau tmp;
         * bottom -= tau * essential * tmp */
        for (int j = k + 1; j < S; ++j) {
            double *a = A + (size_t)j * rows;
            double tmp = 0.0;
            for (int64_t i = k + 1; i < rows; ++i) tmp += x[i] * a[i];
            tmp += a[k];
            a[k] -= t * tmp;
            for (int64_t i = k + 1; i < rows; ++i) a[i] -= t * x[i] * tmp;
        }
    }
}

/* dbse.cpp:39-60 */
void fdo_morph_weights(const double *QR, int64_t N, int S, const float *P_xyz, const float *rest_xyz, double *w)
{
    const int64_t rows = 3 * N;
    for (int s = 0; s < S; ++s) {
        const double *q = QR + (size_t)s * rows;
        double acc = 0.0;
        for (int64_t e = 0; e < rows; ++e) {
            const float d = P_xyz[e] - rest_xyz[e];              /* :49-51, fp32 then widened */
            acc += (double)d * q[e];                             /* asDiagonal() * matrixQR(), colwise sum :55-56 */
        }
        w[s] = acc;
    }
}

/* dbse.cpp:62-77 and SOP_FaceDeform.cpp:458-473 */
void fdo_morph_displace(const double *shapes, int64_t N, int S, const double *w, const float *clamp_lo_hi,
                        int add_delta, float falloffradius, const float *rest_xyz, float *P_xyz)
{
    const int64_t rows = 3 * N;
    for (int64_t i = 0; i < N; ++i) {
        float disp[3] = {0.f, 0.f, 0.f};
        for (int s = 0; s < S; ++s) {
            const float xd = (float)shapes[(size_t)s * rows + 3 * i];
            const float yd = (float)shapes[(size_t)s * rows + 3 * i + 1];
            const float zd = (float)shapes[(size_t)s * rows + 3 * i + 2];
            const float ws = (float)(w[s] * 3);                  /* :70, the magic number */
            float cw = ws;
            if (clamp_lo_hi) cw = ws < clamp_lo_hi[0] ? clamp_lo_hi[0] : (ws > clamp_lo_hi[1] ? clamp_lo_hi[1] : ws);
            disp[0] += xd * cw; disp[1] += yd * cw; disp[2] += zd * cw;
        }
        const float *rest = rest_xyz + 3 * i;
        float *pos = P_xyz + 3 * i;
        if (add_delta) {                                         /* SOP :467-470 */
            disp[0] += (pos[0] - rest[0]) * falloffradius;
            disp[1] += (pos[1] - rest[1]) * falloffradius;
            disp[2] += (pos[2] - rest[2]) * falloffradius;
        }
        pos[0] = rest[0] + disp[0]; pos[1] = rest[1] + disp[1]; pos[2] = rest[2] + disp[2];   /* :471 */
    }
}

/* ---- next row N2: dist2 producer (reference src/capture.cpp:46-99) ------------------------ */
static double tri_dist2(const double p[3], const float *t)
{
    double a[3], b[3], c[3], ab[3], ac[3], ap[3], bp[3], cp[3], q[3];
    for (int k = 0; k < 3; ++k) { a[k] = t[k]; b[k] = t[3 + k]; c[k] = t[6 + k]; }
    for (int k = 0; k < 3; ++k) { ab[k] = b[k] - a[k]; ac[k] = c[k] - a[k]; ap[k] = p[k] - a[k]; }
#define DOT(u, v) ((u)[0] * (v)[0] + (u)[1] * (v)[1] + (u)[2] * (v)[2])
    const double d1 = DOT(ab, ap), d2 = DOT(ac, ap);
    if (d1 <= 0.0 && d2 <= 0.0) { for (int k = 0; k < 3; ++k) q[k] = a[k]; goto done; }           /* vertex A */
    for (int k = 0; k < 3; ++k) bp[k] = p[k] - b[k];
    const double d3 = DOT(ab, bp), d4 = DOT(ac, bp);
    if (d3 >= 0.0 && d4 <= d3) { for (int k = 0; k < 3; ++k) q[k] = b[k]; goto done; }            /* vertex B */
    const double vc = d1 * d4 - d3 * d2;
    if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) {                                                     /* edge AB */
        const double v = d1 / (d1 - d3);
        for (int k = 0; k < 3; ++k) q[k] = a[k] + v * ab[k];
        goto done;
    }
    for (int k = 0; k < 3; ++k) cp[k] = p[k] - c[k];
    const double d5 = DOT(ab, cp), d6 = DOT(ac, cp);
    if (d6 >= 0.0 && d5 <= d6) { for (int k = 0; k < 3; ++k) q[k] = c[k]; goto done; }            /* vertex C */
    const double vb = d5 * d2 - d1 * d6;
    if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) {                                                     /* edge AC */
        const double w = d2 / (d2 - d6);
        for (int k = 0; k < 3; ++k) q[k] = a[k] + w * ac[k];
        goto done;
    }
    const double va = d3 * d6 - d5 * d4;
    if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {                                       /* edge BC */
        const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        for (int k = 0; k < 3; ++k) q[k] = b[k] + w * (c[k] - b[k]);
        goto done;
    }
    {
        const double den = va + vb + vc;
        if (den == 0.0) { for (int k = 0; k < 3; ++k) q[k] = a[k]; goto done; }                    /* degenerate */
        const double v = vb / den, w = vc / den;                                                   /* face */
        for (int k = 0; k < 3; ++k) q[k] = a[k] + ab[k] * v + ac[k] * w;
    }
done:
    {
        const double dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
        return dx * dx + dy * dy + dz * dz;
    }
#undef DOT
}

void fdo_capture_dist2(const float *P_xyz, int64_t N, const unsigned char *mask, const float *tri_xyz, int T,
                       float radius2, int dofalloff, float *dist2)
{
    for (int64_t i = 0; i < N; ++i) {
        if (mask && !mask[i]) { dist2[i] = 0.f; continue; }        /* attribute default, capture.cpp:31 */
        if (!dofalloff) { dist2[i] = 0.f; continue; }              /* :71-75 */
        const double p[3] = {P_xyz[3 * i], P_xyz[3 * i + 1], P_xyz[3 * i + 2]};
        double best = INFINITY;
        for (int t = 0; t < T; ++t) {
            const double d = tri_dist2(p, tri_xyz + 9 * (size_t)t);
            if (d < best) best = d;
        }
        const float bf = (float)best;
        dist2[i] = (T > 0 && bf < radius2) ? bf : -1.f;            /* :76-88 */
    }
}

/* src/capture.cpp:101-141 */
void fdo_capture_islands(const float *P_xyz, int64_t N, const int64_t *offsets, const int32_t *neighbours,
                         const float *rig_xyz, int M, int max_edges, unsigned char *mask)
{
    int *level = (int *)malloc(sizeof(int) * (size_t)(N > 0 ? N : 1));
    int64_t *queue = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N > 0 ? N : 1));
    for (int64_t i = 0; i < N; ++i) mask[i] = 0;
    for (int m = 0; m < M && N > 0; ++m) {
        /* :120-121 nearest mesh point of the rig point */
        const double a[3] = {rig_xyz[3 * m], rig_xyz[3 * m + 1], rig_xyz[3 * m + 2]};
        double best = INFINITY;
        int64_t bi = 0;
        for (int64_t i = 0; i < N; ++i) {
            const double dx = P_xyz[3 * i] - a[0], dy = P_xyz[3 * i + 1] - a[1], dz = P_xyz[3 * i + 2] - a[2];
            const double d = dx * dx + dy * dy + dz * dz;
            if (d < best) { best = d; bi = i; }
        }
        /* :132 groupEdgePoints(target, max_edges): breadth first over the edges */
        for (int64_t i = 0; i < N; ++i) level[i] = -1;
        int64_t head = 0, tail = 0;
        level[bi] = 0; queue[tail++] = bi;
        while (head < tail) {
            const int64_t v = queue[head++];
            mask[v] = 1;                                             /* :133-134 combine into the handle's group */
            if (level[v] >= max_edges) continue;
            for (int64_t e = offsets[v]; e < offsets[v + 1]; ++e) {
                const int32_t u = neighbours[e];
                if (level[u] < 0) { level[u] = level[v] + 1; queue[tail++] = u; }
            }
        }
    }
    free(level);
    free(queue);
}

