/*
 * fd_oracle.h -- CPU restatement (plain C, fp64) of the RBF deformation hot path
 * of symek/facedeform's SOP_FaceDeform::cookMySop.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and there only as the checker / the timed CPU baseline.  The shipped path
 * (facedeform_amd/, include/facedeform_hip.h) never links or calls this file.
 *
 * PARITY UNPINNED against the reference: the reference has no tests, fixtures
 * or golden vectors, and its RBF arithmetic lives in ALGLIB, an un-vendored,
 * un-pinned third-party dependency (reference CMakeLists.txt:9-19) that is not
 * present here, so the reference path cannot be compiled or run (needs the
 * Houdini HDK as well).  What pins this oracle instead: (1) golden vectors
 * generated in the build container by an independent implementation of the
 * same dense system, SciPy 1.15.3 scipy.interpolate.RBFInterpolator
 * (tests/golden/make_golden.py, fixtures committed under tests/golden/),
 * (2) closed-form known answers, (3) one known-answer test per reference
 * epilogue behaviour.
 *
 * What is restated, and from where (paths relative to /root/reference):
 *   fdo_control_table      src/SOP_FaceDeform.cpp:268-287  (M x 6 table, fp32 delta)
 *   fdo_build              src/SOP_FaceDeform.cpp:331-368  (model config + solve; dense
 *                          formulation of north_star instead of ALGLIB QNN/ML)
 *   fdo_deform             src/SOP_FaceDeform.cpp:396-439  (gate, evaluate, tangent,
 *                          fall-off, write-back -- in the reference's order)
 *   fdo_project_to_tangents src/SOP_FaceDeform.hpp:28-41
 *   fdo_morph_*            src/dbse.cpp:9-87, src/SOP_FaceDeform.cpp:444-473 (next row N1:
 *                          morph-space reprojection).  The Householder QR there is Eigen's
 *                          (un-vendored, absent); its packed storage and reflector convention
 *                          are those of LAPACK dgeqr2, which is what pins this restatement
 *                          (SciPy scipy.linalg.qr(mode='raw'), tests/golden/make_golden.py).
 */
#ifndef FD_ORACLE_H
#define FD_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Radial kernels.  d2 = squared distance to centre j.
 * GAUSSIAN      exp(-d2 / R^2), one radius R for all centres   params: {R[, lambda]}
 * GAUSSIAN_QNN  exp(-d2 / R_j^2), R_j = min(q*nn_j, z*median_k(q*nn_k)), nn_j = distance
 *               to the nearest other centre   params: {q, z[, lambda]}
 *               Built in ALGLIB's order: the term's polynomial is a least-squares fit to the
 *               deltas, removed first; the Gaussians then fit the remainder,
 *               (Phi + lambda I) w = f - P a.  (Every other kind solves polynomial and weights
 *               together: the constrained saddle-point system of north_star.)
 * THIN_PLATE    r^2 ln r = 0.5 * d2 * ln d2, 0 at d2 = 0       params: {[lambda]}
 * BIHARMONIC    -r   (SciPy 'linear' sign convention)          params: {[lambda]}
 * CUBIC         r^3                                             params: {[lambda]}
 */
enum {
    FDO_KERNEL_GAUSSIAN = 0,
    FDO_KERNEL_GAUSSIAN_QNN = 1,
    FDO_KERNEL_THIN_PLATE = 2,
    FDO_KERNEL_BIHARMONIC = 3,
    FDO_KERNEL_CUBIC = 4
};

/* Same integers as ALGLIB_TERM_* in src/SOP_FaceDeform.hpp:16-18. */
enum { FDO_TERM_LINEAR = 0, FDO_TERM_CONST = 1, FDO_TERM_ZERO = 2 };

/* A2: table[i] = (rest_i.xyz, float(deform_i - rest_i).xyz) widened to fp64. */
void fdo_control_table(const float *rest_xyz, const float *deform_xyz, int M, double *table);

/* Per-centre Gaussian radii for `kind`; radii[M].  Non-Gaussian kinds get 1.0. */
int fdo_radii(const double *table, int M, int kind, const double *params, int nparams,
              double *radii);

/* A3-A6: dense assemble + LU (partial pivoting) solve.
 * W is (M+4) x 3 row-major: rows 0..M-1 RBF weights, row M the constant
 * coefficient, rows M+1..M+3 the x,y,z linear coefficients (zero when the
 * term does not carry them).  radii_out[M] receives the Gaussian radii.
 * Returns 0 and *terminationtype = 1 on success; *terminationtype = -5 for
 * coincident centres, -4 for a singular system (then returns nonzero). */
int fdo_build(const double *table, int M, int kind, const double *params, int nparams,
              int term, double *W, double *radii_out, int *terminationtype);

/* A8 alone: delta_out[i] = RBF(x_i) in fp64 (no epilogue). */
void fdo_eval(const double *table, int M, int kind, const double *radii, const double *W,
              int64_t N, const double *x_xyz, double *delta_out);

/* A9. u, v, n must already be normalised by the caller, as in the reference. */
void fdo_project_to_tangents(const float u[3], const float v[3], const float n[3],
                             float disp[3]);

/* A7-A10: the evaluation loop.  P_out may alias P_in.  dist2 / falloff_out /
 * (tu, tv, nrm) may be NULL.  nthreads <= 1 runs on the calling thread (the
 * reference is single-threaded: src/SOP_FaceDeform.hpp:11). */
int fdo_deform(const double *table, int M, int kind, const double *radii, const double *W,
               int64_t N, const float *P_in, float *P_out, const float *dist2,
               float *falloff_out, const float *tu, const float *tv, const float *nrm,
               float radius2, float falloffrate, int nthreads);

/* ---- next row N1: morph-space reprojection (DirectBSEdit) -------------------------
 * dbse.cpp:9-37: shapes matrix A (3N x S, column-major, lda = 3N), A[3i+c][s] =
 * double(float(shape_s[i][c] - rest[i][c])), then HouseholderQR in place: on return A holds
 * Eigen's matrixQR() -- R in the upper triangle, the essential parts of the reflectors
 * below the diagonal; tau[S] are the Householder coefficients.  S <= 3N. */
void fdo_morph_shapes_matrix(const float *rest_xyz, const float *const *shapes_xyz, int64_t N, int S, double *A);
void fdo_morph_qr(double *A, int64_t rows, int S, double *tau);
/* dbse.cpp:39-60: delta = float(P - rest) per component, w_s = sum_i delta_i * QR[i][s]
 * (the packed matrix itself, as the reference uses it). */
void fdo_morph_weights(const double *QR, int64_t N, int S, const float *P_xyz, const float *rest_xyz, double *w);
/* dbse.cpp:62-77 + SOP_FaceDeform.cpp:458-473, fp32 as there: disp = sum_s float(A0[.][s]) *
 * clamp(float(3 w_s)); if (add_delta) disp += (P - rest) * falloffradius; P = rest + disp.
 * shapes = the UNfactored matrix of fdo_morph_shapes_matrix; clamp_lo_hi NULL = no clamping. */
void fdo_morph_displace(const double *shapes, int64_t N, int S, const double *w, const float *clamp_lo_hi,
                        int add_delta, float falloffradius, const float *rest_xyz, float *P_xyz);

/* ---- next row N2: the dist2 producer (ProximityCapture::capture, src/capture.cpp:46-99) ----
 * For every mesh point of an island (mask[i] != 0; mask NULL = every point): with dofalloff off
 * the attribute gets 0 (:71-75); otherwise the squared distance to the closest point of the rest
 * rig's surface if that is below radius2, else -1 (:76-88: GU_RayIntersect::minimumPoint with
 * GU_MinInfo(radius2) either finds a closer point or leaves distance_sqrt at -1).  Points outside
 * every island keep the detached attribute's default 0 (:31).  The rig surface is given as T
 * triangles, 9 floats each; the HDK's own primitive evaluation is not reproduced.  Closest point
 * on a triangle by Voronoi regions (Ericson, Real-Time Collision Detection 5.1.5), in fp64. */
void fdo_capture_dist2(const float *P_xyz, int64_t N, const unsigned char *mask, const float *tri_xyz, int T,
                       float radius2, int dofalloff, float *dist2);

/* ProximityCapture::findIslands (src/capture.cpp:101-141): for every rig point the nearest mesh
 * point (GEO_PointTree::findNearestIdx; ties go to the lower index here), then every mesh point
 * within max_edges edges of it (GQ_Detail::groupEdgePoints; the start point included).  The
 * handle classes only partition the result into groups that capture() treats alike, so the
 * product is their union: mask[i] = 1 for island points.  The mesh's edges come as a CSR
 * adjacency (offsets[N+1], neighbours[offsets[N]]). */
void fdo_capture_islands(const float *P_xyz, int64_t N, const int64_t *offsets, const int32_t *neighbours,
                         const float *rig_xyz, int M, int max_edges, unsigned char *mask);

#ifdef __cplusplus
}
#endif
#endif
