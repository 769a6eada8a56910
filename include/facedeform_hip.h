/*
 * facedeform_hip.h -- C ABI of the MI355X (gfx950) RBF deformation engine.
 *
 * This is the drop-in boundary for the hot path of symek/facedeform's
 * SOP_FaceDeform::cookMySop.  Each entry point names the reference interface
 * it replaces (paths relative to the reference tree).  Plain C: no C++ types,
 * no torch types, no exceptions cross it.  All functions return 0 (FD_OK) on
 * success and a negative FD_E_* code on failure unless stated otherwise;
 * fd_last_error() gives the text.
 *
 * Threading: an fd_ctx is not thread-safe; distinct contexts are independent
 * and re-entrant (different SOP instances may cook concurrently).
 *
 * Memory: "host" entry points take caller-owned host pointers and copy.
 * "_dev" entry points take device pointers valid on the context's device and
 * enqueue on the context's stream without synchronising.
 */
#ifndef FACEDEFORM_HIP_H
#define FACEDEFORM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FD_ABI_VERSION 9   /* 3: fd_config.solver, FD_KERNEL_GAUSSIAN_ML, fd_model_centres; 4: fd_mesh_capture, capture inputs at the end of fdsop_geo; 5: fd_batch_wait_consumed, fd_batch_prepare_shared; 6: fd_batch_set_eval_cus; 7: fd_batch_cook_group, FD_SOLVER_REGISTER / FD_SOLVER_CHAIN; 8: fd_report grows by the fp32 estimate (callers built against 7 pass a shorter struct: rebuild), fd_set_eval_precision, fd_fp32_holds, fd_report.reserved becomes solver_used; 9: fd_shared_kernel_name, fd_set_output, fd_batch_set_shared_factor */

/* ---- error codes ---------------------------------------------------------- */
enum {
    FD_OK = 0,
    FD_E_INVALID = -1,    /* bad argument / call order                           */
    FD_E_NOMEM = -2,      /* host or device allocation failed                    */
    FD_E_DEVICE = -3,     /* HIP runtime error (text in fd_last_error)           */
    FD_E_SINGULAR = -4,   /* solver failure; report.terminationtype = -4         */
    FD_E_DUPLICATE = -5,  /* coincident control points; terminationtype = -5     */
    FD_E_NOT_BUILT = -6,  /* fd_deform before a successful fd_build              */
    FD_E_NO_DEVICE = -7   /* no usable gfx950 device / kernels cannot load       */
};

/* ---- radial kernels (d2 = squared distance to centre j) --------------------
 * The reference exposes ALGLIB's two Gaussian models through the `model` parm
 * (src/SOP_FaceDeform.cpp:48-53,342-349).  The engine solves the dense system
 * north_star describes; kinds 0/1 are the reference's family, 2..4 BASELINE's.
 *   FD_KERNEL_GAUSSIAN      exp(-d2/R^2)            params {R [, lambda]}
 *       (model=1 "Multilayer" collapsed to one layer: R = radius, lambda = lambda)
 *   FD_KERNEL_GAUSSIAN_QNN  exp(-d2/R_j^2), R_j = min(q*nn_j, z*median_k(q*nn_k))
 *       (model=0 "QNN": q = qcoef, z = zcoef)      params {q, z [, lambda]}
 *       Built in ALGLIB's order, like FD_KERNEL_GAUSSIAN_ML below: the term's polynomial is a
 *       least-squares fit to the deltas, removed first; then (Phi + lambda I) w = f - P a (Phi is
 *       not symmetric with per-centre radii: pivoted LU).  The median is element M/2 of the
 *       sorted radii.
 *   FD_KERNEL_THIN_PLATE    r^2 ln r                params {[lambda]}
 *   FD_KERNEL_BIHARMONIC    -r                      params {[lambda]}
 *   FD_KERNEL_CUBIC         r^3                     params {[lambda]}
 * lambda is added to the diagonal of the kernel block (smoothing); default 0.
 *   FD_KERNEL_GAUSSIAN_ML   model=1 "Multilayer" as rbfsetalgomultilayer(model, radius, layers, lambda)
 *       (src/SOP_FaceDeform.cpp:346-348) lays it out, in dense form   params {R [, layers [, lambda]]}
 *       -- the term's polynomial is fitted to the deltas first, by least squares, and removed
 *       (ALGLIB's order; the kinds above solve it together with the weights); then layer
 *       l = 0 .. layers-1 fits what is left with exp(-d2/R_l^2), R_l = R / 2^l, on every centre:
 *       (Phi_l + lambda I) w_l = r_l,  r_{l+1} = r_l - Phi_l w_l.  The solved model is M * layers
 *       Gaussian records (fd_model_centres).  1 <= layers <= 8; defaults 4 and 0.1 as the SOP's.
 *       Not bit-parity with ALGLIB's truncated-Gaussian LSQR fit (absent here): parity with this
 *       dense statement, pinned by tests/golden/ml_golden.npz. */
enum {
    FD_KERNEL_GAUSSIAN = 0,
    FD_KERNEL_GAUSSIAN_QNN = 1,
    FD_KERNEL_THIN_PLATE = 2,
    FD_KERNEL_BIHARMONIC = 3,
    FD_KERNEL_CUBIC = 4,
    FD_KERNEL_GAUSSIAN_ML = 5
};

/* Same integers as ALGLIB_TERM_LINEAR/CONST/ZERO, src/SOP_FaceDeform.hpp:16-18. */
enum { FD_TERM_LINEAR = 0, FD_TERM_CONST = 1, FD_TERM_ZERO = 2 };

/* Evaluation arithmetic.  The solve is always fp64.
 * FP32, thin-plate with 49 or more centres (the default kernel of BASELINE's configurations,
 *   k_deform32_tps_mfma): the squared distances come from the matrix pipe as
 *   |x|^2 - 2 x.c + |c|^2 on operands split into two fp16 pieces (22 significant bits, about
 *   1.4e-6 absolute in normalised coordinates); logarithm, weights and accumulation are fp32,
 *   partial sums folded into a second fp32 level every <= 96 terms.  No fp64 anywhere.
 * FP32, every other kernel and small rigs (k_deform32): fp32 coordinate differences and kernel,
 *   partial sums folded into fp64 accumulators every 64 centres.
 * FP64: everything in fp64 (ill-conditioned weights). */
enum { FD_EVAL_FP32 = 0, FD_EVAL_FP64 = 1 };

/* Direct solver of the dense system.  AUTO: where the kernel is conditionally positive definite
 * of the order its polynomial term covers (thin-plate and cubic with the linear term, biharmonic
 * with a constant or linear term, the fixed-radius Gaussian with any term; lambda >= 0, M >= 16)
 * the polynomial constraints are eliminated with Householder reflectors and the projected kernel
 * block is Cholesky-factorised -- no pivot search; up to 256 control points that whole build is ONE
 * launch of one workgroup per model with the matrix in registers (FD_SOLVER_REGISTER names it; FD_SOLVER_CHAIN
 * keeps the launch chain at any size: same mathematics, weights equal to rounding, 1e-12);
 * everything else (QNN radii make Phi non-symmetric) goes through LU (the QNN model without the pivot
 * search where that is exact, see below).  LU forces the partially pivoted LU everywhere.  A
 * system on which the Cholesky loses definiteness to rounding (centres one fp32 step apart, a
 * fixed-radius Gaussian wider than the rig) is rebuilt with the LU before anything is reported,
 * and the context keeps the LU until its kernel, term or M change: nothing the LU accepts fails.
 * The rebuild happens where the status is first seen: in fd_build / fd_build_result /
 * fd_batch_build_result, or -- in a pipeline that enqueues builds and evaluations without collecting
 * results -- in the first fd_deform* / fd_batch_deform* call made after the build has executed
 * (every enqueued build posts its status to page-locked memory; every later call polls it without
 * waiting).  Evaluations enqueued BEFORE the status could be known pass the mesh through; from then
 * on the rig is evaluated correctly, or, if the LU fails too, fd_deform* returns FD_E_SINGULAR /
 * FD_E_DUPLICATE (sticky until the next fd_set_points / fd_set_deltas).  Both
 * are fp64 direct solves of the same system: their weights agree to rounding (~1e-12 relative
 * on the benchmark rigs), far inside the parity tolerance. */
/* ONE_WORKGROUP (thin-plate / Gaussian family as for AUTO's Cholesky path, up to 512 control points; anything else
 * behaves as AUTO): everything after the assembly of K -- projection, Cholesky, substitution, packing -- in ONE
 * launch of ONE workgroup per model.  A single build is slower that way (0.37 against 0.25 ms at 256 control
 * points), but a batch of 32 costs what one does, as 32 workgroups on 32 CUs and nothing else on the device: the
 * choice for a pipeline that keeps evaluating on the other CUs while the next frames' models are solved (bench.py).
 * Same arithmetic within this choice for single and batched builds (bit-identical weights); against AUTO the
 * weights agree to rounding (1e-12 relative). */
/* QNN model (FD_KERNEL_GAUSSIAN_QNN, the SOP's default: rbfsetalgoqnn, src/SOP_FaceDeform.cpp:342-345), up to 1024 control
 * points, under AUTO: the LU of its kernel block runs WITHOUT pivot search -- with q <= 1 partial pivoting never interchanges
 * (profiles/r02_qnn_pivot_stats.txt) and the factors are bit-identical to the pivoted ones.  A multiplier above 4 or a pivot
 * below the threshold ends that build with -4 and the pivoted LU repeats it, through the same path as above (q = 2 rigs).
 * FD_SOLVER_LU_NOPIVOT is a value of fd_report.solver_used only, not a choice. */
enum { FD_SOLVER_AUTO = 0, FD_SOLVER_LU = 1, FD_SOLVER_ONE_WORKGROUP = 2, FD_SOLVER_REGISTER = 3, FD_SOLVER_CHAIN = 4, FD_SOLVER_LU_NOPIVOT = 5 };

typedef struct fd_ctx fd_ctx;

typedef struct fd_config {
    int struct_size;     /* = sizeof(fd_config); 0 is accepted as "version 1"   */
    int device;          /* HIP device ordinal; -1 = the calling thread's current */
    int eval_precision;  /* FD_EVAL_*                                            */
    int eval_variant;    /* 0 = auto; otherwise a kernel variant id (tuning/tests) */
    int solver;          /* FD_SOLVER_*                                           */
    int reserved[3];
} fd_config;

/* Replaces alglib::rbfreport as read at src/SOP_FaceDeform.cpp:365-373. */
typedef struct fd_report {
    int terminationtype; /* 1 ok; -5 coincident centres; -4 solver failure       */
    int iterationscount; /* unknowns eliminated by the direct solver (n when done) */
    int n;               /* order of the solved system (M + term columns)        */
    int solver_used;     /* FD_SOLVER_* the build actually ran (FD_SOLVER_LU after a fallback; FD_SOLVER_LU_NOPIVOT: QNN) */
    double pivot_ratio;  /* min|pivot| / max|pivot| of the LU (cheap rcond proxy) */
    float t_assemble_ms; /* device time: prepare + kernel-matrix assembly        */
    float t_solve_ms;    /* device time: factorisation + substitution + pack     */
    /* What the fp32 evaluation (FD_EVAL_FP32) can be trusted with for THIS model.  It adds up M terms w_j phi(d_j) whose
     * magnitudes sum to S = sum_j |w_j| max phi over the rig's extent (+ the polynomial); each carries a relative 2^-24,
     * so a displacement comes out with an absolute error of up to 2^-24 S whatever its own size; the kernels measure at
     * 0.3 .. 0.4 of that and fp32_error = 2^-25 S is what is reported.  The
     * reference (fp64 inside ALGLIB, src/SOP_FaceDeform.cpp:404-439) has no such floor.  fd_fp32_holds() turns these into
     * the decision fdsop_cook takes; 0 in all four for imported models (no control table to measure against). */
    double fp32_error;   /* ~ absolute error of an fp32-evaluated displacement, in position units        */
    double cancellation; /* S / max_i |delta_i|: how much larger the summed terms are than what they add up to */
    double delta_min;    /* smallest |delta_i| among the control points that move (|delta_i| >= 0.1 delta_max): a
                          * stationary or barely moving control point -- most of a face rig in most frames, the fringe of a
                          * localised deformation -- does not count */
    double delta_max;    /* largest                                                                       */
    double extent;       /* largest |rest_i|: the size of the positions the displacement is added to     */
} fd_report;

/* 1 when the fp32 evaluation of the reported model is expected to hold `tol` (the reference's 1e-5) of every
 * vertex's own displacement: fp32_error <= tol * delta_min / 2 (vertices between the moving control points move less than the
 * least of those; around a stationary one the field is zero to within the next term in fp32 too) + one fp32 ulp of the positions (both sides round P + d to
 * fp32, :438, which shelters errors below that).  0: evaluate this model with FD_EVAL_FP64 (fd_set_eval_precision).
 * Conservative where the displacement field has zeros between control points -- no estimate from M points sees those. */
int fd_fp32_holds(const fd_report *report, double tol);

/* ---- lifetime ---------------------------------------------------------------
 * fd_create replaces `alglib::rbfmodel model; alglib::rbfcreate(3, 3, model)`
 * (src/SOP_FaceDeform.cpp:332,335).  NULL on failure (no device, out of
 * memory); fd_last_error(NULL) then holds the reason. */
fd_ctx *fd_create(const fd_config *cfg);
void fd_destroy(fd_ctx *ctx);
const char *fd_last_error(const fd_ctx *ctx);
int fd_abi_version(void);

/* Launch everything on `hip_stream` (a hipStream_t) instead of the context's
 * own stream.  NULL restores the context's stream. */
int fd_set_stream(fd_ctx *ctx, void *hip_stream);

/* FD_EVAL_FP32 / FD_EVAL_FP64 for the evaluations from here on (fd_config.eval_precision is the initial value).  A built
 * model carries the records of both: no rebuild.  fdsop_cook switches per cook on fd_fp32_holds(). */
int fd_set_eval_precision(fd_ctx *ctx, int eval_precision);
/* What the evaluation calls write into P_out.  FD_OUTPUT_POSITION (default): P + d f, the reference's write-back
 * (src/SOP_FaceDeform.cpp:438).  FD_OUTPUT_DISPLACEMENT: the addend alone, d f -- the fp32 displacement after the tangent projection
 * and the fall-off (:415-437), BEFORE it is added to the position; gated vertices and frames without a built model, which the
 * reference leaves where they are, get 0.  P_in is still the evaluation point.  For callers that keep a delta attribute, and for
 * parity tests that hold the displacement itself to 1e-5 without the rounding of P + d in the way.  Takes effect with the next
 * fd_deform* call; a batch takes the setting of its first context (all of them must agree). */
enum { FD_OUTPUT_POSITION = 0, FD_OUTPUT_DISPLACEMENT = 1 };
int fd_set_output(fd_ctx *ctx, int what);

/* ---- model set-up -----------------------------------------------------------
 * fd_set_points replaces alglib::rbfsetpoints(model, xy) with the M x 6 table
 * split into its two halves (src/SOP_FaceDeform.cpp:268-287,336): rest_xyz and
 * delta_xyz are AoS M x 3 fp32; delta = float(deformP - restP) computed by the
 * caller in fp32 as the reference does (:278). */
int fd_set_points(fd_ctx *ctx, const float *rest_xyz, const float *delta_xyz, int M);
int fd_set_points_dev(fd_ctx *ctx, const float *d_rest_xyz, const float *d_delta_xyz, int M);

/* Replaces rbfsetalgoqnn / rbfsetalgomultilayer (src/SOP_FaceDeform.cpp:342-349). */
/* New deltas for rest points that have already been factorised: the animated-rig case, where
 * the rest rig (input 1) stands still and only the deformed rig (input 2) moves.  The reference
 * rebuilds its model on every cook (src/SOP_FaceDeform.cpp:331-363); the system matrix depends on
 * the rest points, kernel and term only, so after fd_set_deltas the next fd_build* carries just
 * the new right-hand sides through the stored factorisation (same kernels and operand order as
 * a full build: the weights are bit-identical to fd_set_points + fd_build with the same data).
 * Exception: where FD_SOLVER_AUTO takes the register-resident one-launch build (thin-plate, cubic, biharmonic, fixed-radius
 * Gaussian with a term that makes them definite, up to 256 control points) no factorisation is stored -- the matrix never leaves
 * the registers -- and the next fd_build* simply builds again (0.18 ms at M = 256, less than the stored-factor
 * path's launch chain); the weights are the same bits either way.  FD_SOLVER_CHAIN keeps and reuses the factorisation.
 * FD_E_NOT_BUILT when there is no factorisation to reuse (no build yet, or fd_set_points /
 * fd_set_kernel / fd_set_term / fd_import_model since); M must match; order <= 2048. */
int fd_set_deltas(fd_ctx *ctx, const float *delta_xyz, int M);
int fd_set_deltas_dev(fd_ctx *ctx, const float *d_delta_xyz, int M);
int fd_set_kernel(fd_ctx *ctx, int kind, const double *params, int nparams);

/* Replaces rbfsetlinterm / rbfsetconstterm / rbfsetzeroterm (:351-361). */
int fd_set_term(fd_ctx *ctx, int term);

/* fd_build replaces alglib::rbfbuildmodel(model, report) (:363-368): kernel
 * matrix assembly + dense solve on the device; synchronises; fills *report.
 * Returns FD_OK iff report->terminationtype == 1.
 * fd_build_async enqueues the same work and returns; fd_build_result waits and
 * reports.  A deform enqueued after a failed build passes P through unchanged; once the failure is
 * known to the host (no wait: see FD_SOLVER_AUTO) fd_deform* repairs it or returns its error code. */
int fd_build(fd_ctx *ctx, fd_report *report);
int fd_build_async(fd_ctx *ctx);
int fd_build_result(fd_ctx *ctx, fd_report *report);

/* ---- evaluation -------------------------------------------------------------
 * fd_deform replaces the whole loop body src/SOP_FaceDeform.cpp:404-439
 * (gate :405-410, rbfcalc :411-415, project_to_tangents :416-422 with
 * src/SOP_FaceDeform.hpp:28-41, fall-off :423-425, write-back :437-438).
 *   P_in / P_out   N x 3 AoS fp32; P_out may alias P_in
 *   dist2          N fp32 squared capture distance, or NULL (= 0 everywhere)
 *   falloff_out    N fp32 (the fd_falloff attribute), or NULL; vertices skipped
 *                  by the gate are not written (the attribute keeps its default)
 *   tu, tv, nrm    N x 3 fp32 tangentu / tangentv / N, all three or all NULL
 *   radius2        radius*radius (:402);  falloffrate  the exponent (:424)
 * Synchronises before returning. */
int fd_deform(fd_ctx *ctx, int64_t N, const float *P_in, float *P_out, const float *dist2,
              float *falloff_out, const float *tu, const float *tv, const float *nrm,
              float radius2, float falloffrate);

/* Same, device pointers, asynchronous on the context's stream. */
int fd_deform_dev(fd_ctx *ctx, int64_t N, const float *d_P_in, float *d_P_out,
                  const float *d_dist2, float *d_falloff_out, const float *d_tu,
                  const float *d_tv, const float *d_nrm, float radius2, float falloffrate);

/* Same again, but launched on `hip_stream` (a hipStream_t) instead of the context's stream, so
 * that one stream can evaluate frame after frame while other streams build the next models.
 * Ordering is the caller's: make `hip_stream` wait for this context's build (an event recorded
 * on the context's stream after fd_build_async), and make the next fd_set_points / fd_build on
 * this context wait for the evaluation, which reads the model. */
int fd_deform_dev_stream(fd_ctx *ctx, void *hip_stream, int64_t N, const float *d_P_in, float *d_P_out,
                         const float *d_dist2, float *d_falloff_out, const float *d_tu,
                         const float *d_tv, const float *d_nrm, float radius2, float falloffrate);

/* ---- device-resident mesh (next row N3, engine side) --------------------------
 * In an animated shot the mesh on input 0 -- P, the capture's dist2, the tangent
 * frames -- is the same from cook to cook (Houdini tells by the attributes' data
 * IDs) while the rig moves.  fd_mesh_set uploads those arrays once (host
 * pointers; dist2 and the three frame arrays optional; synchronous);
 * fd_deform_mesh evaluates the current model on them and delivers P_out (and
 * falloff_out, may be NULL) into host arrays: written in place over the host link
 * when they are page-locked, through a device staging copy otherwise.  Same
 * results as fd_deform on the same arrays. */
int fd_mesh_set(fd_ctx *ctx, int64_t N, const float *P, const float *dist2, const float *tu, const float *tv,
                const float *nrm);
int64_t fd_mesh_size(const fd_ctx *ctx);
int fd_deform_mesh(fd_ctx *ctx, float *P_out, float *falloff_out, float radius2, float falloffrate);
/* ProximityCapture::init + capture (src/capture.cpp:10-99; called at src/SOP_FaceDeform.cpp:310-322)
 * on the device-resident mesh of fd_mesh_set: the island mask (fd_capture_islands: nearest mesh
 * point of each of the M rest-rig points, then max_edges edge rings over the CSR adjacency) and,
 * for the island points, the squared distance to the rig's surface (fd_capture_dist2: T triangles,
 * 9 floats each; -1 beyond radius2, 0 with dofalloff off and outside every island).  The result
 * becomes the mesh's dist2 array -- the detached attribute `dist_a` of capture.cpp:31 -- and stays
 * on the device until the next fd_mesh_set / fd_mesh_capture: fd_deform_mesh gates and falls off
 * with it.  Host pointers; synchronous.  dist2_out (may be NULL) receives a copy (npoints). */
int fd_mesh_capture(fd_ctx *ctx, const int64_t *offsets, const int *neighbours, int M, const float *rig_xyz,
                    int max_edges, int T, const float *tri_xyz, float radius2, int dofalloff, float *dist2_out);
/* The device-resident dist2 array (uploaded by fd_mesh_set or produced by fd_mesh_capture) into a
 * host array of fd_mesh_size() floats; FD_E_INVALID when the mesh has none. */
int fd_mesh_get_dist2(fd_ctx *ctx, float *dist2_out);

/* ---- model access -----------------------------------------------------------
 * W is (C+4) x 3 fp64 row-major: C RBF weights, the constant row, the x,y,z
 * linear rows (zero when the term lacks them).  radii (may be NULL) gets C
 * Gaussian radii.  C = fd_model_centres(ctx): the control points M, or M * layers
 * for FD_KERNEL_GAUSSIAN_ML (layer-major: record l*M + j is centre j in layer l).
 * For tests and for RCCL-free replication. */
int fd_model_centres(const fd_ctx *ctx);
int fd_get_weights(fd_ctx *ctx, double *W, double *radii);

/* Solved model as one relocatable blob (header + centres + radii + weights):
 * what a vertex-range split broadcasts from the solving GPU (RCCL over xGMI,
 * or any other transport).  on_device != 0: buf is a device pointer and the
 * copy is enqueued on the context's stream.  The header carries the identity of the rest rig the
 * model was built on (the array fd_batch_set_points_dev read it from): contexts that imported
 * models with the same identity form a batch fd_batch_deform_shared_dev accepts -- the frames of a
 * shot, solved on one GPU, evaluated on every GPU's vertex range with one launch per group. */
size_t fd_model_bytes(const fd_ctx *ctx);
int fd_export_model(fd_ctx *ctx, void *buf, size_t capacity, int on_device);
int fd_import_model(fd_ctx *ctx, const void *buf, size_t bytes, int on_device);

/* Block until everything enqueued on the context's stream has finished. */
int fd_synchronize(fd_ctx *ctx);

/* Page-locked host memory for the arrays handed to fd_deform (replaces nothing
 * in the reference: GA pages are ordinary memory; the wrapper gathers them
 * into a flat array anyway, hdk/SOP_FaceDeformHip.cpp).  When every array of a
 * fd_deform call is page-locked the kernel reads and writes them in place over
 * the host link (no staging copies, traffic in both directions at once);
 * pageable arrays are uploaded, evaluated and downloaded one after the other. */
void *fd_host_alloc(size_t bytes);
void fd_host_free(void *p);

/* ---- batched build ------------------------------------------------------------
 * The reference cooks one node at a time: one rbfbuildmodel per cookMySop
 * (src/SOP_FaceDeform.cpp:363).  A dense system of order 260 keeps one CU of a
 * 256-CU device busy, and the device overlaps only two or three such launch
 * chains.  Contexts that share M, kernel, parameters and term -- the frames of
 * one rig, or several facedeform nodes of one cook graph -- can therefore be
 * assembled and factorised together: one launch chain, one workgroup column per
 * context.  Each context ends up as after its own fd_build_async and is
 * evaluated with the usual fd_deform* calls; those wait for the batch where
 * they run on another stream.  Same kernels and pivots; up to order 512 the
 * results are bit-identical to single builds, above that a batch of four or
 * more groups its panels under deeper trailing updates (they are HBM-bound
 * there) and the weights agree to rounding.  Destroy the batch before its
 * contexts. */
typedef struct fd_batch fd_batch;
#define FD_MAX_BATCH 32
fd_batch *fd_batch_create(fd_ctx *const *ctxs, int n);          /* 1..FD_MAX_BATCH contexts of one device */
void fd_batch_destroy(fd_batch *batch);
int fd_batch_size(const fd_batch *batch);
const char *fd_batch_last_error(const fd_batch *batch);
/* Control points of all contexts from caller-owned DEVICE arrays, one pointer
 * pair per context (host arrays of n device pointers).  No copy is enqueued:
 * the next fd_batch_build_async reads them in place, so they must stay valid
 * and unchanged until that build has executed.  Optional -- contexts whose
 * points were set with fd_set_points(_dev) are built from their own copies. */
int fd_batch_set_points_dev(fd_batch *batch, const float *const *d_rest_xyz,
                            const float *const *d_delta_xyz, int M);
/* Enqueue the build of every context on hip_stream (NULL: the stream of
 * context 0).  FD_E_INVALID if the contexts differ in M, kernel or term. */
int fd_batch_build_async(fd_batch *batch, void *hip_stream);
/* Wait and report per context (reports may be NULL, else n entries).  Returns
 * the first non-zero per-context code. */
int fd_batch_build_result(fd_batch *batch, fd_report *reports);
/* Evaluate every context of the batch on its own vertex arrays with ONE launch
 * (tables of n device pointers, host arrays; the optional tables may be NULL,
 * and so may their entries except P).  A launch over 1M vertices spends ~12 %
 * of its time ramping up and draining; a launch over all frames of a group
 * does not.  Falls back to one launch per context when they do not all take
 * the default thin-plate kernel on equally sized inputs.  Per context the
 * result is exactly that of fd_deform_dev_stream. */
int fd_batch_deform_dev(fd_batch *batch, void *hip_stream, int64_t N, const float *const *d_P_in,
                        float *const *d_P_out, const float *const *d_dist2, float *const *d_falloff_out,
                        const float *const *d_tu, const float *const *d_tv, const float *const *d_nrm,
                        float radius2, float falloffrate);

/* The same for frames that share the MESH and the REST RIG -- the frames of an animated shot, the
 * blendshapes of one head (BASELINE configs 2-4 as SURVEY.md 8d lays them out: "same mesh and rest
 * rig, deltas phase-shifted"): one input mesh (d_P_in, d_dist2, frames: single arrays), one output
 * pair per context.  phi(|x - c_j|^2) depends on the vertex and the centre only, so it is formed
 * once for all frames, and the contraction with the 3 F columns of weights is a dense
 * (N x M) x (M x 3F) product on the matrix pipe (fp16 x 2 split operands, 22 bits, fp32
 * accumulation).  Every frame still has its own model, built by its own assemble + solve.  The
 * contexts must have read their rest points from ONE device array (fd_batch_set_points_dev with
 * the same d_rest_xyz for all; FD_E_INVALID otherwise); thin-plate or Gaussian kernel
 * (FD_KERNEL_GAUSSIAN, FD_KERNEL_GAUSSIAN_QNN -- the SOP's default model: exp(-d2 / R_j^2) from direct
 * coordinate differences, formed once for all frames), fp32 evaluation, 32 or more centres -- anything else
 * (biharmonic, cubic, the multilayer model, fp64) takes fd_batch_deform_dev on the shared arrays.  Parity
 * with the oracle as for fd_deform (1e-5); NOT bit-identical to the one-frame kernels.
 * Fastest form: 17..32 frames (32-row output tiles, rows packed three per frame: 20 frames cost two tiles, not three), no d_dist2, no tangent frames, every d_falloff_out
 * given and 16-byte aligned (results are the same without, through a slower epilogue).  Outputs are written with
 * the non-temporal hint: they are not expected in L2 by whatever runs next. */
int fd_batch_deform_shared_dev(fd_batch *batch, void *hip_stream, int64_t N, const float *d_P_in,
                               float *const *d_P_out, const float *d_dist2, float *const *d_falloff_out,
                               const float *d_tu, const float *d_tv, const float *d_nrm, float radius2,
                               float falloffrate);
/* Makes hip_stream (NULL: context 0's) wait until the batch's last fd_batch_deform_shared_dev no longer reads the
 * contexts' models: that launch copies what it needs of them (weights as fp16 tiles, the rest rig's centre tiles)
 * into the batch's own scratch with a first small kernel, and the evaluation proper reads only that copy.  A pipeline
 * that cooks group after group on the same contexts puts this in front of the next fd_batch_set_points_dev /
 * fd_batch_build_async instead of waiting for the evaluation itself: the next models are assembled and solved while
 * the current ones are still being evaluated.  (The output arrays ARE still being written: they stay the caller's to
 * order.)  No-op when no shared-rig evaluation has been enqueued on the batch. */
int fd_batch_wait_consumed(fd_batch *batch, void *hip_stream);
/* The first, small part of fd_batch_deform_shared_dev on its own, on a stream of the caller's choice (NULL: context
 * 0's) -- typically the build stream, right behind fd_batch_build_async: the models' weights become fp16 tiles in
 * the batch's scratch (two sets, used in turn) together with the frames' output addresses.  A following
 * fd_batch_deform_shared_dev with the SAME output tables then launches the evaluation alone (its stream waits for
 * this kernel), so an evaluation stream runs evaluations back to back while packing happens beside the builds.
 * Invalidated by the next fd_batch_set_points_dev / fd_batch_build_async (and by a model the library had to
 * rebuild); without it, or with other outputs, fd_batch_deform_shared_dev packs by itself as before.  A no-op when
 * the shared-rig launch does not apply to the batch (other kernels). */
int fd_batch_prepare_shared(fd_batch *batch, void *hip_stream, float *const *d_P_out, float *const *d_falloff_out);
/* CU budget of this batch's shared-rig evaluation launches: fd_batch_deform_shared_dev runs one persistent workgroup per
 * CU (it needs a CU's whole LDS), and a pipeline that builds the next group's models beside the evaluation may want to
 * leave some CUs to those builds.  n_cus <= 0: one workgroup per CU (the default).  More than the device's CU count
 * oversubscribes: shares get shorter and the workgroups beyond the resident ones start as CUs come free (measured: no gain over
 * 224 of 256 beside three batches of builds).  Per batch, not
 * per process: two nodes cooking side by side choose independently (round 2 read an environment variable once).
 * The budget also tells the batch's BUILDS where they run (register-resident build, up to 256 control points): with CUs left
 * to them (n_cus below the device's count) each model is built by one workgroup from start to end and stays on those CUs; with
 * none left (the default: nothing evaluates beside the builds) the assembly of K, Y = K V and the projection run as two short
 * launches over all CUs before the factorisation's workgroup -- the same model to rounding, 0.04 ms sooner. */
int fd_batch_set_eval_cus(fd_batch *batch, int n_cus);
/* One factorisation per batched build where the contexts share the rest rig ("factor once and treat frames as extra
 * right-hand sides").  The reference rebuilds its model on every cook (src/SOP_FaceDeform.cpp:331-363); the system matrix depends
 * on the rest points, the kernel and the term only, so the frames of a shot -- one rest rig, other deltas -- can share it.
 * With `on`, fd_batch_build_async (and fd_batch_cook_group) assembles, projects and factorises ONCE, in the register-resident
 * one-launch build of the batch's first context, and carries every other context's right-hand sides through that factor, one
 * workgroup per context (Q^T f, both substitutions, the polynomial, packing).  It applies when the register-resident build does
 * (a definite kernel + term pair, up to 256 control points) AND every context was given the SAME rest array by
 * fd_batch_set_points_dev; otherwise -- other rigs, other kernels, larger rigs -- the call builds every model on its own, as
 * without the switch.  Weights equal the per-context builds' to rounding (1e-12 of max |w|), not bit for bit.  Off by default.
 * fd_batch_last_build_shared_factor: 1 if the batch's last build took the shared path. */
int fd_batch_set_shared_factor(fd_batch *batch, int on);
int fd_batch_last_build_shared_factor(const fd_batch *batch);
/* Which kernel fd_batch_deform_shared_dev launches for `frames` frames of an M-centre model of `kind` (a name for profiles and
 * benchmark lines -- the one rocprofv3 prints): "k_deform32_shared_w1" (17..32 frames, the model resident in LDS),
 * "k_deform32_tps_shared_wide" (17..32 frames, staged in chunks), "k_deform32_tps_shared" (up to 16 frames), or "" where the
 * shared-rig launch does not apply (other kernels, fewer than 32 centres: the per-frame launch runs).  No reference counterpart:
 * the reference has one loop (src/SOP_FaceDeform.cpp:404-439). */
const char *fd_shared_kernel_name(int M, int frames, int kind);
/* One GROUP of frames of a shot in one call -- what a frame pipeline enqueues per group, in the order it must be enqueued:
 *   fd_batch_wait_consumed(batch, build_stream)            the batch's previous evaluation has its own copy of the models
 *   fd_batch_set_points_dev(batch, d_rest, d_delta, M)      (the SAME rest array for every context: frames of one rig)
 *   fd_batch_build_async(batch, build_stream)               every frame's model: assembled, factorised, solved
 *   fd_batch_prepare_shared(batch, build_stream, ...)       weights -> fp16 tiles, beside the builds
 *   eval_stream waits for build_stream;  fd_batch_deform_shared_dev(batch, eval_stream, ...)
 * build_stream / eval_stream: the caller's HIP streams (eval_stream NULL: the evaluation goes on build_stream).  d_rest_xyz:
 * the one rest rig; d_delta_xyz: n pointers, one per context.  No d_dist2 / tangent frames: a group that needs them
 * makes the calls above itself.  Replaces, for n frames, n cooks of reference src/SOP_FaceDeform.cpp:268-287, 331-368, 404-439.
 * A host language pays for ONE foreign call per group instead of five with pointer tables (bench.py reports both).
 * events (may be NULL): four caller-owned hipEvent_t handles recorded around the builds (on build_stream) and around the
 * evaluation launch (on eval_stream), for callers that time the pieces; NULL members are skipped.
 * With the evaluation on build_stream itself (eval_stream NULL or equal: one group, nothing to overlap) the call records NO event
 * of its own in front of the builds, between them and the packing or before the evaluation (the build's reports then carry no
 * phase times) -- stream order does what they do across streams, and every
 * record is a barrier packet the queue idles ~4 us for.  The batch's "build done" event is then recorded BEHIND the evaluation:
 * fd_batch_build_result and any other stream ordered after the builds wait a little longer than needed, never too little, and a
 * failed build's status arrives with the NEXT call on the batch (the evaluation of an unbuilt model passes its frame through,
 * as documented for asynchronous builds above). */
typedef struct fd_group_events {
    void *before_build, *after_build;      /* hipEvent_t, recorded on build_stream */
    void *before_eval, *after_eval;        /* hipEvent_t, recorded on eval_stream, around the evaluation launch alone */
} fd_group_events;
int fd_batch_cook_group(fd_batch *batch, void *build_stream, void *eval_stream, const float *d_rest_xyz,
                        const float *const *d_delta_xyz, int M, int64_t N, const float *d_P_in, float *const *d_P_out,
                        float *const *d_falloff_out, const fd_group_events *events);

/* ---- dist2 producer (next row N2) ---------------------------------------------
 * The per-point body of ProximityCapture::capture (src/capture.cpp:58-97) on the
 * device: for every mesh point of an island (mask[i] != 0; mask NULL = all points)
 * the squared distance to the closest point of the rest rig's surface -- what
 * GU_RayIntersect::minimumPoint returns there -- when it is below radius2, else
 * -1 (:76-88); 0 when dofalloff is off (:71-75) and for points outside every
 * island (the detached attribute's default, :31).  The rig surface is T
 * triangles, 9 floats each (a, b, c).  The result is fd_deform's dist2 input;
 * with the _dev form it never leaves the device.  Island finding (nearest mesh
 * point of every rig point + edge rings, :101-141) needs the mesh topology and
 * stays with the caller. */
int fd_capture_dist2_dev(fd_ctx *ctx, int64_t N, const float *d_P, const unsigned char *d_mask, int T,
                         const float *d_tri_xyz, float radius2, int dofalloff, float *d_dist2);
int fd_capture_dist2(fd_ctx *ctx, int64_t N, const float *P, const unsigned char *mask, int T,
                     const float *tri_xyz, float radius2, int dofalloff, float *dist2);
/* The island mask itself: ProximityCapture::findIslands (src/capture.cpp:101-141).
 * For every rig point the nearest mesh point (GEO_PointTree::findNearestIdx;
 * ties to the lower index), then every mesh point within max_edges edges of it
 * (GQ_Detail::groupEdgePoints, the start point included).  The handle classes
 * only partition the islands into groups capture() treats alike, so the product
 * is their union, mask[i] in {0, 1}.  The mesh's edges come as a CSR adjacency
 * (offsets[N + 1], neighbours[offsets[N]]), which a static mesh needs building
 * once.  max_edges is capped at 250. */
int fd_capture_islands_dev(fd_ctx *ctx, int64_t N, const float *d_P, const int64_t *d_offsets,
                           const int *d_neighbours, int M, const float *d_rig_xyz, int max_edges,
                           unsigned char *d_mask);
int fd_capture_islands(fd_ctx *ctx, int64_t N, const float *P, const int64_t *offsets, const int *neighbours,
                       int M, const float *rig_xyz, int max_edges, unsigned char *mask);

/* ---- morph-space reprojection (next row N1) ---------------------------------
 * Replaces DirectBSEdit (src/dbse.hpp:7-33, src/dbse.cpp:9-87) and the loop
 * that applies it after the RBF pass (src/SOP_FaceDeform.cpp:444-473).  The
 * 3N x S matrix of blendshape deltas lives on the device; the per-cook passes
 * stream over it once each (HBM-bound).
 *   fd_morph_init            DirectBSEdit::init (dbse.cpp:9-37): deltas
 *                            float(shape - rest) widened to fp64, Householder QR
 *                            in Eigen's packed form.  S <= 3N.  Synchronous.
 *   fd_morph_compute_weights_dev  computeWeights (dbse.cpp:39-60):
 *                            w_s = sum_i float(P_i - rest_i) * QRpacked[i][s]
 *   fd_morph_displace_dev    displaceVector + the loop at SOP :458-473:
 *                            P = rest + sum_s delta_s * clamp(float(3 w_s))
 *                                [+ (P - rest) * falloffradius if add_delta]
 *                            clamp_lo_hi: host pointer to {lo, hi} or NULL.
 *   fd_morph_apply           both, on a host array in place; w_out (S) may be NULL.
 * hip_stream NULL = the object's own stream.  is_initialised / is_computed
 * mirror DirectBSEdit::isInitialized / isComputed, which the cook logic tests. */
typedef struct fd_morph fd_morph;
fd_morph *fd_morph_create(const fd_config *cfg);
void fd_morph_destroy(fd_morph *m);
const char *fd_morph_last_error(const fd_morph *m);
int fd_morph_init(fd_morph *m, int64_t N, int S, const float *rest_xyz, const float *const *shapes_xyz);
int fd_morph_init_dev(fd_morph *m, int64_t N, int S, const float *d_rest_xyz, const float *const *d_shapes_xyz);
/* The `rest` point attribute the per-cook passes measure against
 * (SOP_FaceDeform.cpp:178-184, 445-447).  Default (or NULL): the rest pose
 * given to fd_morph_init -- what setupBlends stores when input 0 has no rest
 * attribute of its own.  Reset by fd_morph_init. */
int fd_morph_set_rest(fd_morph *m, const float *rest_xyz, int on_device);
int fd_morph_is_initialised(const fd_morph *m);
int fd_morph_is_computed(const fd_morph *m);
int fd_morph_shape_count(const fd_morph *m);
float fd_morph_last_init_ms(const fd_morph *m);
int fd_morph_compute_weights_dev(fd_morph *m, const float *d_P_xyz, void *hip_stream);
int fd_morph_displace_dev(fd_morph *m, float *d_P_xyz, const float *clamp_lo_hi, int add_delta,
                          float falloffradius, void *hip_stream);
int fd_morph_apply(fd_morph *m, float *P_xyz, const float *clamp_lo_hi, int add_delta, float falloffradius,
                   double *w_out);
int fd_morph_get_weights(fd_morph *m, double *w);
int fd_morph_get_qr(fd_morph *m, double *qr, double *tau);   /* 3N x S column-major packed QR (tests) */

/* ---- host-side cook: the HDK-free mirror of cookMySop ----------------------
 * fdsop_* mirrors the SOP's parm surface (src/SOP_FaceDeform.cpp:99-137) and
 * the cook sequence (:215-489) over plain arrays standing in for GU_Detail:
 * same parm tokens, defaults and clamps, same error / warning texts.  The HDK
 * wrapper (hdk/SOP_FaceDeformHip.cpp) and the tests drive this. */
typedef struct fdsop_node fdsop_node;

/* Severity of the worst message of the last cook, as OP_ERROR orders them. */
enum { FDSOP_OK = 0, FDSOP_MESSAGE = 1, FDSOP_WARNING = 2, FDSOP_ERROR = 3 };

typedef struct fdsop_geo {
    /* input 0: the mesh (cooked in place into the output, duplicatePointSource :226) */
    int64_t npoints;
    const float *P;        /* npoints x 3 */
    const float *tangentu; /* npoints x 3 or NULL */
    const float *tangentv;
    const float *N;
    const float *dist2;    /* ProximityCapture's detached attribute (capture.cpp:31) or NULL */
    /* inputs 1 and 2: rest and deformed rig */
    int64_t rest_npoints, deform_npoints;
    const float *rest_P;
    const float *deform_P;
    /* outputs */
    float *P_out;          /* npoints x 3 */
    float *fd_falloff;     /* npoints, or NULL */
    float *Cd;             /* npoints x 3, or NULL (always white, :386-388) */
    /* inputs 3..: blendshapes of the morph-space pass (:175-213, 444-482); all optional */
    int64_t nshapes;                 /* connected inputs beyond the deformed rig */
    const float *const *shapes_P;    /* nshapes arrays, shapes_npoints[s] x 3 each */
    const int64_t *shapes_npoints;   /* a count different from npoints drops the shape with a warning (:200-204) */
    const float *rest;               /* input 0's `rest` point attribute (npoints x 3) or NULL (:178) */
    int rest_changed;                /* what checkChangedSourceFlags(0) reports: the rest pose changed (:180-184) */
    int blends_changed;              /* ... and for any of inputs 3.. (:186-194) */
    double *weights;                 /* out: the detail array `weights` (:474-481), room for nshapes, or NULL */
    int64_t *weights_count;          /* out: entries written; 0 when the morph pass did not run; or NULL */
    /* The rest rig (input 1) is the one of the previous cook -- what checkChangedSourceFlags(1)
     * tells the wrapper.  Then only the deltas are new and the engine reuses its factorisation
     * (fd_set_deltas); 0 = rebuild, as the reference does every cook (B12). */
    int rig_rest_unchanged;
    /* The arrays of input 0 (P, dist2, tangent frames) are the ones of the previous cook -- their
     * data IDs did not change.  Then the engine's device-resident copy (fd_mesh_set) is used and
     * nothing of the mesh is uploaded; 0 = upload. */
    int mesh_unchanged;
    /* ---- ProximityCapture's inputs (src/capture.cpp:10-44, 101-141; cook :301-322) -- all optional.
     * When `dist2` above is NULL and these are present, the cook captures on the device exactly
     * where the reference does: on the first cook and whenever the rest pose (input 0) or the rest
     * rig (input 1) changed -- NOT when only radius / maxedges / dofalloff changed (the author's
     * FIXME at :309, kept) -- and gates / falls off with the result.
     *   edge_offsets / edge_neighbours   the mesh's edges as a CSR adjacency (what GQ_Detail::
     *                                    groupEdgePoints walks): offsets[npoints + 1], neighbours[offsets[npoints]]
     *   rig_tris                         the rest rig's surface as rig_ntris triangles, 9 floats each
     *                                    (what GU_RayIntersect::minimumPoint searches)
     *   dist2_out                        out: the captured attribute, npoints floats, or NULL */
    const int64_t *edge_offsets;
    const int *edge_neighbours;
    int64_t rig_ntris;
    const float *rig_tris;
    float *dist2_out;
} fdsop_geo;

fdsop_node *fdsop_create(const fd_config *cfg);
void fdsop_destroy(fdsop_node *node);
/* Parm access by token ("radius", "model", ...).  Unknown token: FD_E_INVALID. */
int fdsop_set_float(fdsop_node *node, const char *token, int index, double value);
int fdsop_set_int(fdsop_node *node, const char *token, int value);
int fdsop_set_string(fdsop_node *node, const char *token, const char *value);
int fdsop_get_float(const fdsop_node *node, const char *token, int index, double *value);
int fdsop_get_int(const fdsop_node *node, const char *token, int *value);
int fdsop_parm_count(void);
const char *fdsop_parm_token(int i);
/* Cook.  Returns the FDSOP_* severity.  Messages: "severity\ttext\n" lines. */
int fdsop_cook(fdsop_node *node, const fdsop_geo *geo);
const char *fdsop_messages(const fdsop_node *node);
/* The clamped values the last cook actually used (A1), and the engine underneath. */
int fdsop_effective_float(const fdsop_node *node, const char *token, double *value);
fd_ctx *fdsop_engine(fdsop_node *node);

#ifdef __cplusplus
}
#endif
#endif /* FACEDEFORM_HIP_H */
