// fd_internal.h -- shared declarations of the gfx950 RBF deformation engine.
// Not part of the ABI (that is include/facedeform_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/facedeform_hip.h"

#include "fd_tuning.h"

namespace fd {

// A pointer fetched from a table in memory is a generic pointer to the compiler: every access
// through it becomes a flat_load / flat_store, which counts on BOTH wait counters and so
// serialises against LDS traffic.  The build kernels take their buffers from the batch table, so
// they say explicitly that these live in global memory.
#define FD_GLOBAL __attribute__((address_space(1)))
typedef double FD_GLOBAL gdouble;
typedef const double FD_GLOBAL gcdouble;
template <typename T> __host__ __device__ __forceinline__ T FD_GLOBAL *as_global(T *p) { return (T FD_GLOBAL *)p; }

// ---- solved model as the evaluation kernels read it --------------------------
// One record per centre, 32 B: the unit a scalar s_load_dwordx8 fetches.
//   s  = per-centre kernel scale (Gaussian: -log2(e)/R_j^2, otherwise 0)
//   w* = RBF weight pre-multiplied by the kernel's constant factor
//        (thin-plate: 0.5*ln2, so phi' = d2*log2(d2); biharmonic: -1, phi' = sqrt(d2))
struct Rec32 {
    float cx, cy, cz, s;
    float wx, wy, wz, pad;
};
struct Rec64 {
    double cx, cy, cz, s;
    double wx, wy, wz, pad;
};

// Centres are padded with zero-weight records to a multiple of this.
constexpr int kRecPad = 16;   // also the centre-tile height of the matrix-pipe evaluation

// Centre tile of the thin-plate matrix-pipe evaluation (16 centres): the A operand of one
// v_mfma_f32_16x16x32_bf16 as every lane reads it, and the weights of the tile arranged per
// 16-lane group the way the accumulate consumes them.  1280 B, 16-byte aligned.
struct MfmaTile {
    unsigned int a[64][4];   // lane l: 8 bf16 = k-slots 8*(l>>4) .. +7 of centre (l&15)
    float w[4][12];          // group g: {c0:(r0,r1) c1:(r0,r1) c2:(r0,r1) c0:(r2,r3) c1:(r2,r3) c2:(r2,r3)}, centre = 4g + r
    float pad[16];
};
static_assert(sizeof(MfmaTile) == 1280, "tile image must stay 1280 bytes");

// The same tile for v_mfma_f32_16x16x16_f16: every fp32 value as TWO fp16 pieces (hi + lo, 22
// bits; fp16 subnormals are honoured by the matrix pipe, tools/mfma_f16_subnormal_test.hip), all
// four cross products per coordinate in K = 16.  Half the operand bytes and half the MFMA time
// of the bf16 tile, at ~2x its d2 rounding error.  768 B.
struct MfmaTileH {
    unsigned int a[64][2];   // lane l: 4 fp16 = k-slots 4*(l>>4) .. +3 of centre (l&15)
    float w[4][12];          // as MfmaTile::w
    float pad[16];
};
static_assert(sizeof(MfmaTileH) == 768, "fp16 tile image must stay 768 bytes");

// Device-resident build status + affine part; read by the deform kernel so that
// an asynchronous build needs no host round trip before the deform launch.
struct DevModel {
    int terminationtype;   // 1 ok, -4, -5; 0 = not built
    int dup_flag;          // set by the assembly kernel: coincident centres
    int sing_flag;         // set by the panel kernel: pivot below threshold
    int iterations;        // elimination steps done
    unsigned long long amax_bits;  // max |A_ij| as raw double bits (atomicMax)
    unsigned long long pivmin_bits;
    unsigned long long pivmax_bits;
    // fp64 evaluation, raw coordinates: affine64[c*4 + 0] = constant of output c,
    // affine64[c*4 + 1..3] = x,y,z coefficients
    double affine64[12];
    // fp32 evaluation works in normalised coordinates x' = (x - x0) * inv_s, inv_s a power of
    // two (exact): norm32 = {x0.x, x0.y, x0.z, inv_s}.  poly32[c*5 + 0..4] = {C0, Lx, Ly, Lz, q}
    // of output c: affine part re-expressed in x' plus, for thin-plate, the exact correction
    // kappa * sum_j w_j |x' - c'_j|^2 that the change of length unit brings (kappa = s^2 ln s).
    float norm32[4];
    float poly32[15];
    float wmax32;          // largest |weight| or |polynomial coefficient| as the fp32 records carry them: the shared-rig launch scales by it
    // what the packing code estimates about the fp32 evaluation of this model (fd_pack.h; fd_report carries them)
    double fp32_error, cancellation, delta_min, delta_max, extent;
};

// Blob header for fd_export_model / fd_import_model.
struct ModelHeader {
    uint32_t magic;        // 'FDM2'
    int32_t M, kind, term, nparams;
    int32_t terminationtype;
    int32_t layers;        // multilayer Gaussian model: records per centre (the blob's arrays are layer-major); 0 otherwise
    int32_t reserved;
    double params[4];
    // Identity of the rest rig the model was built on: the exporter's in-place rest array (fd_batch_set_points_dev), 0 when
    // it had none.  Models imported with the same non-zero token came from ONE rest rig -- frames of a shot that one rank
    // solved and the others received -- which is what fd_batch_deform_shared_dev needs to know about them.
    uint64_t rig_token;
};
constexpr uint32_t kModelMagic = 0x324D4446u;

static inline int term_cols(int term) { return term == FD_TERM_LINEAR ? 4 : (term == FD_TERM_CONST ? 1 : 0); }
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ---- build pipeline (fd_build.hip) -------------------------------------------
constexpr int kPanelThreads = 1024;   // one workgroup factors a panel
constexpr int kMaxOrder = 5632;       // back-substitution keeps Y (3 x order fp64) in LDS
constexpr int kRhsCols = 16;          // 3 right-hand sides padded to one MFMA tile
constexpr int kMovesStride = 132;     // ints per elimination step's move list: count + 2 * (2 * 32) pairs
static inline int lu_step_capacity(int npad) { return npad / 4 + 2; }   // steps a system of this order can take

constexpr int kMaxBatch = 32;         // models one launch chain may build together

// Device-side table entry: the buffers of one model.  Every build kernel indexes the table
// with blockIdx.z, so a batch of contexts is assembled and factorised by one launch chain.
struct BatchSlot {
    float *rest, *delta;              // M x 3, the context's own copy of the control points
    double *centres;                  // M x 3
    double *radii;                    // M
    double *A;                        // lda x (ncols + 16): 16 zero columns absorb block overrun
    double *X;                        // 3 x npad solution, column per right-hand side
    double *W;                        // (M+4) x 3
    int *ipiv;                        // npad
    int *moves;                       // [0] = count, then (dst, src) pairs, per panel
    Rec32 *rec32;
    Rec64 *rec64;
    MfmaTile *tiles;                  // Mpad / 16 tiles (thin-plate only)
    MfmaTileH *tiles16;               // the fp16 form of the same tiles
    DevModel *model;
    double *ns;                       // null-space solver state (fd_nullspace.hip): ns_doubles(M) doubles
    int *host_status;                 // the context's page-locked status word as the device sees it (or null): the packing code
                                      // posts terminationtype there itself, so a one-launch build needs no status kernel behind it
};

// compute units of the current device (hipDeviceProp_t::multiProcessorCount: 256 on MI355X), looked up once per device
inline unsigned device_cus()
{
    static unsigned cached[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256u; }
    if (dev >= 0 && dev < 64 && cached[dev]) return cached[dev];
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
    if (dev >= 0 && dev < 64) cached[dev] = (unsigned)n;
    return (unsigned)n;
}

// Caller-owned device arrays of control points, one pair per model of a batch (kernel argument).
struct PointSrc {
    const float *rest[kMaxBatch];
    const float *delta[kMaxBatch];
};

// Dimensions and parameters shared by all models of a launch chain + where their slots are.
struct BuildBuffers {
    int M, T, n, npad, lda, ncols;    // A is lda x ncols column-major; cols npad.. hold the RHS
    int kind, term;
    double lambda;
    double gauss_R, qnn_q, qnn_z;
    int Mpad;
    const BatchSlot *d_slots;         // device table, nbatch entries
    int nbatch;
    // Panels grouped under one deep trailing update (fd_build.hip group_at): pays when the
    // trailing updates are HBM-bound, i.e. for batches; a lone system is launch-bound and stays
    // ungrouped.  The factorisation's layout depends on it, so fd_set_deltas must replay it.
    int group_panels;
    // Symmetric definite path (fd_nullspace.hip): the polynomial constraints are eliminated with
    // Householder reflectors and the projected kernel block, order M - T, is Cholesky-factorised
    // -- no pivot search, so the panel is no longer one workgroup's serial chain.
    int small;                        // FD_SOLVER_ONE_WORKGROUP: k_build_small where it applies
    int reg;                          // the register-resident one-launch build (fd_build_reg.hip) where it applies: M <= 256 on the definite path
    // ... with its parallel front end (k_reg_front1 / k_reg_front2 over all CUs, then the factorisation in one workgroup per model):
    // shorter alone; 0 keeps the whole build in its one workgroup -- what a pipeline wants whose evaluation launches hold most CUs
    // (fd_batch_set_eval_cus below the device's count: the builds then stay on the CUs left to them)
    int reg_front;
    int spd;
    // QNN model, order <= 1024: the LU of its kernel block WITHOUT pivot search (fd_build.hip k_lu_panel_np); a multiplier
    // above kMaxMultiplier or a pivot below the threshold ends the build with -4 and the host repeats it with the pivoted LU.
    int nopivot;
    // Multilayer Gaussian model (FD_KERNEL_GAUSSIAN_ML, fd_nullspace.hip launch_build_ml): number of
    // layers, 0 for every other kind.  `kind` is then FD_KERNEL_GAUSSIAN (what the assembly evaluates).
    int ml_layers;
    // LU look-ahead: second stream + {panel done, rest done} x 2 events; aux_stream == nullptr
    // runs every step on the one stream
    hipStream_t aux_stream;
    hipEvent_t aux_events[4];
};

hipError_t launch_prepare(const BuildBuffers &b, hipStream_t stream, const PointSrc *src);
hipError_t launch_build(const BuildBuffers &b, hipStream_t stream, hipEvent_t ev_mid);
hipError_t launch_resolve(const BuildBuffers &b, hipStream_t stream, const PointSrc *src);
hipError_t launch_pack(const BuildBuffers &b, hipStream_t stream);
hipError_t launch_pack_from_weights(const BuildBuffers &b, hipStream_t stream, int layers = 0);
// pieces of the LU pipeline the null-space path reuses: the kernel block alone (order M, identity
// padding to npad_a) and back-substitution with the upper triangle over rows [0, rows)
hipError_t launch_assemble_block(const BuildBuffers &b, hipStream_t stream, int npad_a, int radii_off = 0);
// k_pack over `records` centres whose weights are already in W (mode 1: imported model, status from
// the values alone; mode 2: built here, the factorisation's flags count too)
hipError_t launch_pack_records(const BuildBuffers &b, hipStream_t stream, int records, int kind, int mode, int layers);
hipError_t launch_backsub_rows(const BuildBuffers &b, hipStream_t stream, int rows);
hipError_t launch_prepare_rhs(const BuildBuffers &b, hipStream_t stream, const PointSrc *src);
hipError_t launch_backsub_update(const BuildBuffers &b, hipStream_t stream, int row_lo, int w);
// the QNN model (fd_nullspace.hip launch_build_qnn) takes the LU apart: radii, then the pivoted
// factorisation + back-substitution of whatever has been assembled, full build or right-hand sides only
hipError_t launch_qnn_radii(const BuildBuffers &b, hipStream_t stream);
hipError_t launch_lu_factor_solve(const BuildBuffers &b, hipStream_t stream);
hipError_t launch_lu_resolve_core(const BuildBuffers &b, hipStream_t stream);

// ---- null-space Cholesky build (fd_nullspace.hip) -----------------------------------
// Which (kernel, term, lambda) make the projected block positive definite; M large enough to project.
bool spd_applicable(int kind, int term, double lambda, int M);
static inline size_t ns_doubles(int M) { return (size_t)12 * (size_t)M + 64 + (size_t)(M / 32 + 2) * 66 * 32; }
hipError_t launch_build_spd(const BuildBuffers &b, hipStream_t stream, hipEvent_t ev_mid);
hipError_t launch_resolve_spd(const BuildBuffers &b, hipStream_t stream, const PointSrc *src);
// ---- register-resident one-launch build (fd_build_reg.hip): the definite path for rigs of up to 256 control points
bool reg_applicable(int kind, int term, double lambda, int M);
// the WHOLE build in one launch, control table included (src == nullptr: the contexts' own copies of the control points)
hipError_t launch_build_reg(const BuildBuffers &b, hipStream_t stream, const PointSrc *src, hipEvent_t ev_mid);
hipError_t launch_build_reg_shared(const BuildBuffers &b, hipStream_t stream, const PointSrc *src, hipEvent_t ev_mid, double *fac);
size_t reg_factor_doubles();
hipError_t reg_build_init();          // once per device, outside stream capture
constexpr int kMaxLayers = 8;
hipError_t launch_build_ml(const BuildBuffers &b, hipStream_t stream, hipEvent_t ev_mid);
// FD_KERNEL_GAUSSIAN_QNN, the SOP's model = 0: least-squares polynomial first, then the pivoted LU
// of the (non-symmetric) kernel block on what is left
hipError_t launch_build_qnn(const BuildBuffers &b, hipStream_t stream, hipEvent_t ev_mid);
hipError_t launch_resolve_qnn(const BuildBuffers &b, hipStream_t stream, const PointSrc *src);

// ---- evaluation (fd_eval.hip) --------------------------------------------------
struct DeformArgs {
    int64_t N;
    const float *P_in; float *P_out;
    const float *dist2; float *falloff_out;
    const float *tu, *tv, *nrm;
    float radius2, falloffrate;
    int M, Mpad, kind;
    const Rec32 *rec32; const Rec64 *rec64;
    const MfmaTile *tiles;
    const MfmaTileH *tiles16;
    const DevModel *model;
    int precision, variant;
    int layers;            // multilayer Gaussian model: records are centre-major, `layers` per centre; 0 otherwise
    int delta_out;         // FD_OUTPUT_DISPLACEMENT: P_out receives the displacement (the addend of :438), not P + displacement
};
hipError_t launch_deform(const DeformArgs &a, hipStream_t stream);
hipError_t launch_deform_batch(const DeformArgs *a, int n, hipStream_t stream);
// frames that share the mesh and the rest rig (fd_eval.hip, k_deform32_tps_shared): phi once per
// (vertex, centre), the 3 F-wide weight contraction on the matrix pipe
struct SharedDeformArgs {
    int64_t N;
    const float *P_in;                    // the one mesh
    const float *dist2;
    const float *tu, *tv, *nrm;
    float radius2, falloffrate;
    int Mpad, nF;
    int kind;                             // FD_KERNEL_THIN_PLATE, or a Gaussian kind (FD_KERNEL_GAUSSIAN / _QNN records)
    const MfmaTileH *ctiles;              // centre tiles of the shared rest rig (thin-plate; the Gaussian kinds read rec32[0])
    const Rec32 *rec32[kMaxBatch];        // per frame: the solved model's records (weights)
    const DevModel *model[kMaxBatch];
    float *P_out[kMaxBatch];
    float *const *falloff_out;            // nF entries or nullptr
    void *wtiles, *frames;                // scratch of shared_wtile_bytes / shared_frame_bytes
    hipEvent_t packed_ev;                 // recorded once the pack kernel has read the models (may be null): from then on the
                                          // launch reads nothing of the contexts -- their next build may start
    int mode;                             // 0: pack kernel + evaluation; 1: pack kernel only (fd_batch_prepare_shared);
                                          // 2: evaluation only, on a scratch set packed earlier
    int delta_out;                        // FD_OUTPUT_DISPLACEMENT (see DeformArgs)
    int max_wgs;                          // CUs the launch may occupy (one persistent workgroup each); 0 or >= 256: all of them
    // pack kernel only: every model's centres (fp64, M x 3) are compared with model 0's -- "one rest rig" checked by content --
    // and 1 + the index of a model that differs is posted to *mismatch (device address of a page-locked word; may be null)
    const double *centres[kMaxBatch];
    int M;
    int *mismatch;
};
hipError_t launch_deform_shared(const SharedDeformArgs &a, hipStream_t stream);
size_t shared_wtile_bytes(int Mpad, int nF);
size_t shared_frame_bytes(int nF);
const char *shared_kernel_name(int Mpad, int nF, int kind);
// island mask (fd_capture.hip): nearest mesh point per rig point + max_edges breadth-first rings
hipError_t launch_capture_islands(const float *d_P, int64_t N, const int64_t *d_offsets, const int *d_neighbours,
                                  const float *d_rig, int M, int max_edges, unsigned char *d_mask, hipStream_t stream);
// dist2 producer (fd_capture.hip): all pointers are device pointers, d_mask may be null
hipError_t launch_capture_dist2(const float *d_P, int64_t N, const unsigned char *d_mask, const float *d_tri, int T,
                                float radius2, int dofalloff, float *d_dist2, hipStream_t stream);
const char *deform_kernel_name(int kind, int precision, int variant);

}  // namespace fd
