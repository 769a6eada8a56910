// fd_capi.hip -- the C ABI of include/facedeform_hip.h over the HIP kernels.
// No CPU fallback anywhere: without a gfx950 device fd_create fails loudly.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "fd_internal.h"

using namespace fd;

struct fd_batch;
struct fd_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;   // created on first use when the caller sets no stream
    hipStream_t stream_ = nullptr;          // fd_set_stream; nullptr: own_stream
    int eval_precision = FD_EVAL_FP32;
    int eval_variant = 0;
    int output = FD_OUTPUT_POSITION;    // fd_set_output
    int solver = FD_SOLVER_AUTO;
    int imported_layers = 0;         // a multilayer model that came in through fd_import_model
    uint64_t rig_build_id = 0;       // nonzero: the id of the ONE batched build that read this context's rest rig, in place, from the same array as
                                     // its batch mates (their centres are equal by construction); 0: built on its own / from its own copy
    const float *rest_src = nullptr; // caller's device array the rest points were last read from in place (fd_batch_set_points_dev), else null
    bool prefer_lu = false;          // the Cholesky path lost definiteness on this rig: LU until kernel, term or M change
    unsigned long long model_gen = 0; // counts the models this context has held (every enqueued build, every import): what a
                                     // batch's packed copy of the weights is checked against before it is reused
    bool last_spd = false;           // the build in flight / last finished took the Cholesky path
    bool last_nopivot = false;       // ... the LU without pivot search (QNN): a -4 from it is answered with the pivoted LU, like the Cholesky's
    bool last_reg = false;           // ... in its register-resident form (no factorisation is kept: fd_set_deltas builds again)

    // model configuration
    int M = 0, kind = FD_KERNEL_GAUSSIAN_QNN, term = FD_TERM_LINEAR, nparams = 0;
    double params[4] = {1.0, 5.0, 0.0, 0.0};
    bool points_set = false;
    bool build_pending = false;   // enqueued, status not read back yet
    bool built = false;           // status read back and == 1
    bool have_report = false;
    fd_report report{};
    // fd_set_deltas: a full build with the current rest points / kernel / term has been enqueued
    // (its factorisation sits in d_A), and the next build only has new right-hand sides
    bool have_factor = false;
    bool factor_grouped = false;   // the factorisation in d_A was built with grouped panels (a batched build)
    bool deltas_only = false;
    hipGraphExec_t resolve_exec = nullptr;

    // device buffers (grow-only)
    int cap_M = 0;                // capacity in centres
    int cap_npad = 0;
    int cap_records = 0;             // centres / radii / weights / records: M, or M * layers for the multilayer model
    float *d_rest = nullptr, *d_delta = nullptr;
    double *d_centres = nullptr, *d_radii = nullptr, *d_W = nullptr;
    double *d_A = nullptr, *d_X = nullptr;
    int *d_ipiv = nullptr, *d_moves = nullptr;
    Rec32 *d_rec32 = nullptr;
    Rec64 *d_rec64 = nullptr;
    MfmaTile *d_tiles = nullptr;
    MfmaTileH *d_tiles16 = nullptr;
    double *d_ns = nullptr;          // null-space solver state (fd_nullspace.hip)
    DevModel *d_model = nullptr;
    DevModel *h_model = nullptr;  // pinned mirror
    // where the build kernels find the buffers above: a one-entry device table (a single build
    // is a batch of one); re-uploaded whenever a buffer is reallocated
    BatchSlot *d_slot = nullptr;
    BatchSlot h_slot{};
    uint64_t alloc_gen = 0;
    // set while the model comes from a batched build on another stream (fd_batch_build_async)
    hipEvent_t wait_event = nullptr;
    hipStream_t wait_stream = nullptr;
    struct fd_batch *wait_batch = nullptr;
    hipEvent_t tev0 = nullptr, tev_mid = nullptr, tev1 = nullptr;   // events the report's timings come from
    ModelHeader *h_header = nullptr;  // pinned, for device-side export
    // Asynchronous status: after every enqueued build a one-thread kernel drops the model's
    // terminationtype into this page-locked word and status_ev is recorded behind it.  Every later
    // call on the context polls the event (no wait): a Cholesky that lost definiteness is then
    // rebuilt with the LU right away, on the stream, before whatever the call enqueues; a failure
    // the LU confirms becomes a sticky error that fd_deform* returns until the next set-up call.
    int *h_status = nullptr;
    int *d_status_alias = nullptr;      // the same word as the device sees it (looked up once: hipHostGetDevicePointer costs a runtime call per build)
    hipEvent_t status_ev = nullptr;
    hipEvent_t status_poll = nullptr;   // the event that says h_status is in: status_ev, or the one of the batch that built the model
    bool status_inflight = false;
    int sticky_rc = FD_OK;

    // staging for the host-pointer deform
    int64_t cap_N = 0;
    float *d_P = nullptr, *d_dist2 = nullptr, *d_fall = nullptr;
    float *d_tu = nullptr, *d_tv = nullptr, *d_nrm = nullptr;
    // fd_mesh_set: the mesh arrays that stay the same from cook to cook live in their own
    // device buffers (the staging above is evaluated in place, so it cannot serve as a cache)
    int64_t mesh_N = 0, mesh_cap = 0;
    bool mesh_has_dist2 = false, mesh_has_frames = false;
    float *m_P = nullptr, *m_dist2 = nullptr, *m_tu = nullptr, *m_tv = nullptr, *m_nrm = nullptr;
    float *m_out = nullptr, *m_fall = nullptr;      // result staging for pageable outputs

    hipEvent_t ev0 = nullptr, ev_mid = nullptr, ev1 = nullptr;
    // LU look-ahead (fd_build.hip lu_step): a second stream and four events, only with
    // FD_LOOKAHEAD set (see make_lookahead)
    hipStream_t lu_stream = nullptr;
    hipEvent_t lu_events[4] = {nullptr, nullptr, nullptr, nullptr};

    // The build is ~25 dependent launches whose arguments depend only on the configuration
    // and on buffer addresses: captured once into a hipGraph, replayed on every later build.
    hipGraphExec_t build_exec = nullptr;
    bool use_graph = true;
    struct GraphKey {
        int M, kind, term, nparams;
        double params[4];
        const void *A, *rest, *rec32;
    } graph_key{}, resolve_key{};
    char err[512] = {0};
};

struct fd_batch {
    int n = 0;
    int device = 0;
    fd_ctx *ctxs[kMaxBatch] = {};
    uint64_t gens[kMaxBatch] = {};
    BatchSlot *d_slots = nullptr;
    PointSrc src{};
    bool have_src = false;
    hipEvent_t ev0 = nullptr, ev_mid = nullptr, ev1 = nullptr;
    hipStream_t lu_stream = nullptr;
    hipEvent_t lu_events[4] = {nullptr, nullptr, nullptr, nullptr};
    hipGraphExec_t exec = nullptr;
    bool use_graph = true;
    struct Key { int M, kind, term, nparams; double params[4]; } key{};
    // a stream that already waits for the current build (set by the first evaluation that
    // needed it): the other contexts' evaluations on that stream need no wait of their own
    hipStream_t waited_stream = nullptr;
    hipEvent_t status_ev = nullptr;     // behind the kernel that posts every context's status after a batched build
    // scratch of the shared-rig evaluation (fd_batch_deform_shared_dev): weight tiles + frame records
    // Two sets: the set a launch reads must not be rewritten by the pack kernel of the NEXT models before that launch
    // has finished -- with fd_batch_prepare_shared the next pack runs on the build stream, under the previous evaluation.
    struct SharedSet {
        void *d_wtiles = nullptr, *d_frames = nullptr;
        size_t cap_wtiles = 0, cap_frames = 0;
        hipEvent_t packed_ev = nullptr;  // behind the pack kernel that filled the set
        hipEvent_t eval_ev = nullptr;    // behind the last evaluation that read it
        hipEvent_t eval_done = nullptr;  // the event that says so: eval_ev, or -- a lean group call -- the batch's ev1, recorded right behind
        bool eval_pending = false;
    } sets[2];
    // "One rest rig" is decided on the host by the address the rest points were read from; an address does not identify
    // its contents (the array may have been rewritten, or freed and allocated again, between the set-ups of two contexts:
    // ADVICE r2), so the pack kernel compares every context's centres with context 0's ON THE DEVICE, passes a frame that
    // differs through like a failed build and posts 1 + its index here (page-locked); the next call on the batch reports it.
    int *h_mismatch = nullptr;
    int eval_cus = 0;                    // fd_batch_set_eval_cus: CUs a shared-rig evaluation may occupy (0: all)
    int shared_factor = 0;               // fd_batch_set_shared_factor: one factorisation per build where the contexts share the rest rig
    double *d_fac = nullptr;             // ... and its scratch (reg_factor_doubles())
    int last_shared_factor = 0;          // the last build took that path
    int cur_set = 0;                     // the set packed last
    bool packed_valid = false;           // sets[cur_set].packed_ev is recorded (fd_batch_wait_consumed)
    bool prepared = false;               // sets[cur_set] holds the contexts' CURRENT models and prep_* outputs
    unsigned long long prep_gen[kMaxBatch] = {0};   // the contexts' model_gen when the set was packed
    float *prep_P_out[kMaxBatch] = {nullptr};
    float *prep_fall[kMaxBatch] = {nullptr};
    bool prep_has_fall = false;
    hipEvent_t fallback_ev = nullptr;    // behind the per-frame launches when the shared launch does not apply
    hipEvent_t group_ev = nullptr;       // fd_batch_cook_group: behind the group's builds, for the evaluation stream
    // fd_batch_cook_group with the evaluation on the build stream itself (one unpipelined group): stream order does what the
    // events between build, packing and evaluation do across streams, and every event record is a barrier packet that keeps the
    // queue idle for ~4 us.  `lean`: the build's end (ev1) and the packing's are not recorded where they happen; ev1 goes behind
    // the evaluation (later than needed, never earlier), and "the models are consumed" is the evaluation's own event.
    bool lean = false;
    hipEvent_t consumed_override = nullptr;
    char err[512] = {0};
};


static thread_local char g_err[512] = {0};

static void set_err(fd_ctx *ctx, const char *fmt, ...)
{
    char *dst = ctx ? ctx->err : g_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
}

// The stream work is enqueued on: the caller's (fd_set_stream) or the context's own, which is
// created on first use -- a process with many contexts on caller streams should not spend a
// hardware queue per context (HIP maps streams onto GPU_MAX_HW_QUEUES queues round-robin, and
// two streams on one queue serialise).
static hipStream_t cur_stream(fd_ctx *ctx)
{
    if (ctx->stream_) return ctx->stream_;
    if (!ctx->own_stream && hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        ctx->own_stream = nullptr;     // falls back to the null stream
    }
    return ctx->own_stream;
}

#define FD_HIP(ctx, call)                                                               \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            set_err(ctx, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return FD_E_DEVICE;                                                         \
        }                                                                               \
    } while (0)

static int use_device(fd_ctx *ctx)
{
    int cur = -1;
    FD_HIP(ctx, hipGetDevice(&cur));
    if (cur != ctx->device) FD_HIP(ctx, hipSetDevice(ctx->device));
    return FD_OK;
}

template <typename T>
static int dev_alloc(fd_ctx *ctx, T **p, size_t count)
{
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (count == 0) count = 1;
    hipError_t e = hipMalloc((void **)p, count * sizeof(T));
    if (e != hipSuccess) {
        *p = nullptr;
        set_err(ctx, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
        return FD_E_NOMEM;
    }
    return FD_OK;
}

static int order_of(const fd_ctx *ctx) { return ctx->M + term_cols(ctx->term); }
static bool use_spd(const fd_ctx *ctx);
static int ml_layers(const fd_ctx *ctx) { return ctx->kind == FD_KERNEL_GAUSSIAN_ML ? (int)ctx->params[1] : 0; }
// Gaussian records of the solved model, and the kernel the evaluation sees
static int model_centres(const fd_ctx *ctx) { return ctx->kind == FD_KERNEL_GAUSSIAN_ML ? ctx->M * ml_layers(ctx) : ctx->M; }
static int eval_kind(const fd_ctx *ctx) { return ctx->kind == FD_KERNEL_GAUSSIAN_ML ? FD_KERNEL_GAUSSIAN_QNN : ctx->kind; }
// layers per centre in the evaluation records (built here or imported); 0 = a flat model
static int record_layers(const fd_ctx *ctx) { return ctx->kind == FD_KERNEL_GAUSSIAN_ML ? ml_layers(ctx) : ctx->imported_layers; }

// the solved model's arrays: one entry per Gaussian record (= per centre, times the layers of the
// multilayer model).  Everything in them is produced by a build or an import, so growing them loses nothing.
static int ensure_records_capacity(fd_ctx *ctx, int n)
{
    int rc;
    if (n > ctx->cap_records) {
        const int npad = round_up(n, kRecPad);
        if ((rc = dev_alloc(ctx, &ctx->d_centres, (size_t)n * 3))) return rc;
        if ((rc = dev_alloc(ctx, &ctx->d_radii, (size_t)n))) return rc;
        if ((rc = dev_alloc(ctx, &ctx->d_W, (size_t)(n + 4) * 3))) return rc;
        if ((rc = dev_alloc(ctx, &ctx->d_rec32, (size_t)npad))) return rc;
        if ((rc = dev_alloc(ctx, &ctx->d_rec64, (size_t)npad))) return rc;
        if ((rc = dev_alloc(ctx, &ctx->d_tiles, (size_t)npad / 16))) return rc;
        if ((rc = dev_alloc(ctx, &ctx->d_tiles16, (size_t)npad / 16))) return rc;
        ctx->cap_records = n;
    }
    return FD_OK;
}

static int ensure_model_capacity(fd_ctx *ctx, int M)
{
    int rc;
    if (M > ctx->cap_M) {
        if ((rc = dev_alloc(ctx, &ctx->d_rest, (size_t)M * 3))) return rc;
        if ((rc = dev_alloc(ctx, &ctx->d_delta, (size_t)M * 3))) return rc;
        if ((rc = dev_alloc(ctx, &ctx->d_ns, ns_doubles(M)))) return rc;
        ctx->cap_M = M;
    }
    return ensure_records_capacity(ctx, M);
}

static int ensure_solver_capacity(fd_ctx *ctx, int npad)
{
    int rc;
    if (npad > ctx->cap_npad) {
        const size_t cols = (size_t)npad + kRhsCols + 16;  // + 16 zero columns (block overrun)
        if ((rc = dev_alloc(ctx, &ctx->d_A, (size_t)npad * cols))) return rc;
        if ((rc = dev_alloc(ctx, &ctx->d_X, (size_t)npad * 3))) return rc;
        if ((rc = dev_alloc(ctx, &ctx->d_ipiv, (size_t)npad))) return rc;
        if ((rc = dev_alloc(ctx, &ctx->d_moves, (size_t)kMovesStride * (size_t)lu_step_capacity(npad)))) return rc;
        ctx->cap_npad = npad;
    }
    return FD_OK;
}

// Second stream + events of the LU look-ahead (fd_build.hip lu_step); false leaves the build on
// one stream.  OFF unless FD_LOOKAHEAD is set: measured on MI355X / ROCm 7.2 it loses -- C2 build
// 0.444 vs 0.401 ms, C3 7.65 vs 6.63 ms -- because every cross-queue event wait costs more than
// the overlap of a ~25 us panel with a ~30 us update buys.
static bool make_lookahead(hipStream_t *stream, hipEvent_t events[4])
{
    static const bool on = tuning_env("FD_LOOKAHEAD") != nullptr;
    if (!on) return false;
    if (*stream) return true;
    if (hipStreamCreateWithFlags(stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); *stream = nullptr; return false; }
    for (int q = 0; q < 4; ++q)
        if (hipEventCreateWithFlags(&events[q], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            for (int r = 0; r < q; ++r) { (void)hipEventDestroy(events[r]); events[r] = nullptr; }
            (void)hipStreamDestroy(*stream);
            *stream = nullptr;
            return false;
        }
    return true;
}

static int sync_slot(fd_ctx *ctx)
{
    BatchSlot t{};
    t.rest = ctx->d_rest; t.delta = ctx->d_delta;
    t.centres = ctx->d_centres; t.radii = ctx->d_radii;
    t.A = ctx->d_A; t.X = ctx->d_X; t.W = ctx->d_W;
    t.ipiv = ctx->d_ipiv; t.moves = ctx->d_moves;
    t.rec32 = ctx->d_rec32; t.rec64 = ctx->d_rec64; t.tiles = ctx->d_tiles; t.tiles16 = ctx->d_tiles16;
    t.model = ctx->d_model;
    t.ns = ctx->d_ns;
    t.host_status = ctx->d_status_alias;
    if (ctx->alloc_gen != 0 && memcmp(&t, &ctx->h_slot, sizeof(t)) == 0) return FD_OK;
    // a buffer moved: hipFree in dev_alloc has drained the device, nothing reads the old table
    FD_HIP(ctx, hipMemcpy(ctx->d_slot, &t, sizeof(t), hipMemcpyHostToDevice));
    ctx->h_slot = t;
    ++ctx->alloc_gen;
    return FD_OK;
}

// a model that a batched build is producing on another stream: make `s` wait for it
static bool host_is_pinned(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

static int order_after_batch(fd_ctx *ctx, hipStream_t s)
{
    if (!ctx->wait_event || s == ctx->wait_stream) return FD_OK;
    if (ctx->wait_batch && ctx->wait_batch->waited_stream == s) return FD_OK;
    FD_HIP(ctx, hipStreamWaitEvent(s, ctx->wait_event, 0));
    if (ctx->wait_batch) ctx->wait_batch->waited_stream = s;
    return FD_OK;
}

// the model's status into page-locked host memory (one thread; the build's last kernel wrote it)
__global__ void k_post_status(const DevModel *model, int *host_word) { *host_word = model->terminationtype; }
struct StatusTable { const DevModel *model[kMaxBatch]; int *host_word[kMaxBatch]; };
__global__ void k_post_status_batch(const StatusTable t, int n)
{
    const int i = threadIdx.x;
    if (i < n) *t.host_word[i] = t.model[i]->terminationtype;
}

static int post_status(fd_ctx *ctx, hipStream_t s)
{
    int *dword = ctx->d_status_alias;
    if (!dword) return FD_OK;
    hipLaunchKernelGGL(k_post_status, dim3(1), dim3(1), 0, s, ctx->d_model, dword);
    if (hipGetLastError() != hipSuccess || hipEventRecord(ctx->status_ev, s) != hipSuccess) { (void)hipGetLastError(); return FD_OK; }
    ctx->status_poll = ctx->status_ev;
    ctx->status_inflight = true;
    return FD_OK;
}

static int poll_status(fd_ctx *ctx);

extern "C" {

int fd_abi_version(void) { return FD_ABI_VERSION; }

const char *fd_last_error(const fd_ctx *ctx) { return ctx ? ctx->err : g_err; }

fd_ctx *fd_create(const fd_config *cfg)
{
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        set_err(nullptr, "fd_create: no HIP device visible (%s); this engine has no CPU path",
                e != hipSuccess ? hipGetErrorString(e) : "0 devices");
        return nullptr;
    }
    fd_ctx *ctx = new (std::nothrow) fd_ctx();
    if (!ctx) { set_err(nullptr, "fd_create: out of host memory"); return nullptr; }
    int dev = cfg ? cfg->device : -1;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= ndev) {
        set_err(nullptr, "fd_create: device %d out of range (%d visible)", dev, ndev);
        delete ctx;
        return nullptr;
    }
    ctx->device = dev;
    if (cfg) {
        ctx->eval_precision = cfg->eval_precision == FD_EVAL_FP64 ? FD_EVAL_FP64 : FD_EVAL_FP32;
        ctx->eval_variant = cfg->eval_variant;
        ctx->solver = (cfg->solver == FD_SOLVER_LU || cfg->solver == FD_SOLVER_ONE_WORKGROUP || cfg->solver == FD_SOLVER_REGISTER ||
                       cfg->solver == FD_SOLVER_CHAIN) ? cfg->solver : FD_SOLVER_AUTO;
    }
    hipDeviceProp_t prop;
    if (hipSetDevice(dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        set_err(nullptr, "fd_create: cannot query device %d", dev);
        delete ctx;
        return nullptr;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_err(nullptr, "fd_create: device %d is %s; kernels are built for gfx950 (MI355X) only",
                dev, prop.gcnArchName);
        delete ctx;
        return nullptr;
    }
    bool ok = true;
    ok = ok && hipEventCreate(&ctx->ev0) == hipSuccess && hipEventCreate(&ctx->ev_mid) == hipSuccess &&
         hipEventCreate(&ctx->ev1) == hipSuccess;
    ok = ok && hipMalloc((void **)&ctx->d_model, sizeof(DevModel)) == hipSuccess;
    ok = ok && hipMalloc((void **)&ctx->d_slot, sizeof(BatchSlot)) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&ctx->h_model, sizeof(DevModel), hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&ctx->h_header, sizeof(ModelHeader), hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&ctx->h_status, sizeof(int), hipHostMallocDefault) == hipSuccess;
    if (ok && hipHostGetDevicePointer((void **)&ctx->d_status_alias, ctx->h_status, 0) != hipSuccess) { (void)hipGetLastError(); ctx->d_status_alias = nullptr; }
    ok = ok && hipEventCreateWithFlags(&ctx->status_ev, hipEventDisableTiming) == hipSuccess;
    if (ok) ok = hipMemset(ctx->d_model, 0, sizeof(DevModel)) == hipSuccess;
    if (!ok) {
        set_err(nullptr, "fd_create: device resource allocation failed: %s",
                hipGetErrorString(hipGetLastError()));
        fd_destroy(ctx);
        return nullptr;
    }
    ctx->tev0 = ctx->ev0; ctx->tev_mid = ctx->ev_mid; ctx->tev1 = ctx->ev1;
    ctx->use_graph = tuning_env("FD_NO_GRAPH") == nullptr;
    return ctx;
}

void fd_destroy(fd_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream_ || ctx->own_stream) (void)hipStreamSynchronize(cur_stream(ctx));
    void *bufs[] = {ctx->d_rest, ctx->d_delta, ctx->d_centres, ctx->d_radii, ctx->d_W, ctx->d_A,
                    ctx->d_X, ctx->d_ipiv, ctx->d_moves, ctx->d_rec32, ctx->d_rec64, ctx->d_tiles, ctx->d_tiles16, ctx->d_ns, ctx->d_model, ctx->d_slot,
                    ctx->d_P, ctx->d_dist2, ctx->d_fall, ctx->d_tu, ctx->d_tv, ctx->d_nrm,
                    ctx->m_P, ctx->m_dist2, ctx->m_tu, ctx->m_tv, ctx->m_nrm, ctx->m_out, ctx->m_fall};
    for (void *p : bufs) if (p) (void)hipFree(p);
    if (ctx->h_model) (void)hipHostFree(ctx->h_model);
    if (ctx->h_header) (void)hipHostFree(ctx->h_header);
    if (ctx->h_status) (void)hipHostFree(ctx->h_status);
    if (ctx->status_ev) (void)hipEventDestroy(ctx->status_ev);
    if (ctx->build_exec) (void)hipGraphExecDestroy(ctx->build_exec);
    if (ctx->resolve_exec) (void)hipGraphExecDestroy(ctx->resolve_exec);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev_mid) (void)hipEventDestroy(ctx->ev_mid);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (hipEvent_t e : ctx->lu_events) if (e) (void)hipEventDestroy(e);
    if (ctx->lu_stream) (void)hipStreamDestroy(ctx->lu_stream);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int fd_set_stream(fd_ctx *ctx, void *hip_stream)
{
    if (!ctx) return FD_E_INVALID;
    int rc = use_device(ctx);
    if (rc) return rc;
    if (ctx->stream_ || ctx->own_stream) FD_HIP(ctx, hipStreamSynchronize(cur_stream(ctx)));
    ctx->stream_ = (hipStream_t)hip_stream;
    return FD_OK;
}

int fd_synchronize(fd_ctx *ctx)
{
    if (!ctx) return FD_E_INVALID;
    int rc = use_device(ctx);
    if (rc) return rc;
    FD_HIP(ctx, hipStreamSynchronize(cur_stream(ctx)));
    return FD_OK;
}

static int set_points_common(fd_ctx *ctx, const float *rest, const float *delta, int M, bool on_device)
{
    if (!ctx) return FD_E_INVALID;
    if (!rest || !delta || M <= 0) { set_err(ctx, "fd_set_points: need M > 0 and both arrays"); return FD_E_INVALID; }
    if (M + 4 > kMaxOrder) { set_err(ctx, "fd_set_points: M = %d exceeds the supported %d", M, kMaxOrder - 4); return FD_E_INVALID; }
    int rc = use_device(ctx);
    if (rc) return rc;
    if ((rc = ensure_model_capacity(ctx, M))) return rc;
    const hipMemcpyKind k = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    FD_HIP(ctx, hipMemcpyAsync(ctx->d_rest, rest, sizeof(float) * 3 * (size_t)M, k, cur_stream(ctx)));
    FD_HIP(ctx, hipMemcpyAsync(ctx->d_delta, delta, sizeof(float) * 3 * (size_t)M, k, cur_stream(ctx)));
    if (!on_device) FD_HIP(ctx, hipStreamSynchronize(cur_stream(ctx)));  // caller may reuse its arrays
    if (M != ctx->M) ctx->prefer_lu = false;
    ctx->M = M;
    ctx->rest_src = nullptr;
    ctx->rig_build_id = 0;
    ctx->sticky_rc = FD_OK; ctx->status_inflight = false;
    ctx->points_set = true;
    ctx->built = false;
    ctx->build_pending = false;
    ctx->have_factor = false;
    ctx->deltas_only = false;
    return FD_OK;
}

// New deltas for the rest points already factorised (the animated-rig case: the rest rig stands
// still, the deformed rig moves).  The next fd_build* only carries the new right-hand sides
// through the stored factorisation.
static int set_deltas_common(fd_ctx *ctx, const float *delta, int M, bool on_device)
{
    if (!ctx) return FD_E_INVALID;
    if (!delta || M <= 0) { set_err(ctx, "fd_set_deltas: need M > 0 and the array"); return FD_E_INVALID; }
    if (!ctx->have_factor || M != ctx->M) {
        set_err(ctx, "fd_set_deltas: no factorisation for %d control points (call fd_set_points + fd_build first; "
                     "fd_set_kernel / fd_set_term / fd_import_model discard it)", M);
        return FD_E_NOT_BUILT;
    }
    if (!ctx->last_spd && round_up(order_of(ctx), 32) > 2048) {
        set_err(ctx, "fd_set_deltas: supported up to order 2048 (M = %d here); use fd_set_points", M);
        return FD_E_INVALID;
    }
    int rc = use_device(ctx);
    if (rc) return rc;
    const hipMemcpyKind k = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    FD_HIP(ctx, hipMemcpyAsync(ctx->d_delta, delta, sizeof(float) * 3 * (size_t)M, k, cur_stream(ctx)));
    if (!on_device) FD_HIP(ctx, hipStreamSynchronize(cur_stream(ctx)));
    ctx->deltas_only = true;
    ctx->built = false;
    ctx->build_pending = false;
    ctx->sticky_rc = FD_OK; ctx->status_inflight = false;
    return FD_OK;
}

int fd_set_deltas(fd_ctx *ctx, const float *delta_xyz, int M) { return set_deltas_common(ctx, delta_xyz, M, false); }
int fd_set_deltas_dev(fd_ctx *ctx, const float *d_delta_xyz, int M) { return set_deltas_common(ctx, d_delta_xyz, M, true); }

int fd_set_points(fd_ctx *ctx, const float *rest_xyz, const float *delta_xyz, int M)
{
    return set_points_common(ctx, rest_xyz, delta_xyz, M, false);
}

int fd_set_points_dev(fd_ctx *ctx, const float *d_rest_xyz, const float *d_delta_xyz, int M)
{
    return set_points_common(ctx, d_rest_xyz, d_delta_xyz, M, true);
}

int fd_set_kernel(fd_ctx *ctx, int kind, const double *params, int nparams)
{
    if (!ctx) return FD_E_INVALID;
    if (kind < FD_KERNEL_GAUSSIAN || kind > FD_KERNEL_GAUSSIAN_ML || nparams < 0 || nparams > 4 ||
        (nparams > 0 && !params)) {
        set_err(ctx, "fd_set_kernel: bad kind %d / nparams %d", kind, nparams);
        return FD_E_INVALID;
    }
    double p[4] = {0, 0, 0, 0};
    for (int i = 0; i < nparams; ++i) p[i] = params[i];
    if (kind == FD_KERNEL_GAUSSIAN) {
        if (nparams < 1) p[0] = 1.0;
        if (!(p[0] > 0.0)) { set_err(ctx, "fd_set_kernel: Gaussian radius must be > 0"); return FD_E_INVALID; }
    } else if (kind == FD_KERNEL_GAUSSIAN_QNN) {
        if (nparams < 1) p[0] = 1.0;
        if (nparams < 2) p[1] = 5.0;
        if (!(p[0] > 0.0) || !(p[1] > 0.0)) { set_err(ctx, "fd_set_kernel: q and z must be > 0"); return FD_E_INVALID; }
    } else if (kind == FD_KERNEL_GAUSSIAN_ML) {
        if (nparams < 1) p[0] = 1.0;
        if (nparams < 2) p[1] = 4.0;
        if (nparams < 3) p[2] = 0.1;     // the SOP's defaults for radius, layers, lambda (src/SOP_FaceDeform.cpp:125-131)
        p[1] = floor(p[1]);
        if (!(p[0] > 0.0) || !(p[1] >= 1.0 && p[1] <= (double)kMaxLayers) || !(p[2] >= 0.0)) {
            set_err(ctx, "fd_set_kernel: multilayer needs radius > 0, 1 <= layers <= %d, lambda >= 0", kMaxLayers);
            return FD_E_INVALID;
        }
    }
    if (kind == ctx->kind && nparams == ctx->nparams && memcmp(ctx->params, p, sizeof(p)) == 0)
        return FD_OK;                 // nothing changes: the model and its factorisation stay valid
    ctx->kind = kind;
    ctx->nparams = nparams;
    memcpy(ctx->params, p, sizeof(p));
    ctx->imported_layers = 0;
    ctx->prefer_lu = false;
    ctx->built = false;
    ctx->build_pending = false;
    ctx->have_factor = false;
    ctx->deltas_only = false;
    return FD_OK;
}

int fd_set_term(fd_ctx *ctx, int term)
{
    if (!ctx) return FD_E_INVALID;
    if (term < FD_TERM_LINEAR || term > FD_TERM_ZERO) { set_err(ctx, "fd_set_term: bad term %d", term); return FD_E_INVALID; }
    if (term == ctx->term) return FD_OK;
    ctx->term = term;
    ctx->prefer_lu = false;
    ctx->built = false;
    ctx->build_pending = false;
    ctx->have_factor = false;
    ctx->deltas_only = false;
    return FD_OK;
}

static double ctx_lambda(const fd_ctx *ctx)
{
    const int idx = ctx->kind == FD_KERNEL_GAUSSIAN ? 1 : (ctx->kind == FD_KERNEL_GAUSSIAN_QNN || ctx->kind == FD_KERNEL_GAUSSIAN_ML ? 2 : 0);
    return ctx->nparams > idx ? ctx->params[idx] : 0.0;
}

// FD_SOLVER=lu keeps every system on the pivoted LU (A/B measurements, tests of that path)
static bool use_spd(const fd_ctx *ctx)
{
    static const char *env = tuning_env("FD_SOLVER");
    if ((env && strcmp(env, "lu") == 0) || ctx->solver == FD_SOLVER_LU || ctx->prefer_lu) return false;
    if (ctx->kind == FD_KERNEL_GAUSSIAN_ML) return false;          // its own pipeline (launch_build_ml)
    return spd_applicable(ctx->kind, ctx->term, ctx_lambda(ctx), ctx->M);
}

// The QNN model's kernel block without pivot search (fd_build.hip k_lu_panel_np): one row per thread, order <= 1024; not once
// a build of this rig has asked for the pivoted LU (prefer_lu), nor under FD_SOLVER_LU.
static bool use_nopivot(const fd_ctx *ctx)
{
    static const char *env = tuning_env("FD_SOLVER");
    if ((env && strcmp(env, "lu") == 0) || ctx->solver == FD_SOLVER_LU || ctx->prefer_lu) return false;
    return ctx->kind == FD_KERNEL_GAUSSIAN_QNN && round_up(ctx->M, 32) <= 1024;
}

// The register-resident one-launch build (fd_build_reg.hip): what FD_SOLVER_AUTO takes on the definite path up to 256
// control points; FD_SOLVER_CHAIN keeps the launch chain, FD_SOLVER_ONE_WORKGROUP the round-2 one-workgroup build
// (FD_REG_BUILD=0 in the environment: never, for A/B measurements).
static bool use_reg(const fd_ctx *ctx)
{
    static const bool off = [] { const char *e = tuning_env("FD_REG_BUILD"); return e && atoi(e) == 0; }();
    if (off || !(ctx->solver == FD_SOLVER_AUTO || ctx->solver == FD_SOLVER_REGISTER)) return false;
    return reg_applicable(ctx->kind, ctx->term, ctx_lambda(ctx), ctx->M);
}

static void fill_build_buffers(const fd_ctx *ctx, BuildBuffers &b)
{
    b.M = ctx->M;
    b.T = term_cols(ctx->term);
    b.n = order_of(ctx);
    b.npad = round_up(b.n, 32);
    b.lda = b.npad;
    b.ncols = b.npad + kRhsCols;
    b.ml_layers = ml_layers(ctx);
    b.kind = b.ml_layers ? FD_KERNEL_GAUSSIAN : ctx->kind;       // what the assembly evaluates
    b.term = ctx->term;
    b.lambda = ctx_lambda(ctx);
    b.gauss_R = (ctx->kind == FD_KERNEL_GAUSSIAN || b.ml_layers) ? ctx->params[0] : 1.0;
    b.qnn_q = ctx->params[0];
    b.qnn_z = ctx->params[1];
    b.Mpad = round_up(ctx->M, kRecPad);
    b.d_slots = ctx->d_slot;
    b.nbatch = 1;
    b.group_panels = 0;
    b.spd = use_spd(ctx) ? 1 : 0;
    b.small = ctx->solver == FD_SOLVER_ONE_WORKGROUP ? 1 : 0;
    b.reg = (b.spd && use_reg(ctx)) ? 1 : 0;
    b.reg_front = 1;
    b.nopivot = use_nopivot(ctx) ? 1 : 0;
    b.aux_stream = nullptr;
    for (hipEvent_t &e : b.aux_events) e = nullptr;
}

int fd_build_async(fd_ctx *ctx)
{
    if (!ctx) return FD_E_INVALID;
    if (!ctx->points_set) { set_err(ctx, "fd_build: fd_set_points has not been called"); return FD_E_INVALID; }
    ctx->imported_layers = 0;        // whatever was imported is about to be replaced
    int rc = use_device(ctx);
    if (rc) return rc;
    const int npad = round_up(order_of(ctx), 32);
    const bool grew = npad > ctx->cap_npad;
    if ((rc = ensure_solver_capacity(ctx, npad))) return rc;
    if ((rc = ensure_records_capacity(ctx, model_centres(ctx)))) return rc;
    if ((rc = sync_slot(ctx))) return rc;
    BuildBuffers b;
    fill_build_buffers(ctx, b);
    if (b.reg) FD_HIP(ctx, reg_build_init());
    if (ctx->deltas_only && ctx->have_factor && ctx->last_reg) {
        // the register-resident build keeps no factorisation (it is faster than the stored-factor path was): new deltas
        // are a new build from the context's own copy of the points -- bit-identical to fd_set_points + fd_build by construction
        ctx->deltas_only = false;
    }
    if (ctx->deltas_only && ctx->have_factor) {
        // fd_set_deltas: right-hand sides only, through the factorisation of the last full build
        b.group_panels = ctx->factor_grouped ? 1 : 0;
        b.spd = ctx->last_spd ? 1 : 0;                // the path that left the factorisation (a batch may have overruled this context's own choice)
        hipStream_t st = cur_stream(ctx);
        fd_ctx::GraphKey rkey{};
        rkey.M = ctx->M; rkey.kind = ctx->kind; rkey.term = ctx->term | (b.group_panels << 8) | (b.spd << 9); rkey.nparams = ctx->nparams;
        memcpy(rkey.params, ctx->params, sizeof(rkey.params));
        rkey.A = ctx->d_A; rkey.rest = ctx->d_rest; rkey.rec32 = ctx->d_rec32;
        if (ctx->use_graph && (!ctx->resolve_exec || memcmp(&rkey, &ctx->resolve_key, sizeof(rkey)) != 0)) {
            if (ctx->resolve_exec) { (void)hipGraphExecDestroy(ctx->resolve_exec); ctx->resolve_exec = nullptr; }
            hipGraph_t graph = nullptr;
            hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess) {
                hipError_t e1 = launch_resolve(b, st, nullptr);
                e = hipStreamEndCapture(st, &graph);
                if (e == hipSuccess && e1 != hipSuccess) e = e1;
            }
            if (e == hipSuccess && graph) e = hipGraphInstantiate(&ctx->resolve_exec, graph, nullptr, nullptr, 0);
            if (graph) (void)hipGraphDestroy(graph);
            if (e != hipSuccess) { (void)hipGetLastError(); ctx->resolve_exec = nullptr; }
            else ctx->resolve_key = rkey;
        }
        FD_HIP(ctx, hipEventRecord(ctx->ev0, st));
        FD_HIP(ctx, hipEventRecord(ctx->ev_mid, st));
        if (ctx->use_graph && ctx->resolve_exec) FD_HIP(ctx, hipGraphLaunch(ctx->resolve_exec, st));
        else FD_HIP(ctx, launch_resolve(b, st, nullptr));
        FD_HIP(ctx, hipEventRecord(ctx->ev1, st));
        // a later evaluation on ANOTHER stream (fd_deform_dev_stream, fd_batch_deform*) is ordered behind this build
        ctx->wait_event = ctx->ev1; ctx->wait_stream = st; ctx->wait_batch = nullptr;
        ctx->tev0 = ctx->ev0; ctx->tev_mid = ctx->ev_mid; ctx->tev1 = ctx->ev1;
        ctx->last_spd = b.spd != 0;
        ctx->build_pending = true;
        ++ctx->model_gen;
        ctx->built = false;
        ctx->have_report = false;
        ctx->sticky_rc = FD_OK;
        return post_status(ctx, st);
    }
    if (b.reg) {
        // the register-resident build is ONE launch, control table included: nothing to capture, nothing to prepare
        hipStream_t st = cur_stream(ctx);
        FD_HIP(ctx, hipEventRecord(ctx->ev0, st));
        FD_HIP(ctx, launch_build_reg(b, st, nullptr, ctx->ev_mid));
        FD_HIP(ctx, hipEventRecord(ctx->ev1, st));
        ctx->wait_event = ctx->ev1; ctx->wait_stream = st; ctx->wait_batch = nullptr;
        ctx->tev0 = ctx->ev0; ctx->tev_mid = ctx->ev_mid; ctx->tev1 = ctx->ev1;
        ctx->have_factor = true;          // (fd_set_deltas is allowed: it builds again, see above)
        ctx->last_spd = true; ctx->last_reg = true; ctx->last_nopivot = false;
        ctx->factor_grouped = false;
        ctx->deltas_only = false;
        ctx->build_pending = true;
        ++ctx->model_gen;
        ctx->built = false;
        ctx->have_report = false;
        ctx->sticky_rc = FD_OK;
        if (ctx->h_slot.host_status) {
            // the build's own last thread has posted the status word: the event behind the build is the one to poll
            ctx->status_poll = ctx->ev1;
            ctx->status_inflight = true;
            return FD_OK;
        }
        return post_status(ctx, st);
    }
    if (make_lookahead(&ctx->lu_stream, ctx->lu_events)) {
        b.aux_stream = ctx->lu_stream;
        for (int q = 0; q < 4; ++q) b.aux_events[q] = ctx->lu_events[q];
    }
    if (grew) {
        // the 16 overrun columns past the RHS block must read as zero forever
        const size_t cols = (size_t)ctx->cap_npad + kRhsCols + 16;
        FD_HIP(ctx, hipMemsetAsync(ctx->d_A, 0, sizeof(double) * (size_t)ctx->cap_npad * cols, cur_stream(ctx)));
    }
    fd_ctx::GraphKey key{};
    key.M = ctx->M; key.kind = ctx->kind; key.term = ctx->term | (b.spd << 9) | (b.reg << 11) | (b.nopivot << 12); key.nparams = ctx->nparams;
    memcpy(key.params, ctx->params, sizeof(key.params));
    key.A = ctx->d_A; key.rest = ctx->d_rest; key.rec32 = ctx->d_rec32;
    if (ctx->use_graph && (!ctx->build_exec || memcmp(&key, &ctx->graph_key, sizeof(key)) != 0)) {
        if (ctx->build_exec) { (void)hipGraphExecDestroy(ctx->build_exec); ctx->build_exec = nullptr; }
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamBeginCapture(cur_stream(ctx), hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
            hipError_t e1 = launch_prepare(b, cur_stream(ctx), nullptr);
            hipError_t e2 = launch_build(b, cur_stream(ctx), nullptr);
            e = hipStreamEndCapture(cur_stream(ctx), &graph);
            if (e == hipSuccess && (e1 != hipSuccess || e2 != hipSuccess)) e = e1 != hipSuccess ? e1 : e2;
        }
        if (e == hipSuccess && graph) e = hipGraphInstantiate(&ctx->build_exec, graph, nullptr, nullptr, 0);
        if (graph) (void)hipGraphDestroy(graph);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            ctx->build_exec = nullptr;
            ctx->use_graph = false;      // this runtime cannot capture the sequence: launch directly
        } else {
            ctx->graph_key = key;
        }
    }
    FD_HIP(ctx, hipEventRecord(ctx->ev0, cur_stream(ctx)));
    if (ctx->use_graph && ctx->build_exec) {
        FD_HIP(ctx, hipEventRecord(ctx->ev_mid, cur_stream(ctx)));   // phases are not split inside a graph
        FD_HIP(ctx, hipGraphLaunch(ctx->build_exec, cur_stream(ctx)));
    } else {
        FD_HIP(ctx, launch_prepare(b, cur_stream(ctx), nullptr));
        FD_HIP(ctx, launch_build(b, cur_stream(ctx), ctx->ev_mid));
    }
    FD_HIP(ctx, hipEventRecord(ctx->ev1, cur_stream(ctx)));
    // A later evaluation on ANOTHER stream is ordered behind this build (order_after_batch): a single build as much as
    // a batched one, and in particular the LU rebuild that poll_status enqueues here when the Cholesky lost definiteness --
    // the evaluation that triggered the repair may launch on a stream that knows nothing of this one.
    ctx->wait_event = ctx->ev1; ctx->wait_stream = cur_stream(ctx); ctx->wait_batch = nullptr;
    ctx->tev0 = ctx->ev0; ctx->tev_mid = ctx->ev_mid; ctx->tev1 = ctx->ev1;
    ctx->have_factor = b.ml_layers == 0;     // the multilayer model keeps no single factorisation to reuse
    ctx->last_spd = b.spd != 0;
    ctx->last_reg = b.spd != 0 && b.reg != 0;
    ctx->last_nopivot = b.nopivot != 0;
    ctx->factor_grouped = false;
    ctx->deltas_only = false;
    ctx->build_pending = true;
    ++ctx->model_gen;
    ctx->built = false;
    ctx->have_report = false;
    ctx->sticky_rc = FD_OK;
    return post_status(ctx, cur_stream(ctx));
}

int fd_build_result(fd_ctx *ctx, fd_report *report)
{
    if (!ctx) return FD_E_INVALID;
    int rc = use_device(ctx);
    if (rc) return rc;
    if (ctx->build_pending) {
        if ((rc = order_after_batch(ctx, cur_stream(ctx)))) return rc;
        FD_HIP(ctx, hipMemcpyAsync(ctx->h_model, ctx->d_model, sizeof(DevModel), hipMemcpyDeviceToHost, cur_stream(ctx)));
        FD_HIP(ctx, hipStreamSynchronize(cur_stream(ctx)));
        ctx->status_inflight = false;
        ctx->wait_event = nullptr; ctx->wait_stream = nullptr; ctx->wait_batch = nullptr;   // the batched build it named is complete
        fd_report r{};
        r.terminationtype = ctx->h_model->terminationtype;
        r.iterationscount = ctx->h_model->iterations;
        r.n = order_of(ctx);
        r.solver_used = ctx->last_reg ? FD_SOLVER_REGISTER
                        : ctx->last_spd ? (ctx->solver == FD_SOLVER_ONE_WORKGROUP ? FD_SOLVER_ONE_WORKGROUP : FD_SOLVER_CHAIN)
                        : ctx->last_nopivot ? FD_SOLVER_LU_NOPIVOT : FD_SOLVER_LU;
        double pmin, pmax;
        memcpy(&pmin, &ctx->h_model->pivmin_bits, 8);
        memcpy(&pmax, &ctx->h_model->pivmax_bits, 8);
        r.pivot_ratio = (pmax > 0.0 && pmin <= pmax) ? pmin / pmax : 0.0;
        float ms = 0.f;
        r.fp32_error = ctx->h_model->fp32_error; r.cancellation = ctx->h_model->cancellation;
        r.delta_min = ctx->h_model->delta_min; r.delta_max = ctx->h_model->delta_max; r.extent = ctx->h_model->extent;
        if (ctx->tev0 && ctx->tev_mid && hipEventElapsedTime(&ms, ctx->tev0, ctx->tev_mid) == hipSuccess) r.t_assemble_ms = ms;
        if (ctx->tev_mid && ctx->tev1 && hipEventElapsedTime(&ms, ctx->tev_mid, ctx->tev1) == hipSuccess) r.t_solve_ms = ms;
        ctx->report = r;
        ctx->have_report = true;
        ctx->build_pending = false;
        ctx->built = r.terminationtype == 1;
        // The Cholesky pivot of two nearly coincident centres is the SQUARE of what partial
        // pivoting sees (1e-12 vs 2e-7 of the largest at 1e-7 apart), and a wide fixed-radius
        // Gaussian loses definiteness to rounding before the LU gives up.  Nothing that the LU
        // accepts may fail here: build again with it, and keep it for this rig.
        if (r.terminationtype == -4 && (ctx->last_spd || ctx->last_nopivot) && ctx->points_set) {
            ctx->prefer_lu = true;
            ctx->have_factor = false;
            ctx->deltas_only = false;
            if ((rc = fd_build_async(ctx))) return rc;
            return fd_build_result(ctx, report);
        }
    }
    if (!ctx->have_report) { set_err(ctx, "fd_build_result: no build has been enqueued"); return FD_E_NOT_BUILT; }
    if (report) *report = ctx->report;
    if (ctx->report.terminationtype == 1) return FD_OK;
    if (ctx->report.terminationtype == -5) { set_err(ctx, "fd_build: coincident control points"); return FD_E_DUPLICATE; }
    set_err(ctx, "fd_build: singular system (terminationtype %d)", ctx->report.terminationtype);
    return FD_E_SINGULAR;
}

int fd_fp32_holds(const fd_report *report, double tol)
{
    if (!report || report->terminationtype != 1) return 0;
    if (!(report->delta_max > 0.0)) return 1;             // nothing to measure against (imported model, or no displacement at all)
    const double ulp = 5.9604644775390625e-08 * (report->extent + report->delta_max);
    return report->fp32_error <= tol * 0.5 * report->delta_min + ulp ? 1 : 0;     // (vertices between control points move less than the least of those)
}

int fd_set_eval_precision(fd_ctx *ctx, int eval_precision)
{
    if (!ctx) return FD_E_INVALID;
    if (eval_precision != FD_EVAL_FP32 && eval_precision != FD_EVAL_FP64) { set_err(ctx, "fd_set_eval_precision: unknown precision %d", eval_precision); return FD_E_INVALID; }
    ctx->eval_precision = eval_precision;
    return FD_OK;
}

int fd_set_output(fd_ctx *ctx, int what)
{
    if (!ctx) return FD_E_INVALID;
    if (what != FD_OUTPUT_POSITION && what != FD_OUTPUT_DISPLACEMENT) { set_err(ctx, "fd_set_output: unknown output %d", what); return FD_E_INVALID; }
    ctx->output = what;
    return FD_OK;
}

int fd_build(fd_ctx *ctx, fd_report *report)
{
    int rc = fd_build_async(ctx);
    if (rc) {
        if (report) { memset(report, 0, sizeof(*report)); report->terminationtype = -4; }
        return rc;
    }
    return fd_build_result(ctx, report);
}

}  // extern "C"

// Non-blocking look at the status the last enqueued build posted (see fd_ctx::h_status).
static int poll_status(fd_ctx *ctx)
{
    if (ctx->sticky_rc != FD_OK) {
        set_err(ctx, ctx->sticky_rc == FD_E_DUPLICATE ? "the model's build failed: coincident control points"
                                                      : "the model's build failed: singular system");
        return ctx->sticky_rc;
    }
    if (!ctx->status_inflight) return FD_OK;
    const hipError_t q = hipEventQuery(ctx->status_poll);
    if (q == hipErrorNotReady) { (void)hipGetLastError(); return FD_OK; }     // not known yet: nothing to act on
    ctx->status_inflight = false;
    if (q != hipSuccess) { (void)hipGetLastError(); return FD_OK; }
    const int tt = *ctx->h_status;
    if (tt == 1 || tt == 0) return FD_OK;
    if (tt == -4 && (ctx->last_spd || ctx->last_nopivot) && ctx->points_set) {
        // the Cholesky lost definiteness on this rig (header: FD_SOLVER_AUTO): the LU now, enqueued on
        // the context's stream ahead of whatever the caller is about to enqueue
        ctx->prefer_lu = true;
        ctx->have_factor = false;
        ctx->deltas_only = false;
        // rare path: evaluations of the failed model enqueued earlier (pass-throughs) may still be reading
        // its buffers on streams this context knows nothing about
        if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
        ctx->wait_event = nullptr; ctx->wait_stream = nullptr; ctx->wait_batch = nullptr;
        return fd_build_async(ctx);       // leaves wait_event = the rebuild's end on cur_stream(ctx): callers order their stream behind it
    }
    ctx->sticky_rc = tt == -5 ? FD_E_DUPLICATE : FD_E_SINGULAR;
    set_err(ctx, tt == -5 ? "the model's build failed: coincident control points" : "the model's build failed: singular system");
    return ctx->sticky_rc;
}

extern "C" {

int fd_deform_dev(fd_ctx *ctx, int64_t N, const float *d_P_in, float *d_P_out, const float *d_dist2,
                  float *d_falloff_out, const float *d_tu, const float *d_tv, const float *d_nrm,
                  float radius2, float falloffrate)
{
    if (!ctx) return FD_E_INVALID;
    return fd_deform_dev_stream(ctx, cur_stream(ctx), N, d_P_in, d_P_out, d_dist2, d_falloff_out, d_tu, d_tv,
                                d_nrm, radius2, falloffrate);
}

int fd_deform_dev_stream(fd_ctx *ctx, void *hip_stream, int64_t N, const float *d_P_in, float *d_P_out,
                         const float *d_dist2, float *d_falloff_out, const float *d_tu,
                         const float *d_tv, const float *d_nrm, float radius2, float falloffrate)
{
    if (!ctx) return FD_E_INVALID;
    hipStream_t launch_stream = hip_stream ? (hipStream_t)hip_stream : cur_stream(ctx);
    if (N < 0 || (N > 0 && (!d_P_in || !d_P_out))) { set_err(ctx, "fd_deform: bad N / P pointers"); return FD_E_INVALID; }
    const int ntan = (d_tu != nullptr) + (d_tv != nullptr) + (d_nrm != nullptr);
    if (ntan != 0 && ntan != 3) { set_err(ctx, "fd_deform: tu, tv, nrm must be all set or all NULL"); return FD_E_INVALID; }
    if (!ctx->built && !ctx->build_pending) { set_err(ctx, "fd_deform: no successfully built model"); return FD_E_NOT_BUILT; }
    if (N == 0) return FD_OK;
    int rc = use_device(ctx);
    if (rc) return rc;
    if ((rc = poll_status(ctx))) return rc;           // a failed asynchronous build: repaired here, or reported
    DeformArgs a;
    a.N = N;
    a.P_in = d_P_in; a.P_out = d_P_out;
    a.dist2 = d_dist2; a.falloff_out = d_falloff_out;
    a.tu = d_tu; a.tv = d_tv; a.nrm = d_nrm;
    a.radius2 = radius2; a.falloffrate = falloffrate;
    a.M = model_centres(ctx); a.Mpad = round_up(a.M, kRecPad); a.kind = eval_kind(ctx); a.layers = record_layers(ctx);
    a.rec32 = ctx->d_rec32; a.rec64 = ctx->d_rec64; a.tiles = ctx->d_tiles; a.tiles16 = ctx->d_tiles16;
    a.model = ctx->d_model;
    a.precision = ctx->eval_precision;
    a.variant = ctx->eval_variant;
    a.delta_out = ctx->output == FD_OUTPUT_DISPLACEMENT;
    if ((rc = order_after_batch(ctx, launch_stream))) return rc;
    FD_HIP(ctx, launch_deform(a, launch_stream));
    return FD_OK;
}

int fd_deform(fd_ctx *ctx, int64_t N, const float *P_in, float *P_out, const float *dist2,
              float *falloff_out, const float *tu, const float *tv, const float *nrm, float radius2,
              float falloffrate)
{
    if (!ctx) return FD_E_INVALID;
    if (N < 0 || (N > 0 && (!P_in || !P_out))) { set_err(ctx, "fd_deform: bad N / P pointers"); return FD_E_INVALID; }
    const int ntan = (tu != nullptr) + (tv != nullptr) + (nrm != nullptr);
    if (ntan != 0 && ntan != 3) { set_err(ctx, "fd_deform: tu, tv, nrm must be all set or all NULL"); return FD_E_INVALID; }
    if (!ctx->built && !ctx->build_pending) { set_err(ctx, "fd_deform: no successfully built model"); return FD_E_NOT_BUILT; }
    if (N == 0) return FD_OK;
    int rc = use_device(ctx);
    if (rc) return rc;
    // Page-locked caller arrays (fd_host_alloc, hipHostMalloc, hipHostRegister): the kernel reads
    // and writes them in place over the host link -- reads and writes travel in both directions
    // at once and nothing is staged.  Measured at C2: 0.44 ms against 0.62 ms for upload +
    // evaluate + download (chunking those copies over two streams did not overlap them at all).
    static const bool no_zero_copy = tuning_env("FD_NO_ZEROCOPY") != nullptr;
    const bool all_pinned = host_is_pinned(P_in) && host_is_pinned(P_out) && (!dist2 || host_is_pinned(dist2)) &&
                            (!falloff_out || host_is_pinned(falloff_out)) &&
                            (!tu || (host_is_pinned(tu) && host_is_pinned(tv) && host_is_pinned(nrm)));
    if (all_pinned && !no_zero_copy) {
        bool ok = true;
        auto dp = [&ok](const void *h) -> void * {
            void *d = nullptr;
            if (!h) return nullptr;
            if (hipHostGetDevicePointer(&d, const_cast<void *>(h), 0) != hipSuccess) { (void)hipGetLastError(); ok = false; }
            return d;
        };
        const float *zP = (const float *)dp(P_in), *zD = (const float *)dp(dist2), *zU = (const float *)dp(tu),
                    *zV = (const float *)dp(tv), *zN = (const float *)dp(nrm);
        float *zO = (float *)dp(P_out), *zF = (float *)dp(falloff_out);
        if (ok) {
            hipStream_t s = cur_stream(ctx);
            rc = fd_deform_dev_stream(ctx, s, N, zP, zO, zD, zF, zU, zV, zN, radius2, falloffrate);
            if (rc) return rc;
            FD_HIP(ctx, hipStreamSynchronize(s));
            return FD_OK;
        }
    }
    if (N > ctx->cap_N) {
        if ((rc = dev_alloc(ctx, &ctx->d_P, (size_t)N * 3))) return rc;
        if ((rc = dev_alloc(ctx, &ctx->d_dist2, (size_t)N))) return rc;
        if ((rc = dev_alloc(ctx, &ctx->d_fall, (size_t)N))) return rc;
        // tangent frames are optional and 3x as large: allocate on first use
        if (ctx->d_tu) { (void)hipFree(ctx->d_tu); ctx->d_tu = nullptr; }
        if (ctx->d_tv) { (void)hipFree(ctx->d_tv); ctx->d_tv = nullptr; }
        if (ctx->d_nrm) { (void)hipFree(ctx->d_nrm); ctx->d_nrm = nullptr; }
        ctx->cap_N = N;
    }
    if (tu && !ctx->d_tu) {
        if ((rc = dev_alloc(ctx, &ctx->d_tu, (size_t)ctx->cap_N * 3))) return rc;
        if ((rc = dev_alloc(ctx, &ctx->d_tv, (size_t)ctx->cap_N * 3))) return rc;
        if ((rc = dev_alloc(ctx, &ctx->d_nrm, (size_t)ctx->cap_N * 3))) return rc;
    }
    // A vertex whose gate fails keeps the caller's fd_falloff entry, so that array goes up as
    // well -- unless no vertex can be gated (no dist2 and a non-negative radius^2).
    const bool fall_up = falloff_out && (dist2 || radius2 < 0.f);
    hipStream_t s = cur_stream(ctx);
    const size_t b3 = sizeof(float) * 3 * (size_t)N, b1 = sizeof(float) * (size_t)N;
    FD_HIP(ctx, hipMemcpyAsync(ctx->d_P, P_in, b3, hipMemcpyHostToDevice, s));
    if (dist2) FD_HIP(ctx, hipMemcpyAsync(ctx->d_dist2, dist2, b1, hipMemcpyHostToDevice, s));
    if (fall_up) FD_HIP(ctx, hipMemcpyAsync(ctx->d_fall, falloff_out, b1, hipMemcpyHostToDevice, s));
    if (tu) {
        FD_HIP(ctx, hipMemcpyAsync(ctx->d_tu, tu, b3, hipMemcpyHostToDevice, s));
        FD_HIP(ctx, hipMemcpyAsync(ctx->d_tv, tv, b3, hipMemcpyHostToDevice, s));
        FD_HIP(ctx, hipMemcpyAsync(ctx->d_nrm, nrm, b3, hipMemcpyHostToDevice, s));
    }
    rc = fd_deform_dev(ctx, N, ctx->d_P, ctx->d_P, dist2 ? ctx->d_dist2 : nullptr,
                       falloff_out ? ctx->d_fall : nullptr, tu ? ctx->d_tu : nullptr,
                       tu ? ctx->d_tv : nullptr, tu ? ctx->d_nrm : nullptr, radius2, falloffrate);
    if (rc) return rc;
    FD_HIP(ctx, hipMemcpyAsync(P_out, ctx->d_P, b3, hipMemcpyDeviceToHost, s));
    if (falloff_out) FD_HIP(ctx, hipMemcpyAsync(falloff_out, ctx->d_fall, b1, hipMemcpyDeviceToHost, s));
    FD_HIP(ctx, hipStreamSynchronize(s));
    return FD_OK;
}

// ---- dist2 producer (next row N2; kernel in fd_capture.hip) -----------------------------------
int fd_capture_dist2_dev(fd_ctx *ctx, int64_t N, const float *d_P, const unsigned char *d_mask, int T,
                         const float *d_tri_xyz, float radius2, int dofalloff, float *d_dist2)
{
    if (!ctx) return FD_E_INVALID;
    if (N < 0 || T < 0 || (N > 0 && (!d_P || !d_dist2)) || (T > 0 && !d_tri_xyz)) {
        set_err(ctx, "fd_capture_dist2: bad sizes or NULL arrays");
        return FD_E_INVALID;
    }
    int rc = use_device(ctx);
    if (rc) return rc;
    FD_HIP(ctx, launch_capture_dist2(d_P, N, d_mask, d_tri_xyz, T, radius2, dofalloff, d_dist2, cur_stream(ctx)));
    return FD_OK;
}

int fd_capture_dist2(fd_ctx *ctx, int64_t N, const float *P, const unsigned char *mask, int T, const float *tri_xyz,
                     float radius2, int dofalloff, float *dist2)
{
    if (!ctx) return FD_E_INVALID;
    if (N < 0 || T < 0 || (N > 0 && (!P || !dist2)) || (T > 0 && !tri_xyz)) {
        set_err(ctx, "fd_capture_dist2: bad sizes or NULL arrays");
        return FD_E_INVALID;
    }
    if (N == 0) return FD_OK;
    int rc = use_device(ctx);
    if (rc) return rc;
    hipStream_t s = cur_stream(ctx);
    float *d_P = nullptr, *d_tri = nullptr, *d_out = nullptr;
    unsigned char *d_mask = nullptr;
    auto cleanup = [&]() {
        if (d_P) (void)hipFree(d_P);
        if (d_tri) (void)hipFree(d_tri);
        if (d_out) (void)hipFree(d_out);
        if (d_mask) (void)hipFree(d_mask);
    };
    hipError_t e = hipMalloc((void **)&d_P, sizeof(float) * 3 * (size_t)N);
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, sizeof(float) * (size_t)N);
    if (e == hipSuccess && T > 0) e = hipMalloc((void **)&d_tri, sizeof(float) * 9 * (size_t)T);
    if (e == hipSuccess && mask) e = hipMalloc((void **)&d_mask, (size_t)N);
    if (e == hipSuccess) e = hipMemcpyAsync(d_P, P, sizeof(float) * 3 * (size_t)N, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && T > 0) e = hipMemcpyAsync(d_tri, tri_xyz, sizeof(float) * 9 * (size_t)T, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && mask) e = hipMemcpyAsync(d_mask, mask, (size_t)N, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = launch_capture_dist2(d_P, N, d_mask, d_tri, T, radius2, dofalloff, d_out, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dist2, d_out, sizeof(float) * (size_t)N, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    cleanup();
    if (e != hipSuccess) { set_err(ctx, "fd_capture_dist2 failed: %s", hipGetErrorString(e)); return FD_E_DEVICE; }
    return FD_OK;
}

int fd_capture_islands_dev(fd_ctx *ctx, int64_t N, const float *d_P, const int64_t *d_offsets, const int *d_neighbours,
                           int M, const float *d_rig_xyz, int max_edges, unsigned char *d_mask)
{
    if (!ctx) return FD_E_INVALID;
    if (N < 0 || M < 0 || max_edges < 0 || (N > 0 && (!d_P || !d_offsets || !d_mask)) || (M > 0 && !d_rig_xyz)) {
        set_err(ctx, "fd_capture_islands: bad sizes or NULL arrays");
        return FD_E_INVALID;
    }
    int rc = use_device(ctx);
    if (rc) return rc;
    FD_HIP(ctx, launch_capture_islands(d_P, N, d_offsets, d_neighbours, d_rig_xyz, M, max_edges, d_mask, cur_stream(ctx)));
    return FD_OK;
}

int fd_capture_islands(fd_ctx *ctx, int64_t N, const float *P, const int64_t *offsets, const int *neighbours, int M,
                       const float *rig_xyz, int max_edges, unsigned char *mask)
{
    if (!ctx) return FD_E_INVALID;
    if (N < 0 || M < 0 || max_edges < 0 || (N > 0 && (!P || !offsets || !mask)) || (M > 0 && !rig_xyz)) {
        set_err(ctx, "fd_capture_islands: bad sizes or NULL arrays");
        return FD_E_INVALID;
    }
    if (N == 0) return FD_OK;
    const int64_t E = offsets[N];
    if (E < 0 || (E > 0 && !neighbours)) { set_err(ctx, "fd_capture_islands: bad adjacency"); return FD_E_INVALID; }
    int rc = use_device(ctx);
    if (rc) return rc;
    hipStream_t s = cur_stream(ctx);
    float *d_P = nullptr, *d_rig = nullptr;
    int64_t *d_off = nullptr;
    int *d_nb = nullptr;
    unsigned char *d_mask = nullptr;
    hipError_t e = hipMalloc((void **)&d_P, sizeof(float) * 3 * (size_t)N);
    if (e == hipSuccess) e = hipMalloc((void **)&d_off, sizeof(int64_t) * (size_t)(N + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&d_nb, sizeof(int) * (size_t)(E > 0 ? E : 1));
    if (e == hipSuccess) e = hipMalloc((void **)&d_mask, (size_t)N);
    if (e == hipSuccess && M > 0) e = hipMalloc((void **)&d_rig, sizeof(float) * 3 * (size_t)M);
    if (e == hipSuccess) e = hipMemcpyAsync(d_P, P, sizeof(float) * 3 * (size_t)N, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, offsets, sizeof(int64_t) * (size_t)(N + 1), hipMemcpyHostToDevice, s);
    if (e == hipSuccess && E > 0) e = hipMemcpyAsync(d_nb, neighbours, sizeof(int) * (size_t)E, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && M > 0) e = hipMemcpyAsync(d_rig, rig_xyz, sizeof(float) * 3 * (size_t)M, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = launch_capture_islands(d_P, N, d_off, d_nb, d_rig, M, max_edges, d_mask, s);
    if (e == hipSuccess) e = hipMemcpyAsync(mask, d_mask, (size_t)N, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    for (void *p : {(void *)d_P, (void *)d_off, (void *)d_nb, (void *)d_mask, (void *)d_rig}) if (p) (void)hipFree(p);
    if (e != hipSuccess) { set_err(ctx, "fd_capture_islands failed: %s", hipGetErrorString(e)); return FD_E_DEVICE; }
    return FD_OK;
}

void *fd_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        set_err(nullptr, "fd_host_alloc(%zu) failed: %s", bytes, hipGetErrorString(hipGetLastError()));
        return nullptr;
    }
    return p;
}

void fd_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

// ---- device-resident mesh (next row N3, engine side) ----------------------------------------
int fd_mesh_set(fd_ctx *ctx, int64_t N, const float *P, const float *dist2, const float *tu, const float *tv,
                const float *nrm)
{
    if (!ctx) return FD_E_INVALID;
    if (N <= 0 || !P) { set_err(ctx, "fd_mesh_set: need N > 0 and P"); return FD_E_INVALID; }
    const int ntan = (tu != nullptr) + (tv != nullptr) + (nrm != nullptr);
    if (ntan != 0 && ntan != 3) { set_err(ctx, "fd_mesh_set: tu, tv, nrm must be all set or all NULL"); return FD_E_INVALID; }
    int rc = use_device(ctx);
    if (rc) return rc;
    ctx->mesh_N = 0;
    if (N > ctx->mesh_cap) {
        if ((rc = dev_alloc(ctx, &ctx->m_P, (size_t)N * 3))) return rc;
        if (ctx->m_dist2) { (void)hipFree(ctx->m_dist2); ctx->m_dist2 = nullptr; }
        if (ctx->m_tu) { (void)hipFree(ctx->m_tu); ctx->m_tu = nullptr; }
        if (ctx->m_tv) { (void)hipFree(ctx->m_tv); ctx->m_tv = nullptr; }
        if (ctx->m_nrm) { (void)hipFree(ctx->m_nrm); ctx->m_nrm = nullptr; }
        if (ctx->m_out) { (void)hipFree(ctx->m_out); ctx->m_out = nullptr; }
        if (ctx->m_fall) { (void)hipFree(ctx->m_fall); ctx->m_fall = nullptr; }
        ctx->mesh_cap = N;
    }
    if (dist2 && !ctx->m_dist2 && (rc = dev_alloc(ctx, &ctx->m_dist2, (size_t)ctx->mesh_cap))) return rc;
    if (tu && !ctx->m_tu) {
        if ((rc = dev_alloc(ctx, &ctx->m_tu, (size_t)ctx->mesh_cap * 3))) return rc;
        if ((rc = dev_alloc(ctx, &ctx->m_tv, (size_t)ctx->mesh_cap * 3))) return rc;
        if ((rc = dev_alloc(ctx, &ctx->m_nrm, (size_t)ctx->mesh_cap * 3))) return rc;
    }
    hipStream_t s = cur_stream(ctx);
    const size_t b3 = sizeof(float) * 3 * (size_t)N, b1 = sizeof(float) * (size_t)N;
    FD_HIP(ctx, hipMemcpyAsync(ctx->m_P, P, b3, hipMemcpyHostToDevice, s));
    if (dist2) FD_HIP(ctx, hipMemcpyAsync(ctx->m_dist2, dist2, b1, hipMemcpyHostToDevice, s));
    if (tu) {
        FD_HIP(ctx, hipMemcpyAsync(ctx->m_tu, tu, b3, hipMemcpyHostToDevice, s));
        FD_HIP(ctx, hipMemcpyAsync(ctx->m_tv, tv, b3, hipMemcpyHostToDevice, s));
        FD_HIP(ctx, hipMemcpyAsync(ctx->m_nrm, nrm, b3, hipMemcpyHostToDevice, s));
    }
    FD_HIP(ctx, hipStreamSynchronize(s));      // the caller's arrays may change after this returns
    ctx->mesh_N = N;
    ctx->mesh_has_dist2 = dist2 != nullptr;
    ctx->mesh_has_frames = tu != nullptr;
    return FD_OK;
}

int64_t fd_mesh_size(const fd_ctx *ctx) { return ctx ? ctx->mesh_N : 0; }

int fd_deform_mesh(fd_ctx *ctx, float *P_out, float *falloff_out, float radius2, float falloffrate)
{
    if (!ctx) return FD_E_INVALID;
    if (ctx->mesh_N <= 0) { set_err(ctx, "fd_deform_mesh: fd_mesh_set has not been called"); return FD_E_INVALID; }
    if (!P_out) { set_err(ctx, "fd_deform_mesh: P_out is NULL"); return FD_E_INVALID; }
    if (!ctx->built && !ctx->build_pending) { set_err(ctx, "fd_deform_mesh: no successfully built model"); return FD_E_NOT_BUILT; }
    int rc = use_device(ctx);
    if (rc) return rc;
    const int64_t N = ctx->mesh_N;
    hipStream_t s = cur_stream(ctx);
    const float *d2 = ctx->mesh_has_dist2 ? ctx->m_dist2 : nullptr;
    const float *tu = ctx->mesh_has_frames ? ctx->m_tu : nullptr, *tv = ctx->mesh_has_frames ? ctx->m_tv : nullptr,
                *nr = ctx->mesh_has_frames ? ctx->m_nrm : nullptr;
    // page-locked outputs: the kernel reads the mesh from HBM and writes the results straight
    // into the caller's arrays -- the only traffic on the host link is the result itself
    static const bool no_zero_copy = tuning_env("FD_NO_ZEROCOPY") != nullptr;
    if (!no_zero_copy && host_is_pinned(P_out) && (!falloff_out || host_is_pinned(falloff_out))) {
        void *zo = nullptr, *zf = nullptr;
        bool ok = hipHostGetDevicePointer(&zo, P_out, 0) == hipSuccess;
        if (ok && falloff_out) ok = hipHostGetDevicePointer(&zf, falloff_out, 0) == hipSuccess;
        if (ok) {
            rc = fd_deform_dev_stream(ctx, s, N, ctx->m_P, (float *)zo, d2, (float *)zf, tu, tv, nr, radius2, falloffrate);
            if (rc) return rc;
            FD_HIP(ctx, hipStreamSynchronize(s));
            return FD_OK;
        }
        (void)hipGetLastError();
    }
    if (!ctx->m_out && (rc = dev_alloc(ctx, &ctx->m_out, (size_t)ctx->mesh_cap * 3))) return rc;
    if (falloff_out && !ctx->m_fall && (rc = dev_alloc(ctx, &ctx->m_fall, (size_t)ctx->mesh_cap))) return rc;
    const size_t b3 = sizeof(float) * 3 * (size_t)N, b1 = sizeof(float) * (size_t)N;
    // a gated vertex keeps the caller's fd_falloff entry (see fd_deform)
    if (falloff_out && (d2 || radius2 < 0.f)) FD_HIP(ctx, hipMemcpyAsync(ctx->m_fall, falloff_out, b1, hipMemcpyHostToDevice, s));
    rc = fd_deform_dev_stream(ctx, s, N, ctx->m_P, ctx->m_out, d2, falloff_out ? ctx->m_fall : nullptr, tu, tv, nr, radius2,
                              falloffrate);
    if (rc) return rc;
    FD_HIP(ctx, hipMemcpyAsync(P_out, ctx->m_out, b3, hipMemcpyDeviceToHost, s));
    if (falloff_out) FD_HIP(ctx, hipMemcpyAsync(falloff_out, ctx->m_fall, b1, hipMemcpyDeviceToHost, s));
    FD_HIP(ctx, hipStreamSynchronize(s));
    return FD_OK;
}

// ProximityCapture on the device-resident mesh: islands, then squared distances, into m_dist2
int fd_mesh_capture(fd_ctx *ctx, const int64_t *offsets, const int *neighbours, int M, const float *rig_xyz,
                    int max_edges, int T, const float *tri_xyz, float radius2, int dofalloff, float *dist2_out)
{
    if (!ctx) return FD_E_INVALID;
    const int64_t N = ctx->mesh_N;
    if (N <= 0) { set_err(ctx, "fd_mesh_capture: fd_mesh_set has not been called"); return FD_E_INVALID; }
    if (!offsets || M < 0 || T < 0 || max_edges < 0 || (M > 0 && !rig_xyz) || (T > 0 && !tri_xyz)) {
        set_err(ctx, "fd_mesh_capture: bad sizes or NULL arrays");
        return FD_E_INVALID;
    }
    const int64_t E = offsets[N];
    if (offsets[0] != 0 || E < 0 || (E > 0 && !neighbours)) { set_err(ctx, "fd_mesh_capture: bad adjacency"); return FD_E_INVALID; }
    int rc = use_device(ctx);
    if (rc) return rc;
    if (!ctx->m_dist2 && (rc = dev_alloc(ctx, &ctx->m_dist2, (size_t)ctx->mesh_cap))) return rc;
    hipStream_t s = cur_stream(ctx);
    float *d_rig = nullptr, *d_tri = nullptr;
    int64_t *d_off = nullptr;
    int *d_nb = nullptr;
    unsigned char *d_mask = nullptr;
    hipError_t e = hipMalloc((void **)&d_off, sizeof(int64_t) * (size_t)(N + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&d_nb, sizeof(int) * (size_t)(E > 0 ? E : 1));
    if (e == hipSuccess) e = hipMalloc((void **)&d_mask, (size_t)N);
    if (e == hipSuccess) e = hipMalloc((void **)&d_rig, sizeof(float) * 3 * (size_t)(M > 0 ? M : 1));
    if (e == hipSuccess) e = hipMalloc((void **)&d_tri, sizeof(float) * 9 * (size_t)(T > 0 ? T : 1));
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, offsets, sizeof(int64_t) * (size_t)(N + 1), hipMemcpyHostToDevice, s);
    if (e == hipSuccess && E > 0) e = hipMemcpyAsync(d_nb, neighbours, sizeof(int) * (size_t)E, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && M > 0) e = hipMemcpyAsync(d_rig, rig_xyz, sizeof(float) * 3 * (size_t)M, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && T > 0) e = hipMemcpyAsync(d_tri, tri_xyz, sizeof(float) * 9 * (size_t)T, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = launch_capture_islands(ctx->m_P, N, d_off, d_nb, d_rig, M, max_edges, d_mask, s);
    if (e == hipSuccess) e = launch_capture_dist2(ctx->m_P, N, d_mask, d_tri, T, radius2, dofalloff, ctx->m_dist2, s);
    if (e == hipSuccess && dist2_out) e = hipMemcpyAsync(dist2_out, ctx->m_dist2, sizeof(float) * (size_t)N, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    for (void *p : {(void *)d_off, (void *)d_nb, (void *)d_mask, (void *)d_rig, (void *)d_tri}) if (p) (void)hipFree(p);
    if (e != hipSuccess) { (void)hipGetLastError(); set_err(ctx, "fd_mesh_capture failed: %s", hipGetErrorString(e)); return FD_E_DEVICE; }
    ctx->mesh_has_dist2 = true;
    return FD_OK;
}

int fd_mesh_get_dist2(fd_ctx *ctx, float *dist2_out)
{
    if (!ctx || !dist2_out) return FD_E_INVALID;
    if (ctx->mesh_N <= 0 || !ctx->mesh_has_dist2 || !ctx->m_dist2) { set_err(ctx, "fd_mesh_get_dist2: the mesh has no dist2 array"); return FD_E_INVALID; }
    int rc = use_device(ctx);
    if (rc) return rc;
    hipStream_t s = cur_stream(ctx);
    FD_HIP(ctx, hipMemcpyAsync(dist2_out, ctx->m_dist2, sizeof(float) * (size_t)ctx->mesh_N, hipMemcpyDeviceToHost, s));
    FD_HIP(ctx, hipStreamSynchronize(s));
    return FD_OK;
}

static int require_built(fd_ctx *ctx, const char *who)
{
    if (ctx->build_pending) {
        int rc = fd_build_result(ctx, nullptr);
        if (rc) return rc;
    }
    if (!ctx->built) { set_err(ctx, "%s: no successfully built model", who); return FD_E_NOT_BUILT; }
    return FD_OK;
}

int fd_model_centres(const fd_ctx *ctx) { return ctx ? model_centres(ctx) : 0; }

int fd_get_weights(fd_ctx *ctx, double *W, double *radii)
{
    if (!ctx || !W) return FD_E_INVALID;
    int rc = use_device(ctx);
    if (rc) return rc;
    if ((rc = require_built(ctx, "fd_get_weights"))) return rc;
    const int n = model_centres(ctx);
    FD_HIP(ctx, hipMemcpyAsync(W, ctx->d_W, sizeof(double) * 3 * (size_t)(n + 4), hipMemcpyDeviceToHost, cur_stream(ctx)));
    if (radii) FD_HIP(ctx, hipMemcpyAsync(radii, ctx->d_radii, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, cur_stream(ctx)));
    FD_HIP(ctx, hipStreamSynchronize(cur_stream(ctx)));
    return FD_OK;
}

static size_t model_bytes_for(int M)
{
    return sizeof(ModelHeader) + sizeof(double) * ((size_t)M * 3 + (size_t)M + (size_t)(M + 4) * 3);
}

size_t fd_model_bytes(const fd_ctx *ctx) { return ctx ? model_bytes_for(model_centres(ctx)) : 0; }

int fd_export_model(fd_ctx *ctx, void *buf, size_t capacity, int on_device)
{
    if (!ctx || !buf) return FD_E_INVALID;
    int rc = use_device(ctx);
    if (rc) return rc;
    if ((rc = require_built(ctx, "fd_export_model"))) return rc;
    // a multilayer model travels as what it is once solved: M * layers Gaussians with their own radii
    const int M = model_centres(ctx);
    if (capacity < model_bytes_for(M)) { set_err(ctx, "fd_export_model: buffer too small"); return FD_E_INVALID; }
    // the pinned header may still be the source of an earlier in-flight copy
    FD_HIP(ctx, hipStreamSynchronize(cur_stream(ctx)));
    ModelHeader *h = ctx->h_header;
    memset(h, 0, sizeof(*h));
    h->magic = kModelMagic;
    h->M = M; h->kind = eval_kind(ctx); h->term = ctx->term; h->nparams = ctx->nparams;
    h->terminationtype = 1;
    h->layers = record_layers(ctx);
    memcpy(h->params, ctx->params, sizeof(h->params));
    h->rig_token = (uint64_t)(uintptr_t)ctx->rest_src;
    char *p = (char *)buf;
    const hipMemcpyKind kh = on_device ? hipMemcpyHostToDevice : hipMemcpyHostToHost;
    const hipMemcpyKind kd = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    FD_HIP(ctx, hipMemcpyAsync(p, h, sizeof(*h), kh, cur_stream(ctx)));
    p += sizeof(*h);
    FD_HIP(ctx, hipMemcpyAsync(p, ctx->d_centres, sizeof(double) * 3 * (size_t)M, kd, cur_stream(ctx)));
    p += sizeof(double) * 3 * (size_t)M;
    FD_HIP(ctx, hipMemcpyAsync(p, ctx->d_radii, sizeof(double) * (size_t)M, kd, cur_stream(ctx)));
    p += sizeof(double) * (size_t)M;
    FD_HIP(ctx, hipMemcpyAsync(p, ctx->d_W, sizeof(double) * 3 * (size_t)(M + 4), kd, cur_stream(ctx)));
    if (!on_device) FD_HIP(ctx, hipStreamSynchronize(cur_stream(ctx)));
    return FD_OK;
}

int fd_import_model(fd_ctx *ctx, const void *buf, size_t bytes, int on_device)
{
    if (!ctx || !buf || bytes < sizeof(ModelHeader)) return FD_E_INVALID;
    int rc = use_device(ctx);
    if (rc) return rc;
    ModelHeader h;
    if (on_device) {
        FD_HIP(ctx, hipMemcpyAsync(ctx->h_header, buf, sizeof(h), hipMemcpyDeviceToHost, cur_stream(ctx)));
        FD_HIP(ctx, hipStreamSynchronize(cur_stream(ctx)));
        h = *ctx->h_header;
    } else {
        memcpy(&h, buf, sizeof(h));
    }
    if (h.magic != kModelMagic || h.M <= 0 || h.M > kMaxOrder * kMaxLayers || h.kind < 0 || h.kind > FD_KERNEL_CUBIC ||
        h.term < 0 || h.term > 2 || h.terminationtype != 1 || bytes < model_bytes_for(h.M) ||
        h.layers < 0 || h.layers > kMaxLayers || (h.layers > 1 && h.M % h.layers != 0)) {
        set_err(ctx, "fd_import_model: not a valid model blob");
        return FD_E_INVALID;
    }
    const int M = h.M;
    if ((rc = ensure_model_capacity(ctx, M))) return rc;
    ctx->M = M; ctx->kind = h.kind; ctx->term = h.term; ctx->nparams = h.nparams;
    ctx->imported_layers = h.layers > 1 ? h.layers : 0;
    memcpy(ctx->params, h.params, sizeof(h.params));
    const char *p = (const char *)buf + sizeof(ModelHeader);
    const hipMemcpyKind kd = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    FD_HIP(ctx, hipMemcpyAsync(ctx->d_centres, p, sizeof(double) * 3 * (size_t)M, kd, cur_stream(ctx)));
    p += sizeof(double) * 3 * (size_t)M;
    FD_HIP(ctx, hipMemcpyAsync(ctx->d_radii, p, sizeof(double) * (size_t)M, kd, cur_stream(ctx)));
    p += sizeof(double) * (size_t)M;
    FD_HIP(ctx, hipMemcpyAsync(ctx->d_W, p, sizeof(double) * 3 * (size_t)(M + 4), kd, cur_stream(ctx)));
    if ((rc = sync_slot(ctx))) return rc;
    BuildBuffers b;
    fill_build_buffers(ctx, b);
    FD_HIP(ctx, launch_pack_from_weights(b, cur_stream(ctx), ctx->imported_layers));
    FD_HIP(ctx, hipEventRecord(ctx->ev1, cur_stream(ctx)));
    ctx->wait_event = ctx->ev1; ctx->wait_stream = cur_stream(ctx); ctx->wait_batch = nullptr;      // evaluations on other streams wait for the pack
    if (!on_device) FD_HIP(ctx, hipStreamSynchronize(cur_stream(ctx)));
    ctx->have_factor = false;
    ctx->deltas_only = false;
    ++ctx->model_gen;
    ctx->points_set = false;   // no rest/delta on this context: it can deform, not rebuild
    // never dereferenced, only compared: bit 0 marks an imported identity (arrays are at least 4-byte aligned), so that it
    // can only ever equal another imported model's
    ctx->rig_build_id = 0;
    ctx->rest_src = h.rig_token ? (const float *)(uintptr_t)(h.rig_token | 1u) : nullptr;
    ctx->build_pending = false;
    ctx->built = true;
    fd_report r{};
    r.terminationtype = 1;
    r.n = order_of(ctx);
    ctx->report = r;
    ctx->have_report = true;
    return FD_OK;
}

// ---- batched build ----------------------------------------------------------------
// Several contexts with the same M, kernel, parameters and term -- frames of one rig, or the
// nodes of one cook graph -- are assembled and factorised by ONE launch chain: grid z selects
// the context.  A lone build of this size keeps one CU busy through ~25 dependent launches and
// the device overlaps only two or three such chains, so throughput-oriented callers batch.
static void batch_err(fd_batch *b, const char *fmt, ...)
{
    char *dst = b ? b->err : g_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
}

fd_batch *fd_batch_create(fd_ctx *const *ctxs, int n)
{
    if (!ctxs || n <= 0 || n > kMaxBatch) { set_err(nullptr, "fd_batch_create: need 1..%d contexts", kMaxBatch); return nullptr; }
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i] || ctxs[i]->device != ctxs[0]->device) { set_err(nullptr, "fd_batch_create: contexts must exist and share one device"); return nullptr; }
        for (int j = 0; j < i; ++j)
            if (ctxs[j] == ctxs[i]) { set_err(nullptr, "fd_batch_create: context listed twice"); return nullptr; }
    }
    fd_batch *b = new (std::nothrow) fd_batch();
    if (!b) { set_err(nullptr, "fd_batch_create: out of host memory"); return nullptr; }
    b->n = n;
    b->device = ctxs[0]->device;
    for (int i = 0; i < n; ++i) b->ctxs[i] = ctxs[i];
    bool ok = hipSetDevice(b->device) == hipSuccess;
    ok = ok && hipMalloc((void **)&b->d_slots, sizeof(BatchSlot) * (size_t)n) == hipSuccess;
    ok = ok && hipEventCreate(&b->ev0) == hipSuccess && hipEventCreate(&b->ev_mid) == hipSuccess &&
         hipEventCreate(&b->ev1) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&b->status_ev, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&b->h_mismatch, sizeof(int), hipHostMallocMapped) == hipSuccess;
    if (ok) *b->h_mismatch = 0;
    if (!ok) {
        set_err(nullptr, "fd_batch_create: device resource allocation failed: %s", hipGetErrorString(hipGetLastError()));
        fd_batch_destroy(b);
        return nullptr;
    }
    b->use_graph = tuning_env("FD_NO_GRAPH") == nullptr;
    return b;
}

void fd_batch_destroy(fd_batch *b)
{
    if (!b) return;
    (void)hipSetDevice(b->device);
    for (int i = 0; i < b->n; ++i) {
        fd_ctx *c = b->ctxs[i];
        if (c && c->wait_event == b->ev1) {
            if (c->wait_stream) (void)hipStreamSynchronize(c->wait_stream);
            c->wait_event = nullptr; c->wait_stream = nullptr; c->wait_batch = nullptr;
        }
        if (c && c->tev0 == b->ev0) { c->tev0 = c->ev0; c->tev_mid = c->ev_mid; c->tev1 = c->ev1; }
        if (c && (c->status_poll == b->status_ev || c->status_poll == b->ev1)) { c->status_poll = c->status_ev; c->status_inflight = false; }
    }
    if (b->status_ev) (void)hipEventDestroy(b->status_ev);
    if (b->exec) (void)hipGraphExecDestroy(b->exec);
    for (hipEvent_t e : b->lu_events) if (e) (void)hipEventDestroy(e);
    if (b->lu_stream) (void)hipStreamDestroy(b->lu_stream);
    if (b->d_slots) (void)hipFree(b->d_slots);
    for (auto &st : b->sets) {
        if (st.d_wtiles) (void)hipFree(st.d_wtiles);
        if (st.d_frames) (void)hipFree(st.d_frames);
        if (st.packed_ev) (void)hipEventDestroy(st.packed_ev);
        if (st.eval_ev) (void)hipEventDestroy(st.eval_ev);
    }
    if (b->d_fac) (void)hipFree(b->d_fac);
    if (b->fallback_ev) (void)hipEventDestroy(b->fallback_ev);
    if (b->group_ev) (void)hipEventDestroy(b->group_ev);
    if (b->h_mismatch) (void)hipHostFree(b->h_mismatch);
    if (b->ev0) (void)hipEventDestroy(b->ev0);
    if (b->ev_mid) (void)hipEventDestroy(b->ev_mid);
    if (b->ev1) (void)hipEventDestroy(b->ev1);
    delete b;
}

const char *fd_batch_last_error(const fd_batch *b) { return b ? b->err : g_err; }

int fd_batch_set_points_dev(fd_batch *b, const float *const *d_rest_xyz, const float *const *d_delta_xyz, int M)
{
    if (!b || !d_rest_xyz || !d_delta_xyz) return FD_E_INVALID;
    if (M <= 0 || M + 4 > kMaxOrder) { batch_err(b, "fd_batch_set_points_dev: M = %d outside 1..%d", M, kMaxOrder - 4); return FD_E_INVALID; }
    b->prepared = false;
    if (b->h_mismatch && *b->h_mismatch != 0) {
        // The pack kernel of the batch's PREVIOUS evaluation found a context built on other rest points than context 0 (same
        // address, other contents) and passed its frame through.  It posts the word asynchronously; a pipeline comes through
        // here before it ever polls (the word was cleared unseen): report it now, once, then start clean.
        const int who = *b->h_mismatch - 1;
        *b->h_mismatch = 0;
        batch_err(b, "shared-rig evaluation of the previous group: context %d was built on other rest points than context 0 (same "
                     "address, other contents); its frame was passed through", who);
        return FD_E_INVALID;
    }
    for (int i = 0; i < b->n; ++i)
        if (!d_rest_xyz[i] || !d_delta_xyz[i]) { batch_err(b, "fd_batch_set_points_dev: null array for context %d", i); return FD_E_INVALID; }
    for (int i = 0; i < b->n; ++i) {
        fd_ctx *c = b->ctxs[i];
        int rc = use_device(c);
        if (!rc) rc = ensure_model_capacity(c, M);
        if (rc) { batch_err(b, "context %d: %s", i, c->err); return rc; }
        c->M = M;
        c->points_set = true;
        c->built = false;
        c->build_pending = false;
        c->have_factor = false;
        c->deltas_only = false;
        c->rest_src = d_rest_xyz[i];
        c->rig_build_id = 0;
        b->src.rest[i] = d_rest_xyz[i];
        b->src.delta[i] = d_delta_xyz[i];
    }
    b->have_src = true;
    return FD_OK;
}

int fd_batch_build_async(fd_batch *b, void *hip_stream)
{
    if (!b) return FD_E_INVALID;
    b->prepared = false;
    fd_ctx *c0 = b->ctxs[0];
    int rc = use_device(c0);
    if (rc) { batch_err(b, "%s", c0->err); return rc; }
    hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : cur_stream(c0);
    for (int i = 0; i < b->n; ++i) {
        fd_ctx *c = b->ctxs[i];
        if (!c->points_set) { batch_err(b, "fd_batch_build: context %d has no control points", i); return FD_E_INVALID; }
        c->imported_layers = 0;
        if (c->M != c0->M || c->kind != c0->kind || c->term != c0->term || c->nparams != c0->nparams ||
            memcmp(c->params, c0->params, sizeof(c->params)) != 0) {
            batch_err(b, "fd_batch_build: context %d differs from context 0 in M, kernel, parameters or term", i);
            return FD_E_INVALID;
        }
    }
    const int npad = round_up(order_of(c0), 32);
    bool table_stale = false;
    for (int i = 0; i < b->n; ++i) {
        fd_ctx *c = b->ctxs[i];
        const bool grew = npad > c->cap_npad;
        if ((rc = ensure_solver_capacity(c, npad)) || (rc = ensure_records_capacity(c, model_centres(c))) || (rc = sync_slot(c))) {
            batch_err(b, "context %d: %s", i, c->err);
            return rc;
        }
        if (grew) {
            const size_t cols = (size_t)c->cap_npad + kRhsCols + 16;
            hipError_t e = hipMemsetAsync(c->d_A, 0, sizeof(double) * (size_t)c->cap_npad * cols, stream);
            if (e != hipSuccess) { batch_err(b, "hipMemsetAsync failed: %s", hipGetErrorString(e)); return FD_E_DEVICE; }
        }
        if (b->gens[i] != c->alloc_gen) table_stale = true;
    }
    if (table_stale) {
        BatchSlot tab[kMaxBatch];
        for (int i = 0; i < b->n; ++i) { tab[i] = b->ctxs[i]->h_slot; b->gens[i] = b->ctxs[i]->alloc_gen; }
        // buffers only move through hipFree, which has drained the device: nobody reads the old table
        hipError_t e = hipMemcpy(b->d_slots, tab, sizeof(BatchSlot) * (size_t)b->n, hipMemcpyHostToDevice);
        if (e != hipSuccess) { batch_err(b, "slot table upload failed: %s", hipGetErrorString(e)); return FD_E_DEVICE; }
    }
    BuildBuffers bb;
    fill_build_buffers(c0, bb);
    bb.d_slots = b->d_slots;
    bb.nbatch = b->n;
    for (int i = 0; i < b->n; ++i)
        if (!use_spd(b->ctxs[i])) bb.spd = 0;        // one context that fell back to the LU takes the batch with it
    for (int i = 0; i < b->n; ++i)
        if (b->ctxs[i]->solver != FD_SOLVER_ONE_WORKGROUP) bb.small = 0;      // ... and the one-workgroup build is everybody's choice or nobody's
    for (int i = 0; i < b->n; ++i)
        if (!use_reg(b->ctxs[i])) bb.reg = 0;                                  // ... and so is the register-resident one
    if (!bb.spd) bb.reg = 0;
    for (int i = 0; i < b->n; ++i)
        if (!use_nopivot(b->ctxs[i])) bb.nopivot = 0;
    if (bb.reg && reg_build_init() != hipSuccess) { (void)hipGetLastError(); bb.reg = 0; }
    // a caller that leaves CUs to the builds (fd_batch_set_eval_cus below the device's count) runs them BESIDE evaluation launches:
    // one workgroup per model then stays on those CUs; the front end's short wide launches would queue for CUs the evaluation holds
    bb.reg_front = (b->eval_cus <= 0 || b->eval_cus >= (int)device_cus()) ? 1 : 0;
    static const bool no_groups = tuning_env("FD_NO_PANEL_PAIRS") != nullptr;
    bb.group_panels = (b->n >= 4 && !no_groups && !tuning_env("FD_LOOKAHEAD")) ? 1 : 0;
    if (make_lookahead(&b->lu_stream, b->lu_events)) {
        bb.aux_stream = b->lu_stream;
        for (int q = 0; q < 4; ++q) bb.aux_events[q] = b->lu_events[q];
    }

    fd_batch::Key key{};
    key.M = c0->M; key.kind = c0->kind; key.term = c0->term | (bb.spd << 9) | (bb.small << 10) | (bb.reg << 11) | (bb.nopivot << 12); key.nparams = c0->nparams;
    memcpy(key.params, c0->params, sizeof(key.params));
    if (!bb.reg && b->use_graph && (!b->exec || memcmp(&key, &b->key, sizeof(key)) != 0)) {
        if (b->exec) { (void)hipGraphExecDestroy(b->exec); b->exec = nullptr; }
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
            hipError_t e2 = launch_build(bb, stream, nullptr);
            e = hipStreamEndCapture(stream, &graph);
            if (e == hipSuccess && e2 != hipSuccess) e = e2;
        }
        if (e == hipSuccess && graph) e = hipGraphInstantiate(&b->exec, graph, nullptr, nullptr, 0);
        if (graph) (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { (void)hipGetLastError(); b->exec = nullptr; b->use_graph = false; }
        else b->key = key;
    }
#define FD_BHIP(call)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) { batch_err(b, "%s failed: %s", #call, hipGetErrorString(e_)); return FD_E_DEVICE; } \
    } while (0)
    // One factorisation for the group (fd_batch_set_shared_factor): the register-resident build applies and every context reads
    // its rest rig, in place, from the SAME device array (fd_batch_set_points_dev with one rest pointer); anything else builds
    // every model on its own as ever.
    // every context reads its rest rig in place from ONE array, in this one launch sequence: their centres are equal by construction
    bool same_rest_launch = b->have_src;
    for (int i = 1; i < b->n && same_rest_launch; ++i) same_rest_launch = b->src.rest[i] == b->src.rest[0];
    const bool shared_fac_ok = same_rest_launch;
    bool shared_fac = b->shared_factor != 0 && bb.reg && shared_fac_ok && b->n > 1;
    if (shared_fac && !b->d_fac && hipMalloc((void **)&b->d_fac, sizeof(double) * reg_factor_doubles()) != hipSuccess) {
        (void)hipGetLastError(); b->d_fac = nullptr; shared_fac = false;
    }
    b->last_shared_factor = shared_fac ? 1 : 0;
    if (!b->lean) FD_BHIP(hipEventRecord(b->ev0, stream));      // (lean: the group's first packet is its first kernel; the reports carry no phase times)
    if (shared_fac) {
        FD_BHIP(launch_build_reg_shared(bb, stream, &b->src, nullptr, b->d_fac));
        b->have_src = false;
    } else if (bb.reg) {
        // one launch, one workgroup per model, control table included
        FD_BHIP(launch_build_reg(bb, stream, b->have_src ? &b->src : nullptr, nullptr));      // (no phases to split: the report's assembly time is 0)
        b->have_src = false;
    } else {
    // the control points are kernel arguments (they change every call), so k_prepare stays
    // outside the captured graph
    FD_BHIP(launch_prepare(bb, stream, b->have_src ? &b->src : nullptr));
    b->have_src = false;
    if (b->use_graph && b->exec) {
        FD_BHIP(hipEventRecord(b->ev_mid, stream));
        FD_BHIP(hipGraphLaunch(b->exec, stream));
    } else {
        FD_BHIP(launch_build(bb, stream, b->ev_mid));
    }
    }
    if (!b->lean) FD_BHIP(hipEventRecord(b->ev1, stream));      // (lean: fd_batch_cook_group records it behind the evaluation)
    b->waited_stream = nullptr;
#undef FD_BHIP
    static uint64_t build_ids = 0;
    const uint64_t this_build = same_rest_launch ? ++build_ids : 0;
    for (int i = 0; i < b->n; ++i) {
        fd_ctx *c = b->ctxs[i];
        c->rig_build_id = this_build;
        c->wait_event = b->ev1; c->wait_stream = stream; c->wait_batch = b;
        c->tev0 = b->lean ? nullptr : b->ev0; c->tev_mid = b->lean ? nullptr : (bb.reg ? b->ev0 : b->ev_mid); c->tev1 = b->lean ? nullptr : b->ev1;
        c->have_factor = bb.ml_layers == 0;   // a batched build leaves a factorisation fd_set_deltas can reuse (not the multilayer model)
        c->factor_grouped = bb.group_panels != 0;
        c->last_spd = bb.spd != 0;
        c->last_reg = bb.spd != 0 && bb.reg != 0;
        c->last_nopivot = bb.nopivot != 0;
        c->deltas_only = false;
        c->build_pending = true;
        ++c->model_gen;
        c->built = false;
        c->have_report = false;
        c->sticky_rc = FD_OK;
    }
    // the statuses of all contexts: one launch, one event (recorded on every context's own event
    // object would cost a stream operation each)
    bool posted_in_kernel = bb.reg != 0;
    for (int i = 0; i < b->n; ++i) posted_in_kernel = posted_in_kernel && b->ctxs[i]->h_slot.host_status != nullptr;
    if (posted_in_kernel) {
        // (the register-resident build posts every context's status word itself: no status kernel behind it)
        for (int i = 0; i < b->n; ++i) { b->ctxs[i]->status_poll = b->ev1; b->ctxs[i]->status_inflight = true; }
    } else {
        StatusTable st{};
        bool ok = true;
        for (int i = 0; i < b->n && ok; ++i) {
            st.model[i] = b->ctxs[i]->d_model;
            int *dword = b->ctxs[i]->d_status_alias;
            ok = dword != nullptr;
            st.host_word[i] = dword;
        }
        if (ok) {
            hipLaunchKernelGGL(k_post_status_batch, dim3(1), dim3(kMaxBatch), 0, stream, st, b->n);
            ok = hipGetLastError() == hipSuccess && hipEventRecord(b->status_ev, stream) == hipSuccess;
            for (int i = 0; i < b->n && ok; ++i) { b->ctxs[i]->status_poll = b->status_ev; b->ctxs[i]->status_inflight = true; }
        }
        if (!ok) (void)hipGetLastError();
    }
    return FD_OK;
}

// the statuses a batched build posted: one event query while they are in flight, then every context on its own
static int batch_poll(fd_batch *b)
{
    if (b->h_mismatch && *b->h_mismatch != 0) {
        batch_err(b, "shared-rig evaluation: context %d was built on other rest points than context 0 (same address, other "
                     "contents); its frame was passed through", *b->h_mismatch - 1);
        return FD_E_INVALID;
    }
    bool any = false;
    for (int i = 0; i < b->n; ++i) any = any || b->ctxs[i]->status_inflight || b->ctxs[i]->sticky_rc != FD_OK;
    if (!any) return FD_OK;
    bool shared_pending = false;
    hipEvent_t shared_ev = nullptr;
    for (int i = 0; i < b->n; ++i)
        if (b->ctxs[i]->status_inflight && (b->ctxs[i]->status_poll == b->status_ev || b->ctxs[i]->status_poll == b->ev1)) {
            shared_pending = true; shared_ev = b->ctxs[i]->status_poll;
        }
    if (shared_pending) {
        const hipError_t q = hipEventQuery(shared_ev);
        if (q == hipErrorNotReady) { (void)hipGetLastError(); return FD_OK; }
    }
    for (int i = 0; i < b->n; ++i) {
        const bool lu_before = b->ctxs[i]->prefer_lu;
        const int rc = poll_status(b->ctxs[i]);
        if (b->ctxs[i]->prefer_lu != lu_before) b->prepared = false;     // a model was rebuilt: a set packed from the old one is stale
        if (rc) { batch_err(b, "context %d: %s", i, b->ctxs[i]->err); return rc; }
    }
    return FD_OK;
}

int fd_batch_build_result(fd_batch *b, fd_report *reports)
{
    if (!b) return FD_E_INVALID;
    int first = FD_OK;
    for (int i = 0; i < b->n; ++i) {
        const int rc = fd_build_result(b->ctxs[i], reports ? &reports[i] : nullptr);
        if (rc && !first) { first = rc; batch_err(b, "context %d: %s", i, b->ctxs[i]->err); }
    }
    return first;
}

// One launch evaluates every context of the batch on its own vertex arrays (grid y = context)
// when they all take the default thin-plate kernel on equally sized inputs; otherwise the
// single launches are enqueued one after the other.  Results are those of fd_deform_dev_stream
// per context, bit for bit.
int fd_batch_deform_dev(fd_batch *b, void *hip_stream, int64_t N, const float *const *d_P_in, float *const *d_P_out,
                        const float *const *d_dist2, float *const *d_falloff_out, const float *const *d_tu,
                        const float *const *d_tv, const float *const *d_nrm, float radius2, float falloffrate)
{
    if (!b || !d_P_in || !d_P_out) return FD_E_INVALID;
    if (N < 0) { batch_err(b, "fd_batch_deform_dev: N < 0"); return FD_E_INVALID; }
    const int ntab = (d_tu != nullptr) + (d_tv != nullptr) + (d_nrm != nullptr);
    if (ntab != 0 && ntab != 3) { batch_err(b, "fd_batch_deform_dev: tu, tv, nrm tables must be all set or all NULL"); return FD_E_INVALID; }
    if (N == 0) return FD_OK;
    fd_ctx *c0 = b->ctxs[0];
    int rc = use_device(c0);
    if (rc) { batch_err(b, "%s", c0->err); return rc; }
    hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : cur_stream(c0);
    DeformArgs args[kMaxBatch];
    for (int i = 0; i < b->n; ++i) {
        fd_ctx *c = b->ctxs[i];
        if (!d_P_in[i] || !d_P_out[i]) { batch_err(b, "fd_batch_deform_dev: NULL vertex array for context %d", i); return FD_E_INVALID; }
        if (!c->built && !c->build_pending) { batch_err(b, "fd_batch_deform_dev: context %d has no built model", i); return FD_E_NOT_BUILT; }
        const float *tu = d_tu ? d_tu[i] : nullptr, *tv = d_tv ? d_tv[i] : nullptr, *nr = d_nrm ? d_nrm[i] : nullptr;
        const int ntan = (tu != nullptr) + (tv != nullptr) + (nr != nullptr);
        if (ntan != 0 && ntan != 3) { batch_err(b, "fd_batch_deform_dev: context %d: tu, tv, nrm must be all set or all NULL", i); return FD_E_INVALID; }
        DeformArgs &a = args[i];
        a.N = N;
        a.P_in = d_P_in[i]; a.P_out = d_P_out[i];
        a.dist2 = d_dist2 ? d_dist2[i] : nullptr; a.falloff_out = d_falloff_out ? d_falloff_out[i] : nullptr;
        a.tu = tu; a.tv = tv; a.nrm = nr;
        a.radius2 = radius2; a.falloffrate = falloffrate;
        a.M = model_centres(c); a.Mpad = round_up(a.M, kRecPad); a.kind = eval_kind(c); a.layers = record_layers(c);
        a.rec32 = c->d_rec32; a.rec64 = c->d_rec64; a.tiles = c->d_tiles; a.tiles16 = c->d_tiles16;
        a.model = c->d_model;
        a.precision = c->eval_precision;
        a.variant = c->eval_variant;
        a.delta_out = c0->output == FD_OUTPUT_DISPLACEMENT;
        if (c->output != c0->output) { batch_err(b, "fd_batch_deform_dev: context %d has another fd_set_output setting than context 0", i); return FD_E_INVALID; }
    }
    // statuses first: a model the poll had to rebuild (on its context's stream) is then ordered like any other build
    if ((rc = batch_poll(b))) return rc;
    for (int i = 0; i < b->n; ++i)
        if ((rc = order_after_batch(b->ctxs[i], stream))) { batch_err(b, "context %d: %s", i, b->ctxs[i]->err); return rc; }
    hipError_t e = launch_deform_batch(args, b->n, stream);
    if (e != hipSuccess) { batch_err(b, "launch_deform_batch failed: %s", hipGetErrorString(e)); return FD_E_DEVICE; }
    return FD_OK;
}

// Frames of ONE mesh and ONE rest rig: phi(|x - c|^2) is formed once per (vertex, centre) for all of
// them and the weight contraction runs on the matrix pipe (fd_eval.hip, k_deform32_tps_shared).
// ---- frames of one mesh and one rest rig (fd_batch_deform_shared_dev, fd_batch_prepare_shared) ----
// 1: the shared-rig launch applies; 0: the per-frame launches do; < 0: error
static int shared_applies(fd_batch *b, const char *who, float *const *d_P_out, int *ek_out)
{
    fd_ctx *c0 = b->ctxs[0];
    const int ek = eval_kind(c0);
    *ek_out = ek;
    bool fast = (ek == FD_KERNEL_THIN_PLATE || ek == FD_KERNEL_GAUSSIAN || ek == FD_KERNEL_GAUSSIAN_QNN) && round_up(c0->M, kRecPad) >= 32;
    for (int i = 0; i < b->n; ++i) {
        fd_ctx *c = b->ctxs[i];
        if (!d_P_out[i]) { batch_err(b, "%s: NULL output array for context %d", who, i); return FD_E_INVALID; }
        if (!c->built && !c->build_pending) { batch_err(b, "%s: context %d has no built model", who, i); return FD_E_NOT_BUILT; }
        // one rest rig: every context read its rest points in place from the SAME device array
        if (!c->rest_src || c->rest_src != c0->rest_src || c->M != c0->M || c->kind != c0->kind ||
            c->term != c0->term || c->nparams != c0->nparams || memcmp(c->params, c0->params, sizeof(c->params)) != 0) {
            batch_err(b, "%s: the contexts must share one rest rig (fd_batch_set_points_dev with the "
                         "same rest array for all), kernel and term; context %d does not", who, i);
            return FD_E_INVALID;
        }
        if (c->output != c0->output) { batch_err(b, "%s: context %d has another fd_set_output setting than context 0", who, i); return FD_E_INVALID; }
        if (c->eval_precision != FD_EVAL_FP32 || c->eval_variant > 0 || record_layers(c) != 0) fast = false;
    }
    return fast ? 1 : 0;
}

static bool make_event(hipEvent_t *ev)
{
    if (*ev) return true;
    if (hipEventCreateWithFlags(ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); *ev = nullptr; return false; }
    return true;
}

// the pack kernel of the contexts' current models into the set NOT read by the evaluation before; on `stream`
static int shared_pack(fd_batch *b, hipStream_t stream, int ek, float *const *d_P_out, float *const *d_falloff_out)
{
    fd_ctx *c0 = b->ctxs[0];
    const int si = b->packed_valid ? (b->cur_set ^ 1) : 0;
    fd_batch::SharedSet &st = b->sets[si];
    SharedDeformArgs a{};
    a.N = 1; a.Mpad = round_up(c0->M, kRecPad); a.nF = b->n; a.kind = ek; a.ctiles = c0->d_tiles16;
    a.falloff_out = d_falloff_out;
    int rc;
    // (lean: the build's event is recorded behind the evaluation -- a query now would see the PREVIOUS build's completion; the
    //  status arrives with the next call, and an evaluation of an unbuilt model passes its frame through: header)
    if (!b->lean && (rc = batch_poll(b))) return rc;           // before the ordering: a repaired model's rebuild is ordered with the rest
    for (int i = 0; i < b->n; ++i) {
        fd_ctx *c = b->ctxs[i];
        a.rec32[i] = c->d_rec32; a.model[i] = c->d_model; a.P_out[i] = d_P_out[i]; a.centres[i] = c->d_centres;
        if ((rc = order_after_batch(c, stream))) { batch_err(b, "context %d: %s", i, c->err); return rc; }
    }
    // "one rest rig" by content is what the pack kernel checks (an address does not identify contents); contexts that ONE batched
    // build read from one array are equal by construction, and the comparison -- the pack kernel's longest chain -- is skipped
    {
        bool one_build = c0->rig_build_id != 0;
        for (int i = 1; i < b->n && one_build; ++i) one_build = b->ctxs[i]->rig_build_id == c0->rig_build_id;
        if (one_build) for (int i = 1; i < b->n; ++i) a.centres[i] = a.centres[0];
    }
    const size_t wb = shared_wtile_bytes(a.Mpad, a.nF), fb = shared_frame_bytes(a.nF);
    if (wb > st.cap_wtiles || fb > st.cap_frames) {
        // (hipFree drains the device: no launch still reads the old scratch)
        if (st.d_wtiles) (void)hipFree(st.d_wtiles);
        if (st.d_frames) (void)hipFree(st.d_frames);
        st.d_wtiles = st.d_frames = nullptr; st.cap_wtiles = st.cap_frames = 0; st.eval_pending = false;
        if (hipMalloc(&st.d_wtiles, wb) != hipSuccess || hipMalloc(&st.d_frames, fb) != hipSuccess) {
            (void)hipGetLastError();
            batch_err(b, "shared-rig evaluation: scratch allocation (%zu bytes) failed", wb + fb);
            return FD_E_NOMEM;
        }
        st.cap_wtiles = wb; st.cap_frames = fb;
    }
    // the evaluation that last read this set must be through with it
    if (st.eval_pending && st.eval_done && hipStreamWaitEvent(stream, st.eval_done, 0) != hipSuccess) {
        batch_err(b, "shared-rig evaluation: hipStreamWaitEvent failed: %s", hipGetErrorString(hipGetLastError()));
        return FD_E_DEVICE;
    }
    st.eval_pending = false;
    a.wtiles = st.d_wtiles; a.frames = st.d_frames;
    a.packed_ev = b->lean ? nullptr : (make_event(&st.packed_ev) ? st.packed_ev : nullptr);
    b->consumed_override = nullptr;
    a.mode = 1;
    a.M = c0->M;
    if (b->h_mismatch && hipHostGetDevicePointer((void **)&a.mismatch, b->h_mismatch, 0) != hipSuccess) { (void)hipGetLastError(); a.mismatch = nullptr; }
    hipError_t e = launch_deform_shared(a, stream);
    if (e != hipSuccess) { batch_err(b, "launch_deform_shared (pack) failed: %s", hipGetErrorString(e)); return FD_E_DEVICE; }
    b->cur_set = si;
    b->packed_valid = b->lean || st.packed_ev != nullptr;
    b->prepared = true;
    for (int i = 0; i < b->n; ++i) {
        b->prep_P_out[i] = d_P_out[i]; b->prep_fall[i] = d_falloff_out ? d_falloff_out[i] : nullptr;
        b->prep_gen[i] = b->ctxs[i]->model_gen;
    }
    b->prep_has_fall = d_falloff_out != nullptr;
    return FD_OK;
}

int fd_batch_prepare_shared(fd_batch *b, void *hip_stream, float *const *d_P_out, float *const *d_falloff_out)
{
    if (!b || !d_P_out) return FD_E_INVALID;
    fd_ctx *c0 = b->ctxs[0];
    int rc = use_device(c0);
    if (rc) { batch_err(b, "%s", c0->err); return rc; }
    hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : cur_stream(c0);
    int ek = 0;
    const int applies = shared_applies(b, "fd_batch_prepare_shared", d_P_out, &ek);
    if (applies < 0) return applies;
    if (applies == 0) return FD_OK;              // the per-frame launches read the models themselves: nothing to prepare
    return shared_pack(b, stream, ek, d_P_out, d_falloff_out);
}

int fd_batch_deform_shared_dev(fd_batch *b, void *hip_stream, int64_t N, const float *d_P_in, float *const *d_P_out,
                               const float *d_dist2, float *const *d_falloff_out, const float *d_tu,
                               const float *d_tv, const float *d_nrm, float radius2, float falloffrate)
{
    if (!b || !d_P_out) return FD_E_INVALID;
    if (N < 0 || (N > 0 && !d_P_in)) { batch_err(b, "fd_batch_deform_shared_dev: bad N / P_in"); return FD_E_INVALID; }
    const int ntan = (d_tu != nullptr) + (d_tv != nullptr) + (d_nrm != nullptr);
    if (ntan != 0 && ntan != 3) { batch_err(b, "fd_batch_deform_shared_dev: tu, tv, nrm must be all set or all NULL"); return FD_E_INVALID; }
    if (N == 0) return FD_OK;
    fd_ctx *c0 = b->ctxs[0];
    int rc = use_device(c0);
    if (rc) { batch_err(b, "%s", c0->err); return rc; }
    hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : cur_stream(c0);
    int ek = 0;
    const int applies = shared_applies(b, "fd_batch_deform_shared_dev", d_P_out, &ek);
    if (applies < 0) return applies;
    if (applies == 0) {
        // any other kernel / precision: the per-frame launches on the shared arrays (same results as fd_deform_dev)
        const float *pin[kMaxBatch], *pd2[kMaxBatch], *ptu[kMaxBatch], *ptv[kMaxBatch], *pnr[kMaxBatch];
        for (int i = 0; i < b->n; ++i) { pin[i] = d_P_in; pd2[i] = d_dist2; ptu[i] = d_tu; ptv[i] = d_tv; pnr[i] = d_nrm; }
        rc = fd_batch_deform_dev(b, hip_stream, N, pin, d_P_out, d_dist2 ? pd2 : nullptr, d_falloff_out, d_tu ? ptu : nullptr,
                                 d_tu ? ptv : nullptr, d_tu ? pnr : nullptr, radius2, falloffrate);
        // these launches read the models to their end: fd_batch_wait_consumed waits for all of them
        if (rc == FD_OK && make_event(&b->fallback_ev)) {
            if (hipEventRecord(b->fallback_ev, stream) != hipSuccess) { (void)hipGetLastError(); rc = FD_E_DEVICE; }
            b->packed_valid = false;
            b->consumed_override = nullptr;
        }
        return rc;
    }
    // the prepared set, if it was packed from these models for these outputs; else pack now, on this stream
    bool reuse = b->prepared && b->prep_has_fall == (d_falloff_out != nullptr);
    for (int i = 0; reuse && i < b->n; ++i)
        reuse = b->prep_P_out[i] == d_P_out[i] && (!d_falloff_out || b->prep_fall[i] == d_falloff_out[i]) &&
                b->prep_gen[i] == b->ctxs[i]->model_gen;          // a context rebuilt or re-imported on its own since: pack again
    if (reuse) {
        if (!b->lean && (rc = batch_poll(b))) return rc;
        if (!b->prepared) reuse = false;         // the poll repaired a model: pack again (shared_pack orders the stream)
        for (int i = 0; reuse && i < b->n; ++i)
            if ((rc = order_after_batch(b->ctxs[i], stream))) { batch_err(b, "context %d: %s", i, b->ctxs[i]->err); return rc; }
    }
    if (reuse) {
        fd_batch::SharedSet &ps = b->sets[b->cur_set];
        if (!b->lean && ps.packed_ev && hipStreamWaitEvent(stream, ps.packed_ev, 0) != hipSuccess) {
            batch_err(b, "fd_batch_deform_shared_dev: hipStreamWaitEvent failed: %s", hipGetErrorString(hipGetLastError()));
            return FD_E_DEVICE;
        }
    } else if ((rc = shared_pack(b, stream, ek, d_P_out, d_falloff_out))) {
        return rc;
    }
    fd_batch::SharedSet &st = b->sets[b->cur_set];
    SharedDeformArgs a{};
    a.N = N; a.P_in = d_P_in; a.dist2 = d_dist2; a.tu = d_tu; a.tv = d_tv; a.nrm = d_nrm;
    a.radius2 = radius2; a.falloffrate = falloffrate;
    a.Mpad = round_up(c0->M, kRecPad); a.nF = b->n;
    a.ctiles = c0->d_tiles16;
    a.kind = ek;
    a.falloff_out = d_falloff_out;
    for (int i = 0; i < b->n; ++i) { a.rec32[i] = b->ctxs[i]->d_rec32; a.model[i] = b->ctxs[i]->d_model; a.P_out[i] = d_P_out[i]; }
    a.wtiles = st.d_wtiles; a.frames = st.d_frames;
    a.packed_ev = nullptr;
    a.mode = 2;
    a.delta_out = c0->output == FD_OUTPUT_DISPLACEMENT;
    a.max_wgs = b->eval_cus;
    hipError_t e = launch_deform_shared(a, stream);
    if (e != hipSuccess) { batch_err(b, "launch_deform_shared failed: %s", hipGetErrorString(e)); return FD_E_DEVICE; }
    if (b->lean) {
        // (one stream for the whole group: fd_batch_cook_group records the batch's ev1 right behind this launch -- that record says
        //  "evaluated" as well as "built"; one trailing packet fewer in front of the caller's wait.  A later re-record of ev1 belongs to
        //  a build that was itself ordered behind this evaluation by fd_batch_wait_consumed: waiting for it still implies this.)
        st.eval_done = b->ev1;
        st.eval_pending = true;
    } else if (make_event(&st.eval_ev)) {
        st.eval_done = st.eval_ev;
        if (hipEventRecord(st.eval_ev, stream) != hipSuccess) { (void)hipGetLastError(); st.eval_pending = false; }
        else st.eval_pending = true;
    } else {
        // no event to order the next pack of this set by: be safe
        (void)hipStreamSynchronize(stream);
    }
    return FD_OK;
}

int fd_batch_wait_consumed(fd_batch *b, void *hip_stream)
{
    if (!b) return FD_E_INVALID;
    hipEvent_t ev = b->consumed_override ? b->consumed_override : (b->packed_valid ? b->sets[b->cur_set].packed_ev : b->fallback_ev);
    if (!ev) return FD_OK;          // no shared-rig evaluation enqueued: nothing reads the models beyond stream order
    fd_ctx *c0 = b->ctxs[0];
    int rc = use_device(c0);
    if (rc) { batch_err(b, "%s", c0->err); return rc; }
    hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : cur_stream(c0);
    if (hipStreamWaitEvent(stream, ev, 0) != hipSuccess) {
        batch_err(b, "fd_batch_wait_consumed: hipStreamWaitEvent failed: %s", hipGetErrorString(hipGetLastError()));
        return FD_E_DEVICE;
    }
    return FD_OK;
}

int fd_batch_cook_group(fd_batch *b, void *build_stream, void *eval_stream, const float *d_rest_xyz,
                        const float *const *d_delta_xyz, int M, int64_t N, const float *d_P_in, float *const *d_P_out,
                        float *const *d_falloff_out, const fd_group_events *events)
{
    if (!b || !d_rest_xyz || !d_delta_xyz || !d_P_out) return FD_E_INVALID;
    fd_ctx *c0 = b->ctxs[0];
    hipStream_t bs = build_stream ? (hipStream_t)build_stream : cur_stream(c0);
    hipStream_t es = eval_stream ? (hipStream_t)eval_stream : bs;
    auto mark = [&](void *ev, hipStream_t st) { if (ev && hipEventRecord((hipEvent_t)ev, st) != hipSuccess) (void)hipGetLastError(); };
#ifdef FD_TUNING
    // FD_COOK_TIMING=1: host time of the call's pieces on stderr (us): wait_consumed | set_points | build (launches) | packing | evaluation
    static const bool cook_timing = tuning_env("FD_COOK_TIMING") != nullptr;
    struct CookClock {
        bool on; double t[6]; int n = 0;
        static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
        void tick() { if (on && n < 6) t[n++] = now(); }
        ~CookClock() { if (on && n == 6) fprintf(stderr, "[cook_group host us] wait_consumed %.1f | set_points %.1f | build %.1f | packing %.1f | evaluation %.1f\n", t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4]); }
    } clk{cook_timing};
#define FD_COOK_TICK() clk.tick()
#else
#define FD_COOK_TICK()
#endif
    FD_COOK_TICK();
    int rc = fd_batch_wait_consumed(b, build_stream);
    if (rc) return rc;
    FD_COOK_TICK();
    const float *rest[kMaxBatch];
    for (int i = 0; i < b->n; ++i) rest[i] = d_rest_xyz;
    if ((rc = fd_batch_set_points_dev(b, rest, d_delta_xyz, M))) return rc;
    // one stream for everything: no event between the build, the packing and the evaluation (see fd_batch::lean)
    const bool lean = es == bs;
    struct LeanScope {
        fd_batch *b; hipStream_t s; bool on, built = false;
        ~LeanScope() {
            if (!on) return;
            // the build's end, as every waiter and the status poll know it: behind whatever of the group was enqueued
            if (built && hipEventRecord(b->ev1, s) != hipSuccess) (void)hipGetLastError();
            b->prepared = false;          // the set was packed without its event: a later call on another stream packs again
            b->lean = false;
        }
    } scope{b, bs, lean};
    b->lean = lean;
    FD_COOK_TICK();
    if (events) mark(events->before_build, bs);
    if ((rc = fd_batch_build_async(b, build_stream))) return rc;
    scope.built = true;
    if (events) mark(events->after_build, bs);
    FD_COOK_TICK();
    if ((rc = fd_batch_prepare_shared(b, build_stream, d_P_out, d_falloff_out))) return rc;
    FD_COOK_TICK();
    if (es != bs) {
        // (the evaluation waits for the pack kernel's event inside fd_batch_deform_shared_dev when the set was prepared; the
        // per-frame fallback -- other kernels -- needs the builds themselves)
        if (!make_event(&b->group_ev) || hipEventRecord(b->group_ev, bs) != hipSuccess || hipStreamWaitEvent(es, b->group_ev, 0) != hipSuccess) {
            batch_err(b, "fd_batch_cook_group: ordering the evaluation stream failed: %s", hipGetErrorString(hipGetLastError()));
            return FD_E_DEVICE;
        }
    }
    if (events) mark(events->before_eval, es);
    rc = fd_batch_deform_shared_dev(b, es, N, d_P_in, d_P_out, nullptr, d_falloff_out, nullptr, nullptr, nullptr, 1.0f, 1.0f);
    if (events) mark(events->after_eval, es);
    FD_COOK_TICK();
    if (lean && rc == FD_OK && b->packed_valid && b->sets[b->cur_set].eval_pending) b->consumed_override = b->ev1;      // (recorded by `scope` on the way out)
    return rc;
}

int fd_batch_set_eval_cus(fd_batch *b, int n_cus)
{
    if (!b) return FD_E_INVALID;
    b->eval_cus = n_cus > 0 ? n_cus : 0;
    return FD_OK;
}

int fd_batch_set_shared_factor(fd_batch *b, int on)
{
    if (!b) return FD_E_INVALID;
    b->shared_factor = on ? 1 : 0;
    return FD_OK;
}

int fd_batch_last_build_shared_factor(const fd_batch *b) { return b ? b->last_shared_factor : 0; }

int fd_batch_size(const fd_batch *b) { return b ? b->n : 0; }

const char *fd_shared_kernel_name(int M, int frames, int kind)
{
    const bool takes = (kind == FD_KERNEL_THIN_PLATE || kind == FD_KERNEL_GAUSSIAN || kind == FD_KERNEL_GAUSSIAN_QNN) && M > 0 && round_up(M, kRecPad) >= 32 &&
                       frames >= 1 && frames <= kMaxBatch;
    return takes ? shared_kernel_name(round_up(M, kRecPad), frames, kind) : "";
}

}  // extern "C"
