// fd_sop_host.cpp -- HDK-free mirror of SOP_FaceDeform's cook over plain arrays.
//
// Same parm tokens, defaults and clamps as the reference's PRM_Template list
// (reference src/SOP_FaceDeform.cpp:99-137), same cook order and the same
// error / warning / message texts (:215-489), with the ALGLIB call sequence and
// the evaluation loop replaced by the fd_* C ABI.  The HDK wrapper
// (hdk/SOP_FaceDeformHip.cpp) gathers GU_Detail attributes into an fdsop_geo
// and calls fdsop_cook; the tests drive the same entry points through ctypes.
//
// The morph-space pass (setupBlends :175-213, the loop at :444-482, DirectBSEdit in
// src/dbse.cpp) runs on the device through fd_morph_*.  ProximityCapture (:301-322,
// src/capture.cpp) runs on the device too, through fd_mesh_capture, when the caller hands over
// what it needs (the mesh's edge adjacency and the rig's triangles) and no dist2 array of its
// own; cached across cooks exactly as the reference caches m_mesh_capture.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/facedeform_hip.h"

namespace {

enum ParmType { P_STRING, P_ORD, P_FLOAT, P_INT, P_TOGGLE, P_FLOAT2 };

struct ParmDef {
    const char *token;
    ParmType type;
    double def0, def1;
    const char *label;
};

// reference src/SOP_FaceDeform.cpp:99-137; the last four are additions that the
// survey allows (new parms, nothing renamed or removed).
const ParmDef kParms[] = {
    {"group", P_STRING, 0, 0, "Group"},
    {"model", P_ORD, 0, 0, "Model"},                       // 0 QNN, 1 Multilayer   (:48-53)
    {"term", P_ORD, 0, 0, "RBF Term"},                     // 0 Linear, 1 Constant, 2 Zero (:55-61)
    {"qcoef", P_FLOAT, 1, 0, "Q (Smoothness)"},
    {"zcoef", P_FLOAT, 5, 0, "Z (Deviation)"},
    {"radius", P_FLOAT, 1, 0, "Radius"},
    {"maxedges", P_INT, 4, 0, "Max edges"},
    {"layers", P_INT, 4, 0, "Layers"},
    {"lambda", P_FLOAT, 0.1, 0, "Lambda"},
    {"tangent", P_TOGGLE, 0, 0, "Tangent space"},
    {"morphspace", P_TOGGLE, 0, 0, "Blendshapes subspace"},
    {"doclampweight", P_TOGGLE, 0, 0, "Clamp weights"},
    {"weightrange", P_FLOAT2, 0, 1, "Range"},
    {"dofalloff", P_TOGGLE, 0, 0, "Falloff"},
    {"falloffradius", P_FLOAT, 1, 0, "Falloff radius"},
    {"falloffrate", P_FLOAT, 1, 0, "Falloff rate (exponent)"},
    // -- additions
    {"kernel", P_ORD, 0, 0, "Kernel"},        // 0 = per `model` (Gaussian), 1 thin-plate, 2 biharmonic, 3 cubic
    {"smoothing", P_FLOAT, 0, 0, "Smoothing"},  // diagonal lambda for kernel != 0 and for QNN
    {"precision", P_ORD, 0, 0, "Evaluation precision"},  // 0 fp32 while its error estimate holds the tolerance (else fp64 for that cook), 1 fp64, 2 fp32 always
    {"device", P_INT, -1, 0, "GPU device"},
};
constexpr int kNumParms = (int)(sizeof(kParms) / sizeof(kParms[0]));

int find_parm(const char *token)
{
    if (!token) return -1;
    for (int i = 0; i < kNumParms; ++i)
        if (strcmp(kParms[i].token, token) == 0) return i;
    return -1;
}

template <typename T>
T sys_max(T a, T b) { return a > b ? a : b; }

}  // namespace

struct fdsop_node {
    fd_config cfg{};
    fd_ctx *engine = nullptr;
    // what the engine's device-resident mesh (fd_mesh_set) was uploaded with
    bool mesh_had_dist2 = false, mesh_had_frames = false;
    // m_mesh_capture.isInitialized() && isCaptured() (:310-311): the device-resident dist2 is the
    // product of a capture that is still valid
    bool captured = false;
    fd_morph *morph = nullptr;        // m_direct_blends (:SOP_FaceDeform.hpp), lives as long as the node
    int morph_device = -2;
    double fval[kNumParms][2];
    std::string sval[kNumParms];
    std::string messages;
    int severity = FDSOP_OK;
    // clamped values of the last cook (A1)
    float e_qcoef = 0, e_zcoef = 0, e_radius = 0, e_lambda = 0;
    int e_layers = 0, e_maxedges = 0;
    int engine_precision = -1, engine_device = -2;
    bool fp64_warned = false;         // the "evaluating in fp64" warning has been given for the current rest rig (once per rig, not per cook)

    void add(int sev, const char *text)
    {
        const char *name = sev == FDSOP_ERROR ? "error" : (sev == FDSOP_WARNING ? "warning" : "message");
        messages += name;
        messages += '\t';
        messages += text;
        messages += '\n';
        if (sev > severity) severity = sev;
    }
    // ordinal parms are strings parsed with atoi, as reference :247-248
    int ord(int idx) const { return atoi(sval[idx].c_str()); }
};

extern "C" {

int fdsop_parm_count(void) { return kNumParms; }
const char *fdsop_parm_token(int i) { return (i >= 0 && i < kNumParms) ? kParms[i].token : nullptr; }

fdsop_node *fdsop_create(const fd_config *cfg)
{
    fdsop_node *n = new (std::nothrow) fdsop_node();
    if (!n) return nullptr;
    if (cfg) n->cfg = *cfg;
    else { memset(&n->cfg, 0, sizeof(n->cfg)); n->cfg.device = -1; }
    for (int i = 0; i < kNumParms; ++i) {
        n->fval[i][0] = kParms[i].def0;
        n->fval[i][1] = kParms[i].def1;
        if (kParms[i].type == P_ORD) n->sval[i] = std::to_string((int)kParms[i].def0);
    }
    n->fval[find_parm("device")][0] = n->cfg.device;
    n->sval[find_parm("precision")] = std::to_string(n->cfg.eval_precision == FD_EVAL_FP64 ? 1 : 0);
    return n;
}

void fdsop_destroy(fdsop_node *node)
{
    if (!node) return;
    if (node->engine) fd_destroy(node->engine);
    if (node->morph) fd_morph_destroy(node->morph);
    delete node;
}

int fdsop_set_float(fdsop_node *node, const char *token, int index, double value)
{
    const int i = find_parm(token);
    if (!node || i < 0 || index < 0 || index > 1) return FD_E_INVALID;
    if (kParms[i].type == P_STRING) return FD_E_INVALID;
    if (index == 1 && kParms[i].type != P_FLOAT2) return FD_E_INVALID;
    node->fval[i][index] = value;
    if (kParms[i].type == P_ORD) node->sval[i] = std::to_string((int)value);
    return FD_OK;
}

int fdsop_set_int(fdsop_node *node, const char *token, int value)
{
    return fdsop_set_float(node, token, 0, (double)value);
}

int fdsop_set_string(fdsop_node *node, const char *token, const char *value)
{
    const int i = find_parm(token);
    if (!node || i < 0 || !value) return FD_E_INVALID;
    if (kParms[i].type != P_STRING && kParms[i].type != P_ORD) return FD_E_INVALID;
    node->sval[i] = value;
    if (kParms[i].type == P_ORD) node->fval[i][0] = atoi(value);
    return FD_OK;
}

int fdsop_get_float(const fdsop_node *node, const char *token, int index, double *value)
{
    const int i = find_parm(token);
    if (!node || i < 0 || !value || index < 0 || index > 1) return FD_E_INVALID;
    *value = node->fval[i][index];
    return FD_OK;
}

int fdsop_get_int(const fdsop_node *node, const char *token, int *value)
{
    const int i = find_parm(token);
    if (!node || i < 0 || !value) return FD_E_INVALID;
    *value = kParms[i].type == P_ORD ? node->ord(i) : (int)node->fval[i][0];
    return FD_OK;
}

const char *fdsop_messages(const fdsop_node *node) { return node ? node->messages.c_str() : ""; }

fd_ctx *fdsop_engine(fdsop_node *node) { return node ? node->engine : nullptr; }

int fdsop_effective_float(const fdsop_node *node, const char *token, double *value)
{
    if (!node || !token || !value) return FD_E_INVALID;
    if (!strcmp(token, "qcoef")) *value = node->e_qcoef;
    else if (!strcmp(token, "zcoef")) *value = node->e_zcoef;
    else if (!strcmp(token, "radius")) *value = node->e_radius;
    else if (!strcmp(token, "lambda")) *value = node->e_lambda;
    else if (!strcmp(token, "layers")) *value = node->e_layers;
    else if (!strcmp(token, "maxedges")) *value = node->e_maxedges;
    else return FD_E_INVALID;
    return FD_OK;
}

// The cook: reference src/SOP_FaceDeform.cpp:215-489, step for step.
int fdsop_cook(fdsop_node *node, const fdsop_geo *geo)
{
    if (!node || !geo) return FDSOP_ERROR;
    node->messages.clear();
    node->severity = FDSOP_OK;
    if (geo->npoints < 0 || (geo->npoints > 0 && (!geo->P || !geo->P_out)) || !geo->rest_P || !geo->deform_P) {
        node->add(FDSOP_ERROR, "Invalid geometry arrays.");
        return node->severity;
    }
    // duplicatePointSource(0) (:226): the output starts as a copy of input 0.  On the success
    // path fd_deform writes every point of P_out from P (gated points are copied through), so
    // the 12 B/point host copy is only made where the cook stops early.
    struct PassThrough {
        const fdsop_geo *g; bool armed;
        ~PassThrough() {
            if (armed && g->P_out != g->P && g->npoints > 0)
                memcpy(g->P_out, g->P, sizeof(float) * 3 * (size_t)g->npoints);
        }
    } pass{geo, true};

    // :228-234
    if (geo->rest_npoints != geo->deform_npoints) {
        node->add(FDSOP_ERROR, "Rest and deform geometry should match.");
        return node->severity;
    }

    // :244-263 -- SYSmax(0.1, fpreal) is evaluated in double, then narrowed to float
    const int model_index = node->ord(find_parm("model"));
    const int term_index = node->ord(find_parm("term"));
    const float qcoef = (float)sys_max(0.1, node->fval[find_parm("qcoef")][0]);
    const float zcoef = (float)sys_max(0.1, node->fval[find_parm("zcoef")][0]);
    const float radius = (float)sys_max(0.01, node->fval[find_parm("radius")][0]);
    const int layers = sys_max(1, (int)node->fval[find_parm("layers")][0]);
    const float lambda = (float)sys_max(0.01, node->fval[find_parm("lambda")][0]);
    const int tangent_disp = (int)node->fval[find_parm("tangent")][0];
    const int morph_space = (int)node->fval[find_parm("morphspace")][0];
    const int max_edges = sys_max(1, (int)node->fval[find_parm("maxedges")][0]);
    const float falloffrate = (float)node->fval[find_parm("falloffrate")][0];
    node->e_qcoef = qcoef; node->e_zcoef = zcoef; node->e_radius = radius; node->e_lambda = lambda;
    node->e_layers = layers; node->e_maxedges = max_edges;
    const int kernel_ext = node->ord(find_parm("kernel"));
    const double smoothing = node->fval[find_parm("smoothing")][0];
    const int precision_parm = node->ord(find_parm("precision"));
    const int precision = precision_parm == 1 ? FD_EVAL_FP64 : FD_EVAL_FP32;
    const int device = (int)node->fval[find_parm("device")][0];

    // :268-287 -- M x 6 table: rest position and fp32 delta
    const int M = (int)geo->rest_npoints;
    std::vector<float> delta((size_t)(M > 0 ? M : 0) * 3);
    for (int i = 0; i < M; ++i)
        for (int c = 0; c < 3; ++c) delta[3 * (size_t)i + c] = geo->deform_P[3 * (size_t)i + c] - geo->rest_P[3 * (size_t)i + c];

    // :289-298
    const bool do_tangent_disp = tangent_disp && geo->tangentu && geo->tangentv && geo->N;
    if (tangent_disp && !do_tangent_disp)
        node->add(FDSOP_WARNING, "Append PolyFrameSOP and enable tangent[u/v] and N attribute to allow tangent displacement.");

    // :323-329 -- morph space needs blendshapes on inputs 3..; setupBlends (:175-213)
    const bool have_blends = geo->nshapes > 0 && geo->shapes_P && geo->shapes_npoints;
    if (geo->weights_count) *geo->weights_count = 0;
    // the `rest` attribute: input 0's own unless the rest pose changed or there is none, in
    // which case it is a copy of the incoming P (:178-184)
    const float *rest_attr = (geo->rest && !geo->rest_changed) ? geo->rest : geo->P;
    bool morph_inited_this_cook = false;
    if (morph_space && have_blends) {
        if (!node->morph || node->morph_device != device) {
            if (node->morph) { fd_morph_destroy(node->morph); node->morph = nullptr; }
            fd_config mcfg = node->cfg;
            mcfg.struct_size = (int)sizeof(fd_config);
            mcfg.device = device;
            node->morph = fd_morph_create(&mcfg);
            node->morph_device = device;
        }
        if (node->morph && (geo->blends_changed || !fd_morph_is_initialised(node->morph))) {
            std::vector<const float *> shapes;
            bool mismatch = false;
            for (int64_t sidx = 0; sidx < geo->nshapes; ++sidx) {
                if (geo->shapes_npoints[sidx] != geo->npoints || !geo->shapes_P[sidx]) { mismatch = true; continue; }
                shapes.push_back(geo->shapes_P[sidx]);
            }
            if (mismatch)   // :201-203
                node->add(FDSOP_WARNING, "Some blendshapes don't match rest pose point count. Ignoring them.");
            // DirectBSEdit::init measures the shapes against the mesh's CURRENT P (dbse.cpp:22), not
            // against the rest attribute
            if (geo->npoints <= 0 ||
                fd_morph_init(node->morph, geo->npoints, (int)shapes.size(), geo->P, shapes.empty() ? nullptr : shapes.data()) != FD_OK)
                node->add(FDSOP_WARNING, "Can't proceed with morph space deformation. Ingoring it.");   // :209-211
            else
                morph_inited_this_cook = true;
        }
    } else if (morph_space) {
        node->add(FDSOP_WARNING, "No blendshapes found. Ignoring morphspace deformation.");
    }

    // engine (re)creation when the device changed (the precision is set per cook, below)
    if (!node->engine || node->engine_device != device) {
        if (node->engine) { fd_destroy(node->engine); node->engine = nullptr; }
        fd_config cfg = node->cfg;
        cfg.struct_size = (int)sizeof(fd_config);
        cfg.device = device;
        cfg.eval_precision = precision;
        node->engine = fd_create(&cfg);
        node->engine_precision = precision;
        node->engine_device = device;
        if (!node->engine) {
            std::string t = std::string("Can't create the GPU deformation engine: ") + fd_last_error(nullptr);
            node->add(FDSOP_ERROR, t.c_str());
            return node->severity;
        }
    }
    fd_ctx *ctx = node->engine;

    // :301-322 -- proximity capture, where the reference has it: before the model.  Input 0's arrays
    // go to the device first (fd_mesh_set; skipped when the caller vouches they are the previous
    // cook's), then -- with no dist2 array from the caller but the capture's own inputs at hand --
    // islands and squared distances are produced there.  Re-captured on the first cook and when the
    // rest pose or the rest rig changed; NOT when radius / maxedges / dofalloff alone changed (the
    // FIXME at :309: kept, B12).
    const int dofalloff_parm = (int)node->fval[find_parm("dofalloff")][0];
    const bool want_d2 = geo->dist2 != nullptr;
    const bool use_capture = !want_d2 && geo->edge_offsets && geo->rig_ntris >= 0 && (geo->rig_ntris == 0 || geo->rig_tris);
    bool mesh_ready = false;
    if (geo->npoints > 0) {
        // (a mesh that still carries an earlier cook's CAPTURED dist2 is not reusable once the capture's inputs are gone:
        // the reference then applies neither radius nor fall-off, :396-399 -- fd_mesh_set drops the stale attribute)
        const bool reuse = geo->mesh_unchanged && fd_mesh_size(ctx) == geo->npoints &&
                           node->mesh_had_dist2 == want_d2 && node->mesh_had_frames == do_tangent_disp &&
                           !(node->captured && !use_capture);
        int mrc = FD_OK;
        if (!reuse) {
            mrc = fd_mesh_set(ctx, geo->npoints, geo->P, geo->dist2, do_tangent_disp ? geo->tangentu : nullptr,
                              do_tangent_disp ? geo->tangentv : nullptr, do_tangent_disp ? geo->N : nullptr);
            node->mesh_had_dist2 = want_d2;
            node->mesh_had_frames = do_tangent_disp;
            node->captured = false;
        }
        if (mrc != FD_OK) {
            std::string t = std::string("GPU deformation failed: ") + fd_last_error(ctx);
            node->add(FDSOP_ERROR, t.c_str());
            return node->severity;
        }
        mesh_ready = true;
        if (!use_capture) node->captured = false;
        if (use_capture && (!node->captured || !geo->rig_rest_unchanged)) {
            // capture(max_edges, radius, dofalloff, falloffrate) (:317): radius^2 is the search bound
            mrc = fd_mesh_capture(ctx, geo->edge_offsets, geo->edge_neighbours, M > 0 ? M : 0, geo->rest_P, max_edges,
                                  (int)geo->rig_ntris, geo->rig_tris, radius * radius, dofalloff_parm, nullptr);
            if (mrc != FD_OK) {
                node->add(FDSOP_ERROR, "Can't capture geometry with a rig!");     // :318
                return node->severity;
            }
            node->captured = true;
        }
    }

    // :342-349 -- model select; `kernel` (addition) overrides the Gaussian family.  (Set before
    // the points here: an unchanged kernel / term leaves the engine's factorisation in place, and
    // fd_set_deltas below depends on that; every failure ends in the same message as at :337-340.)
    int rc = FD_OK;
    if (kernel_ext == 1 || kernel_ext == 2 || kernel_ext == 3) {
        const int kind = kernel_ext == 1 ? FD_KERNEL_THIN_PLATE : (kernel_ext == 2 ? FD_KERNEL_BIHARMONIC : FD_KERNEL_CUBIC);
        const double p[1] = {smoothing};
        rc = fd_set_kernel(ctx, kind, p, 1);
    } else if (model_index == 1) {   // ALGLIB_MODEL_ML: rbfsetalgomultilayer(model, radius, layers, lambda)
        // the engine's multilayer model takes up to 8 layers: beyond that the radius has shrunk
        // 256-fold and the remaining layers see only lambda^8 of the residual
        const double p[3] = {(double)radius, (double)(layers < 8 ? layers : 8), (double)lambda};
        rc = fd_set_kernel(ctx, FD_KERNEL_GAUSSIAN_ML, p, 3);
    } else {                         // ALGLIB_MODEL_QNN: rbfsetalgoqnn(model, qcoef, zcoef)
        const double p[3] = {(double)qcoef, (double)zcoef, smoothing};
        rc = fd_set_kernel(ctx, FD_KERNEL_GAUSSIAN_QNN, p, 3);
    }
    // :351-361 -- any other ordinal leaves ALGLIB's default (linear) in place
    const int term = (term_index == 1) ? FD_TERM_CONST : (term_index == 2 ? FD_TERM_ZERO : FD_TERM_LINEAR);
    if (rc == FD_OK) rc = fd_set_term(ctx, term);
    // :331-340 -- rbfcreate + rbfsetpoints.  The reference rebuilds its model on every cook (B12);
    // when the caller vouches that the rest rig has not changed since the last cook
    // (checkChangedSourceFlags(1) in the wrapper) only the deltas are new and the engine reuses
    // its factorisation -- bit-identical weights, half the cook.  Anything that invalidates it
    // (first cook, another M, kernel or term) falls back to the full path.
    if (rc == FD_OK) {
        rc = FD_E_NOT_BUILT;
        if (M > 0 && geo->rig_rest_unchanged) rc = fd_set_deltas(ctx, delta.data(), M);
        if (rc != FD_OK) rc = M > 0 ? fd_set_points(ctx, geo->rest_P, delta.data(), M) : FD_E_INVALID;
    }
    if (rc != FD_OK) {
        node->add(FDSOP_ERROR, "Can't build RBF model.");
        return node->severity;
    }
    // :363-368
    fd_report report;
    memset(&report, 0, sizeof(report));
    rc = fd_build(ctx, &report);
    if (report.terminationtype != 1 || rc != FD_OK) {
        node->add(FDSOP_ERROR, "Can't solve the problem.");
        return node->severity;
    }
    // The reference evaluates in fp64 inside ALGLIB (:404-439); the fp32 evaluation has an absolute error floor that the
    // build reports (fd_report.fp32_error).  Where that floor exceeds the reference's 1e-5 of the smallest displacements of
    // the control table, this cook is evaluated in fp64 -- unless the artist asked for fp32 outright (precision = 2).
    int cook_precision = precision;
    if (!geo->rig_rest_unchanged) node->fp64_warned = false;       // another rest rig: say it again if it applies
    if (precision_parm == 0 && !fd_fp32_holds(&report, 1e-5)) {
        cook_precision = FD_EVAL_FP64;
        if (!node->fp64_warned) {                                  // once per rig: an animated shot cooks the same rig every frame
            char t[240];
            snprintf(t, sizeof(t), "fp32 evaluation would not hold 1e-5 of this rig's displacements (error ~%.2g, smallest delta %.2g): "
                                   "evaluating in fp64.", report.fp32_error, report.delta_min);
            node->add(FDSOP_WARNING, t);
            node->fp64_warned = true;
        }
    }
    fd_set_eval_precision(ctx, cook_precision);
    // :370-373
    char info[200];
    snprintf(info, sizeof(info), "Termination type: %d, Iterations: %d", report.terminationtype, report.iterationscount);
    node->add(FDSOP_MESSAGE, info);

    // :386-388 -- Cd is added white and never written again
    if (geo->Cd)
        for (int64_t i = 0; i < geo->npoints * 3; ++i) geo->Cd[i] = 1.f;
    // :396-399
    if (!geo->dist2 && !(use_capture && node->captured))
        node->add(FDSOP_WARNING, "Can't find distance capture attribute. Won't apply radius nor falloff.");
    // :401 -- a fresh float attribute reads 0 until written
    if (geo->fd_falloff && geo->npoints > 0) memset(geo->fd_falloff, 0, sizeof(float) * (size_t)geo->npoints);
    const float radius_sqrt = radius * radius;   // :402 (a square, despite the name)

    // :404-439.  The mesh arrays go to the device through fd_mesh_set; when the caller vouches
    // that input 0 is the one of the previous cook (same data IDs) the upload is skipped and the
    // cook moves only its results over the host link.
    pass.armed = false;
    rc = FD_OK;
    if (mesh_ready) rc = fd_deform_mesh(ctx, geo->P_out, geo->fd_falloff, radius_sqrt, falloffrate);
    if (rc == FD_OK && geo->dist2_out && use_capture && node->captured)
        rc = fd_mesh_get_dist2(ctx, geo->dist2_out);
    if (rc != FD_OK) {
        std::string t = std::string("GPU deformation failed: ") + fd_last_error(ctx);
        node->add(FDSOP_ERROR, t.c_str());
        pass.armed = true;
        return node->severity;
    }

    // :444-482 -- morph-space reprojection of the deformed mesh
    if (morph_space && have_blends && node->morph && fd_morph_is_initialised(node->morph)) {
        // :448-450: the weights are computed only while isComputed() is false, i.e. on the first
        // cook after DirectBSEdit::init; every later cook with unchanged blendshapes takes the
        // warning branch below.  Kept as the reference has it.
        bool weights_done = false;
        const int S = fd_morph_shape_count(node->morph);
        if (!fd_morph_is_computed(node->morph)) {
            const int dofalloff = dofalloff_parm;
            const int doclampweight = (int)node->fval[find_parm("doclampweight")][0];
            const float falloffradius = (float)node->fval[find_parm("falloffradius")][0];
            const float weightrange[2] = {(float)node->fval[find_parm("weightrange")][0],
                                          (float)node->fval[find_parm("weightrange")][1]};
            std::vector<double> w((size_t)(S > 0 ? S : 1));
            // the engine already holds this cook's P as its rest pose when it was initialised in
            // this cook and input 0 has no rest attribute of its own; otherwise send the attribute
            const bool same_rest = morph_inited_this_cook && rest_attr == geo->P;
            int mrc = fd_morph_set_rest(node->morph, same_rest ? nullptr : rest_attr, 0);
            // computeWeights (dbse.cpp:39-60) + the displacement loop (:458-473), P_out in place
            if (mrc == FD_OK)
                mrc = fd_morph_apply(node->morph, geo->P_out, doclampweight ? weightrange : nullptr,
                                     dofalloff && falloffradius != 0.f, falloffradius, w.data());
            weights_done = mrc == FD_OK;
            if (weights_done) {   // :474-481, the detail array attribute `weights`
                if (geo->weights) memcpy(geo->weights, w.data(), sizeof(double) * (size_t)S);
                if (geo->weights_count) *geo->weights_count = S;
            }
        }
        if (!weights_done)
            node->add(FDSOP_WARNING, "Can't compute weights for morphspace deformation. Ingoring it.");   // :452
    }
    return node->severity;
}

}  // extern "C"
