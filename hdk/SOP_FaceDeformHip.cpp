// SOP_FaceDeformHip.cpp -- Houdini-side wrapper: the `facedeform` SOP with its RBF hot path
// running in libfacedeform_hip.so (MI355X) instead of ALGLIB.
//
// Not part of libfacedeform_hip.so: a plugin needs the Houdini HDK ($HT), which the build container does not have.  What the
// repository does with it: tests/test_hdk_wrapper_compiles.py compiles it (g++ -fsyntax-only -Wall -Wextra) and
// tests/test_gpu_hdk_wrapper.py cooks a paged detail through it on the GPU, both against tests/hdk_mock/ -- a mock of exactly the
// HDK types this file touches; against the real HDK it has never been built.  It is the reference-side binding INTEGRATION.md
// refers to.  It keeps the outer plugin contract of the reference (operator name/label/inputs:
// reference src/SOP_FaceDeform.cpp:35-46; parm tokens and defaults: :99-137) and hands the cook
// to fdsop_cook(), the HDK-free mirror of cookMySop in facedeform_amd/csrc/fd_sop_host.cpp.
//
// What stays on the host, as in the reference: input locking, duplicatePointSource, the point
// group of the `group` parm (cookInputGroups, :155-173) and the conditional data-ID bump it
// governs (:485-486).  ProximityCapture (reference src/capture.cpp) runs on the device: this file
// only gathers what it walks -- the mesh's edges as a CSR adjacency (what
// GQ_Detail::groupEdgePoints follows) and the rest rig's surface as triangles (what
// GU_RayIntersect::minimumPoint searches) -- and fdsop_cook captures where the reference does
// (:301-322), cached across cooks as m_mesh_capture is.  Morph space (DirectBSEdit, :175-213,
// 444-482) goes through the same geometry struct (inputs 3..).
#include <UT/UT_DSOVersion.h>

#include <GA/GA_AIFNumericArray.h>
#include <GA/GA_Handle.h>
#include <GA/GA_PageHandle.h>
#include <GA/GA_SplittableRange.h>
#include <GU/GU_Detail.h>
#include <GEO/GEO_Primitive.h>
#include <UT/UT_Array.h>
#include <OP/OP_AutoLockInputs.h>
#include <OP/OP_Operator.h>
#include <OP/OP_OperatorTable.h>
#include <PRM/PRM_Include.h>
#include <SOP/SOP_Node.h>

#include <algorithm>
#include <string>
#include <memory>
#include <vector>

#include "facedeform_hip.h"

namespace fdhip {

// ---- parm surface: driven by the engine's own table so the two cannot drift -------------
// (token, label, kind) -- kinds: s string, o ordinal, f float, l log-float, i int, t toggle, 2 float2
struct ParmRow { const char *token, *label; char kind; float def0, def1; const char *help; };
// The help texts of the reference's templates (src/SOP_FaceDeform.cpp:67-94, attached at :121-137), as the user of the node reads
// them in the parameter pane: part of the parm surface like the tokens and labels.  (Two of them name ALGLIB, whose models the
// engine restates densely: DESIGN.md 1.)
static const char *const kModelHelp = "QNN and Multilayer are different algorithms to perform RBF interpolation in ALGLIB. "
                                      "Multilayer is more robust and thus more expensive.";
static const char *const kTermHelp = "By appending small linear or constant term to RBF system one can stabalize it and help to solve smooth solution.";
static const char *const kRadiusHelp = "Radius controls not only RBF solution (how far to reach for a scattered data), "
                                       "but also radius of deformation applied to geometry.";
static const char *const kMaxEdgesHelp = "Number of edges deformation affects geometry. This is applied before radius.";
static const char *const kTangentHelp = "Project deformation into tangential space of a rest geometry. This helps to remove extreme deformations. ";
static const char *const kMorphHelp = "Projects deformation into subspace defined by blendshapes of a base mesh. They should be connected after second "
                                      "and third input and match rest mesh topology (unlike 2d and 3rd input which are typically sparser than first "
                                      "input (rest pose geo)).";
static const char *const kWeightRangeHelp = "Clamps total blendshape weights, so that deformation will be constrained strictly to blends' poses.";
static const char *const kFalloffHelp = "This is exponent of distance ratio (distance / radius) with which displacement falls off.";
static const ParmRow kRows[] = {
    {"model", "Model", 'o', 0, 0, kModelHelp},          {"term", "RBF Term", 'o', 0, 0, kTermHelp},
    {"qcoef", "Q (Smoothness)", 'f', 1, 0, nullptr},    {"zcoef", "Z (Deviation)", 'f', 5, 0, nullptr},
    {"radius", "Radius", 'l', 1, 0, kRadiusHelp},       {"maxedges", "Max edges", 'i', 4, 0, kMaxEdgesHelp},
    {"layers", "Layers", 'i', 4, 0, nullptr},           {"lambda", "Lambda", 'f', 0.1f, 0, nullptr},
    {"tangent", "Tangent space", 't', 0, 0, kTangentHelp}, {"morphspace", "Blendshapes subspace", 't', 0, 0, kMorphHelp},
    {"doclampweight", "Clamp weights", 't', 0, 0, kWeightRangeHelp}, {"weightrange", "Range", '2', 0, 1, kWeightRangeHelp},
    {"dofalloff", "Falloff", 't', 0, 0, nullptr},       {"falloffradius", "Falloff radius", 'l', 1, 0, kRadiusHelp},
    {"falloffrate", "Falloff rate (exponent)", 'f', 1, 0, kFalloffHelp},
    // additions (SURVEY.md 8b allows new parms; nothing above is renamed or removed)
    {"kernel", "Kernel", 'o', 0, 0, "Radial kernel of the dense system. Gaussian follows the Model parameter (QNN radii or multilayer); thin plate, "
                                    "biharmonic and cubic are solved with the RBF Term's polynomial."},
    {"smoothing", "Smoothing", 'f', 0, 0, "Added to the diagonal of the kernel matrix (0 interpolates the control points exactly)."},
    {"precision", "Evaluation precision", 'o', 0, 0, "fp32 evaluation is ~80x faster; the default falls back to fp64 for a rig whose displacements fp32 cannot "
                                                     "hold to 1e-5 (a warning names the two numbers)."},
    {"device", "GPU device", 'i', -1, 0, "HIP device ordinal (-1: the process's current device)."},
};
static constexpr int kNumRows = sizeof(kRows) / sizeof(kRows[0]);

static PRM_Name sModel[] = {PRM_Name("0", "QNN"), PRM_Name("1", "Multilayer"), PRM_Name(0)};
static PRM_Name sTerm[] = {PRM_Name("0", "Linear"), PRM_Name("1", "Constant"), PRM_Name("2", "Zero"), PRM_Name(0)};
static PRM_Name sKernel[] = {PRM_Name("0", "Gaussian (per Model)"), PRM_Name("1", "Thin plate"),
                             PRM_Name("2", "Biharmonic"), PRM_Name("3", "Cubic"), PRM_Name(0)};
static PRM_Name sPrecision[] = {PRM_Name("0", "fp32 (fp64 where fp32 cannot hold 1e-5)"), PRM_Name("1", "fp64"), PRM_Name("2", "fp32 always"), PRM_Name(0)};
static PRM_ChoiceList sModelMenu(PRM_CHOICELIST_SINGLE, sModel), sTermMenu(PRM_CHOICELIST_SINGLE, sTerm),
    sKernelMenu(PRM_CHOICELIST_SINGLE, sKernel), sPrecisionMenu(PRM_CHOICELIST_SINGLE, sPrecision);
static PRM_Range sRadiusRange(PRM_RANGE_RESTRICTED, 0.0, PRM_RANGE_UI, 10.0);
static PRM_Range sFalloffRange(PRM_RANGE_RESTRICTED, 0.0, PRM_RANGE_UI, 2.0);

static std::vector<PRM_Name> sNames;
static std::vector<PRM_Default> sDefaults;
static std::vector<PRM_Template> sTemplates;

static PRM_Template *buildTemplates()
{
    if (!sTemplates.empty()) return sTemplates.data();
    sNames.reserve(kNumRows);
    sDefaults.reserve(2 * kNumRows);
    sTemplates.emplace_back(PRM_STRING, 1, &PRMgroupName, nullptr, &SOP_Node::pointGroupMenu, nullptr, nullptr,
                            SOP_Node::getGroupSelectButton(GA_GROUP_POINT));
    for (const ParmRow &r : kRows) {
        sNames.emplace_back(r.token, r.label);
        sDefaults.emplace_back(r.def0);
        PRM_Default *def = &sDefaults.back();
        if (r.kind == '2') sDefaults.emplace_back(r.def1);
        PRM_Name *name = &sNames.back();
        const std::string tok(r.token);
        switch (r.kind) {
        case 'o': {
            PRM_ChoiceList *menu = tok == "model" ? &sModelMenu : tok == "term" ? &sTermMenu
                                   : tok == "kernel" ? &sKernelMenu : &sPrecisionMenu;
            sTemplates.emplace_back(PRM_ORD, 1, name, def, menu, nullptr, nullptr, nullptr, 0, r.help);
            break;
        }
        // (type, size, name, defaults, menu, range, callback, spare data, parm group, help text -- the reference's argument order, :121-137)
        case 'f': sTemplates.emplace_back(PRM_FLT_J, 1, name, def, nullptr, tok == "falloffrate" ? &sFalloffRange : nullptr, nullptr, nullptr, 0, r.help); break;
        case 'l': sTemplates.emplace_back(PRM_FLT_LOG, 1, name, def, nullptr, &sRadiusRange, nullptr, nullptr, 0, r.help); break;
        case 'i': sTemplates.emplace_back(PRM_INT_J, 1, name, def, nullptr, nullptr, nullptr, nullptr, 0, r.help); break;
        case 't': sTemplates.emplace_back(PRM_TOGGLE, 1, name, def, nullptr, nullptr, nullptr, nullptr, 0, r.help); break;
        case '2': sTemplates.emplace_back(PRM_FLT_J, 2, name, def, nullptr, nullptr, nullptr, nullptr, 0, r.help); break;
        }
    }
    sTemplates.emplace_back();
    return sTemplates.data();
}

// ---- page-wise gather / scatter between GA attributes and flat arrays --------------------
// (the access pattern the reference's dead threaded evaluator sketches,
//  src/SOP_FaceDeform.hpp:116-188: GA pages are contiguous runs of 1024 elements)
// Grow-only page-locked float array (fd_host_alloc): when every array of the cook is page-locked
// the evaluation kernel reads and writes them in place over the host link, with no staging copy
// (include/facedeform_hip.h, fd_host_alloc).  Kept as node members so the pages stay locked
// from cook to cook.
class PinnedF
{
public:
    PinnedF() = default;
    PinnedF(const PinnedF &) = delete;
    PinnedF &operator=(const PinnedF &) = delete;
    ~PinnedF() { fd_host_free(myData); }
    void resize(size_t n)
    {
        if (n > myCap) {
            fd_host_free(myData);
            myData = (float *)fd_host_alloc(sizeof(float) * n);
            myCap = myData ? n : 0;
        }
        mySize = myData ? n : 0;
    }
    void clear() { mySize = 0; }
    bool empty() const { return mySize == 0; }
    float *data() { return myData; }
    const float *data() const { return myData; }
    float &operator[](size_t i) { return myData[i]; }
    const float &operator[](size_t i) const { return myData[i]; }
private:
    float *myData = nullptr;
    size_t mySize = 0, myCap = 0;
};

static void gatherV3(const GU_Detail *gdp, const GA_Attribute *attr, PinnedF &out)
{
    out.resize(3 * (size_t)gdp->getNumPoints());
    GA_ROPageHandleV3 h(attr);
    GA_Offset start, end;
    for (GA_Iterator it(gdp->getPointRange()); it.blockAdvance(start, end);) {
        h.setPage(start);
        for (GA_Offset o = start; o < end; ++o) {
            const UT_Vector3 v = h.get(o);
            const size_t i = 3 * (size_t)gdp->pointIndex(o);
            out[i] = v.x(); out[i + 1] = v.y(); out[i + 2] = v.z();
        }
    }
}

static void scatterV3(GU_Detail *gdp, GA_Attribute *attr, const PinnedF &in)
{
    GA_RWPageHandleV3 h(attr);
    GA_Offset start, end;
    for (GA_Iterator it(gdp->getPointRange()); it.blockAdvance(start, end);) {
        h.setPage(start);
        for (GA_Offset o = start; o < end; ++o) {
            const size_t i = 3 * (size_t)gdp->pointIndex(o);
            h.set(o, UT_Vector3(in[i], in[i + 1], in[i + 2]));
        }
    }
}

// The mesh's edges as a CSR adjacency over point INDICES: every primitive contributes the edges
// between consecutive vertices (closed: last to first as well), both directions, duplicates
// removed.  This is the graph GQ_Detail::groupEdgePoints walks in the reference
// (src/capture.cpp:24,134).  Rebuilt only when input 0's topology changed.
static void gatherEdgeAdjacency(const GU_Detail *gdp, std::vector<int64_t> &offsets, std::vector<int> &neighbours)
{
    const size_t n = (size_t)gdp->getNumPoints();
    std::vector<std::vector<int>> adj(n);
    const GEO_Primitive *prim;
    GA_FOR_ALL_PRIMITIVES(gdp, prim) {
        const GA_Size nv = prim->getVertexCount();
        if (nv < 2) continue;
        const bool closed = prim->isClosed();
        for (GA_Size v = 0; v + 1 < nv || (closed && v < nv); ++v) {
            const GA_Index a = gdp->pointIndex(prim->getPointOffset(v));
            const GA_Index b = gdp->pointIndex(prim->getPointOffset((v + 1) % nv));
            if (a == b) continue;
            adj[(size_t)a].push_back((int)b);
            adj[(size_t)b].push_back((int)a);
        }
    }
    offsets.assign(n + 1, 0);
    neighbours.clear();
    for (size_t i = 0; i < n; ++i) {
        std::vector<int> &row = adj[i];
        std::sort(row.begin(), row.end());
        row.erase(std::unique(row.begin(), row.end()), row.end());
        neighbours.insert(neighbours.end(), row.begin(), row.end());
        offsets[i + 1] = (int64_t)neighbours.size();
    }
}

// The rest rig's surface as triangles (a fan per polygon), 9 floats each: what
// GU_RayIntersect::minimumPoint searches in the reference (src/capture.cpp:19,81).  Curves and
// points have no surface and contribute nothing (every distance then reads -1, as a miss does).
static void gatherRigTriangles(const GU_Detail *rig, std::vector<float> &tris)
{
    tris.clear();
    const GEO_Primitive *prim;
    GA_FOR_ALL_PRIMITIVES(rig, prim) {
        const GA_Size nv = prim->getVertexCount();
        if (nv < 3 || !prim->isClosed()) continue;
        const UT_Vector3 a = rig->getPos3(prim->getPointOffset(0));
        for (GA_Size v = 1; v + 1 < nv; ++v) {
            const UT_Vector3 b = rig->getPos3(prim->getPointOffset(v)), c = rig->getPos3(prim->getPointOffset(v + 1));
            const float t[9] = {a.x(), a.y(), a.z(), b.x(), b.y(), b.z(), c.x(), c.y(), c.z()};
            tris.insert(tris.end(), t, t + 9);
        }
    }
}

class SOP_FaceDeformHip : public SOP_Node
{
public:
    static OP_Node *create(OP_Network *net, const char *name, OP_Operator *op) { return new SOP_FaceDeformHip(net, name, op); }

    SOP_FaceDeformHip(OP_Network *net, const char *name, OP_Operator *op) : SOP_Node(net, name, op)
    {
        mySopFlags.setManagesDataIDs(true);     // as the reference: it bumps P's data ID itself
        // fdsop_geo and fd_report are plain structs without a size field, written whole by the library: a wrapper compiled
        // against another header must not call into it (the node then cooks to an error, as with no device)
        myNode = fd_abi_version() == FD_ABI_VERSION ? fdsop_create(nullptr) : nullptr;
    }
    ~SOP_FaceDeformHip() override { fdsop_destroy(myNode); }

protected:
    // reference src/SOP_FaceDeform.cpp:155-173, verbatim in effect: the point group of the `group`
    // parm into myGroup.  The evaluation ignores it (the reference's loop runs over every point,
    // :404); it only decides whether P's data ID is bumped at the end (:485).
    OP_ERROR cookInputGroups(OP_Context &context, int alone = 0) override
    {
        return cookInputPointGroups(context, myGroup, alone, true, 0, -1, true, false, true, 0);
    }

    OP_ERROR cookMySop(OP_Context &context) override
    {
        OP_AutoLockInputs inputs(this);
        if (inputs.lock(context) >= UT_ERROR_ABORT) return error();
        const fpreal t = context.getTime();
        duplicatePointSource(0, context);
        const GU_Detail *rest = inputGeo(1);
        const GU_Detail *deform = inputGeo(2);
        if (!myNode) { addError(SOP_MESSAGE, "GPU deformation engine unavailable."); return error(); }

        // forward every parm by token
        for (const ParmRow &r : kRows) {
            if (r.kind == 'o') { UT_String s; evalString(s, r.token, 0, t); fdsop_set_string(myNode, r.token, s.buffer()); }
            else if (r.kind == 'i' || r.kind == 't') fdsop_set_int(myNode, r.token, (int)evalInt(r.token, 0, t));
            else { fdsop_set_float(myNode, r.token, 0, evalFloat(r.token, 0, t)); if (r.kind == '2') fdsop_set_float(myNode, r.token, 1, evalFloat(r.token, 1, t)); }
        }

        PinnedF &P = myP, &restP = myRestP, &deformP = myDeformP, &tu = myTu, &tv = myTv, &nn = myNn,
                &dist2 = myDist2, &Pout = myPout, &falloff = myFalloff;
        tu.clear(); tv.clear(); nn.clear(); dist2.clear();
        gatherV3(gdp, gdp->getP(), P);
        gatherV3(rest, rest->getP(), restP);
        gatherV3(deform, deform->getP(), deformP);
        const GA_Attribute *aU = gdp->findFloatTuple(GA_ATTRIB_POINT, "tangentu", 3);
        const GA_Attribute *aV = gdp->findFloatTuple(GA_ATTRIB_POINT, "tangentv", 3);
        const GA_Attribute *aN = gdp->findFloatTuple(GA_ATTRIB_POINT, "N", 3);
        if (aU && aV && aN) { gatherV3(gdp, aU, tu); gatherV3(gdp, aV, tv); gatherV3(gdp, aN, nn); }
        // ProximityCapture's inputs (reference :310-322 -> src/capture.cpp): gathered when the
        // topology they come from changed; the capture itself runs in fdsop_cook on the device
        // and its result, the detached attribute `dist_a` (capture.cpp:31), never leaves it.
        {
            int topoChanged = myEdgeOffsets.empty(), rigChanged = myRigTris.empty() && !myRigGathered;
            int c = 0;
            checkChangedSourceFlags(0, context, &c); topoChanged |= c;
            checkChangedSourceFlags(1, context, &c); rigChanged |= c;
            if (topoChanged || myEdgeOffsets.size() != (size_t)gdp->getNumPoints() + 1) gatherEdgeAdjacency(gdp, myEdgeOffsets, myEdgeNeighbours);
            if (rigChanged) { gatherRigTriangles(rest, myRigTris); myRigGathered = true; }
        }
        const size_t n = (size_t)gdp->getNumPoints();
        Pout.resize(3 * n); falloff.resize(n);

        fdsop_geo geo{};
        geo.npoints = (int64_t)n;
        geo.P = P.data();
        geo.tangentu = tu.empty() ? nullptr : tu.data();
        geo.tangentv = tv.empty() ? nullptr : tv.data();
        geo.N = nn.empty() ? nullptr : nn.data();
        geo.dist2 = nullptr;                   // no attribute of the caller's: the cook captures on the device
        geo.edge_offsets = myEdgeOffsets.data();
        geo.edge_neighbours = myEdgeNeighbours.empty() ? nullptr : myEdgeNeighbours.data();
        geo.rig_ntris = (int64_t)(myRigTris.size() / 9);
        geo.rig_tris = myRigTris.empty() ? nullptr : myRigTris.data();
        geo.rest_npoints = rest->getNumPoints();
        geo.deform_npoints = deform->getNumPoints();
        geo.rest_P = restP.data();
        geo.deform_P = deformP.data();
        geo.P_out = Pout.data();
        geo.fd_falloff = falloff.data();
        geo.Cd = nullptr;      // the attribute is created with white defaults below (:386-388)

        // inputs 3..: blendshapes of the morph-space pass (reference setupBlends, :175-213)
        std::vector<const float *> shapePtrs;
        std::vector<int64_t> shapeCounts;
        std::vector<double> weights;
        int64_t weightsCount = 0;
        int restChanged = 0, blendsChanged = 0;
        if (evalInt("morphspace", 0, t) && nConnectedInputs() > 3) {
            checkChangedSourceFlags(0, context, &restChanged);
            const unsigned nshapes = nConnectedInputs() - 3;
            if (myShapes.size() < nshapes) myShapes.resize(nshapes);
            for (unsigned i = 3; i < nConnectedInputs(); ++i) {
                int changed = 0;
                checkChangedSourceFlags(i, context, &changed);
                blendsChanged |= changed;
                const GU_Detail *shape = inputGeo(i);
                PinnedF &buf = *myShapes[i - 3].get(this);
                // the arrays are only read when the engine (re)initialises; gather them then
                if (changed || buf.empty()) gatherV3(shape, shape->getP(), buf);
                shapePtrs.push_back(buf.data());
                shapeCounts.push_back((int64_t)shape->getNumPoints());
            }
            if (const GA_Attribute *aRest = gdp->findFloatTuple(GA_ATTRIB_POINT, "rest", 3)) {
                gatherV3(gdp, aRest, myRestAttr);
                geo.rest = myRestAttr.data();
            }
            weights.resize(nshapes);
            geo.nshapes = (int64_t)nshapes;
            geo.shapes_P = shapePtrs.data();
            geo.shapes_npoints = shapeCounts.data();
            geo.rest_changed = restChanged;
            geo.blends_changed = blendsChanged;
            geo.weights = weights.data();
            geo.weights_count = &weightsCount;
        }
        {
            // the rest rig of the previous cook again?  Then only the deltas are new (fd_set_deltas)
            int rigChanged = 1;
            checkChangedSourceFlags(1, context, &rigChanged);
            geo.rig_rest_unchanged = !rigChanged;
            // input 0 again, untouched (its attributes keep their data IDs when nothing upstream
            // recooked)?  Then the engine's device-resident mesh is still good.
            int meshChanged = 1;
            checkChangedSourceFlags(0, context, &meshChanged);
            geo.mesh_unchanged = !meshChanged;
        }
        // :380 -- the group parm, before the evaluation as in the reference
        if (cookInputGroups(context) >= UT_ERROR_ABORT) return error();
        fdsop_cook(myNode, &geo);

        // replay the engine's messages through the node's own channels
        std::string msgs(fdsop_messages(myNode));
        for (size_t p = 0; p < msgs.size();) {
            const size_t e = msgs.find('\n', p), tab = msgs.find('\t', p);
            const std::string sev = msgs.substr(p, tab - p), text = msgs.substr(tab + 1, e - tab - 1);
            if (sev == "error") addError(sev == "error" && text == "Rest and deform geometry should match." ? SOP_ERR_MISMATCH_POINT : SOP_ERR_NO_DEFORM_EFFECT, text.c_str());
            else if (sev == "warning") addWarning(SOP_MESSAGE, text.c_str());
            else addMessage(SOP_MESSAGE, text.c_str());
            p = e + 1;
        }
        if (error() >= UT_ERROR_ABORT) return error();

        scatterV3(gdp, gdp->getP(), Pout);
        GA_RWHandleF hf(gdp->addFloatTuple(GA_ATTRIB_POINT, "fd_falloff", 1));
        GA_RWHandleV3 hc(gdp->addFloatTuple(GA_ATTRIB_POINT, "Cd", 3, GA_Defaults(GA_STORE_REAL32, 3, 1.f, 1.f, 1.f)));
        GA_Offset o;
        GA_FOR_ALL_PTOFF(gdp, o) hf.set(o, falloff[(size_t)gdp->pointIndex(o)]);
        (void)hc;
        if (weightsCount > 0) {
            // :474-481, the detail array attribute `weights`
            GA_Attribute *wAttrib = gdp->addFloatArray(GA_ATTRIB_DETAIL, "weights", 1);
            const GA_AIFNumericArray *wAif = wAttrib->getAIFNumericArray();
            UT_FprealArray arr;
            arr.setSize(weightsCount);
            for (int64_t i = 0; i < weightsCount; ++i) arr(i) = weights[(size_t)i];
            wAif->set(wAttrib, 0, arr);
            wAttrib->bumpDataId();
        }
        // :483-486 -- we manage our own data IDs: P is bumped unless the group parm names an empty group
        if (!myGroup || !myGroup->isEmpty()) gdp->getP()->bumpDataId();
        return error();
    }

private:
    fdsop_node *myNode = nullptr;
    const GA_PointGroup *myGroup = nullptr;       // reference src/SOP_FaceDeform.hpp:108
    // ProximityCapture's inputs, kept from cook to cook (a static mesh builds them once)
    std::vector<int64_t> myEdgeOffsets;
    std::vector<int> myEdgeNeighbours;
    std::vector<float> myRigTris;
    bool myRigGathered = false;
    PinnedF myP, myRestP, myDeformP, myTu, myTv, myNn, myDist2, myPout, myFalloff, myRestAttr;
    // one gather buffer per blendshape input (PinnedF is not movable: held by pointer)
    struct ShapeSlot {
        std::unique_ptr<PinnedF> p;
        PinnedF *get(SOP_FaceDeformHip *) { if (!p) p.reset(new PinnedF()); return p.get(); }
    };
    std::vector<ShapeSlot> myShapes;
};

}  // namespace fdhip

void newSopOperator(OP_OperatorTable *table)
{
    table->addOperator(new OP_Operator("facedeform", "Face Deform", fdhip::SOP_FaceDeformHip::create,
                                       fdhip::buildTemplates(), 3, 1000, nullptr));
}
