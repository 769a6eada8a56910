// cook_harness.cpp -- cooks hdk/SOP_FaceDeformHip.cpp (the HDK-side wrapper, compiled against the mock in hdk_mock.h) on a PAGED
// detail whose point offsets are not point indices, against the real libfacedeform_hip.so.  Test infrastructure
// (tests/test_gpu_hdk_wrapper.py): SURVEY.md H5 asked for "a paged mock in the harness"; VERDICT r2 #10 for a compiler to read
// the wrapper at all.
//   cook_harness <in.bin> <out.bin>
// in:  int64 N, M, hole_every; float P[3N], rest[3M], deform_a[3M], deform_b[3M]
// out: float P_a[3N], P_b[3N], falloff_b[N]        (two cooks: everything new; then only the animated rig changed)
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "../../hdk/SOP_FaceDeformHip.cpp"

PRM_Name PRMgroupName("group", "Group");
static PRM_Name sNoNames[] = {PRM_Name(0)};
PRM_ChoiceList SOP_Node::pointGroupMenu(PRM_CHOICELIST_SINGLE, sNoNames);

static bool read_all(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n; }

static void fill_points(GU_Detail &d, const std::vector<float> &xyz, int hole_every)
{
    const GA_Size n = (GA_Size)(xyz.size() / 3);
    d.createPoints(n, hole_every);
    // every offset of every page gets a sentinel first: holes must still carry it after the cook
    for (auto &page : d.getP()->pages) for (float &v : page) v = -777.0f;
    for (GA_Size i = 0; i < n; ++i) d.setPos3(d.pointOffset(i), UT_Vector3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: cook_harness in.bin out.bin\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("in"); return 2; }
    int64_t hdr[3];
    if (!read_all(f, hdr, sizeof(hdr))) return 2;
    const int64_t N = hdr[0], M = hdr[1];
    const int hole = (int)hdr[2];
    std::vector<float> P(3 * N), rest(3 * M), da(3 * M), db(3 * M);
    if (!read_all(f, P.data(), 4 * P.size()) || !read_all(f, rest.data(), 4 * rest.size()) || !read_all(f, da.data(), 4 * da.size()) ||
        !read_all(f, db.data(), 4 * db.size())) return 2;
    fclose(f);

    GU_Detail mesh, rigRest, rigA, rigB;
    fill_points(mesh, P, hole);
    fill_points(rigRest, rest, 5);
    fill_points(rigA, da, 5);               // (same offset -> index map as the rest rig: the reference assumes it, :276-277)
    fill_points(rigB, db, 5);
    for (GA_Size i = 0; i + 2 < N; i += 2) {   // a strip of triangles: the edges ProximityCapture's flood walks
        GEO_Primitive tri;
        tri.verts = {mesh.pointOffset(i), mesh.pointOffset(i + 1), mesh.pointOffset(i + 2)};
        mesh.prims.push_back(tri);
    }

    OP_OperatorTable table;
    newSopOperator(&table);
    if (table.ops.size() != 1 || table.ops[0]->name != "facedeform" || table.ops[0]->label != "Face Deform" || table.ops[0]->minIn != 3 ||
        table.ops[0]->maxIn != 1000) { fprintf(stderr, "operator registration differs from the reference's (:38-45)\n"); return 3; }
    int nparms = 0, nhelp = 0;
    // the reference attaches a help text to these tokens (src/SOP_FaceDeform.cpp:121-137)
    const char *documented[] = {"model", "term", "radius", "maxedges", "tangent", "morphspace", "doclampweight", "weightrange", "falloffradius", "falloffrate"};
    for (PRM_Template *t = table.ops[0]->templates; t->type != PRM_LIST_TERMINATOR; ++t) {
        ++nparms;
        for (const char *d : documented)
            if (t->name && t->name->token && std::strcmp(t->name->token, d) == 0) {
                if (!t->help || !*t->help) { fprintf(stderr, "parm %s carries no help text\n", d); return 3; }
                ++nhelp;
            }
    }
    printf("parms: %d (%d with the reference's help texts)\n", nparms, nhelp);
    if (nhelp != (int)(sizeof(documented) / sizeof(documented[0]))) return 3;
    OP_Network net;
    std::unique_ptr<OP_Node> holder(table.ops[0]->ctor(&net, "facedeform1", table.ops[0].get()));
    SOP_Node *node = dynamic_cast<SOP_Node *>(holder.get());
    if (!node) return 3;
    // defaults of the reference's parm list (:117-137), thin-plate kernel (this repository's addition) for the oracle's sake
    node->mockParms = {{"model", {0}}, {"term", {0}}, {"qcoef", {1}}, {"zcoef", {5}}, {"radius", {1}}, {"maxedges", {4}}, {"layers", {4}},
                       {"lambda", {0.1}}, {"tangent", {0}}, {"morphspace", {0}}, {"doclampweight", {0}}, {"weightrange", {0, 1}},
                       {"dofalloff", {0}}, {"falloffradius", {1}}, {"falloffrate", {1}}, {"kernel", {1}}, {"smoothing", {0}},
                       {"precision", {0}}, {"device", {-1}}};
    OP_Context ctx;
    std::vector<float> outA(3 * N), outB(3 * N), fall(N);
    auto cook = [&](const GU_Detail &rigDeform, std::vector<int> changed, std::vector<float> &out) -> int {
        node->mockInputs = {&mesh, &rigRest, &rigDeform};
        node->mockChanged = changed;
        OP_ERROR e;
        try { e = node->cook(ctx); } catch (const std::string &s) { fprintf(stderr, "mock: %s\n", s.c_str()); return 4; }
        for (auto &m : node->mockMessages) printf("%s: %s\n", m.first.c_str(), m.second.c_str());
        if (e >= UT_ERROR_ABORT) return 5;
        GU_Detail *g = node->detail();
        if (g->getNumPoints() != N) return 6;
        for (GA_Size i = 0; i < N; ++i) { const UT_Vector3 v = g->getPos3(g->pointOffset(i)); out[3 * i] = v.x(); out[3 * i + 1] = v.y(); out[3 * i + 2] = v.z(); }
        // holes keep their sentinel: the scatter went through the page handles offset by offset
        for (GA_Offset o = 0; o < g->offsetEnd(); ++o)
            if (g->indexOf[(size_t)o] < 0 && g->getP()->at(o)[0] != -777.0f) { fprintf(stderr, "hole at offset %lld was written\n", (long long)o); return 7; }
        const GA_Attribute *fa = g->findFloatTuple(GA_ATTRIB_POINT, "fd_falloff", 1);
        const GA_Attribute *cd = g->findFloatTuple(GA_ATTRIB_POINT, "Cd", 3);
        if (!fa || !cd) { fprintf(stderr, "fd_falloff / Cd missing (:386-388, :401)\n"); return 8; }
        for (GA_Size i = 0; i < N; ++i) fall[i] = *fa->at(g->pointOffset(i));
        if (cd->at(g->pointOffset(0))[0] != 1.0f) return 8;      // white (:387)
        printf("P data id: %lld\n", (long long)g->getP()->dataId);
        return 0;
    };
    int rc = cook(rigA, {1, 1, 1}, outA);
    if (rc) return rc;
    rc = cook(rigB, {0, 0, 1}, outB);         // mesh and rest rig untouched: device-resident mesh + fd_set_deltas path
    if (rc) return rc;
    f = fopen(argv[2], "wb");
    if (!f) return 2;
    fwrite(outA.data(), 4, outA.size(), f); fwrite(outB.data(), 4, outB.size(), f); fwrite(fall.data(), 4, fall.size(), f);
    fclose(f);
    return 0;
}
