// hdk_mock.h -- a MOCK of the dozen Houdini HDK types hdk/SOP_FaceDeformHip.cpp touches, written for this repository's tests
// (SURVEY.md H5): enough for a compiler to read the wrapper and for a harness to cook it on a PAGED detail.
// Nothing here comes from the HDK (which is not available): names and signatures are those the wrapper uses, behaviour is the
// minimum the wrapper relies on -- point offsets that are not point indices (holes left by deleted points), attribute storage
// in pages of 1024 offsets reached through page handles, data ids, change flags per input, messages collected per node.
// Test infrastructure only.
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

typedef double fpreal;
typedef int64_t GA_Offset;
typedef int64_t GA_Index;
typedef int64_t GA_Size;
constexpr GA_Size GA_PAGE_SIZE = 1024;

enum GA_AttributeOwner { GA_ATTRIB_POINT, GA_ATTRIB_DETAIL };
enum GA_GroupType { GA_GROUP_POINT };
enum GA_Storage { GA_STORE_REAL32 };
enum UT_ErrorSeverity { UT_ERROR_NONE = 0, UT_ERROR_MESSAGE, UT_ERROR_WARNING, UT_ERROR_ABORT };
typedef UT_ErrorSeverity OP_ERROR;
enum { SOP_MESSAGE = 1, SOP_ERR_MISMATCH_POINT, SOP_ERR_NO_DEFORM_EFFECT };

struct UT_Vector3 {
    float v[3];
    UT_Vector3() : v{0, 0, 0} {}
    UT_Vector3(float x, float y, float z) : v{x, y, z} {}
    float x() const { return v[0]; }
    float y() const { return v[1]; }
    float z() const { return v[2]; }
};
struct UT_String {
    std::string s;
    const char *buffer() const { return s.c_str(); }
};
struct UT_FprealArray {
    std::vector<fpreal> a;
    void setSize(int64_t n) { a.resize((size_t)n); }
    fpreal &operator()(int64_t i) { return a[(size_t)i]; }
    int64_t size() const { return (int64_t)a.size(); }
};
template <typename T> struct UT_Array : std::vector<T> {};

struct GA_Defaults {
    float d[4] = {0, 0, 0, 0};
    GA_Defaults() {}
    GA_Defaults(GA_Storage, int n, float a, float b = 0, float c = 0) { d[0] = a; d[1] = b; d[2] = c; (void)n; }
};

struct GA_Attribute;
struct GA_AIFNumericArray {
    bool set(GA_Attribute *attr, GA_Offset off, const UT_FprealArray &arr) const;
};
// float tuples stored PAGE by PAGE: page p holds offsets [1024 p, 1024 p + 1023], tuple-interleaved
struct GA_Attribute {
    std::string name;
    int tuple = 1;
    int64_t dataId = 0;
    std::vector<std::vector<float>> pages;
    std::vector<fpreal> detailArray;                  // detail float array (addFloatArray)
    GA_AIFNumericArray aif;
    void ensure(GA_Offset noffsets, const GA_Defaults &def)
    {
        const size_t np = (size_t)((noffsets + GA_PAGE_SIZE - 1) / GA_PAGE_SIZE);
        while (pages.size() < np) {
            pages.emplace_back((size_t)GA_PAGE_SIZE * tuple);
            for (size_t i = 0; i < pages.back().size(); ++i) pages.back()[i] = def.d[i % tuple];
        }
    }
    float *at(GA_Offset o) { return &pages[(size_t)(o / GA_PAGE_SIZE)][(size_t)(o % GA_PAGE_SIZE) * tuple]; }
    const float *at(GA_Offset o) const { return &pages[(size_t)(o / GA_PAGE_SIZE)][(size_t)(o % GA_PAGE_SIZE) * tuple]; }
    void bumpDataId() { ++dataId; }
    const GA_AIFNumericArray *getAIFNumericArray() const { return &aif; }
};
inline bool GA_AIFNumericArray::set(GA_Attribute *attr, GA_Offset, const UT_FprealArray &arr) const { attr->detailArray = arr.a; return true; }

struct GA_PointGroup {
    std::vector<GA_Offset> members;
    bool isEmpty() const { return members.empty(); }
};

// a range of point offsets = the valid offsets of a detail, walked in runs that never cross a page
struct GA_Range { const std::vector<GA_Offset> *valid = nullptr; };
struct GA_SplittableRange : GA_Range {};
struct GA_Iterator {
    const std::vector<GA_Offset> *valid;
    size_t pos = 0;
    explicit GA_Iterator(const GA_Range &r) : valid(r.valid) {}
    bool blockAdvance(GA_Offset &start, GA_Offset &end)
    {
        if (!valid || pos >= valid->size()) return false;
        start = (*valid)[pos];
        end = start + 1;
        ++pos;
        while (pos < valid->size() && (*valid)[pos] == end && end / GA_PAGE_SIZE == start / GA_PAGE_SIZE) { ++end; ++pos; }
        return true;
    }
};

// page handles: bound to one page at a time, as the HDK's (an access outside the bound page is a bug the mock catches)
template <bool RW> struct GA_PageHandleV3T {
    GA_Attribute *attr;
    GA_Offset page = -1;
    explicit GA_PageHandleV3T(const GA_Attribute *a) : attr(const_cast<GA_Attribute *>(a)) {}
    void setPage(GA_Offset start) { page = start / GA_PAGE_SIZE; }
    UT_Vector3 get(GA_Offset o) const
    {
        if (o / GA_PAGE_SIZE != page) throw std::string("page handle read outside its page");
        const float *p = attr->at(o);
        return UT_Vector3(p[0], p[1], p[2]);
    }
    void set(GA_Offset o, const UT_Vector3 &v)
    {
        static_assert(RW, "read-only page handle");
        if (o / GA_PAGE_SIZE != page) throw std::string("page handle write outside its page");
        float *p = attr->at(o);
        p[0] = v.x(); p[1] = v.y(); p[2] = v.z();
    }
};
typedef GA_PageHandleV3T<false> GA_ROPageHandleV3;
typedef GA_PageHandleV3T<true> GA_RWPageHandleV3;
struct GA_RWHandleF {
    GA_Attribute *attr;
    explicit GA_RWHandleF(GA_Attribute *a) : attr(a) {}
    void set(GA_Offset o, float v) { *attr->at(o) = v; }
    float get(GA_Offset o) const { return *attr->at(o); }
};
struct GA_RWHandleV3 {
    GA_Attribute *attr;
    explicit GA_RWHandleV3(GA_Attribute *a) : attr(a) {}
    void set(GA_Offset o, const UT_Vector3 &v) { float *p = attr->at(o); p[0] = v.x(); p[1] = v.y(); p[2] = v.z(); }
};

struct GU_Detail;
struct GEO_Primitive {
    std::vector<GA_Offset> verts;
    bool closed = true;
    GA_Size getVertexCount() const { return (GA_Size)verts.size(); }
    bool isClosed() const { return closed; }
    GA_Offset getPointOffset(GA_Size v) const { return verts[(size_t)v]; }
};

struct GU_Detail {
    std::vector<GA_Offset> valid;                     // point index -> offset (ascending; holes where points were deleted)
    std::vector<GA_Index> indexOf;                    // offset -> index (-1: hole)
    std::map<std::string, std::unique_ptr<GA_Attribute>> pointAttribs, detailAttribs;
    std::vector<GEO_Primitive> prims;
    GA_Attribute *P = nullptr;
    GU_Detail() { P = addFloatTuple(GA_ATTRIB_POINT, "P", 3); }
    // n points, every `hole_every`-th offset left empty
    void createPoints(GA_Size n, int hole_every = 0)
    {
        valid.clear(); indexOf.clear();
        GA_Offset o = 0;
        for (GA_Size i = 0; i < n; ++i, ++o) {
            if (hole_every > 0 && o % hole_every == hole_every - 1) { indexOf.push_back(-1); ++o; }
            valid.push_back(o);
            indexOf.push_back(i);
        }
        for (auto &kv : pointAttribs) kv.second->ensure(o + 1, GA_Defaults());
    }
    GA_Offset offsetEnd() const { return valid.empty() ? 0 : valid.back() + 1; }
    GA_Size getNumPoints() const { return (GA_Size)valid.size(); }
    GA_Attribute *getP() { return P; }
    const GA_Attribute *getP() const { return P; }
    GA_Index pointIndex(GA_Offset o) const { return indexOf[(size_t)o]; }
    GA_Offset pointOffset(GA_Index i) const { return valid[(size_t)i]; }
    GA_Range getPointRange() const { GA_Range r; r.valid = &valid; return r; }
    UT_Vector3 getPos3(GA_Offset o) const { const float *p = P->at(o); return UT_Vector3(p[0], p[1], p[2]); }
    void setPos3(GA_Offset o, const UT_Vector3 &v) { float *p = P->at(o); p[0] = v.x(); p[1] = v.y(); p[2] = v.z(); }
    const GA_Attribute *findFloatTuple(GA_AttributeOwner, const char *name, int size) const
    {
        auto it = pointAttribs.find(name);
        return (it != pointAttribs.end() && it->second->tuple == size) ? it->second.get() : nullptr;
    }
    GA_Attribute *addFloatTuple(GA_AttributeOwner, const char *name, int size, const GA_Defaults &def = GA_Defaults())
    {
        auto it = pointAttribs.find(name);
        if (it != pointAttribs.end()) return it->second.get();
        std::unique_ptr<GA_Attribute> a(new GA_Attribute());
        a->name = name; a->tuple = size;
        a->ensure(offsetEnd() > 0 ? offsetEnd() : 1, def);
        GA_Attribute *raw = a.get();
        pointAttribs[name] = std::move(a);
        return raw;
    }
    GA_Attribute *addFloatArray(GA_AttributeOwner, const char *name, int)
    {
        auto &slot = detailAttribs[name];
        if (!slot) { slot.reset(new GA_Attribute()); slot->name = name; }
        return slot.get();
    }
    void copyFrom(const GU_Detail &src)
    {
        valid = src.valid; indexOf = src.indexOf; prims = src.prims;
        pointAttribs.clear();
        for (const auto &kv : src.pointAttribs) pointAttribs[kv.first].reset(new GA_Attribute(*kv.second));
        P = pointAttribs["P"].get();
    }
};
#define GA_FOR_ALL_PRIMITIVES(gdp, prim) for (size_t fd_pi_ = 0; fd_pi_ < (gdp)->prims.size() && ((prim) = &(gdp)->prims[fd_pi_], true); ++fd_pi_)
#define GA_FOR_ALL_PTOFF(gdp, o) for (size_t fd_oi_ = 0; fd_oi_ < (gdp)->valid.size() && ((o) = (gdp)->valid[fd_oi_], true); ++fd_oi_)

// ---- PRM ----
enum PRM_Type { PRM_STRING, PRM_ORD, PRM_FLT_J, PRM_FLT_LOG, PRM_INT_J, PRM_TOGGLE, PRM_LIST_TERMINATOR };
enum PRM_ChoiceListType { PRM_CHOICELIST_SINGLE };
enum PRM_RangeFlag { PRM_RANGE_UI, PRM_RANGE_RESTRICTED };
struct PRM_Name {
    const char *token, *label;
    PRM_Name(const char *t = nullptr, const char *l = nullptr) : token(t), label(l) {}
};
struct PRM_Default {
    float f;
    PRM_Default(float v = 0.f) : f(v) {}
};
struct PRM_ChoiceList {
    PRM_ChoiceListType type; PRM_Name *names;
    PRM_ChoiceList(PRM_ChoiceListType t, PRM_Name *n) : type(t), names(n) {}
};
struct PRM_Range {
    PRM_Range(PRM_RangeFlag, double lo, PRM_RangeFlag, double hi) : lo_(lo), hi_(hi) {}
    double lo_, hi_;
};
struct PRM_SpareData {};
typedef int (*PRM_Callback)(void *, int, fpreal, const void *);
struct PRM_Template {
    PRM_Type type = PRM_LIST_TERMINATOR; int size = 0; PRM_Name *name = nullptr; PRM_Default *def = nullptr;
    PRM_ChoiceList *menu = nullptr; PRM_Range *range = nullptr; const char *help = nullptr;
    PRM_Template() {}
    PRM_Template(PRM_Type t, int n, PRM_Name *nm, PRM_Default *d = nullptr, PRM_ChoiceList *m = nullptr, PRM_Range *r = nullptr,
                 PRM_Callback = nullptr, PRM_SpareData * = nullptr, int /*parm group*/ = 0, const char *h = nullptr)
        : type(t), size(n), name(nm), def(d), menu(m), range(r), help(h) {}
};
extern PRM_Name PRMgroupName;

// ---- OP / SOP ----
struct OP_Context {
    fpreal t = 0;
    fpreal getTime() const { return t; }
};
struct OP_Network {};
struct OP_Operator;
struct OP_Node { virtual ~OP_Node() {} };
typedef OP_Node *(*OP_Constructor)(OP_Network *, const char *, OP_Operator *);
struct CH_LocalVariable;
struct OP_Operator {
    std::string name, label; OP_Constructor ctor; PRM_Template *templates; unsigned minIn, maxIn;
    OP_Operator(const char *n, const char *l, OP_Constructor c, PRM_Template *t, unsigned mn, unsigned mx, CH_LocalVariable * = nullptr)
        : name(n), label(l), ctor(c), templates(t), minIn(mn), maxIn(mx) {}
};
struct OP_OperatorTable {
    std::vector<std::unique_ptr<OP_Operator>> ops;
    void addOperator(OP_Operator *op) { ops.emplace_back(op); }
};
struct SOP_Flags {
    bool managesDataIDs = false;
    void setManagesDataIDs(bool v) { managesDataIDs = v; }
};
class SOP_Node : public OP_Node
{
public:
    SOP_Node(OP_Network *, const char *, OP_Operator *) { gdp = &myDetail; }
    // ---- what the harness sets up (not HDK API) ----
    std::vector<const GU_Detail *> mockInputs;
    std::vector<int> mockChanged;                                   // per input: changed since the last cook
    std::map<std::string, std::vector<double>> mockParms;           // token -> values (ordinals as their number)
    std::map<std::string, std::string> mockStrings;
    std::vector<std::pair<std::string, std::string>> mockMessages;  // (severity, text)
    GA_PointGroup mockGroup;
    bool mockHasGroup = false;
    OP_ERROR cook(OP_Context &c) { mockMessages.clear(); mySeverity = UT_ERROR_NONE; return cookMySop(c); }
    GU_Detail *detail() { return gdp; }
    // ---- the API the wrapper uses ----
    static PRM_ChoiceList pointGroupMenu;
    static PRM_SpareData *getGroupSelectButton(GA_GroupType) { return nullptr; }
protected:
    GU_Detail *gdp;
    SOP_Flags mySopFlags;
    virtual OP_ERROR cookMySop(OP_Context &) = 0;
    virtual OP_ERROR cookInputGroups(OP_Context &, int = 0) { return error(); }
    OP_ERROR error() const { return mySeverity; }
    void addError(int, const char *m = nullptr) { mockMessages.emplace_back("error", m ? m : ""); mySeverity = UT_ERROR_ABORT; }
    void addWarning(int, const char *m = nullptr) { mockMessages.emplace_back("warning", m ? m : ""); if (mySeverity < UT_ERROR_WARNING) mySeverity = UT_ERROR_WARNING; }
    void addMessage(int, const char *m = nullptr) { mockMessages.emplace_back("message", m ? m : ""); if (mySeverity < UT_ERROR_MESSAGE) mySeverity = UT_ERROR_MESSAGE; }
    void duplicatePointSource(unsigned idx, OP_Context &) { myDetail.copyFrom(*mockInputs[idx]); }
    const GU_Detail *inputGeo(unsigned idx) const { return mockInputs[idx]; }
    unsigned nConnectedInputs() const { return (unsigned)mockInputs.size(); }
    void evalString(UT_String &s, const char *tok, int, fpreal) const
    {
        auto it = mockStrings.find(tok);
        if (it != mockStrings.end()) { s.s = it->second; return; }
        auto jt = mockParms.find(tok);
        s.s = jt != mockParms.end() ? std::to_string((int)jt->second[0]) : "0";
    }
    int64_t evalInt(const char *tok, int i, fpreal) const { auto it = mockParms.find(tok); return it != mockParms.end() ? (int64_t)it->second[(size_t)i] : 0; }
    fpreal evalFloat(const char *tok, int i, fpreal) const { auto it = mockParms.find(tok); return it != mockParms.end() ? it->second[(size_t)i] : 0.0; }
    void checkChangedSourceFlags(unsigned idx, OP_Context &, int *changed) { *changed = idx < mockChanged.size() ? mockChanged[idx] : 1; }
    OP_ERROR cookInputPointGroups(OP_Context &, const GA_PointGroup *&group, int, bool, int, int, bool, bool, bool, int)
    {
        group = mockHasGroup ? &mockGroup : nullptr;
        return error();
    }
private:
    GU_Detail myDetail;
    OP_ERROR mySeverity = UT_ERROR_NONE;
    friend struct OP_AutoLockInputs;
};
struct OP_AutoLockInputs {
    explicit OP_AutoLockInputs(SOP_Node *) {}
    OP_ERROR lock(OP_Context &) { return UT_ERROR_NONE; }
};
