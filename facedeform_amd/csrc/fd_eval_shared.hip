// fd_eval_shared.hip -- evaluation of ALL frames of a shot in one launch (fd_batch_deform_shared_dev): the frames
// share the mesh and the rest rig, so phi(|x - c|^2) is formed once per (vertex, centre) and contracted with every
// frame's weights on the fp16 matrix pipe.  Replaces F runs of the loop body of SOP_FaceDeform::cookMySop,
// reference src/SOP_FaceDeform.cpp:404-439, for F control-point delta sets on one rest rig (:268-287).
//
// Its own translation unit (the file had outgrown one, and the kernel's compile flags can differ from the one-frame
// kernels').  Accumulators stay in ordinary registers: AGPR accumulators would cost the matrix pipe and the issue
// port less per instruction (tools/ubench_mfma16.hip: 16.7 vs 18.6 cycles alone, 7.6 vs 10.1 cycles of issue time),
// but once a kernel uses AGPRs this compiler splits the 256 registers of a two-wave SIMD 128 / 128, and the kernel
// needs ~150 ordinary registers beside its 96 accumulators (tried: 19 spills, 540 accumulator reads).
// Built with -ffp-contract=off like the rest (explicit fmas only).
#include <cstdio>
#include <cstdlib>

#include <type_traits>

#include "fd_eval_common.h"

namespace fd {

namespace {

// ---- frames that share the mesh AND the rest rig: the contraction on the matrix pipe ------------------
// The frames of an animated shot, the blendshapes of one head (BASELINE configs 4 and 2-as-benchmarked):
// the same vertices against the same centres, only the deltas -- hence the weights -- differ.  Then
//     Delta_f(x_v) = poly_f(x_v) + sum_j phi(|x_v - c_j|^2) w_f[j]
// is Phi (N x M) times W (M x 3F): phi is formed ONCE per (vertex, centre) -- d2 on the matrix pipe as
// in k_deform32_tps_mfma, then one v_log_f32 and one multiply -- and the 3F-wide contraction, which is
// what costs 24 of the 38 vector instructions per 4 pairs in the one-frame kernel, becomes
// v_mfma_f32_16x16x32_f16 work: north_star's "N x M evaluation recast as a dense GEMM-like contraction".
// fp16 has 11 significant bits, so both operands go in as two pieces (hi = RN16(v), lo = RN16(v - hi):
// 22 bits) and a product is three instructions, hi*hi + hi*lo + lo*hi (the dropped lo*lo is 2^-22
// relative); accumulation is fp32 inside the matrix pipe.  Per-frame weights are scaled by a power of
// two so that their largest piece sits at 2^13 (fp16 range), undone exactly in the epilogue.
//
// Layout: a workgroup of 8 waves (two per SIMD) keeps the centre tiles and the weight tiles of a chunk
// of centres in LDS -- the whole model at M = 256, F = 32: 12 + 128 KiB -- and walks vertex groups of
// 512; a wave owns 64 vertices = 4 vertex tiles of 16.  Per 32 centres (one K block) and vertex tile:
// two d2 instructions, 8 log + 8 multiplies + the fp16 split per lane, and that lane's 8 phi values
// ARE its B operand (the k-slot <-> centre map is a free choice as long as the weight tiles use the
// same one: slot 8g + s = centre 32 kb + 16 (s >> 2) + 4 g + (s & 3)).  An output tile is 16 rows
// x 16 vertices with rows = 4 frames x (x, y, z, unused): lane group g' of the accumulator then holds
// all three components of frame 4 T + g' for its vertex and writes them as 12 contiguous bytes.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int kSharedThreads = 512;
constexpr int kWideWaves = 8;      // waves per workgroup of the two-tile 32-row kernel: two per SIMD, up to 256 registers each
constexpr size_t kSharedLdsBudget = 158 * 1024;   // of 160 KiB (one workgroup per CU)

// Two row layouts of the output tiles (16 rows x 16 vertices each):
//   padded (up to 12 frames): tile T holds frames 4 T .. 4 T + 3, row = 4 (frame - 4 T) + component, one row in
//     four unused -- ceil(F / 4) tiles;
//   dense (13 frames and more): frames come in blocks of 16 and a block is three tiles, one per component:
//     tile 3 B + c holds component c of frames 16 B .. 16 B + 15, row = frame - 16 B -- 3 ceil(F / 16) tiles, no
//     unused rows at F = 16, 32 (6 tiles instead of 8 at 32 frames: a quarter fewer matrix instructions).
// Either way the 4 x 4 transpose across lane groups in the epilogue leaves every lane with all rows of
// every tile for ONE vertex.
constexpr int kGaussShift = 10;      // Gaussian kinds: phi (<= 1) enters the matrix pipe as 2^10 phi, clear of the fp16 subnormals
constexpr bool shared_dense(int nF) { return nF > 12; }
constexpr int shared_tiles(int nF) { return shared_dense(nF) ? 3 * ((nF + 15) / 16) : (nF + 3) / 4; }
constexpr int shared_slots(int nT, bool dense) { return dense ? nT / 3 * 16 : nT * 4; }     // frame records

struct SharedFrame {              // one per frame slot (nT * 4), written by k_pack_shared
    float inv_scale;              // 2^-k: undoes the scaling of the frame's weights and polynomial
    int built;                    // terminationtype == 1
    int pad[2];
    float *P_out, *falloff_out;   // the frame's outputs (read from LDS inside the frame loop: 64 pointers
                                  // as kernel arguments end up hoisted into SGPRs all at once and spilled)
};
static_assert(sizeof(SharedFrame) == 32, "frame record");

struct SharedOut {                // per-frame outputs (kernel argument)
    float *P_out[kMaxBatch];
    float *falloff_out[kMaxBatch];
};

struct SharedParams {
    int64_t N;
    const float *P_in;
    const float *dist2;
    const float *tu, *tv, *nrm;
    float radius2, falloffrate;
    int ntiles;                   // centre tiles (Mpad / 16)
    int nkb;                      // K blocks of 32 centres = ceil(ntiles / 2)
    int nF, nT;                   // frames, output tiles (4 frames each)
    int kchunk;                   // K blocks staged in LDS at a time
    const MfmaTileH *ctiles;      // centre tiles of the shared rest rig: the pack kernel's copy, 2 nkb tiles (zeros beyond ntiles)
    const float *norm;            // normalisation of the shared rest rig (DevModel::norm32), the pack kernel's copy
    const uint4 *wtiles;          // [nkb][nT][2 (hi, lo)][64 lanes] x 16 B, then the polynomial tiles [nT][64 lanes] x 16 B
    const SharedFrame *frames;    // [nT * 4]
    int dbg;                      // FD_SHARED_DBG (diagnostics, tests/tools/shared_eval_timing.py): 1 = no stores, 2 = no K loop
    int fast;                     // no dist2, no tangent frames, every frame slot in use and built, fd_falloff wanted everywhere:
                                  // full vertex groups take the branch-free epilogue (below)
    int delta;                    // FD_OUTPUT_DISPLACEMENT: write d f instead of P + d f (general epilogue only: the host clears `fast`)
    int stagger;                  // waves 4..7 start this many x 8192 cycles late (resident model only)
    unsigned long long *stamps;   // diagnostics (FD_SHARED_STAMPS): shader-clock shares of the phases, per wave of workgroup 0
};

struct SharedSlots {              // the models of the frames (kernel argument of the pack kernel)
    const Rec32 *rec32[kMaxBatch];
    const DevModel *model[kMaxBatch];
    const double *centres[kMaxBatch];     // fp64 centres as built / imported (M x 3): compared with model 0's
    int M;
    int nreal;                            // frames of the launch (the slots beyond repeat frame nreal - 1)
    int *mismatch;                        // page-locked word (device address) or null
};

// "One rest rig" by content: frame f's centres against frame 0's, all M of them, by the `per` lanes (a power of two inside
// one wave, lane index l) that also scan the frame's weights in the pack workgroup that writes its frame record.  A frame
// that differs is reported (1 + index) and recorded as unbuilt, i.e. passed through.  Every lane must call it.
__device__ __forceinline__ bool rig_matches(const SharedSlots &slots, int f, int nF, int l, int per)
{
    bool same = true;
    if (f > 0 && f < nF && slots.centres[f] != slots.centres[0]) {
        // (eight pairs in flight per lane and no short circuit: one pair at a time behind `same &&` this comparison was
        //  the packing kernel's 30 us -- 96 dependent round trips per lane at 8 lanes per frame)
        const double *a = slots.centres[f], *b0 = slots.centres[0];
        const int n = 3 * slots.M;
        bool diff = false;
        int e = l;
        for (; e + 7 * per < n; e += 8 * per) {
            double va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { va[u] = a[e + u * per]; vb[u] = b0[e + u * per]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) diff |= !(va[u] == vb[u]);
        }
        for (; e < n; e += per) diff |= !(a[e] == b0[e]);
        same = !diff;
    }
    for (int off = per / 2; off >= 1; off >>= 1) same = (__shfl_xor((int)same, off) != 0) && same;
    if (!same && l == 0 && slots.mismatch) *slots.mismatch = (f < slots.nreal ? f : slots.nreal - 1) + 1;
    return same;
}

// weight tiles, polynomial tiles and frame records from the solved models.  grid (nkb, nT), 256 threads (the first 64 write the tile).
// The polynomial part of a frame (DevModel::poly32: C0 + L.x' + q |x'|^2 per output) rides in the
// same matrix product as five more "centres" whose phi are (1, x', y', z', |x'|^2): one K = 32
// instruction per output tile and vertex tile holds all three split products -- lane group 0
// pairs hi x hi, group 1 lo(vertex) x hi(coefficient), group 2 hi(vertex) x lo(coefficient).
__global__ __launch_bounds__(256) void k_pack_shared(const SharedSlots slots, const SharedOut out, int nF, int Mpad, int dense,
                                                      uint4 *wtiles, SharedFrame *frames, const MfmaTileH *ctiles, int gauss)
{
    const int kb = blockIdx.x, T = blockIdx.y, nT = gridDim.y, nkb = gridDim.x;
    if (T == 0 && gauss) {
        // Gaussian kinds: the slot of a K block's two centre tiles holds its 32 centre records instead
        // ({c'x, c'y, c'z, -log2(e) s^2 / R_j^2}: the first half of Rec32), read by direct differences
        constexpr int per = (int)(sizeof(MfmaTileH) / 16);
        uint4 *dst = wtiles + (size_t)nkb * nT * 128 + (size_t)nT * 64;
        if (threadIdx.x < 32) {
            const int centre = 32 * kb + (int)threadIdx.x;
            dst[(size_t)2 * kb * per + threadIdx.x] =
                centre < Mpad ? *reinterpret_cast<const uint4 *>(&slots.rec32[0][centre]) : make_uint4(0u, 0u, 0u, 0u);
        }
        if (kb == 0 && threadIdx.x == 255) {
            const float *nn = slots.model[0]->norm32;
            dst[(size_t)2 * nkb * per] = make_uint4(__float_as_uint(nn[0]), __float_as_uint(nn[1]), __float_as_uint(nn[2]), __float_as_uint(nn[3]));
        }
    } else if (T == 0) {
        // everything else the evaluation reads of the contexts: the rest rig's centre tiles and normalisation.  With
        // these in the batch's scratch the contexts are free for their next build as soon as THIS kernel has run.
        constexpr int per = (int)(sizeof(MfmaTileH) / 16);
        uint4 *dst = wtiles + (size_t)nkb * nT * 128 + (size_t)nT * 64;
        const int ntiles = Mpad / 16;
        if ((int)threadIdx.x < 2 * per) {
            const int tile = 2 * kb + (int)threadIdx.x / per;
            dst[(size_t)2 * kb * per + threadIdx.x] =
                tile < ntiles ? reinterpret_cast<const uint4 *>(ctiles + tile)[threadIdx.x % per] : make_uint4(0u, 0u, 0u, 0u);
        }
        if (kb == 0 && threadIdx.x == 255) {
            const float *nn = slots.model[0]->norm32;
            dst[(size_t)2 * nkb * per] = make_uint4(__float_as_uint(nn[0]), __float_as_uint(nn[1]), __float_as_uint(nn[2]), __float_as_uint(nn[3]));
        }
    }
    const int lane = threadIdx.x & 63, g = lane >> 4, rho = lane & 15;
    // row rho of tile T: frame f0 + fi, component c
    const int nfr = dense ? 16 : 4, f0 = dense ? 16 * (T / 3) : 4 * T;
    const int fi = dense ? rho : rho >> 2, c = dense ? T % 3 : rho & 3;
    // scale of each of this tile's frames: largest |weight| or |polynomial coefficient| to [2^13, 2^14).
    // 256 / nfr lanes per frame, all of a frame's records requested at once.
    __shared__ float s_scale[16];
    {
        const int per = 256 / nfr, q = threadIdx.x / per, l = threadIdx.x % per;
        const int f = f0 + q;
        float m = 0.f;
        if (f < nF) {
            m = slots.model[f]->wmax32;          // (left there by the packing code of the build: fd_pack.h)
        }
        for (int off = per / 2; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));      // per is 16 or 64: inside a wave
        const bool same_rig = (kb == 0 && (!dense || T % 3 == 0)) ? rig_matches(slots, f, nF, l, per) : true;
        if (l == 0) {
            int k = 0;
            if (m > 0.f && m < INFINITY) k = 13 - (__builtin_amdgcn_frexp_expf(m) - 1);
            k = k < -100 ? -100 : (k > 100 ? 100 : k);
            s_scale[q] = ldexpf(1.f, k);
            if (kb == 0 && (!dense || T % 3 == 0)) {
                SharedFrame fr;
                fr.inv_scale = ldexpf(1.f, -k - (gauss ? kGaussShift : 0));
                fr.built = (f < nF && same_rig && slots.model[f]->terminationtype == 1) ? 1 : 0;
                fr.pad[0] = fr.pad[1] = 0;
                fr.P_out = f < nF ? out.P_out[f] : nullptr;
                fr.falloff_out = f < nF ? out.falloff_out[f] : nullptr;
                frames[f] = fr;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int f = f0 + fi;
    const float sc = s_scale[fi];
    f16x8 hi, lo;
#pragma unroll
    for (int sidx = 0; sidx < 8; ++sidx) {
        const int centre = 32 * kb + 16 * (sidx >> 2) + 4 * g + (sidx & 3);
        float w = 0.f;
        if (f < nF && c < 3 && centre < Mpad) {
            const Rec32 r = slots.rec32[f][centre];
            w = (c == 0 ? r.wx : (c == 1 ? r.wy : r.wz)) * sc;
        }
        const _Float16 h = (_Float16)w;
        hi[sidx] = h;
        lo[sidx] = (_Float16)(w - (float)h);
    }
    uint4 *dst = wtiles + ((size_t)kb * nT + T) * 128;
    dst[lane] = __builtin_bit_cast(uint4, hi);
    dst[64 + lane] = __builtin_bit_cast(uint4, lo);
    if (kb == 0) {
        // polynomial tile of output tile T: k-slot s < 5 of lane group g carries coefficient s
        // ({C0, Lx, Ly, Lz, q}) of row rho -- hi piece in groups 0 and 1, lo piece in group 2
        f16x8 pt;
#pragma unroll
        for (int sidx = 0; sidx < 8; ++sidx) {
            float w = 0.f;
            if (f < nF && c < 3 && sidx < 5 && g < 3) w = slots.model[f]->poly32[5 * c + sidx] * sc;
            const _Float16 h = (_Float16)w;
            pt[sidx] = g == 2 ? (_Float16)(w - (float)h) : h;
        }
        wtiles[(size_t)nkb * nT * 128 + (size_t)T * 64 + lane] = __builtin_bit_cast(uint4, pt);
    }
}

// fp32 pair -> its two fp16 pieces, packed: hi = RN16(v), lo = RN16(v - hi).  One v_cvt_pk_f16_f32 and
// two mixed-precision fmas that subtract the fp16 piece from the fp32 value and round the
// remainder to fp16 in the same instruction (v_fma_mixlo/hi_f16 write one half of the destination
// and keep the other) -- three instructions for two values, no unpacking, no repacking.
// PLAIN: the same two pieces from instructions the compiler sees (conversions and a subtraction: about twice as
// many).  The Gaussian kinds take this form: there the inputs come straight from v_exp_f32, and a vector
// instruction hidden in an asm string that reads a transcendental's result gets none of the wait states the
// compiler pads that pair with (hipcc pads nothing inside or around asm strings).
template <bool PLAIN = false, bool NOP = true>
__device__ __forceinline__ void split_pair_f16(float v0, float v1, unsigned &hi, unsigned &lo)
{
    if constexpr (PLAIN) {
        const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
        const _Float16 l0 = (_Float16)(v0 - (float)h0), l1 = (_Float16)(v1 - (float)h1);
        hi = __builtin_bit_cast(unsigned, (f16x2){h0, h1});
        lo = __builtin_bit_cast(unsigned, (f16x2){l0, l1});
        return;
    }
    const f16x2 hh = __builtin_convertvector((f32x2){v0, v1}, f16x2);
    hi = __builtin_bit_cast(unsigned, hh);
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(v0));
    // (s_nop 1 inside the string: a register written by a vector instruction needs two wait states before a matrix
    // instruction reads it as an operand, and the compiler pads only producers it can see; no measurable cost:
    // 218-229 us per C2 x 32 launch with it, 210-237 without, same box)
    // (NOP = false: the caller fences a whole block of pairs with ONE s_nop behind the last of them, fence_operands below)
    if constexpr (NOP) asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 1" : "+v"(l) : "v"(hi), "v"(v1));
    else asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(v1));
    lo = l;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
// one vertex position as a single 12-byte store (dword-aligned: global_store_dwordx3)
struct __attribute__((packed, aligned(4))) Pos3 { float x, y, z; };
__device__ __forceinline__ void store_pos3(Pos3 FD_GLOBAL *dst, float x, float y, float z)
{
    dst->x = x; dst->y = y; dst->z = z;      // member-wise: a struct assignment through an address-space pointer does not compile on the host pass
}

// the same with the non-temporal hint: the 512 MB a launch writes need not displace what else lives in L2
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef f32x3 f32x3_a4 __attribute__((aligned(4)));
typedef f32x4 f32x4_a16 __attribute__((aligned(16)));
typedef f32x2 f32x2_a8 __attribute__((aligned(8)));
__device__ __forceinline__ void store_pos3_nt(Pos3 FD_GLOBAL *dst, float x, float y, float z)
{
    __builtin_nontemporal_store((f32x3){x, y, z}, (f32x3_a4 FD_GLOBAL *)dst);
}

template <int NT, bool DENSE, bool GAUSS>
__global__ __launch_bounds__(kSharedThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_deform32_tps_shared(const SharedParams p, int ngroups)
{
    constexpr int TV = 4;                        // vertex tiles per wave
    constexpr int kSlots = shared_slots(NT, DENSE);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: [frame records kSlots][polynomial tiles NT*64 x 16 B][centre tiles kchunk*2][weight tiles kchunk*NT*2*64 x 16 B]
    SharedFrame *s_frames = reinterpret_cast<SharedFrame *>(smem);
    uint4 *s_poly = reinterpret_cast<uint4 *>(smem + sizeof(SharedFrame) * (size_t)kSlots);
    MfmaTileH *s_ct = reinterpret_cast<MfmaTileH *>(s_poly + NT * 64);
    uint4 *s_w = reinterpret_cast<uint4 *>(reinterpret_cast<char *>(s_ct) + sizeof(MfmaTileH) * (size_t)(2 * p.kchunk));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform, and the compiler should know it
    const int g = lane >> 4, j = lane & 15;
    const float n0 = p.norm[0], n1 = p.norm[1], n2 = p.norm[2];
    const float inv_s = p.norm[3];
    const bool resident = p.nkb <= p.kchunk;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int kb0, int nk) {
        __syncthreads();
        {   // centre tiles 2 kb0 .. 2 (kb0 + nk) - 1; beyond ntiles: zeros
            const uint4 *src = reinterpret_cast<const uint4 *>(p.ctiles + 2 * kb0);
            uint4 *dst = reinterpret_cast<uint4 *>(s_ct);
            const int per = (int)(sizeof(MfmaTileH) / 16);
            for (int q = tid; q < 2 * nk * per; q += kSharedThreads) dst[q] = src[q];
        }
        {
            // eight loads in flight per thread (native vectors: an array of HIP's uint4 structs goes through scratch memory)
            const u32x4 *src = reinterpret_cast<const u32x4 *>(p.wtiles + (size_t)kb0 * NT * 128);
            u32x4 *dst = reinterpret_cast<u32x4 *>(s_w);
            const int n16 = nk * NT * 128;
            int q = tid;
            for (; q + 7 * kSharedThreads < n16; q += 8 * kSharedThreads) {
                const u32x4 v0 = src[q], v1 = src[q + kSharedThreads], v2 = src[q + 2 * kSharedThreads], v3 = src[q + 3 * kSharedThreads];
                const u32x4 v4 = src[q + 4 * kSharedThreads], v5 = src[q + 5 * kSharedThreads], v6 = src[q + 6 * kSharedThreads], v7 = src[q + 7 * kSharedThreads];
                dst[q] = v0; dst[q + kSharedThreads] = v1; dst[q + 2 * kSharedThreads] = v2; dst[q + 3 * kSharedThreads] = v3;
                dst[q + 4 * kSharedThreads] = v4; dst[q + 5 * kSharedThreads] = v5; dst[q + 6 * kSharedThreads] = v6; dst[q + 7 * kSharedThreads] = v7;
            }
            for (; q < n16; q += kSharedThreads) dst[q] = src[q];
        }
        __syncthreads();
    };

    // The frame records (scale, status, output pointers: 8 dwords x 4 NT frames) live across the
    // lanes of NT / 2 registers for the whole kernel; the epilogue picks a frame's scalars out with
    // v_readlane -- no memory round trip per frame.  (From LDS every frame paid an LDS read behind
    // the other wave's operand traffic; through the scalar cache 800 cycles per pair of frames; as
    // kernel arguments the compiler hoists 32 x 6 scalars above the K loop and spills them.)
    constexpr int kTabRegs = (kSlots * 8 + 63) / 64;
    unsigned tab[kTabRegs];
#pragma unroll
    for (int q = 0; q < kTabRegs; ++q) {
        const int idx = 64 * q + lane;
        tab[q] = idx < kSlots * 8 ? reinterpret_cast<const unsigned *>(p.frames)[idx] : 0u;
    }
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.frames);
        uint4 *dst = reinterpret_cast<uint4 *>(s_frames);
        for (int q = tid; q < kSlots * (int)(sizeof(SharedFrame) / 16); q += kSharedThreads) dst[q] = src[q];
        const uint4 *psrc = p.wtiles + (size_t)p.nkb * NT * 128;
        for (int q = tid; q < NT * 64; q += kSharedThreads) s_poly[q] = psrc[q];
    }
    if (resident) {
        stage(0, p.nkb);
        // The two waves of a SIMD (w and w + 4) run the same program: left alone they reach their
        // logarithm phase together and their matrix phase together, and each phase then has one
        // pipe idle.  A start-up delay for the second half puts one wave's vector work beside
        // the other's matrix work (no barrier follows while the model is resident).
        if (wave >= 4) {
            __builtin_amdgcn_s_sleep(12);
            for (int q = 0; q < (p.stagger & 0xff); ++q) __builtin_amdgcn_s_sleep(127);
        }
        // experiment: workgroups start in four phases ((stagger >> 8) x 8128 cycles apart)
        for (int q = 0; q < (p.stagger >> 8) * (int)(blockIdx.x & 3); ++q) __builtin_amdgcn_s_sleep(127);
    } else {
        __syncthreads();
    }

    const bool stamp = p.stamps != nullptr && blockIdx.x == 0;
    unsigned long long st_prev = 0, st_acc[5] = {0, 0, 0, 0, 0};
#define FD_SSTAMP(K) if (stamp) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[K] += t_ - st_prev; st_prev = t_; __builtin_amdgcn_sched_barrier(0); }
    if (stamp) st_prev = __builtin_amdgcn_s_memtime();
    // Inputs of a vertex group as lane (g, j) holds them: the four vertices (vt, j).  The NEXT group's
    // are requested before this group's stores go out: vector memory operations of a wave retire in
    // issue order, so a load queued behind the epilogue's 64 stores would wait for all of them.
    struct GroupRaw { float p[TV][3]; float d2[TV]; };
    auto load_raw = [&](int grp_, auto fastTag) {
        constexpr bool FAST = decltype(fastTag)::value;
        GroupRaw r;
        const int64_t vb = ((int64_t)grp_ * (kSharedThreads / 64) + wave) * (16 * TV);
#pragma unroll
        for (int t = 0; t < TV; ++t) {
            const int64_t vi = vb + 16 * t + j;
            const int64_t vc = vi < p.N ? vi : p.N - 1;
            r.p[t][0] = p.P_in[3 * vc]; r.p[t][1] = p.P_in[3 * vc + 1]; r.p[t][2] = p.P_in[3 * vc + 2];
            if constexpr (FAST) r.d2[t] = 0.f; else r.d2[t] = p.dist2 ? p.dist2[vc] : 0.f;
        }
        return r;
    };
    // "These loaded registers are needed now": placed in straight-line code right after a group's stores, it lets the
    // compiler count exactly how many stores follow the loads and wait with that count; consumed for the first time at
    // the top of the next iteration -- where the path from the prologue joins -- it would have to assume none.
    auto settle = [&](const GroupRaw &r) {
        asm volatile("" :: "v"(r.p[0][0]), "v"(r.p[0][1]), "v"(r.p[0][2]), "v"(r.p[1][0]), "v"(r.p[1][1]), "v"(r.p[1][2]),
                           "v"(r.p[2][0]), "v"(r.p[2][1]), "v"(r.p[2][2]), "v"(r.p[3][0]), "v"(r.p[3][1]), "v"(r.p[3][2]));
    };
    GroupRaw nxt = load_raw(blockIdx.x < (unsigned)ngroups ? (int)blockIdx.x : 0, std::false_type{});
    settle(nxt);
    // One vertex group (512 vertices of the workgroup, 64 of this wave).  FAST: the group is full and the
    // launch has no gate, fall-off, tangent frames or unbuilt frames -- the epilogue is then straight-line
    // code in which every lane issues every store.  That matters beyond the instruction count: with no
    // branch that could skip a store, the compiler KNOWS 64 stores follow the next group's loads and waits
    // for those loads with vmcnt(63); with conditional stores it has to assume none were issued, waits with
    // vmcnt(0), and every wave sits out the drain of its own 32 KB of stores (all waves at once: the whole
    // chip alternated between a matrix phase with HBM idle and a store phase with the pipes idle).
    auto do_group = [&](int grp, auto fastTag) {
        constexpr bool FAST = decltype(fastTag)::value;
        const int64_t vbase = ((int64_t)grp * (kSharedThreads / 64) + wave) * (16 * TV);
        // The two waves of a SIMD (w, w + 4) take the higher issue priority in turn, group by group: left to the
        // default (oldest first) waves 0..3 finish all their groups a quarter of the kernel early and the others run
        // the rest with nobody to fill their stalls.
        if ((((grp / (int)gridDim.x) ^ (wave >> 2)) & 1) != 0) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
        const GroupRaw cur = nxt;
        // this lane's own vertex in the epilogue is (vt = g, j): one of the four it has loaded
        auto pick = [&](float a0, float a1, float a2, float a3) {        // two levels of v_cndmask, no branches
            const float lo = (g & 1) ? a1 : a0, hi = (g & 1) ? a3 : a2;
            return (g & 2) ? hi : lo;
        };
        const float pos[3] = {pick(cur.p[0][0], cur.p[1][0], cur.p[2][0], cur.p[3][0]), pick(cur.p[0][1], cur.p[1][1], cur.p[2][1], cur.p[3][1]),
                              pick(cur.p[0][2], cur.p[1][2], cur.p[2][2], cur.p[3][2])};
        const float own_d2 = pick(cur.d2[0], cur.d2[1], cur.d2[2], cur.d2[3]);
        // every lane group holds vertex (vt, j): the d2 operand needs one coordinate of it per lane
        // group, the polynomial operand all of them
        f16x4 bop[TV];
        float xn[TV], yn[TV], zn[TV];            // GAUSS: the normalised coordinates, for the direct differences
        f32x4 acc[NT][TV];
        bool lane_live = false;
#pragma unroll
        for (int t = 0; t < TV; ++t) {
            const int64_t vi = vbase + 16 * t + j;
            const float x = (cur.p[t][0] - n0) * inv_s, y = (cur.p[t][1] - n1) * inv_s, z = (cur.p[t][2] - n2) * inv_s;
            xn[t] = x; yn[t] = y; zn[t] = z;
            const float d2v = cur.d2[t];
            if constexpr (FAST) lane_live = true; else lane_live |= (vi < p.N) && !(d2v > p.radius2);
            const float xx = __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x));
            const float v2 = g == 0 ? -2.f * x : (g == 1 ? -2.f * y : (g == 2 ? -2.f * z : xx));
            const _Float16 h = (_Float16)v2;
            const _Float16 l = (_Float16)(v2 - (float)h);
            const _Float16 one = (_Float16)1.0f;
            bop[t] = g < 3 ? (f16x4){h, l, h, l} : (f16x4){one, one, h, l};
            // polynomial operand: k-slots {1, x', y', z', |x'|^2}: hi pieces in lane groups 0 and 2,
            // lo pieces in group 1 (against the coefficients' hi), nothing in group 3
            // (GAUSS: everything times 2^10, the factor phi carries; undone with the frame's scale)
            constexpr float ps = GAUSS ? (float)(1 << kGaussShift) : 1.f;
            constexpr unsigned one16 = GAUSS ? 0x6400u : 0x3c00u;         // fp16 1024 / 1
            unsigned xyh, xyl, zxh, zxl;
            split_pair_f16(x * ps, y * ps, xyh, xyl);
            split_pair_f16(z * ps, xx * ps, zxh, zxl);
            u32x4 pb;
            if (g == 1) pb = (u32x4){xyl << 16, (xyl >> 16) | (zxl << 16), zxl >> 16, 0u};                  // {0, xl, yl, zl, xxl}
            else pb = (u32x4){one16 | (xyh << 16), (xyh >> 16) | (zxh << 16), zxh >> 16, 0u};              // {1, xh, yh, zh, xxh}
            if (g == 3) pb = (u32x4){0u, 0u, 0u, 0u};
            const f16x8 pbv = __builtin_bit_cast(f16x8, pb);
#pragma unroll
            for (int T = 0; T < NT; ++T)
                acc[T][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, s_poly[T * 64 + lane]), pbv, zero4, 0, 0, 0);
        }
        const bool wave_work = FAST ? true : __any(lane_live);
        if constexpr (FAST) {
            // The next group's positions are requested HERE, a whole K loop before the stores of this group: a wave
            // has at most 63 vector-memory operations in flight and they retire in order, so loads requested just
            // before the epilogue's stores would hold up the last of them until their own data (queued behind a
            // chip-wide burst of stores) has come back -- every epilogue then lasts one loaded-memory round trip.
            const int gn = grp + (int)gridDim.x;
            nxt = load_raw(gn < ngroups ? gn : grp, fastTag);
        }
        FD_SSTAMP(0)

        // phi of K block kb (32 centres) for the wave's four vertex tiles, split into fp16 pieces: the B operands
        auto phi_block = [&](int kb, u32x4 (&xh)[TV], u32x4 (&xl)[TV]) {
            if constexpr (GAUSS) {
                // exp(-d2 / R_j^2) from direct coordinate differences (the expanded form of the matrix-pipe d2 has
                // an absolute error the exponent multiplies by 1 / R^2: DESIGN.md 4.1), two vertex tiles per packed
                // instruction; this lane's 8 centres of the block come from LDS once for the four vertex tiles
                const float4 *cr = reinterpret_cast<const float4 *>(&s_ct[2 * kb]);
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {         // centres 4 g .. 4 g + 3 of each 16-centre half of the block
                    float4 c[4];
#pragma unroll
                    for (int sl = 0; sl < 4; ++sl) c[sl] = cr[16 * hf + 4 * g + sl];
#pragma unroll
                    for (int tp = 0; tp < TV; tp += 2) {
                        float ph[4][2];
#pragma unroll
                        for (int sl = 0; sl < 4; ++sl) {
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                // ONE value per instruction.  The packed form (v_pk_add/mul/fma_f32 on two vertex tiles at
                                // once, 7 instructions instead of 14) gave results that differed from launch to launch on a
                                // few vertices per million once this block ran software-pipelined under the matrix
                                // instructions of the previous one -- with v_exp_f32 and with a polynomial exponential
                                // alike, not without the pipelining; this form: no difference in 54 launches of 32M
                                // vertex-frames each (tests/test_gpu_shared.py::test_repeated_launches...).  Cause not
                                // established; the thin-plate block has no packed arithmetic.
                                const float dx = xn[tp + u] - c[sl].x, dy = yn[tp + u] - c[sl].y, dz = zn[tp + u] - c[sl].z;
                                float d2 = dx * dx;
                                d2 = __builtin_fmaf(dy, dy, d2);
                                d2 = __builtin_fmaf(dz, dz, d2);
                                ph[sl][u] = __builtin_amdgcn_exp2f(__builtin_fmaf(d2, c[sl].w, (float)kGaussShift));
                            }
                        }
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            unsigned h, l;
                            split_pair_f16<true>(ph[2 * q][0], ph[2 * q + 1][0], h, l); xh[tp][2 * hf + q] = h; xl[tp][2 * hf + q] = l;
                            split_pair_f16<true>(ph[2 * q][1], ph[2 * q + 1][1], h, l); xh[tp + 1][2 * hf + q] = h; xl[tp + 1][2 * hf + q] = l;
                        }
                    }
                }
                return;
            }
            const f16x4 aopA = *reinterpret_cast<const f16x4 *>(&s_ct[2 * kb].a[lane][0]);
            const f16x4 aopB = *reinterpret_cast<const f16x4 *>(&s_ct[2 * kb + 1].a[lane][0]);
#pragma unroll
            for (int t = 0; t < TV; ++t) {
                const f32x4 da = __builtin_amdgcn_mfma_f32_16x16x16f16(aopA, bop[t], zero4, 0, 0, 0);
                const f32x4 db = __builtin_amdgcn_mfma_f32_16x16x16f16(aopB, bop[t], zero4, 0, 0, 0);
                unsigned h, l;
                split_pair_f16(d2_log_d2(da[0]), d2_log_d2(da[1]), h, l); xh[t][0] = h; xl[t][0] = l;
                split_pair_f16(d2_log_d2(da[2]), d2_log_d2(da[3]), h, l); xh[t][1] = h; xl[t][1] = l;
                split_pair_f16(d2_log_d2(db[0]), d2_log_d2(db[1]), h, l); xh[t][2] = h; xl[t][2] = l;
                split_pair_f16(d2_log_d2(db[2]), d2_log_d2(db[3]), h, l); xh[t][3] = h; xl[t][3] = l;
            }
        };
        // acc += W(kb) x phi(kb): three split products per output tile and vertex tile
        auto contract = [&](int kb, const u32x4 (&xh)[TV], const u32x4 (&xl)[TV]) {
            const uint4 *wk = s_w + (size_t)kb * NT * 128 + lane;
#pragma unroll
            for (int T = 0; T < NT; ++T) {
                const f16x8 ah = __builtin_bit_cast(f16x8, wk[T * 128]), al = __builtin_bit_cast(f16x8, wk[T * 128 + 64]);
#pragma unroll
                for (int t = 0; t < TV; ++t) {
                    const f16x8 vh = __builtin_bit_cast(f16x8, xh[t]), vl = __builtin_bit_cast(f16x8, xl[t]);
                    acc[T][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, vh, acc[T][t], 0, 0, 0);
                    acc[T][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, vh, acc[T][t], 0, 0, 0);
                    acc[T][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, vl, acc[T][t], 0, 0, 0);
                }
            }
        };
        for (int kb0 = 0; kb0 < p.nkb; kb0 += p.kchunk) {
            const int nk = p.nkb - kb0 < p.kchunk ? p.nkb - kb0 : p.kchunk;
            if (!resident) stage(kb0, nk);
            if ((!FAST && !wave_work) || (p.dbg & 2)) continue;
            // Software pipeline over the K blocks: while the matrix pipe contracts block kb with the weights, the
            // vector unit forms phi of block kb + 1 (two d2 instructions per vertex tile, 8 logarithms, 8 multiplies
            // and the fp16 split per lane).  Inside one wave the two would otherwise run back to back -- the
            // logarithms with the matrix pipe idle, then 72 matrix instructions with the vector unit idle -- and two
            // waves per SIMD running the same program do not interleave well enough to hide either.
            auto interleave = [&]() {
                // issue order inside the block just written: one matrix instruction, then the vector work that fits
                // under it (thin-plate only; the Gaussian block is left to the scheduler's own order)
                if constexpr (!GAUSS) {
#pragma unroll
                    for (int q = 0; q < 32; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);      // transcendental
                        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);      // VALU
                    }
#pragma unroll
                    for (int q = 32; q < 8 + NT * TV * 3; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                    }
                }
            };
#ifdef FD_KLOOP_UNROLL2
            // Two operand buffers in turn: the loop is unrolled by two so that neither is ever copied; a full scheduling
            // barrier between the halves keeps the second half's operand loads out of the first (unfenced: 109 spills)
            u32x4 bhA[TV], blA[TV], bhB[TV], blB[TV];
            phi_block(0, bhA, blA);
            int kb = 0;
            for (; kb + 2 < nk; kb += 2) {
                phi_block(kb + 1, bhB, blB);
                contract(kb, bhA, blA);
                interleave();
                __builtin_amdgcn_sched_barrier(0);
                phi_block(kb + 2, bhA, blA);
                contract(kb + 1, bhB, blB);
                interleave();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kb + 1 < nk) {
                phi_block(kb + 1, bhB, blB);
                contract(kb, bhA, blA);
                interleave();
                __builtin_amdgcn_sched_barrier(0);
                contract(kb + 1, bhB, blB);
            } else {
                contract(kb, bhA, blA);
            }
#else
            u32x4 bh[TV], bl[TV];
            phi_block(0, bh, bl);
            for (int kb = 0; kb + 1 < nk; ++kb) {
                u32x4 nbh[TV], nbl[TV];
                phi_block(kb + 1, nbh, nbl);
                contract(kb, bh, bl);
                interleave();
                // (32 register copies per block; FD_KLOOP_UNROLL2 is the variant without them)
#pragma unroll
                for (int t = 0; t < TV; ++t) { bh[t] = nbh[t]; bl[t] = nbl[t]; }
            }
            contract(nk - 1, bh, bl);
#endif
        }

        FD_SSTAMP(1)
        // ---- epilogue.  The accumulators hold, in lane group g, frame 4 T + g for the four vertex
        // tiles; a 4 x 4 transpose across the lane groups (two v_permlane32_swap + two
        // v_permlane16_swap per four registers) turns that into vertex tile g for the four frames
        // of the tile: every lane then owns ONE vertex (vbase + lane), does the per-vertex work
        // (gate, fall-off, tangent axes) once, and a frame's 64 positions leave as one contiguous
        // 768-byte store.  All transposes first, in place (acc[T][k][c] becomes row 4 k + c of tile T for
        // this lane's vertex: padded layout frame 4 T + k, component c; dense layout component T % 3 of
        // frame 16 (T / 3) + 4 k + c), while the wave is still converged.
#pragma unroll
        for (int T = 0; T < NT; ++T) {
#pragma unroll
            for (int c = 0; c < (DENSE ? 4 : 3); ++c) {
                // X_k[g] = (rows 4 g .., vertex tile k)  ->  Y_k[g] = (rows 4 k .., vertex tile g)
                const u32x2 s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[T][0][c]), __float_as_uint(acc[T][2][c]), false, false);
                const u32x2 s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[T][1][c]), __float_as_uint(acc[T][3][c]), false, false);
                const u32x2 y01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
                const u32x2 y23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
                acc[T][0][c] = __uint_as_float(y01[0]); acc[T][1][c] = __uint_as_float(y01[1]);
                acc[T][2][c] = __uint_as_float(y23[0]); acc[T][3][c] = __uint_as_float(y23[1]);
            }
        }
        // the reference's order: gate -> tangent projection -> fall-off -> add (src/SOP_FaceDeform.cpp:405-438)
        const int64_t i = vbase + lane;
        const bool inb = i < p.N;
        const int64_t ic = inb ? i : p.N - 1;
        const bool gated = own_d2 > p.radius2;
        const unsigned off12 = 12u * (unsigned)lane, off4 = 4u * (unsigned)lane;   // byte offsets inside the wave's 64-vertex window
        if constexpr (!FAST) {
            const int gn = grp + (int)gridDim.x;
            nxt = load_raw(gn < ngroups ? gn : grp, fastTag);
        }
        FD_SSTAMP(2)
        if constexpr (FAST) {
            // gate open, fall-off 1 (pow(1 - 0, rate): no dist2 attribute), no tangent frames: P + d * 1.  The sum the
            // matrix pipe holds is 2^k d: one fma with the exact 2^-k gives the same bits as (d * 1) + P.
            // 32 position stores + 16 fall-off stores (two frames each) = 48 operations per group: with the
            // next group's loads they fit the 63 a wave may have in flight, so the epilogue never waits for an
            // acknowledgement.  fd_falloff of frames (fs, fs + 1): lanes 0..31 write vertices 2 l, 2 l + 1 of frame
            // fs, lanes 32..63 the same of frame fs + 1 (8 bytes each: two 256-byte rows per instruction).
            const f32x2 ones = {1.f, 1.f};
            const unsigned off8 = 8u * (unsigned)(lane & 31);
#pragma unroll
            for (int fs = 0; fs < kSlots; ++fs) {
                const float inv = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)tab[(8 * fs) / 64], (8 * fs) % 64));
                const uint64_t pout = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)tab[(8 * fs + 5) / 64], (8 * fs + 5) % 64) << 32) |
                                      (unsigned)__builtin_amdgcn_readlane((int)tab[(8 * fs + 4) / 64], (8 * fs + 4) % 64);
                float d0, d1, d2c;
                if constexpr (DENSE) {
                    const int B = fs / 16, k = (fs % 16) / 4, r = fs % 4;
                    d0 = acc[3 * B][k][r]; d1 = acc[3 * B + 1][k][r]; d2c = acc[3 * B + 2][k][r];
                } else {
                    const int T = fs / 4, k = fs % 4;
                    d0 = acc[T][k][0]; d1 = acc[T][k][1]; d2c = acc[T][k][2];
                }
                Pos3 FD_GLOBAL *dstP = (Pos3 FD_GLOBAL *)((char FD_GLOBAL *)(pout + 12ull * (uint64_t)vbase) + off12);
                if (fs % 2 == 0) {
                    const uint64_t fa = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)tab[(8 * fs + 7) / 64], (8 * fs + 7) % 64) << 32) |
                                        (unsigned)__builtin_amdgcn_readlane((int)tab[(8 * fs + 6) / 64], (8 * fs + 6) % 64);
                    const int fs1 = fs + 1 < kSlots ? fs + 1 : fs;
                    const uint64_t fb = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)tab[(8 * fs1 + 7) / 64], (8 * fs1 + 7) % 64) << 32) |
                                        (unsigned)__builtin_amdgcn_readlane((int)tab[(8 * fs1 + 6) / 64], (8 * fs1 + 6) % 64);
                    const uint64_t fo = (lane < 32 ? fa : fb) + 4ull * (uint64_t)vbase;
                    __builtin_nontemporal_store(ones, (f32x2_a8 FD_GLOBAL *)((char FD_GLOBAL *)fo + off8));
                }
                // (non-temporal, like everything this launch writes: see the 32-row kernel)
                store_pos3_nt(dstP, __builtin_fmaf(d0, inv, pos[0]), __builtin_fmaf(d1, inv, pos[1]), __builtin_fmaf(d2c, inv, pos[2]));
            }
            settle(nxt);
            FD_SSTAMP(3)
            return;
        }
        if (inb && gated) {
            // B2: a gated vertex keeps its position (and no fd_falloff entry is written); as a displacement: zero
            for (int f = 0; f < p.nF; ++f) {
                float *dstp = s_frames[f].P_out;
                if (p.delta) store_pos3((Pos3 FD_GLOBAL *)as_global(dstp) + i, 0.f, 0.f, 0.f);
                else if (dstp != p.P_in) store_pos3((Pos3 FD_GLOBAL *)as_global(dstp) + i, pos[0], pos[1], pos[2]);
            }
        }
        float fall = 1.f;
        float a1[3] = {0.f, 0.f, 0.f}, a2[3] = {0.f, 0.f, 0.f};
        const bool doit = inb && !gated;
        const float base[3] = {p.delta ? 0.f : pos[0], p.delta ? 0.f : pos[1], p.delta ? 0.f : pos[2]};     // (0 + d f = d f exactly)
        if (doit) {
            if (p.dist2 != nullptr || !(p.radius2 != 0.f)) {
                const float q = fminf(own_d2 / p.radius2, 1.f);
                fall = powf(1.f - q, p.falloffrate);
            }
            if (p.tu) {
                // project_to_tangents (src/SOP_FaceDeform.hpp:28-41): the two axes depend on the vertex only
                float u[3] = {p.tu[3 * ic], p.tu[3 * ic + 1], p.tu[3 * ic + 2]};
                float v[3] = {p.tv[3 * ic], p.tv[3 * ic + 1], p.tv[3 * ic + 2]};
                float n[3] = {p.nrm[3 * ic], p.nrm[3 * ic + 1], p.nrm[3 * ic + 2]};
                normalize3(u[0], u[1], u[2]);
                normalize3(v[0], v[1], v[2]);
                normalize3(n[0], n[1], n[2]);
                float gm[3][3];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) gm[r][c] = u[r] * u[c] + v[r] * v[c] + n[r] * n[c];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    a1[c] = u[0] * gm[0][c] + u[1] * gm[1][c] + u[2] * gm[2][c];
                    a2[c] = v[0] * gm[0][c] + v[1] * gm[1][c] + v[2] * gm[2][c];
                }
                normalize3(a1[0], a1[1], a1[2]);
                normalize3(a2[0], a2[1], a2[2]);
            }
        }
#pragma unroll
        for (int fs = 0; fs < kSlots; ++fs) {
            {
                const int f = fs;                    // wave-uniform, compile-time after unrolling
                if (f >= p.nF) continue;
                // word w of frame f sits in lane (8 f + w) % 64 of tab[(8 f + w) / 64]
                const float inv = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)tab[(8 * f) / 64], (8 * f) % 64));
                const bool built = __builtin_amdgcn_readlane((int)tab[(8 * f + 1) / 64], (8 * f + 1) % 64) != 0;
                const uint64_t pout = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)tab[(8 * f + 5) / 64], (8 * f + 5) % 64) << 32) |
                                      (unsigned)__builtin_amdgcn_readlane((int)tab[(8 * f + 4) / 64], (8 * f + 4) % 64);
                const uint64_t fout = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)tab[(8 * f + 7) / 64], (8 * f + 7) % 64) << 32) |
                                      (unsigned)__builtin_amdgcn_readlane((int)tab[(8 * f + 6) / 64], (8 * f + 6) % 64);
                if (!doit) continue;
                // scalar base (the wave's window of the frame's arrays) + a 32-bit lane offset
                Pos3 FD_GLOBAL *dstP = (Pos3 FD_GLOBAL *)((char FD_GLOBAL *)(pout + 12ull * (uint64_t)vbase) + off12);
                if (!built) {
                    if (p.delta) store_pos3(dstP, 0.f, 0.f, 0.f);
                else if (pout != (uint64_t)p.P_in) store_pos3(dstP, pos[0], pos[1], pos[2]);
                    continue;
                }
                // 2^-k is exact: disp is the sum the matrix pipe accumulated, polynomial included
                float disp[3];
                if constexpr (DENSE) {
                    const int B = fs / 16, k = (fs % 16) / 4, r = fs % 4;
                    disp[0] = acc[3 * B][k][r] * inv; disp[1] = acc[3 * B + 1][k][r] * inv; disp[2] = acc[3 * B + 2][k][r] * inv;
                } else {
                    const int T = fs / 4, k = fs % 4;
                    disp[0] = acc[T][k][0] * inv; disp[1] = acc[T][k][1] * inv; disp[2] = acc[T][k][2] * inv;
                }
                if (p.dbg & 1) continue;         // diagnostics: everything but the stores
                if (p.tu) {
                    const float da1 = disp[0] * a1[0] + disp[1] * a1[1] + disp[2] * a1[2];
                    const float da2 = disp[0] * a2[0] + disp[1] * a2[1] + disp[2] * a2[2];
#pragma unroll
                    for (int c = 0; c < 3; ++c) disp[c] = a1[c] * da1 + a2[c] * da2;
                }
                if (fout) __builtin_nontemporal_store(fall, (float FD_GLOBAL *)((char FD_GLOBAL *)(fout + 4ull * (uint64_t)vbase) + off4));
                store_pos3_nt(dstP, base[0] + disp[0] * fall, base[1] + disp[1] * fall, base[2] + disp[2] * fall);
            }
        }
        FD_SSTAMP(3)
    };
    int grp = blockIdx.x;
    // a frame whose build failed passes the mesh through: general path (the status words sit in the frame table)
    bool built_here = true;
#pragma unroll
    for (int q = 0; q < kTabRegs; ++q) {
        const int idx = 64 * q + lane;
        if (idx < kSlots * 8 && (idx & 7) == 1) built_here = built_here && tab[q] != 0u;
    }
    if (p.fast && __all(built_here)) {
        const int nfull = (int)(p.N / kSharedThreads);       // groups in which every wave's 64 vertices exist
        for (; grp < nfull; grp += gridDim.x) do_group(grp, std::true_type{});
    }
    for (; grp < ngroups; grp += gridDim.x) do_group(grp, std::false_type{});
    if (stamp && lane == 0) {
        for (int q = 0; q < 4; ++q) p.stamps[wave * 8 + q] = st_acc[q];
    }
#undef FD_SSTAMP
}


// ---- 17 to 32 frames, thin-plate: 32-row tiles ----------------------------------------------------------------------
// The same contraction on v_mfma_f32_32x32x16_f16: an output tile is 32 frames x 32 vertices, one tile per component,
// and a wave's 64 vertices are two vertex tiles.  The matrix pipe does the same flops in half the instructions, and what
// bounds the K loop of the 16-row kernel above is the issue port, not the pipe (tools/ubench_mfma16.hip: a matrix
// instruction costs the port ~10 cycles whatever its shape: 80 of them per K block and wave there, 40 here, beside the
// same ~150 vector and 32 transcendental instructions).  d2 comes from two v_mfma_f32_32x32x8_f16 per vertex tile
// (K = 16 as x, y | z, |.|^2) and lands as 16 values per lane -- centres 8 (r / 4) + 4 h + r % 4 of the block in register
// r of lane half h -- which after the logarithm ARE the lane's B operands of the two K = 16 steps (registers 0..7 and
// 8..15): the weight tiles are packed to that centre order.  The epilogue needs one v_permlane32_swap per register pair.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr bool shared_wide(int nF, int kind) { (void)kind; return nF > 16; }      // thin-plate and the Gaussian kinds alike
constexpr int kWideSlots = 32;                      // frame records
constexpr int kWideDefaultVar = 49;      // skewed K loop, units from the counter, non-temporal stores and loads: the fastest inside bench.py
// Rows of the output tiles are packed densely: row 3 f + c of the stack of 32-row tiles is component c of frame slot f, so
// 17..20 frames are TWO tiles (60 rows of 64) where one tile per component would be three (r2: 96 rows whatever the frame
// count) -- a third fewer matrix instructions for the driver's 20-frame launch.  After the epilogue's lane-half swap a lane
// holds every row of every tile for its own vertex, so any row <-> (frame, component) map costs nothing there.
// Frame slots come in fours (the fall-off rows leave four frames per store): a launch of nF frames runs
// NSLOT = 4 ceil(nF / 4) slots, and slots nF .. NSLOT - 1 are DUPLICATES of frame nF - 1 (same weights, scale and output
// pointers: the same bits stored twice to the same address) -- so every frame count takes the straight-line epilogue.
constexpr int wide_slots(int nF) { return (nF + 3) / 4 * 4; }
constexpr int wide_tiles(int nslot) { return (3 * nslot + 31) / 32; }               // 2 up to 20 slots (21 frames would fit, 24 slots do not), else 3
constexpr int wide_w16(int nt) { return nt * 2 * 2 * 64; }                          // 16-byte words of weight tiles per K block: [tile][K step][hi, lo][lane]

// grid (nkb, NT row tiles), 256 threads.  Output regions: weight tiles [kb][tile][K step][hi, lo][lane], NT x 64 words of
// polynomial tiles, then the d2 operands of the centres ([kb][2 instructions][64 lanes] x 8 B) and the normalisation.
// nslot frame slots (a multiple of 4); slots.rec32 / model / out of the slots beyond nF point at frame nF - 1 (the host sets them).
// layout 1 (k_deform32_shared_w1, one vertex tile per wave): the d2 operand of a K block is ONE K = 16 instruction -- lane (h, r)
// holds coordinate groups 2 h, 2 h + 1 of centre r, 16 bytes -- and the rows are dealt out per lane half: row rho of tile T is
// register r = 4 (rho / 8) + rho % 4 of half hh = (rho / 4) % 2, flat index k = 16 T + r of that half = component k % 3 of the
// half's local frame k / 3, which is frame slot 2 (k / 3) + hh.
__global__ __launch_bounds__(256) void k_pack_shared_wide(const SharedSlots slots, const SharedOut out, int nF, int nslot, int Mpad,
                                                           uint4 *wtiles, SharedFrame *frames, const MfmaTileH *ctiles, int gauss, int layout)
{
    const int kb = blockIdx.x, T = blockIdx.y, nkb = gridDim.x, NT = gridDim.y;
    const int w16 = wide_w16(NT);
    const size_t poly_at = (size_t)nkb * w16, copy_at = poly_at + (size_t)NT * 64;
    if (T == 0) {
        // lane (h, r) of instruction i: k-slots 4 h .. 4 h + 3 of centre r = coordinate group 2 i + h of the 16-row tile
        uint2 *dst = reinterpret_cast<uint2 *>(wtiles + copy_at) + (size_t)kb * 128;
        if (gauss) {
            // Gaussian kinds: the K block's slot holds its 32 centre records instead ({c'x, c'y, c'z, -log2(e) s^2 / R_j^2}:
            // the first half of Rec32), read by direct differences
            if (threadIdx.x < 32) {
                const int centre = 32 * kb + (int)threadIdx.x;
                wtiles[copy_at + (size_t)kb * 64 + threadIdx.x] =
                    centre < Mpad ? *reinterpret_cast<const uint4 *>(&slots.rec32[0][centre]) : make_uint4(0u, 0u, 0u, 0u);
            }
        } else if (layout == 1) {
            if (threadIdx.x < 64) {
                const int lane = threadIdx.x, h = lane >> 5, r = lane & 31;
                const int tile = 2 * kb + (r >> 4);
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (tile < Mpad / 16) {
                    const unsigned *a0 = ctiles[tile].a[16 * (2 * h) + (r & 15)], *a1 = ctiles[tile].a[16 * (2 * h + 1) + (r & 15)];
                    v = make_uint4(a0[0], a0[1], a1[0], a1[1]);
                }
                wtiles[copy_at + (size_t)kb * 64 + lane] = v;
            }
        } else if (threadIdx.x < 128) {
            const int i = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, r = lane & 31;
            const int tile = 2 * kb + (r >> 4);
            uint2 v = make_uint2(0u, 0u);
            if (tile < Mpad / 16) { const unsigned *a = ctiles[tile].a[16 * (2 * i + h) + (r & 15)]; v = make_uint2(a[0], a[1]); }
            dst[threadIdx.x] = v;
        }
        if (kb == 0 && threadIdx.x == 255) {
            const float *nn = slots.model[0]->norm32;
            wtiles[copy_at + (size_t)nkb * 64] = make_uint4(__float_as_uint(nn[0]), __float_as_uint(nn[1]), __float_as_uint(nn[2]), __float_as_uint(nn[3]));
        }
    }
    // scale of each frame slot: largest |weight| or |polynomial coefficient| to [2^13, 2^14); 8 lanes per slot
    __shared__ float s_scale[kWideSlots];
    {
        const int f = threadIdx.x >> 3, l = threadIdx.x & 7;
        float m = 0.f;
        if (f < nslot) {
            // (the packing code of every build leaves the largest |weight| / |coefficient| of the fp32 records in the model:
            // scanning the records here -- every workgroup all frames -- was most of this kernel's 30 us)
            m = slots.model[f]->wmax32;
        }
        for (int off = 4; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        const bool same_rig = (kb == 0 && T == 0) ? rig_matches(slots, f, nslot, l, 8) : true;
        if (l == 0) {
            int k = 0;
            if (m > 0.f && m < INFINITY) k = 13 - (__builtin_amdgcn_frexp_expf(m) - 1);
            k = k < -100 ? -100 : (k > 100 ? 100 : k);
            s_scale[f] = ldexpf(1.f, k);
            if (kb == 0 && T == 0) {
                SharedFrame fr;
                fr.inv_scale = ldexpf(1.f, -k - (gauss ? kGaussShift : 0));
                fr.built = (f < nslot && same_rig && slots.model[f]->terminationtype == 1) ? 1 : 0;
                fr.pad[0] = fr.pad[1] = 0;
                fr.P_out = f < nslot ? out.P_out[f] : nullptr;
                fr.falloff_out = f < nslot ? out.falloff_out[f] : nullptr;
                frames[f] = fr;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x >= 128) return;
    const int s = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5;
    int f, c;
    if (layout == 1) {
        const int rho = lane & 31, k = 16 * T + 4 * (rho >> 3) + (rho & 3);
        f = 2 * (k / 3) + ((rho >> 2) & 1); c = k % 3;
    } else {
        const int row = 32 * T + (lane & 31);                                      // row 3 f + c = component c of frame slot f
        f = row / 3; c = row % 3;
    }
    const bool live = f < nslot;
    const float sc = live ? s_scale[f] : 0.f;
    f16x8 hi, lo;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int centre = 32 * kb + 8 * (2 * s + (m >> 2)) + 4 * h + (m & 3);
        float w = 0.f;
        if (live && centre < Mpad) {
            const Rec32 r = slots.rec32[f][centre];
            w = (c == 0 ? r.wx : (c == 1 ? r.wy : r.wz)) * sc;
        }
        const _Float16 hh = (_Float16)w;
        hi[m] = hh;
        lo[m] = (_Float16)(w - (float)hh);
    }
    uint4 *dst = wtiles + (size_t)kb * w16 + (size_t)((T * 2 + s) * 2) * 64;
    dst[lane] = __builtin_bit_cast(uint4, hi);
    dst[64 + lane] = __builtin_bit_cast(uint4, lo);
    if (kb == 0 && s == 0) {
        // polynomial tile of row tile T, K = 16: coefficients {C0, Lx, Ly, Lz, q} as (hi, lo) against the vertex
        // operand's {1, x, y, z, |x|^2} as (hi, lo) -- lane half 0: hi[0..4] x hi, then hi[1..3] x lo(x, y, z);
        // lane half 1: hi[4] x lo(|x|^2), lo[0..4] x hi, two unused
        f16x8 pt;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int coef = h == 0 ? (m < 5 ? m : m - 4) : (m == 0 ? 4 : (m < 6 ? m - 1 : -1));
            const bool want_lo = h == 1 && m >= 1;
            float w = 0.f;
            if (live && coef >= 0) w = slots.model[f]->poly32[5 * c + coef] * sc;
            const _Float16 hh = (_Float16)w;
            pt[m] = want_lo ? (_Float16)(w - (float)hh) : hh;
        }
        wtiles[poly_at + (size_t)T * 64 + lane] = __builtin_bit_cast(uint4, pt);
    }
}

#ifndef FD_WIDE_VPM
#define FD_WIDE_VPM 3            // vector instructions placed after each matrix instruction of the K loop
#endif
#ifndef FD_W1_VPM
#define FD_W1_VPM 3              // the same for the one-tile kernel (k_deform32_shared_w1)
#endif

// VAR (build variants kept for A/B runs inside one process: FD_SHARED_WIDE_VAR, tests/tools/wide_variants_timing.py):
//   bit 0  the K loop skewed by half a block, operands written straight into dead registers (clear: phi of block k + 1 under
//          the whole contraction of block k, 32 register copies per block)
//   bit 1  fixed shares of the units per wave (clear: the counter in LDS)
//   bit 2  LDS reads of the weights one (component, K step) pair ahead of their use (skewed loop only)
//   bit 3  the last, partial round dealt out as whole groups (clear: as single units)
//   bit 4  the straight-line epilogue's stores with the non-temporal hint
//   bit 5  the positions read with the non-temporal hint
// Instantiated: 49 only (r2 kept eight for A/B runs; the others lost and are gone from the library).
// NT row tiles of 32 (2 or 3), NSLOT frame slots (20: NT = 2; 24, 28, 32: NT = 3) -- see wide_slots / wide_tiles.
template <int VAR, bool GAUSS, int NT, int NSLOT>
__global__ __launch_bounds__(64 * kWideWaves) __attribute__((amdgpu_waves_per_eu(kWideWaves / 4, kWideWaves / 4)))
void k_deform32_tps_shared_wide(const SharedParams p, int ngroups)
{
    // eight waves per workgroup (two per SIMD), 64 vertices each per group.  (Three per SIMD -- 12 waves, 168 registers, 87 spilled --
    // was measured slower in round 3: 189 -> 206 us; round 4's one-tile kernel below is the three-wave form that pays.)
    constexpr int WAVES = kWideWaves, THREADS = 64 * WAVES;
    static_assert(NSLOT % 4 == 0 && NSLOT <= kWideSlots && 3 * NSLOT <= 32 * NT, "frame slots in fours, three rows each");
    constexpr int kWideW16 = wide_w16(NT);
    constexpr bool SKEWED = (VAR & 1) != 0;
    static_assert(!GAUSS || SKEWED, "the Gaussian kinds take the skewed loop only");
    constexpr bool AHEAD = (VAR & 4) != 0;
    constexpr bool NONTEMPORAL = (VAR & 16) != 0;
    constexpr int TV = 2;                        // vertex tiles (of 32) per wave
    constexpr int kSlots = kWideSlots;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: [frame records 32][polynomial tiles NT x 64 x 16 B][d2 operands kchunk x 2 x 64 x 8 B][weight tiles kchunk x NT x 4 KiB]
    SharedFrame *s_frames = reinterpret_cast<SharedFrame *>(smem);
    uint4 *s_poly = reinterpret_cast<uint4 *>(smem + sizeof(SharedFrame) * (size_t)kSlots);
    uint2 *s_ct = reinterpret_cast<uint2 *>(s_poly + NT * 64);
    uint4 *s_w = reinterpret_cast<uint4 *>(s_ct + (size_t)128 * p.kchunk);
    // fd_falloff pointers of the straight-line epilogue: store q of a group covers frames 4 q .. 4 q + 3, 16 lanes x 16 B each
    uint64_t *s_ftab = reinterpret_cast<uint64_t *>(s_w + (size_t)kWideW16 * p.kchunk);
    unsigned *s_ticket = reinterpret_cast<unsigned *>(s_ftab + 512);      // next 64-vertex unit of this workgroup
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const float n0 = p.norm[0], n1 = p.norm[1], n2 = p.norm[2];
    const float inv_s = p.norm[3];
    const bool resident = p.nkb <= p.kchunk;
    const unsigned long long st_t0 = p.stamps ? __builtin_amdgcn_s_memtime() : 0, st_r0 = p.stamps ? __builtin_amdgcn_s_memrealtime() : 0;     // from the kernel's entry
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;

    auto stage = [&](int kb0, int nk) {
        __syncthreads();
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(p.ctiles) + (size_t)kb0 * 64;
            uint4 *dst = reinterpret_cast<uint4 *>(s_ct);
            for (int q = tid; q < nk * 64; q += THREADS) dst[q] = src[q];
        }
        {
            // eight loads in flight per thread (one at a time the copy of a resident model took 14 round trips to L2,
            // about a twentieth of the launch, with nothing else running)
            // (native vectors: an array of HIP's uint4 structs went through scratch memory)
            const u32x4 *src = reinterpret_cast<const u32x4 *>(p.wtiles + (size_t)kb0 * kWideW16);
            u32x4 *dst = reinterpret_cast<u32x4 *>(s_w);
            const int n16 = nk * kWideW16;
            int q = tid;
            for (; q + 7 * THREADS < n16; q += 8 * THREADS) {
                const u32x4 v0 = src[q], v1 = src[q + THREADS], v2 = src[q + 2 * THREADS], v3 = src[q + 3 * THREADS];
                const u32x4 v4 = src[q + 4 * THREADS], v5 = src[q + 5 * THREADS], v6 = src[q + 6 * THREADS], v7 = src[q + 7 * THREADS];
                dst[q] = v0; dst[q + THREADS] = v1; dst[q + 2 * THREADS] = v2; dst[q + 3 * THREADS] = v3;
                dst[q + 4 * THREADS] = v4; dst[q + 5 * THREADS] = v5; dst[q + 6 * THREADS] = v6; dst[q + 7 * THREADS] = v7;
            }
            for (; q < n16; q += THREADS) dst[q] = src[q];
        }
        __syncthreads();
    };

    // frame records across the lanes of four registers (see k_deform32_tps_shared)
    constexpr int kTabRegs = (kSlots * 8 + 63) / 64;
    unsigned tab[kTabRegs];
#pragma unroll
    for (int q = 0; q < kTabRegs; ++q) tab[q] = reinterpret_cast<const unsigned *>(p.frames)[64 * q + lane];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.frames);
        uint4 *dst = reinterpret_cast<uint4 *>(s_frames);
        for (int q = tid; q < kSlots * (int)(sizeof(SharedFrame) / 16); q += THREADS) dst[q] = src[q];
        const uint4 *psrc = p.wtiles + (size_t)p.nkb * kWideW16;
        for (int q = tid; q < NT * 64; q += THREADS) s_poly[q] = psrc[q];
        if (tid == 0) *s_ticket = 0u;
        __syncthreads();
        if (p.fast && tid < NSLOT * 16) s_ftab[tid] = (uint64_t)p.frames[4 * (tid >> 6) + ((tid & 63) >> 4)].falloff_out + 16u * (unsigned)(tid & 15);
    }
    if (resident) {
        stage(0, p.nkb);
        if (wave >= 4) {                         // the second wave of each SIMD starts out of phase (as above)
            __builtin_amdgcn_s_sleep(12);
            for (int q = 0; q < (p.stagger & 0xff); ++q) __builtin_amdgcn_s_sleep(127);
        }
        if ((p.dbg & 4) && wave >= 4) return;      // diagnostics: one wave per SIMD (no barrier follows while the model is resident)
    } else {
        __syncthreads();
    }

    const bool stamp = p.stamps != nullptr && blockIdx.x == 0;
    unsigned long long st_prev = 0, st_acc[5] = {0, 0, 0, 0, 0};
#define FD_SSTAMP(K) if (stamp) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[K] += t_ - st_prev; st_prev = t_; __builtin_amdgcn_sched_barrier(0); }
    if (stamp) st_prev = __builtin_amdgcn_s_memtime();
    // lane (h, j) holds the two vertices (vt, j) -- both lane halves the same two
    struct GroupRaw { float p[TV][3]; float d2[TV]; };
    // Units: the workgroup's groups (blockIdx + k gridDim) are eight 64-vertex units each; unit u of the workgroup is unit
    // u & 7 of its group u >> 3.  While the model is resident (no barrier in the loop) a wave takes its NEXT unit from a
    // counter in LDS instead of always the one with its own number: the two waves of a SIMD do not run at the same pace
    // (the second starts later and yields more often), and with fixed shares the first four waves of a workgroup
    // finished a group's time before the others, who then ran their last group with the matrix pipe half empty.
    constexpr bool DYNAMIC = (VAR & 2) == 0;
    const bool dynamic = DYNAMIC && resident;
    // The groups that do not fill a last round (ngroups % gridDim of them) are dealt out as single units, workgroup by
    // workgroup (VAR bit 3 clear): every workgroup then ends with one or two lone units at a lone wave's pace (0.6 of a
    // round) instead of a few workgroups running a whole extra round beside idle CUs (at 192 CUs: 34 of them).
    constexpr bool POOL = (VAR & 8) == 0;
    // (A device-wide draw of the groups -- a workgroup fetching its next group from a global counter three rounds ahead, published to
    // its waves through a ring in LDS -- was built in round 3 and measured slower, 245 against 170 us per launch: the ring's LDS
    // word sits between the epilogue's stores and any small wait there waits for the stores in flight.  Removed in round 4;
    // DESIGN.md 4.1c keeps the account.)
    const int whole_rounds = (dynamic && POOL) ? ngroups / (int)gridDim.x : (ngroups + (int)gridDim.x - 1) / (int)gridDim.x;
    const int64_t pool0 = (int64_t)whole_rounds * (int64_t)gridDim.x * WAVES, total_units = (int64_t)ngroups * WAVES;
    auto unit_global = [&](int u) -> int64_t {          // the 64-vertex unit behind local ticket u (>= total_units: none)
        if (u < WAVES * whole_rounds) return ((int64_t)blockIdx.x + (int64_t)(u / WAVES) * (int64_t)gridDim.x) * WAVES + (u % WAVES);
        return pool0 + (int64_t)blockIdx.x + (int64_t)(u - WAVES * whole_rounds) * (int64_t)gridDim.x;
    };
    auto unit_group = [&](int u) -> int64_t { const int64_t g = unit_global(u); return g < total_units ? (g / WAVES) : (int64_t)ngroups; };
    int round_taken = 0;                                 // workgroup-local round of the ticket taken last
    auto next_unit = [&](int u) {
        if (!dynamic) return u + WAVES;
        unsigned v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(s_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int t = (int)__builtin_amdgcn_readfirstlane(v);
        round_taken = t / WAVES;
        return t;
    };
    auto load_raw = [&](int64_t gu, auto fastTag) {
        constexpr bool FAST = decltype(fastTag)::value;
        GroupRaw r;
        const int64_t vb = gu * 64;
#pragma unroll
        for (int t = 0; t < TV; ++t) {
            const int64_t vi = vb + 32 * t + j;
            const int64_t vc = vi < p.N ? vi : p.N - 1;
            if constexpr ((VAR & 32) != 0) {        // read once per launch: streamed past L2 like the outputs
                r.p[t][0] = __builtin_nontemporal_load(&p.P_in[3 * vc]); r.p[t][1] = __builtin_nontemporal_load(&p.P_in[3 * vc + 1]);
                r.p[t][2] = __builtin_nontemporal_load(&p.P_in[3 * vc + 2]);
            } else {
                r.p[t][0] = p.P_in[3 * vc]; r.p[t][1] = p.P_in[3 * vc + 1]; r.p[t][2] = p.P_in[3 * vc + 2];
            }
            if constexpr (FAST) r.d2[t] = 0.f; else r.d2[t] = p.dist2 ? p.dist2[vc] : 0.f;
        }
        return r;
    };
    auto settle = [&](const GroupRaw &r) {
        asm volatile("" :: "v"(r.p[0][0]), "v"(r.p[0][1]), "v"(r.p[0][2]), "v"(r.p[1][0]), "v"(r.p[1][1]), "v"(r.p[1][2]));
    };
    GroupRaw nxt;
    auto do_group = [&](int u, int un, int lround, auto fastTag) {
        constexpr bool FAST = decltype(fastTag)::value;
        const int64_t vbase = unit_global(u) * 64;
        const int64_t gu_next = unit_group(un) < ngroups ? unit_global(un) : unit_global(u);       // whose positions to request
        // (the workgroup's OWN round counter decides: the global group numbers a workgroup draws all have the parity of its
        // index -- 256 workgroups draw in step -- and one wave of each SIMD would keep the priority for the whole launch)
        if (((lround ^ (wave >> 2)) & 1) != 0) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
        const GroupRaw cur = nxt;
        // this lane's own vertex in the epilogue is (vt = h, j)
        const float pos[3] = {h ? cur.p[1][0] : cur.p[0][0], h ? cur.p[1][1] : cur.p[0][1], h ? cur.p[1][2] : cur.p[0][2]};
        const float own_d2 = h ? cur.d2[1] : cur.d2[0];
        f16x4 bop0[TV], bop1[TV];
        float xn[TV], yn[TV], zn[TV];            // GAUSS: the normalised coordinates, for the direct differences
        f32x16 acc[NT][TV];
        bool lane_live = false;
#pragma unroll
        for (int t = 0; t < TV; ++t) {
            const int64_t vi = vbase + 32 * t + j;
            const float x = (cur.p[t][0] - n0) * inv_s, y = (cur.p[t][1] - n1) * inv_s, z = (cur.p[t][2] - n2) * inv_s;
            xn[t] = x; yn[t] = y; zn[t] = z;
            const float d2v = cur.d2[t];
            if constexpr (FAST) lane_live = true; else lane_live |= (vi < p.N) && !(d2v > p.radius2);
            const float xx = __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x));
            // d2 operands: instruction 0 carries (x | y), instruction 1 (z | |x|^2) in the lane halves
            const float va = h == 0 ? -2.f * x : -2.f * y, vb2 = h == 0 ? -2.f * z : xx;
            const _Float16 ha = (_Float16)va, la = (_Float16)(va - (float)ha);
            const _Float16 hb = (_Float16)vb2, lb = (_Float16)(vb2 - (float)hb);
            const _Float16 one = (_Float16)1.0f;
            bop0[t] = (f16x4){ha, la, ha, la};
            bop1[t] = h == 0 ? (f16x4){hb, lb, hb, lb} : (f16x4){one, one, hb, lb};
            // polynomial operand, K = 16 (k_pack_shared_wide): half 0 {1, xh, yh, zh, xxh, xl, yl, zl}, half 1 {xxl, 1, xh, yh, zh, xxh, 0, 0}
            // (GAUSS: everything times 2^10, the factor phi carries; undone with the frame's scale)
            constexpr float ps = GAUSS ? (float)(1 << kGaussShift) : 1.f;
            constexpr unsigned one16 = GAUSS ? 0x6400u : 0x3c00u;         // fp16 1024 / 1
            unsigned xyh, xyl, zxh, zxl;
            split_pair_f16(x * ps, y * ps, xyh, xyl);
            split_pair_f16(z * ps, xx * ps, zxh, zxl);
            u32x4 pb;
            if (h == 0) pb = (u32x4){one16 | (xyh << 16), (xyh >> 16) | (zxh << 16), (zxh >> 16) | (xyl << 16), (xyl >> 16) | (zxl << 16)};
            else pb = (u32x4){(zxl >> 16) | (one16 << 16), xyh, zxh, 0u};
            const f16x8 pbv = __builtin_bit_cast(f16x8, pb);
#pragma unroll
            for (int c = 0; c < NT; ++c)
                acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, s_poly[c * 64 + lane]), pbv, zero16, 0, 0, 0);
        }
        const bool wave_work = FAST ? true : __any(lane_live);
        if constexpr (FAST) {
            nxt = load_raw(gu_next, fastTag);      // a whole K loop ahead of this group's stores (as above)
        }
        FD_SSTAMP(0)

        // phi of K block kb for the two vertex tiles: 16 values per lane and tile, as fp16 pieces (8 + 8 registers)
        auto phi_block = [&](int kb, u32x8 (&xh)[TV], u32x8 (&xl)[TV]) {
            const f16x4 aop0 = __builtin_bit_cast(f16x4, s_ct[(size_t)kb * 128 + lane]);
            const f16x4 aop1 = __builtin_bit_cast(f16x4, s_ct[(size_t)kb * 128 + 64 + lane]);
#pragma unroll
            for (int t = 0; t < TV; ++t) {
                f32x16 d = __builtin_amdgcn_mfma_f32_32x32x8f16(aop0, bop0[t], zero16, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x8f16(aop1, bop1[t], d, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    unsigned hh, ll;
                    split_pair_f16(d2_log_d2(d[2 * q]), d2_log_d2(d[2 * q + 1]), hh, ll);
                    xh[t][q] = hh; xl[t][q] = ll;
                }
            }
        };
        // acc += W(kb) x phi(kb): per component and K step the three split products of both vertex tiles
        auto contract = [&](int kb, const u32x8 (&xh)[TV], const u32x8 (&xl)[TV]) {
            const uint4 *wk = s_w + (size_t)kb * kWideW16 + lane;
#pragma unroll
            for (int c = 0; c < NT; ++c) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const f16x8 ah = __builtin_bit_cast(f16x8, wk[((c * 2 + s) * 2) * 64]), al = __builtin_bit_cast(f16x8, wk[((c * 2 + s) * 2 + 1) * 64]);
                    f16x8 vh[TV], vl[TV];
#pragma unroll
                    for (int t = 0; t < TV; ++t) {
                        vh[t] = __builtin_bit_cast(f16x8, (u32x4){xh[t][4 * s], xh[t][4 * s + 1], xh[t][4 * s + 2], xh[t][4 * s + 3]});
                        vl[t] = __builtin_bit_cast(f16x8, (u32x4){xl[t][4 * s], xl[t][4 * s + 1], xl[t][4 * s + 2], xl[t][4 * s + 3]});
                    }
#pragma unroll
                    for (int t = 0; t < TV; ++t) acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, vh[t], acc[c][t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < TV; ++t) acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, vh[t], acc[c][t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < TV; ++t) acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, vl[t], acc[c][t], 0, 0, 0);
                }
            }
        };
        for (int kb0 = 0; kb0 < p.nkb; kb0 += p.kchunk) {
            const int nk = p.nkb - kb0 < p.kchunk ? p.nkb - kb0 : p.kchunk;
            if (!resident) stage(kb0, nk);
            if ((!FAST && !wave_work) || (p.dbg & 2)) continue;
            // software pipeline as above: phi of block kb + 1 under the matrix instructions of block kb
            // issue order of one pipelined block: the four d2 instructions of block kb + 1 first, two of block kb's
            // contraction to cover their latency, then one logarithm and FD_WIDE_VPM vector instructions under each
            // matrix instruction; the weights of a (component, K step) pair are read from LDS one pair ahead
            auto interleave = [&]() {
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);          // d2 operands, weights of the first pair
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, FD_WIDE_VPM, 0);
                }
#pragma unroll
                for (int q = 0; q < 36; ++q) {
                    if (q % 6 == 0 && q < 30) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (q >= 2 && q < 34) __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, FD_WIDE_VPM, 0);
                }
            };
          if constexpr (SKEWED) {
            // Pipeline skewed by HALF a block, no operand copies: the d2 of block kb + 1 is issued in the middle of
            // block kb; its K step 0 operands are formed under block kb's K step 1 instructions (whose own step 0
            // operands are dead by then) and its K step 1 operands under the first half of block kb + 1.
            u32x4 b0h[TV], b0l[TV], b1h[TV], b1l[TV];
            f32x16 dd[TV];
            auto d2_block = [&](int kb) {
                if constexpr (GAUSS) return;
                const f16x4 aop0 = __builtin_bit_cast(f16x4, s_ct[(size_t)kb * 128 + lane]);
                const f16x4 aop1 = __builtin_bit_cast(f16x4, s_ct[(size_t)kb * 128 + 64 + lane]);
#pragma unroll
                for (int t = 0; t < TV; ++t) {
                    dd[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(aop0, bop0[t], zero16, 0, 0, 0);
                    dd[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(aop1, bop1[t], dd[t], 0, 0, 0);
                }
            };
            auto phi_half = [&](int kbsrc, int s, u32x4 (&xh)[TV], u32x4 (&xl)[TV]) {
                if constexpr (GAUSS) {
                    // exp(-d2 / R_j^2) from direct coordinate differences (the expanded d2 is not accurate enough under the
                    // 1 / R^2 the exponent multiplies it by: DESIGN.md 4.1), ONE value per instruction (packed arithmetic
                    // under in-flight matrix instructions: see k_deform32_tps_shared); this lane's eight centres of the
                    // half block -- 8 (2 s + q / 2) + 4 h + 2 (q % 2) + {0, 1} -- come from LDS once for both vertex tiles
                    const float4 *cr = reinterpret_cast<const float4 *>(s_ct + (size_t)kbsrc * 128) + 4 * h;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 c0 = cr[8 * (2 * s + (q >> 1)) + 2 * (q & 1)], c1 = cr[8 * (2 * s + (q >> 1)) + 2 * (q & 1) + 1];
#pragma unroll
                        for (int t = 0; t < TV; ++t) {
                            float ph[2];
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const float4 c = e ? c1 : c0;
                                const float dx = xn[t] - c.x, dy = yn[t] - c.y, dz = zn[t] - c.z;
                                float d2 = dx * dx;
                                d2 = __builtin_fmaf(dy, dy, d2);
                                d2 = __builtin_fmaf(dz, dz, d2);
                                ph[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(d2, c.w, (float)kGaussShift));
                            }
                            unsigned hh, ll;
                            split_pair_f16<true>(ph[0], ph[1], hh, ll);
                            xh[t][q] = hh; xl[t][q] = ll;
                        }
                    }
                    return;
                }
#pragma unroll
                for (int t = 0; t < TV; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        unsigned hh, ll;
                        split_pair_f16(d2_log_d2(dd[t][8 * s + 2 * q]), d2_log_d2(dd[t][8 * s + 2 * q + 1]), hh, ll);
                        xh[t][q] = hh; xl[t][q] = ll;
                    }
            };
            auto contract_half = [&](int kb, int s, const u32x4 (&xh)[TV], const u32x4 (&xl)[TV]) {
                const uint4 *wk = s_w + (size_t)kb * kWideW16 + lane;
#pragma unroll
                for (int c = 0; c < NT; ++c) {
                    const f16x8 ah = __builtin_bit_cast(f16x8, wk[((c * 2 + s) * 2) * 64]), al = __builtin_bit_cast(f16x8, wk[((c * 2 + s) * 2 + 1) * 64]);
#pragma unroll
                    for (int t = 0; t < TV; ++t) acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(f16x8, xh[t]), acc[c][t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < TV; ++t) acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, __builtin_bit_cast(f16x8, xh[t]), acc[c][t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < TV; ++t) acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(f16x8, xl[t]), acc[c][t], 0, 0, 0);
                }
            };
            d2_block(0);
            phi_half(0, 0, b0h, b0l);
            for (int kb = 0; kb + 1 < nk; ++kb) {
                phi_half(kb, 1, b1h, b1l);
                contract_half(kb, 0, b0h, b0l);
                d2_block(kb + 1);
                phi_half(kb + 1, 0, b0h, b0l);
                contract_half(kb, 1, b1h, b1l);
                if constexpr (GAUSS) continue;       // the Gaussian block is left to the scheduler's own order
                // issue order: 18 matrix instructions of K step 0 with the 16 logarithms of this block's step 1 operands,
                // the four d2 instructions, 18 of K step 1 with the next block's 16 (two instructions after the d2)
              if constexpr (AHEAD) {
                // LDS reads one (component, K step) pair AHEAD of their use: the pair after next is requested behind the
                // first matrix instruction of the current one (as written above, each pair's first instruction waited
                // out the LDS latency of its own operands: six exposed round trips per block)
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                for (int q = 0; q < 18; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (q % 6 == 0 && q < 12) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    if (q == 12) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);      // d2 operands, first pair of K step 1
                    if (q < 16) __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, FD_WIDE_VPM, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                }
#pragma unroll
                for (int q = 0; q < 18; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (q % 6 == 0 && q < 12) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    if (q >= 2) __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, FD_WIDE_VPM, 0);
                }
              } else {
                // NM matrix instructions per K step; the 16 logarithms of a half block go under them one at a time (three
                // row tiles: 18 instructions) or two at a time at first (two row tiles: 12)
                constexpr int NM = 6 * NT, VPM = NT == 3 ? FD_WIDE_VPM : FD_WIDE_VPM + 2;
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                for (int q = 0; q < NM; ++q) {
                    if (q % 6 == 0 && q > 0) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (NT == 2 && q < 4) __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
                    else if (q < 16) __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                }
#pragma unroll
                for (int q = 0; q < NM; ++q) {
                    if (q % 6 == 0 && q > 0) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (NT == 2 && q >= 2 && q < 8) __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
                    else if (q >= 2) __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                }
              }
            }
            phi_half(nk - 1, 1, b1h, b1l);
            contract_half(nk - 1, 0, b0h, b0l);
            contract_half(nk - 1, 1, b1h, b1l);
          } else {
            u32x8 bh[TV], bl[TV];
            phi_block(0, bh, bl);
            for (int kb = 0; kb + 1 < nk; ++kb) {
                u32x8 nbh[TV], nbl[TV];
                phi_block(kb + 1, nbh, nbl);
                contract(kb, bh, bl);
                interleave();
#pragma unroll
                for (int t = 0; t < TV; ++t) { bh[t] = nbh[t]; bl[t] = nbl[t]; }
            }
            contract(nk - 1, bh, bl);
          }
        }

        FD_SSTAMP(1)
        // ---- epilogue.  Register r of acc[T][vt] holds row 8 (r / 4) + 4 h + r % 4 of row tile T for vertex (vt, j); swapping
        // the upper half of vertex tile 0's register with the lower half of vertex tile 1's leaves every lane with its OWN
        // vertex (vbase + lane): acc[T][0][r] = row 8 (r / 4) + r % 4, acc[T][1][r] = row 8 (r / 4) + 4 + r % 4.
        // Row 3 f + c of the stack is component c of frame slot f (row_of below).
#pragma unroll
        for (int c = 0; c < NT; ++c) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[c][0][r]), __float_as_uint(acc[c][1][r]), false, false);
                acc[c][0][r] = __uint_as_float(sw[0]); acc[c][1][r] = __uint_as_float(sw[1]);
            }
        }
        // the reference's order: gate -> tangent projection -> fall-off -> add (src/SOP_FaceDeform.cpp:405-438)
        const int64_t i = vbase + lane;
        const bool inb = i < p.N;
        const int64_t ic = inb ? i : p.N - 1;
        const bool gated = own_d2 > p.radius2;
        const unsigned off12 = 12u * (unsigned)lane, off4 = 4u * (unsigned)lane;
        if constexpr (!FAST) {
            nxt = load_raw(gu_next, fastTag);
        }
        // component c of frame slot fs after the swap (indices are compile-time after unrolling)
        auto row_of = [&](int fs, int c) -> float {
            const int row = 3 * fs + c, T = row / 32, rho = row % 32;
            return acc[T][(rho % 8) / 4][4 * (rho / 8) + rho % 4];
        };
        FD_SSTAMP(2)
        if constexpr (FAST) {
            // straight-line stores (see k_deform32_tps_shared): 32 positions + 8 fall-off stores of four frames each.
            // A frame's scalars come out of the table with v_readlane; its address is the frame's pointer as the
            // scalar base plus ONE 32-bit lane offset for all frames (12 (vbase + lane); the host checks 12 N < 2^32),
            // and the fall-off stores take their per-lane pointers from LDS: ~330 instructions per group, not 640.
            const f32x4 ones = {1.f, 1.f, 1.f, 1.f};
            const unsigned voff = 12u * (unsigned)vbase + off12;
            const uint64_t fbase = 4ull * (uint64_t)vbase;
#pragma unroll
            for (int fs = 0; fs < NSLOT; ++fs) {
                const float inv = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)tab[(8 * fs) / 64], (8 * fs) % 64));
                const uint64_t pout = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)tab[(8 * fs + 5) / 64], (8 * fs + 5) % 64) << 32) |
                                      (unsigned)__builtin_amdgcn_readlane((int)tab[(8 * fs + 4) / 64], (8 * fs + 4) % 64);
                const float d0 = row_of(fs, 0), d1 = row_of(fs, 1), d2c = row_of(fs, 2);
                if constexpr (NONTEMPORAL) {
                    if (fs % 4 == 0) __builtin_nontemporal_store(ones, (f32x4_a16 FD_GLOBAL *)(s_ftab[(fs / 4) * 64 + lane] + fbase));
                    store_pos3_nt((Pos3 FD_GLOBAL *)((char FD_GLOBAL *)pout + voff), __builtin_fmaf(d0, inv, pos[0]), __builtin_fmaf(d1, inv, pos[1]),
                                  __builtin_fmaf(d2c, inv, pos[2]));
                } else {
                    if (fs % 4 == 0) *(f32x4 FD_GLOBAL *)(s_ftab[(fs / 4) * 64 + lane] + fbase) = ones;
                    store_pos3((Pos3 FD_GLOBAL *)((char FD_GLOBAL *)pout + voff), __builtin_fmaf(d0, inv, pos[0]), __builtin_fmaf(d1, inv, pos[1]),
                               __builtin_fmaf(d2c, inv, pos[2]));
                }
            }
            settle(nxt);
            FD_SSTAMP(3)
            return;
        }
        if (inb && gated) {
            // B2: a gated vertex keeps its position (and no fd_falloff entry is written); as a displacement: zero
            for (int f = 0; f < p.nF; ++f) {
                float *dstp = s_frames[f].P_out;
                if (p.delta) store_pos3((Pos3 FD_GLOBAL *)as_global(dstp) + i, 0.f, 0.f, 0.f);
                else if (dstp != p.P_in) store_pos3((Pos3 FD_GLOBAL *)as_global(dstp) + i, pos[0], pos[1], pos[2]);
            }
        }
        float fall = 1.f;
        float a1[3] = {0.f, 0.f, 0.f}, a2[3] = {0.f, 0.f, 0.f};
        const bool doit = inb && !gated;
        const float base[3] = {p.delta ? 0.f : pos[0], p.delta ? 0.f : pos[1], p.delta ? 0.f : pos[2]};     // (0 + d f = d f exactly)
        if (doit) {
            if (p.dist2 != nullptr || !(p.radius2 != 0.f)) {
                const float q = fminf(own_d2 / p.radius2, 1.f);
                fall = powf(1.f - q, p.falloffrate);
            }
            if (p.tu) {
                // project_to_tangents (src/SOP_FaceDeform.hpp:28-41): the two axes depend on the vertex only
                float u[3] = {p.tu[3 * ic], p.tu[3 * ic + 1], p.tu[3 * ic + 2]};
                float v[3] = {p.tv[3 * ic], p.tv[3 * ic + 1], p.tv[3 * ic + 2]};
                float n[3] = {p.nrm[3 * ic], p.nrm[3 * ic + 1], p.nrm[3 * ic + 2]};
                normalize3(u[0], u[1], u[2]);
                normalize3(v[0], v[1], v[2]);
                normalize3(n[0], n[1], n[2]);
                float gm[3][3];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) gm[r][c] = u[r] * u[c] + v[r] * v[c] + n[r] * n[c];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    a1[c] = u[0] * gm[0][c] + u[1] * gm[1][c] + u[2] * gm[2][c];
                    a2[c] = v[0] * gm[0][c] + v[1] * gm[1][c] + v[2] * gm[2][c];
                }
                normalize3(a1[0], a1[1], a1[2]);
                normalize3(a2[0], a2[1], a2[2]);
            }
        }
#pragma unroll
        for (int fs = 0; fs < NSLOT; ++fs) {
            const int f = fs;
            if (f >= p.nF) continue;
            const float inv = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)tab[(8 * f) / 64], (8 * f) % 64));
            const bool built = __builtin_amdgcn_readlane((int)tab[(8 * f + 1) / 64], (8 * f + 1) % 64) != 0;
            const uint64_t pout = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)tab[(8 * f + 5) / 64], (8 * f + 5) % 64) << 32) |
                                  (unsigned)__builtin_amdgcn_readlane((int)tab[(8 * f + 4) / 64], (8 * f + 4) % 64);
            const uint64_t fout = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)tab[(8 * f + 7) / 64], (8 * f + 7) % 64) << 32) |
                                  (unsigned)__builtin_amdgcn_readlane((int)tab[(8 * f + 6) / 64], (8 * f + 6) % 64);
            if (!doit) continue;
            Pos3 FD_GLOBAL *dstP = (Pos3 FD_GLOBAL *)((char FD_GLOBAL *)(pout + 12ull * (uint64_t)vbase) + off12);
            if (!built) {
                if (p.delta) store_pos3(dstP, 0.f, 0.f, 0.f);
                else if (pout != (uint64_t)p.P_in) store_pos3(dstP, pos[0], pos[1], pos[2]);
                continue;
            }
            float disp[3] = {row_of(fs, 0) * inv, row_of(fs, 1) * inv, row_of(fs, 2) * inv};       // 2^-k is exact
            if (p.dbg & 1) continue;
            if (p.tu) {
                const float da1 = disp[0] * a1[0] + disp[1] * a1[1] + disp[2] * a1[2];
                const float da2 = disp[0] * a2[0] + disp[1] * a2[1] + disp[2] * a2[2];
#pragma unroll
                for (int c = 0; c < 3; ++c) disp[c] = a1[c] * da1 + a2[c] * da2;
            }
            if (fout) __builtin_nontemporal_store(fall, (float FD_GLOBAL *)((char FD_GLOBAL *)(fout + 4ull * (uint64_t)vbase) + off4));
            store_pos3_nt(dstP, base[0] + disp[0] * fall, base[1] + disp[1] * fall, base[2] + disp[2] * fall);
        }
        FD_SSTAMP(3)
    };
    bool built_here = true;
#pragma unroll
    for (int q = 0; q < kTabRegs; ++q) {
        const int idx = 64 * q + lane;
        if (idx < NSLOT * 8 && (idx & 7) == 1) built_here = built_here && tab[q] != 0u;
    }
    const bool fast_ok = p.fast && __all(built_here);
    const int nfull = (int)(p.N / THREADS);       // groups in which every wave's 64 vertices exist
    int u = dynamic ? next_unit(0) : wave;
    int ru = dynamic ? round_taken : 0;
    nxt = load_raw(unit_group(u) < ngroups ? unit_global(u) : 0, std::false_type{});
    settle(nxt);
    // two loops, not one with a branch: where the two kinds of group met at the loop's head the compiler had to assume the
    // worst of both for the loads in flight, and waited for every store of the straight-line epilogue again
    if (fast_ok) {
        while (unit_group(u) < nfull) {
            const int un = next_unit(u);     // taken now: its positions are requested a K loop ahead
            const int run = dynamic ? round_taken : ru + 1;
            do_group(u, un, ru, std::true_type{});
            u = un; ru = run;
        }
    }
    while (unit_group(u) < ngroups) {
        const int un = next_unit(u);
        const int run = dynamic ? round_taken : ru + 1;
        do_group(u, un, ru, std::false_type{});
        u = un; ru = run;
    }
    if (stamp && lane == 0 && wave < 8) {
        for (int q = 0; q < 4; ++q) p.stamps[wave * 8 + q] = st_acc[q];
        p.stamps[wave * 8 + 4] = __builtin_amdgcn_s_memtime() - st_t0;          // shader clock against the 100 MHz reference
        p.stamps[wave * 8 + 5] = __builtin_amdgcn_s_memrealtime() - st_r0;
    }
    if (p.stamps != nullptr && lane == 0 && wave < 8 && blockIdx.x < kMaxCUs) {      // every wave's first and last tick of the 100 MHz clock: the spread over the workgroups
        p.stamps[64 + ((size_t)blockIdx.x * 8 + wave) * 2] = st_r0;
        p.stamps[64 + ((size_t)blockIdx.x * 8 + wave) * 2 + 1] = __builtin_amdgcn_s_memrealtime();
    }
#undef FD_SSTAMP
}

// ---- 17 to 32 frames, round 4: ONE 32-vertex tile per wave, three waves per SIMD ------------------------------------------------
// Round 3's 32-row kernel keeps two waves per SIMD (64 vertices = two vertex tiles each, 96 accumulator registers) and loses a
// third of the launch where a wave is outside its K loop -- loads, epilogue, stores -- because the partner alone cannot keep the
// matrix pipe busy (matrix pipe busy 0.48 by PMC).  Here a wave owns ONE vertex tile of 32: 48 accumulator registers, ~130
// registers in all, so THREE waves share a SIMD (twelve per workgroup) and the hardware has two others to issue from while one
// stores.  What changes beside the tile count:
//   * d2 of a K block is ONE v_mfma_f32_32x32x16_f16 (K = 16 holds the four coordinate groups x | y | z | norms that round 3
//     split over two K = 8 instructions): both lane halves carry the same vertex, half 0 supplies groups 0, 1, half 1 groups 2, 3;
//   * no lane-half swap in the epilogue.  Both halves of a lane pair (h, j) hold rows of the SAME vertex j -- register r of
//     row tile c is row 8 (r / 4) + 4 h + r % 4 -- so the rows are dealt out per half: frame slot fs lives in half fs & 1 as local
//     frame lf = fs >> 1, and its component comp is the half's flat accumulator index 3 lf + comp (tile (3 lf + comp) / 16,
//     register (3 lf + comp) % 16).  One position store instruction writes TWO frames: lanes of half 0 frame 2 lf, of half 1
//     frame 2 lf + 1, 384 contiguous bytes each (per-lane 64-bit addresses from a 16-byte table entry per (lf, h) in LDS);
//   * fd_falloff rows of a unit are 128 bytes per frame: one 16-byte store per lane covers EIGHT frames.
// The weight tiles are packed to that row order (k_pack_shared_wide, layout 1); their [K block][tile][K step][hi, lo][lane] order,
// the K-slot <-> centre map and the polynomial tile are round 3's.  The K loop is plain double buffering, unrolled by two so
// that no operand is copied: phi of block k + 1 (16 logarithms, 16 multiplies, the fp16 split) goes under the 18 matrix
// instructions of block k.
constexpr int kW1Waves = 12;                      // three per SIMD (tuning builds also instantiate 8: two per SIMD)
constexpr int kW1Unit = 32;                       // vertices per wave and unit
constexpr size_t w1_fixed_lds(int nt) { return sizeof(SharedFrame) * (size_t)kWideSlots + 32 * 16 + (size_t)nt * 64 * 16 + 256 * sizeof(uint64_t) + 16; }

// The whole model must be resident in LDS (p.nkb <= p.kchunk: M = 256 at 32 frames): no staging and no barrier inside the unit
// loop.  Models that are staged in chunks keep the two-tile kernel: its 512-vertex groups re-stage the model a quarter less often
// (C3, M = 2048: 1.04 against 1.15 ms per 32 frames; C5's ranges, M = 512: 0.40 against 0.44 ms, same process).
template <bool GAUSS, int NT, int NSLOT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(WAVES / 4, WAVES / 4)))
void k_deform32_shared_w1(const SharedParams p, int ngroups)
{
    constexpr int THREADS = 64 * WAVES;
    constexpr int NLF = NSLOT / 2;                 // local frames per lane half
    constexpr int NFQ = (NSLOT + 7) / 8;           // fd_falloff stores per unit (eight frames each)
    static_assert(NSLOT % 4 == 0 && NSLOT <= kWideSlots && 3 * NLF <= 16 * NT, "frame slots in fours, three rows each, half of them per lane half");
    constexpr int kW16 = wide_w16(NT);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: [frame records 32][(lf, h) table 32 x 16 B][polynomial tiles NT x 1 KiB][fd_falloff pointers 4 x 64 x 8 B][ticket]
    //      [d2 operands kchunk x 1 KiB][weight tiles kchunk x NT x 4 KiB]
    SharedFrame *s_frames = reinterpret_cast<SharedFrame *>(smem);
    uint4 *s_ptab = reinterpret_cast<uint4 *>(smem + sizeof(SharedFrame) * (size_t)kWideSlots);
    uint4 *s_poly = s_ptab + 32;
    uint64_t *s_ftab = reinterpret_cast<uint64_t *>(s_poly + NT * 64);
    unsigned *s_ticket = reinterpret_cast<unsigned *>(s_ftab + 256);
    uint4 *s_ct = reinterpret_cast<uint4 *>(s_ticket + 4);
    uint4 *s_w = s_ct + (size_t)64 * p.kchunk;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const float n0 = p.norm[0], n1 = p.norm[1], n2 = p.norm[2];
    const float inv_s = p.norm[3];
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;

    auto stage_model = [&]() {
        const int nk = p.nkb;
        for (int q = tid; q < nk * 64; q += THREADS) s_ct[q] = reinterpret_cast<const uint4 *>(p.ctiles)[q];
        // eight loads in flight per thread (native vectors: an array of HIP's uint4 structs goes through scratch memory)
        const u32x4 *src = reinterpret_cast<const u32x4 *>(p.wtiles);
        u32x4 *dst = reinterpret_cast<u32x4 *>(s_w);
        const int n16 = nk * kW16;
        int q = tid;
        for (; q + 7 * THREADS < n16; q += 8 * THREADS) {
            const u32x4 v0 = src[q], v1 = src[q + THREADS], v2 = src[q + 2 * THREADS], v3 = src[q + 3 * THREADS];
            const u32x4 v4 = src[q + 4 * THREADS], v5 = src[q + 5 * THREADS], v6 = src[q + 6 * THREADS], v7 = src[q + 7 * THREADS];
            dst[q] = v0; dst[q + THREADS] = v1; dst[q + 2 * THREADS] = v2; dst[q + 3 * THREADS] = v3;
            dst[q + 4 * THREADS] = v4; dst[q + 5 * THREADS] = v5; dst[q + 6 * THREADS] = v6; dst[q + 7 * THREADS] = v7;
        }
        for (; q < n16; q += THREADS) dst[q] = src[q];
        __syncthreads();
    };

    {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.frames);
        uint4 *dst = reinterpret_cast<uint4 *>(s_frames);
        for (int q = tid; q < kWideSlots * (int)(sizeof(SharedFrame) / 16); q += THREADS) dst[q] = src[q];
        const uint4 *psrc = p.wtiles + (size_t)p.nkb * kW16;
        for (int q = tid; q < NT * 64; q += THREADS) s_poly[q] = psrc[q];
        if (tid == 0) *s_ticket = 0u;
        if (tid < 32) {
            // entry 2 lf + h: frame slot 2 lf + h = tid itself -- {P_out, 2^-k, built}
            const SharedFrame fr = p.frames[tid < NSLOT ? tid : NSLOT - 1];
            const uint64_t po = (uint64_t)fr.P_out;
            s_ptab[tid] = make_uint4((unsigned)po, (unsigned)(po >> 32), __float_as_uint(fr.inv_scale), (unsigned)fr.built);
        }
        if (p.fast && tid < NFQ * 64) {
            const int f = 8 * (tid >> 6) + ((tid & 63) >> 3);
            s_ftab[tid] = (uint64_t)p.frames[f < NSLOT ? f : NSLOT - 1].falloff_out + 16u * (unsigned)(tid & 7);
        }
    }
    stage_model();

    struct UnitRaw { float p[3]; float d2; };
    // Units: the workgroup's groups (blockIdx + k gridDim) are twelve 32-vertex units each; a wave takes its NEXT unit from a
    // counter in LDS (no barrier in the loop), and the groups that do not fill a last round are dealt out as single units,
    // workgroup by workgroup (as in the two-tile kernel above).
    // (32-bit unit numbers: the host keeps N below 2^31, so there are fewer than 2^26 units; a ticket beyond the last unit maps to
    // total_units or more, never wraps: at most WAVES tickets are taken after the first empty one)
    const int whole_rounds = ngroups / (int)gridDim.x;
    const int pool0 = whole_rounds * (int)gridDim.x * WAVES, total_units = ngroups * WAVES;
    auto unit_global = [&](int u) -> int {          // the 32-vertex unit behind local ticket u (>= total_units: none)
        if (u < WAVES * whole_rounds) return ((int)blockIdx.x + (u / WAVES) * (int)gridDim.x) * WAVES + (u % WAVES);
        const int q = u - WAVES * whole_rounds;
        return q < 2 * WAVES ? pool0 + (int)blockIdx.x + q * (int)gridDim.x : total_units;
    };
    auto unit_group = [&](int u) -> int { const int g = unit_global(u); return g < total_units ? (g / WAVES) : ngroups; };
    auto next_unit = [&](int u) {
        (void)u;
        unsigned v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(s_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return (int)__builtin_amdgcn_readfirstlane(v);
    };
    auto load_raw = [&](int gu, auto fastTag) {
        constexpr bool FAST = decltype(fastTag)::value;
        UnitRaw r;
        const int64_t vi = (int64_t)gu * kW1Unit + j;
        const int64_t vc = vi < p.N ? vi : p.N - 1;
        if constexpr (FAST) {
            // (the straight-line path runs with 12 N < 2^32: the frame's pointer as scalar base, one 32-bit lane offset)
            const char *pb = reinterpret_cast<const char *>(p.P_in) + (size_t)(12u * (unsigned)vc);
            r.p[0] = __builtin_nontemporal_load(reinterpret_cast<const float *>(pb)); r.p[1] = __builtin_nontemporal_load(reinterpret_cast<const float *>(pb + 4));
            r.p[2] = __builtin_nontemporal_load(reinterpret_cast<const float *>(pb + 8));
            r.d2 = 0.f;
        } else {
            r.p[0] = __builtin_nontemporal_load(&p.P_in[3 * vc]); r.p[1] = __builtin_nontemporal_load(&p.P_in[3 * vc + 1]);
            r.p[2] = __builtin_nontemporal_load(&p.P_in[3 * vc + 2]);
            r.d2 = p.dist2 ? p.dist2[vc] : 0.f;
        }
        return r;
    };
    auto settle = [&](const UnitRaw &r) { asm volatile("" :: "v"(r.p[0]), "v"(r.p[1]), "v"(r.p[2])); };
    UnitRaw nxt;
    int uc = 0;                                   // units this wave has done
#ifdef FD_TUNING
    // diagnostics of tuning builds (-DFD_TUNING): shader-clock shares of a unit's phases per wave of workgroup 0, p.dbg bit 0 = no
    // stores, bit 1 = no K loop
    const bool stamp = p.stamps != nullptr && blockIdx.x == 0;
    const unsigned long long st_t0 = p.stamps ? __builtin_amdgcn_s_memtime() : 0, st_r0 = p.stamps ? __builtin_amdgcn_s_memrealtime() : 0;
    unsigned long long st_prev = st_t0, st_acc[4] = {0, 0, 0, 0};
#define FD_W1STAMP(K) if (stamp) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[K] += t_ - st_prev; st_prev = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define FD_W1STAMP(K)
#endif
    auto do_unit = [&](int u, int un, auto fastTag) {
        constexpr bool FAST = decltype(fastTag)::value;
        const int64_t vbase = (int64_t)unit_global(u) * kW1Unit;
        const int gu_next = unit_group(un) < ngroups ? unit_global(un) : unit_global(u);       // whose positions to request
        const UnitRaw cur = nxt;
        const float pos[3] = {cur.p[0], cur.p[1], cur.p[2]};
        const float own_d2 = cur.d2;
        f32x16 acc[NT];
        const float x = (cur.p[0] - n0) * inv_s, y = (cur.p[1] - n1) * inv_s, z = (cur.p[2] - n2) * inv_s;
        const bool lane_live = FAST ? true : ((vbase + j < p.N) && !(own_d2 > p.radius2));
        const float xx = __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x));
        f16x8 bop;
        {
            // d2 operand, K = 16: lane half 0 carries the coordinate groups (x | y), half 1 (z | norms)
            const float va = h == 0 ? -2.f * x : -2.f * z, vb2 = h == 0 ? -2.f * y : xx;
            const _Float16 ha = (_Float16)va, la = (_Float16)(va - (float)ha);
            const _Float16 hb = (_Float16)vb2, lb = (_Float16)(vb2 - (float)hb);
            const _Float16 one = (_Float16)1.0f;
            bop = h == 0 ? (f16x8){ha, la, ha, la, hb, lb, hb, lb} : (f16x8){ha, la, ha, la, one, one, hb, lb};
        }
        {
            // polynomial operand, K = 16 (k_pack_shared_wide): half 0 {1, xh, yh, zh, xxh, xl, yl, zl}, half 1 {xxl, 1, xh, yh, zh, xxh, 0, 0}
            // (GAUSS: everything times 2^10, the factor phi carries; undone with the frame's scale)
            constexpr float ps = GAUSS ? (float)(1 << kGaussShift) : 1.f;
            constexpr unsigned one16 = GAUSS ? 0x6400u : 0x3c00u;         // fp16 1024 / 1
            unsigned xyh, xyl, zxh, zxl;
            split_pair_f16(x * ps, y * ps, xyh, xyl);
            split_pair_f16(z * ps, xx * ps, zxh, zxl);
            u32x4 pb;
            if (h == 0) pb = (u32x4){one16 | (xyh << 16), (xyh >> 16) | (zxh << 16), (zxh >> 16) | (xyl << 16), (xyl >> 16) | (zxl << 16)};
            else pb = (u32x4){(zxl >> 16) | (one16 << 16), xyh, zxh, 0u};
            const f16x8 pbv = __builtin_bit_cast(f16x8, pb);
#pragma unroll
            for (int c = 0; c < NT; ++c)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, s_poly[c * 64 + lane]), pbv, zero16, 0, 0, 0);
        }
        const bool wave_work = FAST ? true : __any(lane_live);
        if constexpr (FAST) nxt = load_raw(gu_next, fastTag);      // a whole K loop ahead of this unit's stores
        FD_W1STAMP(0)

        // One K block of the software pipeline.  Entering step kb: (ch, cl) = the fp16 pieces of phi(kb), 16 values per lane
        // (K step s in registers 4 s .. 4 s + 3); dd = the raw d2 of block kb + 1 (thin-plate); (w0h, w0l) = the weight
        // operands of block kb's first (K step, row tile) pair.  The step forms phi(kb + 1) from dd into (nh, nl) under the 6 NT
        // matrix instructions that contract phi(kb) with the weights, fetches every pair's weights one pair ahead (the first
        // pair of block kb + 1 included), and issues the d2 instruction of block kb + 2 as soon as dd's last value is read.
        // LAST: block kb is the chunk's last -- contraction only.
        f32x16 dd;
        f16x8 w0h, w0l;
        auto d2_block = [&](int kb) {
            if constexpr (!GAUSS) dd = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, s_ct[(size_t)kb * 64 + lane]), bop, zero16, 0, 0, 0);
        };
        auto phi_from = [&](int kb, u32x8 &xh, u32x8 &xl) {
            if constexpr (GAUSS) {
                // exp(-d2 / R_j^2) from direct coordinate differences, one value per instruction; this lane's
                // centres of K step s -- 16 s + 8 (m / 4) + 4 h + m % 4 -- are records 8 (2 s + m / 4) + 4 h + m % 4 of the block
                const float4 *cr = reinterpret_cast<const float4 *>(s_ct + (size_t)kb * 64) + 4 * h;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float ph[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float4 c = cr[8 * (q >> 1) + 2 * (q & 1) + e];
                        const float dx = x - c.x, dy = y - c.y, dz = z - c.z;
                        float d2 = dx * dx;
                        d2 = __builtin_fmaf(dy, dy, d2);
                        d2 = __builtin_fmaf(dz, dz, d2);
                        ph[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(d2, c.w, (float)kGaussShift));
                    }
                    unsigned hh, ll;
                    split_pair_f16<true>(ph[0], ph[1], hh, ll);
                    xh[q] = hh; xl[q] = ll;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    unsigned hh, ll;
                    split_pair_f16<false, false>(d2_log_d2(dd[2 * q]), d2_log_d2(dd[2 * q + 1]), hh, ll);
                    xh[q] = hh; xl[q] = ll;
                }
                // the two wait states a vector write needs before a matrix instruction reads it as an operand, ONCE for the
                // block: the lo pieces come out of asm strings the compiler pads nothing behind, and every later use of them
                // depends on this statement
                asm volatile("s_nop 1" : "+v"(xl[0]), "+v"(xl[1]), "+v"(xl[2]), "+v"(xl[3]), "+v"(xl[4]), "+v"(xl[5]), "+v"(xl[6]), "+v"(xl[7]));
            }
        };
        auto step = [&](int kb, int nk, const u32x8 &ch, const u32x8 &cl, u32x8 &nh, u32x8 &nl, auto lastTag) {
            constexpr bool LAST = decltype(lastTag)::value;
            constexpr int NP = 2 * NT;                         // (K step, row tile) pairs of a block, pair = s * NT + c
            const uint4 *wk = s_w + (size_t)kb * kW16 + lane;
            const int kbn = kb + 1 < nk ? kb + 1 : kb, kbd = kb + 2 < nk ? kb + 2 : nk - 1;
            const uint4 *wkn = s_w + (size_t)kbn * kW16 + lane;
            if constexpr (!LAST) phi_from(kb + 1, nh, nl);
            f16x8 ph_ = w0h, pl_ = w0l;
#pragma unroll
            for (int pr = 0; pr < NP; ++pr) {
                const int s_ = pr / NT, c = pr % NT;
                // the next pair's weights (the last pair fetches the first pair of the next block)
                f16x8 qh, ql;
                if (pr + 1 < NP) {
                    const int s2 = (pr + 1) / NT, c2 = (pr + 1) % NT;
                    qh = __builtin_bit_cast(f16x8, wk[((c2 * 2 + s2) * 2) * 64]); ql = __builtin_bit_cast(f16x8, wk[((c2 * 2 + s2) * 2 + 1) * 64]);
                } else {
                    qh = __builtin_bit_cast(f16x8, wkn[0]); ql = __builtin_bit_cast(f16x8, wkn[64]);
                }
                const f16x8 vh = __builtin_bit_cast(f16x8, (u32x4){ch[4 * s_], ch[4 * s_ + 1], ch[4 * s_ + 2], ch[4 * s_ + 3]});
                const f16x8 vl = __builtin_bit_cast(f16x8, (u32x4){cl[4 * s_], cl[4 * s_ + 1], cl[4 * s_ + 2], cl[4 * s_ + 3]});
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph_, vh, acc[c], 0, 0, 0);
                // (the d2 instruction of block kb + 2 in source order where the schedule below wants it: behind the first
                // instruction of the last pair -- every logarithm of this step has read dd by then)
                if constexpr (!LAST) if (pr == NP - 1) d2_block(kbd);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl_, vh, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph_, vl, acc[c], 0, 0, 0);
                ph_ = qh; pl_ = ql;
            }
            w0h = ph_; w0l = pl_;
            if constexpr (GAUSS || LAST) return;             // (the Gaussian block is left to the scheduler's own order)
            // issue order: under each matrix instruction one logarithm (two at first where a block has only 12) issued BEFORE it
            // -- the matrix instruction is the wait state between the logarithm and the multiply that reads it --, then the
            // other vector work; LDS reads one pair ahead, behind the first instruction of a pair; the d2 instruction of block
            // kb + 2 right behind the matrix instruction that follows the last logarithm
            constexpr int NM = 6 * NT, TR2 = NT == 3 ? 0 : 8;        // slots with two logarithms
            constexpr int QD = 3 * (NP - 1);                         // the d2 instruction goes behind this slot's
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                if (q < TR2) __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
                else if (q < TR2 + (16 - 2 * TR2)) __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (q % 3 == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                if (q == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (q == QD) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, NT == 3 ? FD_W1_VPM : FD_W1_VPM + 1, 0);
            }
        };
#ifdef FD_TUNING
        const bool k_loop = (FAST || wave_work) && !(p.dbg & 2);
#else
        const bool k_loop = FAST || wave_work;
#endif
        if (k_loop) {
            const int nk = p.nkb;
            u32x8 ah_, al_, bh_, bl_;
            d2_block(0);
            phi_from(0, ah_, al_);
            d2_block(nk > 1 ? 1 : 0);
            w0h = __builtin_bit_cast(f16x8, s_w[lane]); w0l = __builtin_bit_cast(f16x8, s_w[64 + lane]);
            int kb = 0;
            while (kb + 2 < nk) {
                step(kb, nk, ah_, al_, bh_, bl_, std::false_type{});
                step(kb + 1, nk, bh_, bl_, ah_, al_, std::false_type{});
                kb += 2;
            }
            if (kb + 2 == nk) {
                step(kb, nk, ah_, al_, bh_, bl_, std::false_type{});
                step(kb + 1, nk, bh_, bl_, ah_, al_, std::true_type{});
            } else {
                step(kb, nk, ah_, al_, bh_, bl_, std::true_type{});
            }
        }

        // ---- epilogue.  Lane (h, j): vertex vbase + j, the frames of its half.  The reference's order: gate -> tangent
        // projection -> fall-off -> add (src/SOP_FaceDeform.cpp:405-438)
        auto accf = [&](int k) -> float { return acc[k / 16][k % 16]; };      // indices are compile-time after unrolling
        const int64_t i = vbase + j;
        FD_W1STAMP(1)
        if constexpr (!FAST) nxt = load_raw(gu_next, fastTag);
        if constexpr (FAST) {
            // straight-line stores: NLF position stores of two frames each + NFQ fall-off stores of eight frames each.  Every table
            // entry is requested before the first store (one LDS round trip, not one per store)
            const f32x4 ones = {1.f, 1.f, 1.f, 1.f};
            uint64_t voff = 12ull * (uint64_t)i, fbase = 4ull * (uint64_t)vbase;
            asm volatile("" : "+v"(voff));                 // (kept as a value: folded into the address it becomes a 64-bit multiply-add per store)
            uint4 ent[NLF];
            uint64_t fptr[NFQ];
#pragma unroll
            for (int lf = 0; lf < NLF; ++lf) ent[lf] = s_ptab[2 * lf + h];
#pragma unroll
            for (int q = 0; q < NFQ; ++q) fptr[q] = s_ftab[q * 64 + lane];
            __builtin_amdgcn_sched_group_barrier(0x100, NLF + NFQ, 0);
#pragma unroll
            for (int lf = 0; lf < NLF; ++lf) {
                const uint4 e = ent[lf];
                const uint64_t dst = (((uint64_t)e.y << 32) | (uint64_t)e.x) + voff;
                const float inv = __uint_as_float(e.z);
#ifdef FD_TUNING
                if (p.dbg & 1) { asm volatile("" :: "v"(accf(3 * lf)), "v"(accf(3 * lf + 1)), "v"(accf(3 * lf + 2)), "v"(dst), "v"(inv)); continue; }
#endif
                if (lf % 4 == 0) __builtin_nontemporal_store(ones, (f32x4_a16 FD_GLOBAL *)(fptr[lf / 4] + fbase));
                store_pos3_nt((Pos3 FD_GLOBAL *)dst, __builtin_fmaf(accf(3 * lf), inv, pos[0]), __builtin_fmaf(accf(3 * lf + 1), inv, pos[1]),
                              __builtin_fmaf(accf(3 * lf + 2), inv, pos[2]));
            }
            settle(nxt);
            ++uc;
            FD_W1STAMP(2)
            return;
        }
        const bool inb = i < p.N;
        const int64_t ic = inb ? i : p.N - 1;
        const bool gated = own_d2 > p.radius2;
        if (inb && gated) {
            // B2: a gated vertex keeps its position (and no fd_falloff entry is written); as a displacement: zero
            for (int f = h; f < p.nF; f += 2) {
                float *dstp = s_frames[f].P_out;
                if (p.delta) store_pos3((Pos3 FD_GLOBAL *)as_global(dstp) + i, 0.f, 0.f, 0.f);
                else if (dstp != p.P_in) store_pos3((Pos3 FD_GLOBAL *)as_global(dstp) + i, pos[0], pos[1], pos[2]);
            }
        }
        float fall = 1.f;
        float a1[3] = {0.f, 0.f, 0.f}, a2[3] = {0.f, 0.f, 0.f};
        const bool doit = inb && !gated;
        const float base[3] = {p.delta ? 0.f : pos[0], p.delta ? 0.f : pos[1], p.delta ? 0.f : pos[2]};     // (0 + d f = d f exactly)
        if (doit) {
            if (p.dist2 != nullptr || !(p.radius2 != 0.f)) {
                const float q = fminf(own_d2 / p.radius2, 1.f);
                fall = powf(1.f - q, p.falloffrate);
            }
            if (p.tu) {
                // project_to_tangents (src/SOP_FaceDeform.hpp:28-41): the two axes depend on the vertex only
                float u3[3] = {p.tu[3 * ic], p.tu[3 * ic + 1], p.tu[3 * ic + 2]};
                float v3[3] = {p.tv[3 * ic], p.tv[3 * ic + 1], p.tv[3 * ic + 2]};
                float n3[3] = {p.nrm[3 * ic], p.nrm[3 * ic + 1], p.nrm[3 * ic + 2]};
                normalize3(u3[0], u3[1], u3[2]);
                normalize3(v3[0], v3[1], v3[2]);
                normalize3(n3[0], n3[1], n3[2]);
                float gm[3][3];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) gm[r][c] = u3[r] * u3[c] + v3[r] * v3[c] + n3[r] * n3[c];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    a1[c] = u3[0] * gm[0][c] + u3[1] * gm[1][c] + u3[2] * gm[2][c];
                    a2[c] = v3[0] * gm[0][c] + v3[1] * gm[1][c] + v3[2] * gm[2][c];
                }
                normalize3(a1[0], a1[1], a1[2]);
                normalize3(a2[0], a2[1], a2[2]);
            }
        }
#pragma unroll
        for (int lf = 0; lf < NLF; ++lf) {
            const int f = 2 * lf + h;                      // this lane's frame of the pair
            if (f >= p.nF || !doit) continue;
            const SharedFrame fr = s_frames[f];
            Pos3 FD_GLOBAL *dstP = (Pos3 FD_GLOBAL *)as_global(fr.P_out) + i;
            if (!fr.built) {
                if (p.delta) store_pos3(dstP, 0.f, 0.f, 0.f);
                else if (fr.P_out != p.P_in) store_pos3(dstP, pos[0], pos[1], pos[2]);
                continue;
            }
            float disp[3] = {accf(3 * lf) * fr.inv_scale, accf(3 * lf + 1) * fr.inv_scale, accf(3 * lf + 2) * fr.inv_scale};       // 2^-k is exact
            if (p.tu) {
                const float da1 = disp[0] * a1[0] + disp[1] * a1[1] + disp[2] * a1[2];
                const float da2 = disp[0] * a2[0] + disp[1] * a2[1] + disp[2] * a2[2];
#pragma unroll
                for (int c = 0; c < 3; ++c) disp[c] = a1[c] * da1 + a2[c] * da2;
            }
            if (fr.falloff_out) __builtin_nontemporal_store(fall, (float FD_GLOBAL *)as_global(fr.falloff_out) + i);
            store_pos3_nt(dstP, base[0] + disp[0] * fall, base[1] + disp[1] * fall, base[2] + disp[2] * fall);
        }
    };
    bool built_here = true;
    if (lane < NSLOT) built_here = p.frames[lane].built != 0;
    const bool fast_ok = p.fast && __all(built_here);
    const int nfull = (int)(p.N / (kW1Unit * WAVES));       // groups in which every wave's 32 vertices exist
    int u = next_unit(0);
    nxt = load_raw(unit_group(u) < ngroups ? unit_global(u) : 0, std::false_type{});
    settle(nxt);
    // two loops, not one with a branch (see the two-tile kernel)
    if (fast_ok) {
        while (unit_group(u) < nfull) {
            const int un = next_unit(u);     // taken now: its positions are requested a K loop ahead
            do_unit(u, un, std::true_type{});
            u = un;
        }
    }
    while (unit_group(u) < ngroups) {
        const int un = next_unit(u);
        do_unit(u, un, std::false_type{});
        u = un;
    }
#ifdef FD_TUNING
    if (stamp && lane == 0 && wave < 16) {
        for (int q = 0; q < 3; ++q) p.stamps[wave * 8 + q] = st_acc[q];
        p.stamps[wave * 8 + 3] = (unsigned long long)uc;
        p.stamps[wave * 8 + 4] = __builtin_amdgcn_s_memtime() - st_t0;          // shader clock against the 100 MHz reference
        p.stamps[wave * 8 + 5] = __builtin_amdgcn_s_memrealtime() - st_r0;
    }
#endif
#undef FD_W1STAMP
}
}  // namespace

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property of a kernel: one process may hold contexts on
// several GPUs (fd_config.device), so "done" is remembered per (kernel, device), not per process.
constexpr int kMaxDevices = 64;
struct LdsAttrOnce {
    bool done[kMaxDevices] = {};
    hipError_t ensure(const void *fn, int bytes)
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < kMaxDevices && done[dev]) return hipSuccess;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess && dev >= 0 && dev < kMaxDevices) done[dev] = true;
        return e;
    }
};

// Frames of one mesh and one rest rig (SharedDeformArgs): pack the weight tiles, then one launch.
hipError_t launch_deform_shared(const SharedDeformArgs &a, hipStream_t stream)
{
    if (a.N <= 0 || a.nF <= 0) return hipSuccess;
    if (a.nF > kMaxBatch || a.Mpad % 16 != 0) return hipErrorInvalidValue;
    const int ntiles = a.Mpad / 16, nkb = (ntiles + 1) / 2;
    const bool gauss = a.kind != FD_KERNEL_THIN_PLATE;       // FD_KERNEL_GAUSSIAN / _QNN: per-record scale, direct differences
    // 17..32 frames: 32-row tiles (r2 kept the 16-row kernel selectable for them, at six tiles: 201 against 175 us at C2 x 32)
    const bool wide = shared_wide(a.nF, a.kind);
    const bool dense = shared_dense(a.nF);
    const int nT = shared_tiles(a.nF);                       // 16-row tiles (also what the scratch is sized by)
    const int wslot = wide_slots(a.nF), wNT = wide_tiles(wslot);
    // Frame slots beyond nF are duplicates of the LAST frame: same weights, same scale, same output pointers, so their
    // stores repeat that frame's bits at the same addresses and the straight-line epilogue needs no "unused slot" case.
    const int nslot = wide ? wslot : shared_slots(nT, dense);
    SharedSlots slots{};
    SharedOut out{};
    for (int f = 0; f < kMaxBatch; ++f) {
        const int q = f < a.nF ? f : a.nF - 1;
        slots.rec32[f] = a.rec32[q]; slots.model[f] = a.model[q]; slots.centres[f] = a.centres[q];
        out.P_out[f] = a.P_out[q]; out.falloff_out[f] = a.falloff_out ? a.falloff_out[q] : nullptr;
    }
    slots.M = a.M; slots.nreal = a.nF; slots.mismatch = a.mismatch;
    // 17..32 frames: one vertex tile per wave, three waves per SIMD (k_deform32_shared_w1); FD_SHARED_W1=0: round 3's two-tile kernel
#ifdef FD_TUNING
    // (tuning builds read the switch on every launch: tests/tools/shared_ab_timing.py alternates the two kernels inside one process)
    const bool w1sel = [] { const char *e = tuning_env("FD_SHARED_W1"); return e ? atoi(e) != 0 : true; }();
#else
    constexpr bool w1sel = true;
#endif
    // ... where the whole model is resident in LDS (M = 256 at 32 frames, 384 at 20); models staged in chunks keep the two-tile kernel
    const bool w1 = wide && w1sel && (kSharedLdsBudget - w1_fixed_lds(wNT)) / ((size_t)1024 + (size_t)wide_w16(wNT) * 16) >= (size_t)nkb;
#ifdef FD_TUNING
    const int w1waves = [] { const char *e = tuning_env("FD_W1_WAVES"); return e && atoi(e) == 8 ? 8 : kW1Waves; }();
#else
    constexpr int w1waves = kW1Waves;
#endif
    if (a.mode != 2) {
        if (wide)
            hipLaunchKernelGGL(k_pack_shared_wide, dim3(nkb, wNT), dim3(256), 0, stream, slots, out, a.nF, wslot, a.Mpad, (uint4 *)a.wtiles,
                               (SharedFrame *)a.frames, a.ctiles, gauss ? 1 : 0, w1 ? 1 : 0);
        else
            hipLaunchKernelGGL(k_pack_shared, dim3(nkb, nT), dim3(256), 0, stream, slots, out, nslot, a.Mpad, dense ? 1 : 0, (uint4 *)a.wtiles,
                               (SharedFrame *)a.frames, a.ctiles, gauss ? 1 : 0);
        if (a.packed_ev) {
            hipError_t e = hipEventRecord(a.packed_ev, stream);
            if (e != hipSuccess) return e;
        }
        if (a.mode == 1) return hipGetLastError();
    }
    SharedParams p{};
    p.N = a.N; p.P_in = a.P_in; p.dist2 = a.dist2; p.tu = a.tu; p.tv = a.tv; p.nrm = a.nrm;
    p.radius2 = a.radius2; p.falloffrate = a.falloffrate; p.delta = a.delta_out;
    p.ntiles = ntiles; p.nkb = nkb; p.nF = a.nF; p.nT = nT;
    if (wide) {
        const uint4 *copy = (const uint4 *)a.wtiles + (size_t)nkb * wide_w16(wNT) + (size_t)wNT * 64;
        p.ctiles = reinterpret_cast<const MfmaTileH *>(copy);
        p.norm = reinterpret_cast<const float *>(copy + (size_t)nkb * 64);
    } else {
        const uint4 *copy = (const uint4 *)a.wtiles + (size_t)nkb * nT * 128 + (size_t)nT * 64;
        p.ctiles = reinterpret_cast<const MfmaTileH *>(copy);
        p.norm = reinterpret_cast<const float *>(copy + (size_t)2 * nkb * (sizeof(MfmaTileH) / 16));
    }
    p.wtiles = (const uint4 *)a.wtiles; p.frames = (const SharedFrame *)a.frames;
#ifdef FD_TUNING
    { const char *e = tuning_env("FD_SHARED_DBG"); p.dbg = e ? atoi(e) : 0; }
#else
    { static const char *e = tuning_env("FD_SHARED_DBG"); p.dbg = e ? atoi(e) : 0; }
#endif
    {
        static const bool no_fast = tuning_env("FD_SHARED_NO_FAST") != nullptr;       // A/B: general epilogue everywhere
        // no gate, no fall-off input, no tangent frames, fd_falloff wanted for every frame: the straight-line epilogue.
        // Any frame count: the slots a launch runs beyond nF repeat the last frame (above).
        // (repeated slots cost stores: worth it up to a quarter of the frames -- 17..32 frames always qualify)
        bool fast = !no_fast && !a.delta_out && (p.dbg & 1) == 0 && a.dist2 == nullptr && a.tu == nullptr && a.radius2 > 0.f && a.falloff_out != nullptr &&
                    a.N < ((int64_t)1 << 28) && 4 * (nslot - a.nF) <= a.nF;
        // (the straight-line epilogues store fd_falloff 8 and 16 bytes at a time: 16-byte aligned arrays, or the general path)
        for (int f = 0; fast && f < a.nF; ++f) fast = a.falloff_out[f] != nullptr && a.P_out[f] != nullptr && ((uintptr_t)a.falloff_out[f] & 15) == 0;
        p.fast = fast ? 1 : 0;
    }
#ifdef FD_TUNING
    { const char *e = tuning_env("FD_SHARED_STAGGER"); p.stagger = e ? atoi(e) : 0; }
#else
    { static const char *e = tuning_env("FD_SHARED_STAGGER"); p.stagger = e ? atoi(e) : 0; }
#endif
#if defined(FD_SHARED_STAMPS_BUILD) || defined(FD_TUNING)
    // diagnostics, compiled in only for profiling builds (-DFD_SHARED_STAMPS_BUILD / -DFD_TUNING): in-kernel clock stamps, printed per launch
    static unsigned long long *d_stamps = nullptr;
    static const bool want_stamps = tuning_env("FD_SHARED_STAMPS") != nullptr;
    constexpr size_t kStampWords = 64 + (size_t)kMaxCUs * 8 * 2;
    if (want_stamps && !d_stamps) (void)hipMalloc((void **)&d_stamps, kStampWords * sizeof(unsigned long long));
    if (want_stamps && d_stamps) (void)hipMemsetAsync(d_stamps, 0, kStampWords * sizeof(unsigned long long), stream);
    p.stamps = want_stamps ? d_stamps : nullptr;
    { static const bool e = tuning_env("FD_SHARED_STAMPS_GENERAL") != nullptr; if (want_stamps && e) p.fast = 0; }
#else
    p.stamps = nullptr;
#endif
    const size_t fixed = w1 ? w1_fixed_lds(wNT) :
                         wide ? sizeof(SharedFrame) * (size_t)kWideSlots + (size_t)wNT * 64 * 16 + 512 * sizeof(uint64_t) + 16
                              : sizeof(SharedFrame) * (size_t)shared_slots(nT, dense) + (size_t)nT * 64 * 16;
    const size_t per_kb = wide ? (size_t)1024 + (size_t)wide_w16(wNT) * 16 : 2 * sizeof(MfmaTileH) + (size_t)nT * 128 * 16;
    int kchunk = (int)((kSharedLdsBudget - fixed) / per_kb);
    if (kchunk < 1) return hipErrorInvalidValue;
    if (kchunk > nkb) kchunk = nkb;
    p.kchunk = kchunk;
    const size_t lds = fixed + per_kb * (size_t)kchunk;
    const int64_t per = w1 ? kW1Unit * w1waves : wide ? 64 * kWideWaves : kSharedThreads / 64 * 64;          // vertices per workgroup and group
    const int64_t ngroups = (a.N + per - 1) / per;
    // One persistent workgroup per CU (150 KiB of LDS, two 240-register waves per SIMD: nothing else fits beside it).
    // a.max_wgs < 256 (fd_batch_set_eval_cus) leaves the other CUs to whatever runs on other streams -- the builds of the
    // next frames in a pipeline (bench.py).
    // More than 256: the workgroups beyond the resident ones go out as CUs come free -- shorter shares, so a workgroup that had to
    // wait for a CU a build holds delays the launch by less.
    const int64_t max_wgs = a.max_wgs > 0 ? (a.max_wgs < 4096 ? a.max_wgs : 4096) : (int64_t)device_cus();
    const unsigned grid = (unsigned)(ngroups < max_wgs ? ngroups : max_wgs);
#define FD_SHARED_CASE(NTV, DNS, GSS)                                                                                \
    {                                                                                                                \
        static LdsAttrOnce once;                                                                                     \
        hipError_t e = once.ensure((const void *)k_deform32_tps_shared<NTV, DNS, GSS>, 160 * 1024);                  \
        if (e != hipSuccess) return e;                                                                               \
        hipLaunchKernelGGL((k_deform32_tps_shared<NTV, DNS, GSS>), dim3(grid), dim3(kSharedThreads), lds, stream, p, (int)ngroups); \
    }
#define FD_SHARED_KIND(NTV, DNS) { if (gauss) FD_SHARED_CASE(NTV, DNS, true) else FD_SHARED_CASE(NTV, DNS, false) }
#define FD_WIDE_CASE(GSS, NTW, NSL)                                                                                  \
    {                                                                                                                \
        static LdsAttrOnce once;                                                                                     \
        hipError_t e = once.ensure((const void *)k_deform32_tps_shared_wide<kWideDefaultVar, GSS, NTW, NSL>, 160 * 1024); \
        if (e != hipSuccess) return e;                                                                               \
        hipLaunchKernelGGL((k_deform32_tps_shared_wide<kWideDefaultVar, GSS, NTW, NSL>), dim3(grid), dim3(64 * kWideWaves), lds, stream, p, (int)ngroups); \
    }
#define FD_WIDE_KIND(NTW, NSL) { if (gauss) FD_WIDE_CASE(true, NTW, NSL) else FD_WIDE_CASE(false, NTW, NSL) }
#define FD_W1_CASE(GSS, NTW, NSL, WVS)                                                                               \
    {                                                                                                                \
        static LdsAttrOnce once;                                                                                     \
        hipError_t e = once.ensure((const void *)k_deform32_shared_w1<GSS, NTW, NSL, WVS>, 160 * 1024);              \
        if (e != hipSuccess) return e;                                                                               \
        hipLaunchKernelGGL((k_deform32_shared_w1<GSS, NTW, NSL, WVS>), dim3(grid), dim3(64 * WVS), lds, stream, p, (int)ngroups); \
    }
#ifdef FD_TUNING
#define FD_W1_WV(GSS, NTW, NSL) { if (w1waves == 8) FD_W1_CASE(GSS, NTW, NSL, 8) else FD_W1_CASE(GSS, NTW, NSL, kW1Waves) }
#else
#define FD_W1_WV(GSS, NTW, NSL) FD_W1_CASE(GSS, NTW, NSL, kW1Waves)
#endif
#define FD_W1_KIND(NTW, NSL) { if (gauss) FD_W1_WV(true, NTW, NSL) else FD_W1_WV(false, NTW, NSL) }
    if (w1) {
        if (wslot == 20) FD_W1_KIND(2, 20)
        else if (wslot == 24) FD_W1_KIND(3, 24)
        else if (wslot == 28) FD_W1_KIND(3, 28)
        else if (wslot == 32) FD_W1_KIND(3, 32)
        else return hipErrorInvalidValue;
    } else if (wide) {
        if (wslot == 20) FD_WIDE_KIND(2, 20)
        else if (wslot == 24) FD_WIDE_KIND(3, 24)
        else if (wslot == 28) FD_WIDE_KIND(3, 28)
        else if (wslot == 32) FD_WIDE_KIND(3, 32)
        else return hipErrorInvalidValue;
    } else if (dense) {
        if (nT == 3) FD_SHARED_KIND(3, true)
        else return hipErrorInvalidValue;
    } else {
        if (nT == 1) FD_SHARED_KIND(1, false)
        else if (nT == 2) FD_SHARED_KIND(2, false)
        else if (nT == 3) FD_SHARED_KIND(3, false)
        else return hipErrorInvalidValue;
    }
#undef FD_SHARED_KIND
#undef FD_SHARED_CASE
#undef FD_WIDE_CASE
#undef FD_WIDE_KIND
#undef FD_W1_CASE
#undef FD_W1_WV
#undef FD_W1_KIND
#if defined(FD_SHARED_STAMPS_BUILD) || defined(FD_TUNING)
    if (want_stamps && d_stamps) {
        static unsigned long long h[kStampWords];
        (void)hipStreamSynchronize(stream);         // (a non-blocking stream: the copy below does not wait for it by itself)
        if (hipMemcpy(h, d_stamps, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
            if (w1) {
                fprintf(stderr, "[w1 stamps, shader cycles per wave of workgroup 0: load+poly | K loop | epilogue | units | whole (counts, 100 MHz ticks)]\n");
                for (int w = 0; w < kW1Waves; ++w)
                    fprintf(stderr, "   wave %2d: %8llu %8llu %8llu  units %llu  whole %llu counts in %llu ticks = %.3f GHz\n", w, h[w * 8], h[w * 8 + 1], h[w * 8 + 2], h[w * 8 + 3],
                            h[w * 8 + 4], h[w * 8 + 5], h[w * 8 + 5] ? (double)h[w * 8 + 4] / (double)h[w * 8 + 5] * 0.1 : 0.0);
            } else if (wide) {
                // first tick anywhere to every workgroup's first and last tick: who starts late, who finishes late (10 ns units)
                unsigned long long t0 = ~0ull, t1 = 0;
                const unsigned sgrid = grid < kMaxCUs ? grid : kMaxCUs;
                for (unsigned b = 0; b < sgrid * 8; ++b) if (h[64 + 2 * b]) { t0 = h[64 + 2 * b] < t0 ? h[64 + 2 * b] : t0; t1 = h[65 + 2 * b] > t1 ? h[65 + 2 * b] : t1; }
                fprintf(stderr, "[shared stamps: %u workgroups, first entry to last exit %.1f us; per workgroup (entry, exit) in us after the first entry:]\n", grid, (t1 - t0) * 0.01);
                for (unsigned b = 0; b < sgrid; ++b) {
                    unsigned long long a = ~0ull, z = 0;
                    for (int w = 0; w < 8; ++w) if (h[64 + 2 * (b * 8 + w)]) { a = h[64 + 2 * (b * 8 + w)] < a ? h[64 + 2 * (b * 8 + w)] : a; z = h[65 + 2 * (b * 8 + w)] > z ? h[65 + 2 * (b * 8 + w)] : z; }
                    fprintf(stderr, "%s%5.1f-%5.1f", b % 8 ? "  " : "\n   ", (a - t0) * 0.01, (z - t0) * 0.01);
                }
                fprintf(stderr, "\n");
            }
            if (!w1) fprintf(stderr, "[shared stamps, shader cycles per wave of workgroup 0: load+poly | K loop | transposes | per-vertex + frames]\n");
            for (int w = 0; w < 8 && !w1; ++w)
                fprintf(stderr, "   wave %d: %8llu %8llu %8llu %8llu   (whole: %llu counts in %llu reference ticks)\n", w, h[w * 8], h[w * 8 + 1], h[w * 8 + 2], h[w * 8 + 3],
                        h[w * 8 + 4], h[w * 8 + 5]);
        }
    }
#endif
    return hipGetLastError();
}

size_t shared_wtile_bytes(int Mpad, int nF)
{
    const int nkb = (Mpad / 16 + 1) / 2, nT = shared_tiles(nF);
    // weight tiles + polynomial tiles + the rest rig's centre tiles (2 nkb) and normalisation (16 B)
    return (size_t)nkb * nT * 128 * 16 + (size_t)nT * 64 * 16 + (size_t)2 * nkb * sizeof(MfmaTileH) + 16;
}
// the kernel launch_deform_shared picks (same decisions as above)
const char *shared_kernel_name(int Mpad, int nF, int kind)
{
    if (!shared_wide(nF, kind)) return "k_deform32_tps_shared";
    const int nkb = (Mpad / 16 + 1) / 2, wNT = wide_tiles(wide_slots(nF));
    const bool w1 = (kSharedLdsBudget - w1_fixed_lds(wNT)) / ((size_t)1024 + (size_t)wide_w16(wNT) * 16) >= (size_t)nkb;
    return w1 ? "k_deform32_shared_w1" : "k_deform32_tps_shared_wide";
}
size_t shared_frame_bytes(int nF)
{
    return sizeof(SharedFrame) * (size_t)shared_slots(shared_tiles(nF), shared_dense(nF));
}

}  // namespace fd
