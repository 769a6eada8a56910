// fd_pack.h -- solution -> weights, evaluation records, centre tiles, status: device bodies shared by
// fd_build.hip (kernels k_pack / k_pack_tiles) and fd_nullspace.hip (last phase of the one-launch build).
// Not part of the ABI.
#pragma once

#include "fd_internal.h"

namespace fd {
namespace packing {

// ---- pack: solution -> weights, evaluation records, status -----------------------
// from_w != 0: the weights are already in W (fd_import_model); only the records,
// the affine part and the status are produced.
//
// The fp32 evaluation runs in normalised coordinates x' = (x - x0) / s, s a power of two:
// thin-plate's log makes fp32 accuracy depend on the length unit (at scale 100 the direct
// form degrades to 2-3e-5), and a far-away origin costs bits in every kernel.  Homogeneous
// kernels only rescale their weights; thin-plate also needs
//     sum_j w_j (1/2) d^2 ln d^2 = s^2 sum_j w_j (1/2) d'^2 ln d'^2 + (s^2 ln s) sum_j w_j d'^2
// whose last sum is the quadratic |x'|^2 m0 - 2 x'.m1 + m2 of three moments of the weights.
// x0 = 0 and s = 1 for unit-scale data near the origin, which leaves those results unchanged.
__device__ __forceinline__ double block_sum(double v, double *scratch, int tid)
{
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((tid & 63) == 0) scratch[tid >> 6] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}
__device__ __forceinline__ double block_max(double v, double *scratch, int tid)
{
    for (int off = 32; off >= 1; off >>= 1) { const double o = __shfl_xor(v, off); v = o > v ? o : v; }
    __syncthreads();
    if ((tid & 63) == 0) scratch[tid >> 6] = v;
    __syncthreads();
    const double a = scratch[0] > scratch[1] ? scratch[0] : scratch[1];
    const double b = scratch[2] > scratch[3] ? scratch[2] : scratch[3];
    return a > b ? a : b;
}

// N sums (or maxima) with ONE pair of barriers: every value goes down its wave by the same butterfly and across the four
// waves in the same order as block_sum / block_max take it -- bit-identical to N calls of those, at a fifteenth of the
// barriers (the packing of the one-launch builds sits on their critical path).
template <int N, bool MAX>
__device__ __forceinline__ void block_reduce_many(double (&v)[N], double *scratch /* 4 N */, int tid)
{
    // (tried in round 4: the six butterfly steps outermost with all N exchanges of a step requested together, swizzles and quad
    //  permutations instead of permutes -- the arrays went to scratch memory and the phase doubled; the values' chains below are
    //  independent and the scheduler interleaves them as they stand)
#pragma unroll
    for (int q = 0; q < N; ++q)
        for (int off = 32; off >= 1; off >>= 1) {
            const double o = __shfl_xor(v[q], off);
            if (MAX) v[q] = o > v[q] ? o : v[q]; else v[q] += o;
        }
    __syncthreads();
    if ((tid & 63) == 0) {
#pragma unroll
        for (int q = 0; q < N; ++q) scratch[(tid >> 6) * N + q] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < N; ++q) {
        if (MAX) {
            const double a = scratch[q] > scratch[N + q] ? scratch[q] : scratch[N + q];
            const double b = scratch[2 * N + q] > scratch[3 * N + q] ? scratch[2 * N + q] : scratch[3 * N + q];
            v[q] = a > b ? a : b;
        } else {
            v[q] = scratch[q] + scratch[N + q] + scratch[2 * N + q] + scratch[3 * N + q];
        }
    }
}

// Where the packing reads its inputs.  PackGlobal: the build's arrays in global memory (every launch-chain build, imports).
// PackLds (fd_build_reg.hip): the register build's workgroup has the same doubles in LDS when its last phase starts -- centres,
// solution, polynomial coefficients, flags -- and every read from global memory there is a round trip to L2 on the build's
// critical path (eight of them in a row, 2 000 cycles each, were half of that phase).  Same values, same order of operations:
// the outputs are bit-identical whichever source is read.
struct PackGlobal {
    const double *X, *W, *centres, *radii;
    const float *dl;
    const DevModel *model;
    int npad, from_w;
    __device__ __forceinline__ double centre(int j, int q) const { return centres[3 * j + q]; }
    __device__ __forceinline__ bool has_delta() const { return dl != nullptr; }
    __device__ __forceinline__ double delta(int j, int q) const { return (double)dl[3 * j + q]; }
    __device__ __forceinline__ double weight(int j, int c) const { return from_w ? W[3 * j + c] : X[(size_t)c * npad + j]; }
    __device__ __forceinline__ double affine(int cc, int k, int M, int T) const
    {
        if (from_w) return W[3 * (M + k) + cc];
        return k < T ? X[(size_t)cc * npad + M + k] : 0.0;
    }
    __device__ __forceinline__ double radius(int j) const { return radii[j]; }
    __device__ __forceinline__ int sing_flag() const { return model->sing_flag; }
    __device__ __forceinline__ int dup_flag() const { return model->dup_flag; }
};

// (256 threads; a kernel of its own in fd_build.hip, the last phase of the one-launch build in fd_nullspace.hip)
template <class Src>
__device__ __forceinline__ void pack_body_from(const Src &in, const BatchSlot &slot, int npad, int M, int Mpad, int T, int kind, int from_w, int layers)
{
    double *W = slot.W;
    Rec32 *rec32 = slot.rec32;
    Rec64 *rec64 = slot.rec64;
    DevModel *model = slot.model;
    __shared__ int s_bad;
    __shared__ double s_red[4 * 18];
    const int tid = threadIdx.x;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    const double kLn2 = 0.6931471805599453;
    const double kLog2e = 1.4426950408889634;
    const bool gauss = kind == FD_KERNEL_GAUSSIAN || kind == FD_KERNEL_GAUSSIAN_QNN;

    // ---- normalisation: x0 = centroid of the centres (0 if they already sit around the
    //      origin), s = power of two nearest to their largest distance from x0
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (int j = tid; j < M; j += 256) { sx += in.centre(j, 0); sy += in.centre(j, 1); sz += in.centre(j, 2); }
    double csum[3] = {sx, sy, sz};
    block_reduce_many<3, false>(csum, s_red, tid);
    double cenx = csum[0] / M, ceny = csum[1] / M, cenz = csum[2] / M;
    double r2 = 0.0, r2o = 0.0;
    // (beside the extent: the largest and the smallest |delta_i| of the control table, for the fp32 estimate below;
    //  the smallest as a maximum of its negative)
    const bool dl = in.has_delta();
    double dmax2 = 0.0, ndmin2 = -INFINITY;
    for (int j = tid; j < M; j += 256) {
        const double x = in.centre(j, 0), y = in.centre(j, 1), z = in.centre(j, 2);
        const double d = (x - cenx) * (x - cenx) + (y - ceny) * (y - ceny) + (z - cenz) * (z - cenz);
        r2 = d > r2 ? d : r2;
        const double o = x * x + y * y + z * z;
        r2o = o > r2o ? o : r2o;
        if (dl) {
            const double a = in.delta(j, 0), b = in.delta(j, 1), c2 = in.delta(j, 2);
            const double dd = a * a + b * b + c2 * c2;
            dmax2 = dd > dmax2 ? dd : dmax2;
            ndmin2 = -dd > ndmin2 ? -dd : ndmin2;
        }
    }
    double ext[4] = {r2, r2o, dmax2, ndmin2};
    block_reduce_many<4, true>(ext, s_red, tid);
    // The smallest displacement the tolerance is asked of: the least |delta_i| among the control points that MOVE, i.e. by at
    // least a tenth of the largest one.  A rig with stationary or barely moving control points (most of a face, in most frames;
    // the fringe of any localised deformation) has min |delta_i| = 0 or next to it, and a floor of zero sends every such cook
    // to fp64 although the field near a stationary point is zero to within half an ulp of the position in fp32 too.
    if (dl) {
        const double moving2 = 1e-2 * ext[2];
        double nmin[1] = {-INFINITY};
        for (int j = tid; j < M; j += 256) {
            const double a = in.delta(j, 0), b = in.delta(j, 1), c2 = in.delta(j, 2);
            const double dd = a * a + b * b + c2 * c2;
            if (dd >= moving2 && dd > 0.0 && -dd > nmin[0]) nmin[0] = -dd;
        }
        block_reduce_many<1, true>(nmin, s_red, tid);
        ext[3] = nmin[0] > -INFINITY ? nmin[0] : 0.0;
    }
    const double rad_c = sqrt(ext[0]);
    const double rad_o = sqrt(ext[1]);
    const double cen = sqrt(cenx * cenx + ceny * ceny + cenz * cenz);
    double x0[3] = {0.0, 0.0, 0.0};
    double rad = rad_o;
    if (cen > rad_c * 0.0625) { x0[0] = (double)(float)cenx; x0[1] = (double)(float)ceny; x0[2] = (double)(float)cenz; rad = rad_c; }
    double sc = 1.0;
    if (rad > 0.0 && isfinite(rad)) {
        int e = (int)rint(log2(rad));
        e = e < -100 ? -100 : (e > 100 ? 100 : e);
        sc = ldexp(1.0, e);
    }
    const double inv_s = 1.0 / sc;
    double w32 = 1.0, w64 = 1.0, kappa = 0.0;
    if (kind == FD_KERNEL_THIN_PLATE) { w32 = 0.5 * kLn2 * sc * sc; w64 = 0.5; kappa = sc * sc * log(sc); }
    if (kind == FD_KERNEL_BIHARMONIC) { w32 = -sc; w64 = -1.0; }
    if (kind == FD_KERNEL_CUBIC) { w32 = sc * sc * sc; }

    // ---- records + moments of the weights in normalised coordinates
    bool bad = false;
    float wmax = 0.f;      // largest |weight| of the fp32 records (k_pack_shared* scale a frame's rows by it)
    double m[18];   // per output c: m0, m1x, m1y, m1z, m2; then per output the sum of |w| as the fp32 evaluation carries it
#pragma unroll
    for (int q = 0; q < 18; ++q) m[q] = 0.0;
    for (int j = tid; j < Mpad; j += 256) {
        Rec32 r32 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        Rec64 r64 = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (j < M) {
            const double w[3] = {in.weight(j, 0), in.weight(j, 1), in.weight(j, 2)};
            bad |= !(isfinite(w[0]) && isfinite(w[1]) && isfinite(w[2]));
            if (!from_w) { W[3 * j] = w[0]; W[3 * j + 1] = w[1]; W[3 * j + 2] = w[2]; }
            const double R = in.radius(j);
            r64.cx = in.centre(j, 0); r64.cy = in.centre(j, 1); r64.cz = in.centre(j, 2);
            r64.s = gauss ? -1.0 / (R * R) : 0.0;
            r64.wx = w[0] * w64; r64.wy = w[1] * w64; r64.wz = w[2] * w64;
            const double cn[3] = {(r64.cx - x0[0]) * inv_s, (r64.cy - x0[1]) * inv_s, (r64.cz - x0[2]) * inv_s};
            r32.cx = (float)cn[0]; r32.cy = (float)cn[1]; r32.cz = (float)cn[2];
            r32.s = gauss ? (float)(-kLog2e * sc * sc / (R * R)) : 0.f;
            r32.wx = (float)(w[0] * w32); r32.wy = (float)(w[1] * w32); r32.wz = (float)(w[2] * w32);
            wmax = fmaxf(wmax, fmaxf(fabsf(r32.wx), fmaxf(fabsf(r32.wy), fabsf(r32.wz))));
            const double cc2 = cn[0] * cn[0] + cn[1] * cn[1] + cn[2] * cn[2];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                m[5 * c] += w[c];
                m[5 * c + 1] += w[c] * cn[0];
                m[5 * c + 2] += w[c] * cn[1];
                m[5 * c + 3] += w[c] * cn[2];
                m[5 * c + 4] += w[c] * cc2;
                m[15 + c] += fabs(w[c] * w32);
            }
        }
        // multilayer model: W is layer-major (record l * Mc + c), the evaluation wants the layers of
        // a centre side by side (c * layers + l) so that they can share its distances
        int jo = j;
        if (layers > 1 && j < M) { const int Mc = M / layers; jo = (j % Mc) * layers + j / Mc; }
        rec32[jo] = r32;
        rec64[jo] = r64;
    }
    block_reduce_many<18, false>(m, s_red, tid);
    double wm[1] = {(double)wmax};
    block_reduce_many<1, true>(wm, s_red, tid);

    // ---- affine part: W rows M..M+3 = const, x, y, z (raw coordinates)
    __shared__ double s_aff[12];
    if (tid < 12) {
        const int cc = tid / 4, k = tid % 4;
        const double v = in.affine(cc, k, M, T);
        bad |= !isfinite(v);
        if (!from_w) W[3 * (M + k) + cc] = v;
        model->affine64[tid] = v;
        s_aff[tid] = v;
    }
    if (bad) s_bad = 1;
    __syncthreads();
    // (threads 0..2 sit in one wave: the largest polynomial coefficient joins the largest weight without a barrier)
    float pmax = 0.f;
    if (tid < 3) {
        const int c = tid;
        const double *a = s_aff + 4 * c;   // {v0, Vx, Vy, Vz}
        const float p5[5] = {(float)(a[0] + a[1] * x0[0] + a[2] * x0[1] + a[3] * x0[2] + kappa * m[5 * c + 4]),
                             (float)(a[1] * sc - 2.0 * kappa * m[5 * c + 1]), (float)(a[2] * sc - 2.0 * kappa * m[5 * c + 2]),
                             (float)(a[3] * sc - 2.0 * kappa * m[5 * c + 3]), (float)(kappa * m[5 * c])};
#pragma unroll
        for (int e = 0; e < 5; ++e) { model->poly32[5 * c + e] = p5[e]; pmax = fmaxf(pmax, fabsf(p5[e])); }
        model->norm32[c] = (float)x0[c];
    }
    pmax = fmaxf(pmax, fmaxf(__shfl(pmax, 1), __shfl(pmax, 2)));
    if (tid == 0) {
        model->wmax32 = fmaxf((float)wm[0], pmax);
        // ---- what the fp32 evaluation can be trusted with (fd_report.fp32_error / cancellation / delta_min).  It adds up
        // M terms w_j phi(d_j) whose magnitudes sum to S = sum_j |w_j| max phi over the rig's extent (normalised
        // coordinates: distances up to 2) plus the polynomial; each carries a relative 2^-24, so the displacement comes out
        // with an ABSOLUTE error of the order of 2^-24 S whatever its own size (measured: 0.3 .. 0.4 of that; 2^-25 S is reported).  The deltas that S answers to are of size
        // delta_max (cancellation = S / delta_max, the verdict's figure); the reference's 1e-5 is asked of every vertex
        // against its OWN displacement, and the control table's smallest |delta| says how small those get.
        const double phimax = kind == FD_KERNEL_THIN_PLATE ? 8.0 : (kind == FD_KERNEL_CUBIC ? 8.0 : (kind == FD_KERNEL_BIHARMONIC ? 2.0 : 1.0));
        double S = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double *a = s_aff + 4 * c;
            const double poly = fabs(a[0]) + (fabs(a[1]) + fabs(a[2]) + fabs(a[3])) * (rad_o > 0.0 ? rad_o : 1.0) + 4.0 * fabs(kappa * m[5 * c]);
            const double v = m[15 + c] * phimax + poly;
            S = v > S ? v : S;
        }
        const double dmax = sqrt(ext[2]), dmin = dl ? sqrt(-ext[3]) : 0.0;
        model->fp32_error = S * 2.98023223876953125e-08;       // 2^-25 S: the kernels measure at 0.3 .. 0.4 of the worst case 2^-24 S
        model->cancellation = dmax > 0.0 ? S / dmax : 0.0;
        model->delta_min = dmin;
        model->delta_max = dmax;
        model->extent = rad_o;
        model->norm32[3] = (float)inv_s;
        int tt = 1;
        if (from_w == 1) {
            if (s_bad) tt = -4;
        } else {
            if (in.sing_flag() || s_bad) tt = -4;
            if (in.dup_flag()) tt = -5;
        }
        model->terminationtype = tt;
        if (slot.host_status) *slot.host_status = tt;        // page-locked: the host polls it behind the build's event (fd_capi.hip)
    }
}

__device__ __forceinline__ void pack_body(const BatchSlot &slot, int npad, int M, int Mpad, int T, int kind, int from_w, int layers)
{
    const PackGlobal in{from_w ? nullptr : slot.X, slot.W, slot.centres, slot.radii, from_w ? nullptr : slot.delta, slot.model, npad, from_w};
    pack_body_from(in, slot, npad, M, Mpad, T, kind, from_w, layers);
}

// ---- thin-plate only: centre tiles for the matrix-pipe evaluation ------------------------
// An fp32 value splits EXACTLY into three bf16 pieces hi + mid + lo (8 significant bits each,
// by truncation).  d2 = |x'|^2 - 2 x'.c' + |c'|^2 then runs on the bf16 MFMA to fp32 accuracy:
// per coordinate the six products (hi,hi) (hi,mid) (mid,hi) (mid,mid) (hi,lo) (lo,hi) -- the
// three dropped ones are below 2^-24 relative -- and, in lane group 3, |c'|^2 against 1 and 1
// against |x'|^2 in three slots each.
__device__ __forceinline__ void split3_bf16(float x, unsigned &hi, unsigned &mid, unsigned &lo)
{
    const unsigned u = __float_as_uint(x);
    const float r1 = x - __uint_as_float(u & 0xffff0000u);     // exact
    const unsigned u1 = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);   // exact, at most 8 significant bits left
    hi = u >> 16; mid = u1 >> 16; lo = __float_as_uint(r2) >> 16;
}

// one centre tile (16 records) by one wave
__device__ __forceinline__ void pack_tiles_body(const BatchSlot &slot, int Mpad, int tile, int lane)
{
    const Rec32 *rec32 = slot.rec32;
    MfmaTile *tiles = slot.tiles;
    const int g = lane >> 4, i = lane & 15;
    const Rec32 r = rec32[16 * tile + i];          // Mpad is a multiple of 16; padding records are zero
    unsigned h, m, l;
    unsigned k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (g < 3) {
        const float c = g == 0 ? r.cx : (g == 1 ? r.cy : r.cz);
        split3_bf16(c, h, m, l);
        k[0] = h; k[1] = h; k[2] = m; k[3] = m; k[4] = h; k[5] = l;     // against x: hi mid hi mid lo hi
    } else {
        const float cc = fmaf(r.cz, r.cz, fmaf(r.cy, r.cy, r.cx * r.cx));
        split3_bf16(cc, h, m, l);
        k[0] = h; k[1] = m; k[2] = l;                                   // against 1, 1, 1
        k[3] = 0x3f80; k[4] = 0x3f80; k[5] = 0x3f80;                    // 1 against the pieces of |x'|^2
    }
    MfmaTile &t = tiles[tile];
    t.a[lane][0] = k[0] | (k[1] << 16);
    t.a[lane][1] = k[2] | (k[3] << 16);
    t.a[lane][2] = k[4] | (k[5] << 16);
    t.a[lane][3] = k[6] | (k[7] << 16);
    if (i < 12) {
        // slot i of group g: pair p = i / 6 (rows 2p, 2p+1), output c = (i % 6) / 2, half = i % 2
        const int pair = i / 6, c = (i % 6) / 2, half = i % 2;
        const Rec32 rw = rec32[16 * tile + 4 * g + 2 * pair + half];
        t.w[g][i] = c == 0 ? rw.wx : (c == 1 ? rw.wy : rw.wz);
    }
    if (lane < 16) t.pad[lane] = 0.f;
    (void)Mpad;

    // fp16 form: pieces hi = RN16(v), lo = RN16(v - hi).  Lane group g < 3: {c_hi, c_hi, c_lo, c_lo}
    // against the vertex's {x_hi, x_lo, x_hi, x_lo}; group 3: {|c|^2_hi, |c|^2_lo, 1, 1} against
    // {1, 1, |x|^2_hi, |x|^2_lo}.
    MfmaTileH *tiles16 = slot.tiles16;
    if (tiles16) {
        MfmaTileH &th = tiles16[tile];
        const float v = g < 3 ? (g == 0 ? r.cx : (g == 1 ? r.cy : r.cz)) : fmaf(r.cz, r.cz, fmaf(r.cy, r.cy, r.cx * r.cx));
        const _Float16 vh = (_Float16)v;
        const _Float16 vl = (_Float16)(v - (float)vh);
        const unsigned uh = (unsigned)__builtin_bit_cast(unsigned short, vh), ul = (unsigned)__builtin_bit_cast(unsigned short, vl);
        if (g < 3) { th.a[lane][0] = uh | (uh << 16); th.a[lane][1] = ul | (ul << 16); }
        else { th.a[lane][0] = uh | (ul << 16); th.a[lane][1] = 0x3c003c00u; }
        if (i < 12) th.w[g][i] = t.w[g][i];
        if (lane < 16) th.pad[lane] = 0.f;
    }
}

}  // namespace packing
}  // namespace fd
