// fd_capture.hip -- next row N2: the dist2 producer on the device.
//
// Replaces the per-point body of ProximityCapture::capture (reference src/capture.cpp:58-97):
// for every mesh point of an island the squared distance to the closest point of the rest rig's
// surface, found by GU_RayIntersect::minimumPoint there (a CPU search per point), brute force
// here -- the rig is a few hundred triangles.  Finding the islands (nearest mesh point of every
// rig point + edge rings, capture.cpp:101-141) needs the mesh topology and stays on the host; its
// product comes in as a byte mask.  The result feeds fd_deform_dev without leaving the device.
//
// Triangles are staged through LDS in chunks as 16-float records (first vertex, two edges, their
// three dot products, bounding sphere), so a point-triangle test is two dot products plus the
// Voronoi-region selection (Ericson 5.1.5) in fp32 on differences from the triangle's first
// vertex -- the error of d2 stays at a few ulps of the squared lengths involved.  A triangle
// whose bounding sphere cannot beat the wave's current best is skipped by the whole wave.
#include "fd_internal.h"

namespace fd {

namespace {

constexpr int kCapBlock = 256;
constexpr int kTriChunk = 1024;       // 64 KiB of LDS
constexpr int kTriRec = 16;           // floats per staged triangle

__device__ __forceinline__ float dot3(const float u[3], const float v[3])
{
    return fmaf(u[2], v[2], fmaf(u[1], v[1], u[0] * v[0]));
}

// record: a[3], ab[3], ac[3], ab.ab, ab.ac, ac.ac, sphere centre (as offset from a)[3], sphere radius
__device__ __forceinline__ void make_record(const float *t, float *r)
{
    const float a[3] = {t[0], t[1], t[2]};
    const float ab[3] = {t[3] - a[0], t[4] - a[1], t[5] - a[2]};
    const float ac[3] = {t[6] - a[0], t[7] - a[1], t[8] - a[2]};
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2];
    r[3] = ab[0]; r[4] = ab[1]; r[5] = ab[2];
    r[6] = ac[0]; r[7] = ac[1]; r[8] = ac[2];
    r[9] = dot3(ab, ab); r[10] = dot3(ab, ac); r[11] = dot3(ac, ac);
    const float g[3] = {(ab[0] + ac[0]) * (1.f / 3.f), (ab[1] + ac[1]) * (1.f / 3.f), (ab[2] + ac[2]) * (1.f / 3.f)};
    const float ga[3] = {g[0], g[1], g[2]};
    const float gb[3] = {g[0] - ab[0], g[1] - ab[1], g[2] - ab[2]};
    const float gc[3] = {g[0] - ac[0], g[1] - ac[1], g[2] - ac[2]};
    const float rr = fmaxf(dot3(ga, ga), fmaxf(dot3(gb, gb), dot3(gc, gc)));
    r[12] = g[0]; r[13] = g[1]; r[14] = g[2];
    r[15] = sqrtf(rr) * 1.000001f;    // a hair outside: the cull must never drop the true minimum
}

// squared distance from a + ap to the triangle of record r
__device__ __forceinline__ float tri_dist2(const float ap[3], const float *r)
{
    const float ab[3] = {r[3], r[4], r[5]}, ac[3] = {r[6], r[7], r[8]};
    const float d1 = dot3(ab, ap), d2 = dot3(ac, ap);
    const float d3 = d1 - r[9], d4 = d2 - r[10];        // ab.(p - b), ac.(p - b)
    const float d5 = d1 - r[10], d6 = d2 - r[11];       // ab.(p - c), ac.(p - c)
    const float vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
    float v, w;                                          // closest point = a + v ab + w ac
    if (d1 <= 0.f && d2 <= 0.f) { v = 0.f; w = 0.f; }                                   // vertex A
    else if (d3 >= 0.f && d4 <= d3) { v = 1.f; w = 0.f; }                               // vertex B
    else if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) { v = d1 / (d1 - d3); w = 0.f; }      // edge AB
    else if (d6 >= 0.f && d5 <= d6) { v = 0.f; w = 1.f; }                               // vertex C
    else if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) { v = 0.f; w = d2 / (d2 - d6); }      // edge AC
    else if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {                       // edge BC
        w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        v = 1.f - w;
    } else {
        const float den = va + vb + vc;                                                 // face
        if (den == 0.f) { v = 0.f; w = 0.f; }
        else { const float inv = 1.f / den; v = vb * inv; w = vc * inv; }
    }
    const float q[3] = {ap[0] - (ab[0] * v + ac[0] * w), ap[1] - (ab[1] * v + ac[1] * w), ap[2] - (ab[2] * v + ac[2] * w)};
    return dot3(q, q);
}

__global__ __launch_bounds__(kCapBlock) void k_capture_dist2(const float *P, int64_t N, const unsigned char *mask,
                                                              const float *tri, int T, float radius2, int dofalloff,
                                                              float *dist2)
{
    extern __shared__ float s_rec[];          // min(T, kTriChunk) x kTriRec
    const int64_t i = (int64_t)blockIdx.x * kCapBlock + threadIdx.x;
    const bool valid = i < N;
    const bool in_island = valid && (!mask || mask[i]);
    const bool search = in_island && dofalloff;
    float p[3] = {0.f, 0.f, 0.f};
    if (search) { p[0] = P[3 * i]; p[1] = P[3 * i + 1]; p[2] = P[3 * i + 2]; }
    // only distances below radius2 are reported, so that is where the search starts
    float best = search ? radius2 : 0.f;
    const bool any = __syncthreads_or(search ? 1 : 0) != 0;
    if (any) {
        for (int t0 = 0; t0 < T; t0 += kTriChunk) {
            const int nt = T - t0 < kTriChunk ? T - t0 : kTriChunk;
            __syncthreads();
            for (int t = threadIdx.x; t < nt; t += kCapBlock) make_record(tri + (size_t)9 * (t0 + t), s_rec + kTriRec * t);
            __syncthreads();
            float sb = sqrtf(best);           // refreshed only when best improves
            for (int t = 0; t < nt; ++t) {
                const float *r = s_rec + kTriRec * t;
                const float ap[3] = {p[0] - r[0], p[1] - r[1], p[2] - r[2]};
                const float gp[3] = {ap[0] - r[12], ap[1] - r[13], ap[2] - r[14]};
                // the triangle lies inside its sphere: it can only beat `best` if
                // |p - centre| - radius < sqrt(best)
                const float reach = sb + r[15];
                const bool need = search && dot3(gp, gp) < reach * reach;
                if (__any(need)) {
                    const float d = tri_dist2(ap, r);
                    if (need && d < best) { best = d; sb = sqrtf(d); }
                }
            }
        }
    }
    if (!valid) return;
    float out = 0.f;                                      // outside every island, or falloff off (:31, :71-75)
    if (search) out = (T > 0 && best < radius2) ? best : -1.f;   // :76-88
    dist2[i] = out;
}

// ---- islands: ProximityCapture::findIslands (reference src/capture.cpp:101-141) ---------------
// one workgroup per rig point: nearest mesh point (ties to the lower index), written as level 1
__global__ __launch_bounds__(kCapBlock) void k_islands_seed(const float *P, int64_t N, const float *rig, unsigned char *level)
{
    __shared__ float s_d[kCapBlock / 64];
    __shared__ long long s_i[kCapBlock / 64];
    const float a[3] = {rig[3 * blockIdx.x], rig[3 * blockIdx.x + 1], rig[3 * blockIdx.x + 2]};
    float best = INFINITY;
    long long bi = 0x7fffffffffffffffll;
    for (int64_t i = threadIdx.x; i < N; i += kCapBlock) {
        const float d[3] = {P[3 * i] - a[0], P[3 * i + 1] - a[1], P[3 * i + 2] - a[2]};
        const float dd = dot3(d, d);
        if (dd < best) { best = dd; bi = i; }          // increasing i: the first minimum of this thread
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const float od = __shfl_xor(best, off);
        const long long oi = __shfl_xor(bi, off);
        if (od < best || (od == best && oi < bi)) { best = od; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { s_d[threadIdx.x >> 6] = best; s_i[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kCapBlock / 64; ++w)
            if (s_d[w] < best || (s_d[w] == best && s_i[w] < bi)) { best = s_d[w]; bi = s_i[w]; }
        if (bi < N) level[bi] = 1;
    }
}

// one breadth-first level: every point at `cur` marks its unreached neighbours cur + 1
// (several parents may write the same value into one neighbour: harmless)
__global__ __launch_bounds__(kCapBlock) void k_islands_expand(unsigned char *level, int64_t N, const int64_t *offsets,
                                                               const int *neighbours, int cur)
{
    const int64_t v = (int64_t)blockIdx.x * kCapBlock + threadIdx.x;
    if (v >= N || level[v] != (unsigned char)cur) return;
    for (int64_t e = offsets[v]; e < offsets[v + 1]; ++e) {
        const int u = neighbours[e];
        if (level[u] == 0) level[u] = (unsigned char)(cur + 1);
    }
}

__global__ __launch_bounds__(kCapBlock) void k_islands_finish(unsigned char *level, int64_t N)
{
    const int64_t v = (int64_t)blockIdx.x * kCapBlock + threadIdx.x;
    if (v < N) level[v] = level[v] ? 1 : 0;
}

}  // namespace

// d_mask doubles as the level array while the rings grow (level + 1; 0 = not reached)
hipError_t launch_capture_islands(const float *d_P, int64_t N, const int64_t *d_offsets, const int *d_neighbours,
                                  const float *d_rig, int M, int max_edges, unsigned char *d_mask, hipStream_t stream)
{
    if (N <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(d_mask, 0, (size_t)N, stream);
    if (e != hipSuccess) return e;
    if (M <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((N + kCapBlock - 1) / kCapBlock);
    hipLaunchKernelGGL(k_islands_seed, dim3((unsigned)M), dim3(kCapBlock), 0, stream, d_P, N, d_rig, d_mask);
    if (max_edges > 250) max_edges = 250;         // levels live in a byte
    for (int cur = 1; cur <= max_edges; ++cur)
        hipLaunchKernelGGL(k_islands_expand, dim3(grid), dim3(kCapBlock), 0, stream, d_mask, N, d_offsets, d_neighbours, cur);
    hipLaunchKernelGGL(k_islands_finish, dim3(grid), dim3(kCapBlock), 0, stream, d_mask, N);
    return hipGetLastError();
}

hipError_t launch_capture_dist2(const float *d_P, int64_t N, const unsigned char *d_mask, const float *d_tri, int T,
                                float radius2, int dofalloff, float *d_dist2, hipStream_t stream)
{
    if (N <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((N + kCapBlock - 1) / kCapBlock);
    const int nt = T < kTriChunk ? (T > 0 ? T : 1) : kTriChunk;
    // 64 KiB of dynamic LDS has to be requested once per device
    static unsigned long long attr_devices = 0ull;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 64 || !(attr_devices & (1ull << dev))) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_capture_dist2), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(sizeof(float) * kTriRec * kTriChunk));
        if (dev < 64) attr_devices |= 1ull << dev;
    }
    hipLaunchKernelGGL(k_capture_dist2, dim3(grid), dim3(kCapBlock), sizeof(float) * kTriRec * (size_t)nt, stream, d_P, N, d_mask,
                       d_tri, T, radius2, dofalloff, d_dist2);
    return hipGetLastError();
}

}  // namespace fd
