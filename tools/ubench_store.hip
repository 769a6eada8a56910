// ubench_store.hip -- what the device's write path gives the shared-rig evaluation's OUTPUT pattern, without any arithmetic.
// The launch of DESIGN.md 4.1d writes 32 frames x (12 MB of float[3] positions + 4 MB of fd_falloff) = 512 MB per 1M vertices
// and reads 12 MB; with its K loop switched off it still takes 135 us (3.9 TB/s) where plain streaming stores reach 6 TB/s
// (MI355X_MICROARCH.md).  This program writes the same bytes to the same addresses in several shapes to see which part of the
// pattern costs: the run of contiguous bytes a wave writes per frame (384 B = 32 vertices), the 12-byte-per-lane store, the
// non-temporal hint, the persistent grid, the waves per workgroup.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_store tools/ubench_store.hip && tools/ubench_store
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int kN = 1000000, kF = 32;
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f32x3 f32x3_a4 __attribute__((aligned(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// ---- (a) the ideal: every lane 16 bytes, consecutive lanes consecutive addresses, grid-stride over the whole output
template <bool NT>
__global__ __launch_bounds__(256) void k_stream16(f32x4 *out, size_t n16)
{
    const f32x4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}

// ---- (b) the evaluation's pattern.  A wave owns a unit of 32 * TILES vertices; per frame it writes 384 * TILES contiguous bytes of
// positions and 128 * TILES of fd_falloff.
//   SHAPE 0: one 12-byte store per lane; lane half h -> frame 2 lf + h, lane & 31 -> vertex (what k_deform32_shared_w1 issues);
//            fd_falloff as 16-byte stores, eight frames per instruction (8 lanes x 4 vertices per frame).
//   SHAPE 1: the same bytes of positions as 16-byte stores: 24 lanes per 384-byte run, two runs (frames) per instruction, lanes
//            24..31 and 56..63 idle (what a transposition through LDS would issue).
//   SHAPE 2: positions AND fd_falloff of a frame interleaved per 32 vertices?  no -- the layout is the reference's; not offered.
// persist: the grid's workgroups x WAVES waves walk the units round-robin (as the kernel's do); otherwise one unit per wave.
template <int SHAPE, bool NT, int TILES, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_pattern(float *pos, float *fall, int n, int frames, int units, int persist)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, l32 = lane & 31;
    const int first = blockIdx.x * WAVES + wave;
    const int stride = persist ? gridDim.x * WAVES : units;
    for (int u = first; u < units; u += stride) {
        for (int t = 0; t < TILES; ++t) {
            const int vt = (u * TILES + t) * 32;
            if (vt >= n) break;
            const f32x3 val3 = {(float)u, (float)lane, (float)t};
            const f32x4 val4 = {(float)u, (float)lane, (float)t, 1.f};
#pragma unroll 4
            for (int lf = 0; lf < frames / 2; ++lf) {
                const int f = 2 * lf + h;
                if (lf % 4 == 0) {
                    // fd_falloff of frames 2 lf .. 2 lf + 7: lane >> 3 -> frame, lane & 7 -> four vertices
                    float *fd = fall + (size_t)(2 * lf + (lane >> 3)) * n + vt + 4 * (lane & 7);
                    if (NT) __builtin_nontemporal_store(val4, (f32x4 *)fd); else *(f32x4 *)fd = val4;
                }
                if (SHAPE == 0) {
                    float *dst = pos + ((size_t)f * n + vt + l32) * 3;
                    if (NT) __builtin_nontemporal_store(val3, (f32x3_a4 *)dst); else *(f32x3_a4 *)dst = val3;
                } else {
                    if (l32 < 24) {
                        float *dst = pos + ((size_t)f * n + vt) * 3 + 4 * l32;
                        if (NT) __builtin_nontemporal_store(val4, (f32x4 *)dst); else *(f32x4 *)dst = val4;
                    }
                }
            }
        }
    }
}

// ---- (c) frame-major inside a workgroup: the WAVES waves of a workgroup own WAVES consecutive units (as the kernel's ticket
// hands them out), but the stores are dealt so that ONE wave writes one frame's run for the whole workgroup: WAVES * 384
// contiguous bytes per (workgroup, frame) as 16-byte stores.  What an epilogue staged through LDS could issue.
template <bool NT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_wg_runs(float *pos, float *fall, int n, int frames, int rounds)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kVerts = WAVES * 32;                 // vertices of a workgroup round
    constexpr int kRun16 = kVerts * 12 / 16;           // 16-byte pieces of a frame's position run
    constexpr int kFall16 = kVerts * 4 / 16;
    for (int r = blockIdx.x; r < rounds; r += gridDim.x) {
        const int v0 = r * kVerts;
        if (v0 + kVerts > n) break;
        const f32x4 val4 = {(float)r, (float)lane, 0.f, 1.f};
        for (int f = wave; f < frames; f += WAVES) {
            float *dst = pos + ((size_t)f * n + v0) * 3;
            for (int q = lane; q < kRun16; q += 64) {
                if (NT) __builtin_nontemporal_store(val4, (f32x4 *)dst + q); else ((f32x4 *)dst)[q] = val4;
            }
            float *fd = fall + (size_t)f * n + v0;
            for (int q = lane; q < kFall16; q += 64) {
                if (NT) __builtin_nontemporal_store(val4, (f32x4 *)fd + q); else ((f32x4 *)fd)[q] = val4;
            }
        }
    }
}

template <typename L>
static double time_us(L launch, int reps = 30)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch();
    (void)hipDeviceSynchronize();
    std::vector<float> ms(reps);
    for (int i = 0; i < reps; ++i) {
        (void)hipEventRecord(a, 0);
        launch();
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms[i], a, b);
    }
    std::sort(ms.begin(), ms.end());
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return ms[reps / 2] * 1e3;
}

template <int SHAPE, bool NT, int TILES, int WAVES>
static void run_pattern(float *pos, float *fall, int grid_persist, const char *tag)
{
    const int units = (kN + 32 * TILES - 1) / (32 * TILES);
    const double bytes = 16.0 * kN * kF;
    for (int persist : {1, 0}) {
        const int grid = persist ? grid_persist : (units + WAVES - 1) / WAVES;
        const double us = time_us([&] { hipLaunchKernelGGL((k_pattern<SHAPE, NT, TILES, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, pos, fall, kN, kF, units, persist); });
        printf("%-46s %s  %3d vertices per wave, %2d waves, %s grid %6d: %7.1f us = %5.2f TB/s\n", tag, NT ? "nt   " : "plain", 32 * TILES, WAVES,
               persist ? "persistent" : "one-shot  ", grid, us, bytes / us * 1e-6);
    }
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs; output of one launch: %d frames x %d vertices x 16 B = %.0f MB\n", prop.gcnArchName, cus, kF, kN, 16.0 * kN * kF * 1e-6);
    float *pos, *fall;
    CK(hipMalloc(&pos, (size_t)kF * kN * 12 + 4096));
    CK(hipMalloc(&fall, (size_t)kF * kN * 4 + 4096));
    const double bytes = 16.0 * kN * kF;
    {
        const double us = time_us([&] { (void)hipMemsetAsync(pos, 0, (size_t)kF * kN * 12, 0); (void)hipMemsetAsync(fall, 0, (size_t)kF * kN * 4, 0); });
        printf("%-46s %7.1f us = %5.2f TB/s\n", "hipMemsetAsync of both arrays", us, bytes / us * 1e-6);
    }
    for (int grid : {cus * 8, cus * 32}) {
        double us = time_us([&] { hipLaunchKernelGGL(k_stream16<false>, dim3(grid), dim3(256), 0, 0, (f32x4 *)pos, (size_t)kF * kN * 12 / 16);
                                  hipLaunchKernelGGL(k_stream16<false>, dim3(grid), dim3(256), 0, 0, (f32x4 *)fall, (size_t)kF * kN * 4 / 16); });
        printf("%-46s plain grid %6d: %7.1f us = %5.2f TB/s\n", "streaming 16 B per lane (two launches)", grid, us, bytes / us * 1e-6);
        us = time_us([&] { hipLaunchKernelGGL(k_stream16<true>, dim3(grid), dim3(256), 0, 0, (f32x4 *)pos, (size_t)kF * kN * 12 / 16);
                           hipLaunchKernelGGL(k_stream16<true>, dim3(grid), dim3(256), 0, 0, (f32x4 *)fall, (size_t)kF * kN * 4 / 16); });
        printf("%-46s nt    grid %6d: %7.1f us = %5.2f TB/s\n", "streaming 16 B per lane (two launches)", grid, us, bytes / us * 1e-6);
    }
    run_pattern<0, true, 1, 12>(pos, fall, cus, "12 B per lane, 384-byte runs (the kernel's)");
    run_pattern<0, false, 1, 12>(pos, fall, cus, "12 B per lane, 384-byte runs (the kernel's)");
    run_pattern<0, true, 1, 8>(pos, fall, cus, "12 B per lane, 384-byte runs");
    run_pattern<0, true, 1, 4>(pos, fall, cus, "12 B per lane, 384-byte runs");
    run_pattern<0, true, 2, 8>(pos, fall, cus, "12 B per lane, 2 x 384-byte runs per frame");
    run_pattern<0, true, 4, 8>(pos, fall, cus, "12 B per lane, 4 x 384-byte runs per frame");
    run_pattern<1, true, 1, 12>(pos, fall, cus, "16 B per lane, 384-byte runs");
    run_pattern<1, false, 1, 12>(pos, fall, cus, "16 B per lane, 384-byte runs");
    run_pattern<1, true, 4, 8>(pos, fall, cus, "16 B per lane, 4 x 384-byte runs per frame");
    for (int grid : {cus, cus * 4}) {
        double us = time_us([&] { hipLaunchKernelGGL((k_wg_runs<true, 12>), dim3(grid), dim3(12 * 64), 0, 0, pos, fall, kN, kF, kN / (12 * 32)); });
        printf("%-46s nt    grid %6d: %7.1f us = %5.2f TB/s\n", "workgroup runs: 4 608 B per (workgroup, frame)", grid, us, bytes / us * 1e-6);
        us = time_us([&] { hipLaunchKernelGGL((k_wg_runs<false, 12>), dim3(grid), dim3(12 * 64), 0, 0, pos, fall, kN, kF, kN / (12 * 32)); });
        printf("%-46s plain grid %6d: %7.1f us = %5.2f TB/s\n", "workgroup runs: 4 608 B per (workgroup, frame)", grid, us, bytes / us * 1e-6);
        us = time_us([&] { hipLaunchKernelGGL((k_wg_runs<true, 16>), dim3(grid), dim3(16 * 64), 0, 0, pos, fall, kN, kF, kN / (16 * 32)); });
        printf("%-46s nt    grid %6d: %7.1f us = %5.2f TB/s\n", "workgroup runs: 6 144 B per (workgroup, frame)", grid, us, bytes / us * 1e-6);
    }
    (void)hipFree(pos); (void)hipFree(fall);
    return 0;
}
