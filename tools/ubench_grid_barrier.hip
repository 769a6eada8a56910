// ubench_grid_barrier.hip -- what a dependent step costs on this device: a kernel boundary against an in-kernel barrier over
// all workgroups of a persistent grid (agent-scope release, one atomic, a bounded spin, agent-scope acquire), each step writing
// 8 KB per workgroup that ANOTHER workgroup reads in the next step (so the data has to cross the XCDs' L2 caches).
// Decides whether a persistent grid with flags would take the C3 Cholesky's 64 dependent launches anywhere.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_grid_barrier tools/ubench_grid_barrier.hip && tools/ubench_grid_barrier
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

constexpr int kDoubles = 1024;      // 8 KB per workgroup and step

__device__ __forceinline__ void step_body(double *buf, int G, int it, int bid, int tid, unsigned long long *bad)
{
    // read what workgroup (bid + 37) % G wrote in the previous step, write my own for this one
    const int src = (bid + 37) % G;
    double s = 0.0;
    if (it > 0) {
        for (int e = tid; e < kDoubles; e += 256) s += buf[((size_t)((it - 1) & 1) * G + src) * kDoubles + e];
        const double want = (double)(it - 1) * 1000.0 + src;
        if (tid == 0 && buf[((size_t)((it - 1) & 1) * G + src) * kDoubles] != want) atomicAdd(bad, 1ull);
    }
    for (int e = tid; e < kDoubles; e += 256) buf[((size_t)(it & 1) * G + bid) * kDoubles + e] = (double)it * 1000.0 + bid + (e ? s * 0.0 : 0.0);
}

__global__ __launch_bounds__(256) void k_persistent(double *buf, unsigned *ctr, unsigned long long *bad, int G, int n)
{
    const int bid = blockIdx.x, tid = threadIdx.x;
    for (int it = 0; it < n; ++it) {
        step_body(buf, G, it, bid, tid, bad);
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(it + 1) * (unsigned)G;
            int spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
            if (spins >= (1 << 22)) atomicAdd(bad, 1ull << 32);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_step(double *buf, unsigned long long *bad, int G, int it) { step_body(buf, G, it, blockIdx.x, threadIdx.x, bad); }

int main()
{
    double *buf; unsigned *ctr; unsigned long long *bad;
    const int n = 2000;
    if (hipMalloc(&buf, sizeof(double) * 2 * 1024 * kDoubles) != hipSuccess) { printf("no device\n"); return 1; }
    (void)hipMalloc(&ctr, 4); (void)hipMalloc(&bad, 8);
    for (int G : {8, 32, 64, 128, 256}) {
        (void)hipMemset(ctr, 0, 4); (void)hipMemset(bad, 0, 8);
        (void)hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_persistent, dim3(G), dim3(256), 0, 0, buf, ctr, bad, G, n);
        (void)hipDeviceSynchronize();
        const double tp = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        unsigned long long hb = 0; (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
        (void)hipMemset(bad, 0, 8);
        (void)hipDeviceSynchronize();
        t0 = std::chrono::steady_clock::now();
        for (int it = 0; it < n; ++it) hipLaunchKernelGGL(k_step, dim3(G), dim3(256), 0, 0, buf, bad, G, it);
        (void)hipDeviceSynchronize();
        const double tl = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        unsigned long long hb2 = 0; (void)hipMemcpy(&hb2, bad, 8, hipMemcpyDeviceToHost);
        printf("%3d workgroups: in-kernel barrier %6.2f us per step (stale reads %llu, timeouts %llu)   kernel boundary %6.2f us per step (stale reads %llu)\n",
               G, tp / n * 1e6, hb & 0xffffffffull, hb >> 32, tl / n * 1e6, hb2);
    }
    return 0;
}
