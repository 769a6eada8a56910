// panel_bench.hip -- diagnostic: where does one LU panel (32 columns) spend its cycles?
// Includes the product kernels with FD_PANEL_STAMPS; prints per-phase shader cycles per column.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include tools/panel_bench.hip facedeform_amd/csrc/fd_nullspace.hip -o tools/panel_bench
//        (fd_build.hip dispatches to the null-space solver, so its translation unit comes along)
#define FD_PANEL_STAMPS 1
#include "../facedeform_amd/csrc/fd_build.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>
using namespace fd;
int main()
{
    const int n = 260, npad = 288, lda = npad, ncols = npad + 16 + 16;
    std::vector<double> A((size_t)lda * ncols, 0.0);
    srand(3);
    for (int j = 0; j < npad; ++j)
        for (int i = 0; i < npad; ++i)
            A[(size_t)j * lda + i] = (i < n && j < n) ? (double)rand() / RAND_MAX - 0.5 : (i == j ? 1.0 : 0.0);
    double *dA; int *dipiv, *dmoves; DevModel *dm; BatchSlot *dslot;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dipiv, npad * 4); hipMalloc(&dmoves, kMovesStride * lu_step_capacity(npad) * 4);
    hipMalloc(&dm, sizeof(DevModel)); hipMalloc(&dslot, sizeof(BatchSlot));
    { BatchSlot hs{}; hs.A = dA; hs.ipiv = dipiv; hs.moves = dmoves; hs.model = dm; hipMemcpy(dslot, &hs, sizeof(hs), hipMemcpyHostToDevice); }
    DevModel hm{}; { const double amax = 0.5; unsigned long long bits; __builtin_memcpy(&bits, &amax, 8); hm.amax_bits = bits; } hm.pivmin_bits = 0x7FF0000000000000ull;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {320, 256, 128, 64}) {
        const int k0 = npad - threads;   // panel with `threads` rows
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
            hipMemcpy(dm, &hm, sizeof(hm), hipMemcpyHostToDevice);
            hipEventRecord(e0);
            hipLaunchKernelGGL((k_lu_panel<32, 512>), dim3(1), dim3(threads), 0, 0, dslot, lda, npad, n, k0 < 0 ? 0 : k0, 0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        unsigned long long st[16];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_panel_stamps), sizeof(st));
        const char *names[10] = {"candidate", "wave max+ballot", "LDS row writes", "barrier", "key+exchange", "prow reads+rcp",
                                 "update fma", "rotate", "(loop exit)", "write-back"};
        printf("rows %d: kernel %.1f us;", threads, best * 1e3);
        unsigned long long tot = 0; for (int q = 0; q < 10; ++q) tot += st[q];
        printf(" stamped total %llu cycles; per column:", tot);
        for (int q = 0; q < 8; ++q) printf(" %s %.0f |", names[q], st[q] / 32.0);
        printf(" ; once: exit %llu write-back %llu\n", st[8], st[9]);
    }
    return 0;
}
