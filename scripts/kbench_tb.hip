// kbench_tb.hip -- timing of the two-iterations-per-launch prototype (scripts/fused_tb.hpp) against two launches of the shipped viscous-limit k_fused3d.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I justrelax.jl_amd/csrc -I scripts scripts/kbench_tb.hip -o scripts/kbench_tb
//   ./scripts/kbench_tb [n=512] [reps=20]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "jrx_internal.hpp"
#include "stokes3d_kernels.hpp"
#include "fused_tb.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_fill(double *p, i64 n, unsigned seed, double lo, double hi, int expo)
{
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x) {
        unsigned long long x = (unsigned long long)t * 6364136223846793005ULL + seed * 1442695040888963407ULL + 1013904223ULL;
        x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
        const double u = (double)(x >> 11) * (1.0 / 9007199254740992.0), v = lo + (hi - lo) * u;
        p[t] = expo ? pow(10.0, v) : v;
    }
}
struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    template <class F> double run(int reps, F f)
    {
        f(); f();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a, 0));
        for (int r = 0; r < reps; r++) f();
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        return ms / reps;
    }
};

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 512, reps = argc > 2 ? atoi(argv[2]) : 20;
    const int nx = n, ny = n, nz = n;
    const double cells = (double)nx * ny * nz;
    jrx_stokes3d_fields f;
    memset(&f, 0, sizeof(f));
    struct Ent { double **p; i64 n; double lo, hi; int expo; };
    const i64 nc = (i64)nx * ny * nz, nvx = (i64)(nx + 1) * (ny + 2) * (nz + 2), nvy = (i64)(nx + 2) * (ny + 1) * (nz + 2),
              nvz = (i64)(nx + 2) * (ny + 2) * (nz + 1), nxy = (i64)(nx + 1) * (ny + 1) * nz, nyz = (i64)nx * (ny + 1) * (nz + 1),
              nxz = (i64)(nx + 1) * ny * (nz + 1);
    std::vector<Ent> ents = {
        {&f.P, nc, -1, 1, 0}, {&f.Vx, nvx, -1, 1, 0}, {&f.Vy, nvy, -1, 1, 0}, {&f.Vz, nvz, -1, 1, 0},
        {&f.txx, nc, -1, 1, 0}, {&f.tyy, nc, -1, 1, 0}, {&f.tzz, nc, -1, 1, 0}, {&f.tyz, nyz, -1, 1, 0}, {&f.txz, nxz, -1, 1, 0}, {&f.txy, nxy, -1, 1, 0},
        {&f.eta, nc, -3, 0, 1}, {&f.fx, nc, -1, 1, 0}, {&f.fy, nc, -1, 1, 0}, {&f.fz, nc, -1, 1, 0}};
    unsigned seed = 1;
    for (auto &e : ents) {
        CK(hipMalloc(e.p, e.n * sizeof(double)));
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, *e.p, e.n, seed++, e.lo, e.hi, e.expo);
    }
    double *etatau;
    CK(hipMalloc(&etatau, nc * sizeof(double)));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, etatau, nc, 99u, 0.5, 1.5, 0);
    SweepArgs a;
    a.f = f; a.etatau = etatau; a._dx = 51.2; a._dy = 49.0; a._dz = 47.5; a.dt = INFINITY; a.r = 0.7; a.theta_dtau = 191.3; a.eta_dtau = 0.0119;
    a.L = make_lay(nx, ny, nz);
    a.i0 = a.j0 = a.k0 = 0;
    Out10 dst;
    const i64 dn[10] = {nc, nc, nc, nc, nyz, nxz, nxy, nvx, nvy, nvz};
    double **dp[10] = {&dst.P, &dst.txx, &dst.tyy, &dst.tzz, &dst.tyz, &dst.txz, &dst.txy, &dst.Vx, &dst.Vy, &dst.Vz};
    for (int q = 0; q < 10; q++) {
        CK(hipMalloc(dp[q], dn[q] * sizeof(double)));
        CK(hipMemset(*dp[q], 0, dn[q] * sizeof(double)));
    }
    a.o = dst;
    CK(hipDeviceSynchronize());
    FusedBC bc;
    memset(&bc, 0, sizeof(bc));
    bc.fsL = bc.fsF = bc.fsK0 = 1;
    Timer T;
    printf("kbench_tb n=%d reps=%d\n", n, reps);
    double t1 = 0.0;
    for (int rep = 0; rep < 2; rep++) {
        {
            constexpr int TX = 64, TY = 4, KZ = 8;
            const int ntx = (nx + TX - 3) / (TX - 2), nty = (ny + TY - 2) / (TY - 1), ntz = (nz + KZ - 1) / KZ;
            t1 = T.run(reps, [&] { hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, 4, 1, false, 1, false, true, 3, 1, 0, true>), dim3(ntx * nty * ntz), dim3(TX * TY), 0, 0, a, bc, ntx, nty, 0, 0, 0); });
            printf("shipped k_fused3d<64,4,8,VISC>: one iteration per launch        %8.3f ms per launch = %8.3f ms per iteration  (%.0f it/s)\n", t1, t1, 1e3 / t1);
        }
#define TB(TY, KZ, XG)                                                                                                                                    \
        {                                                                                                                                                 \
            constexpr int TX = 64;                                                                                                                        \
            const int ntx = (nx + TX - 5) / (TX - 4), nty = (ny + TY - 4) / (TY - 3), ntz = (nz + KZ - 1) / KZ;                                           \
            const double t2 = T.run(reps, [&] { hipLaunchKernelGGL((k_fused3d_tb<TX, TY, KZ, XG>), dim3(ntx * nty * ntz), dim3(TX * TY), 0, 0, a, bc, ntx, nty); }); \
            printf("prototype k_fused3d_tb<64,%d,%d,xg%d>: two iterations per launch     %8.3f ms per launch = %8.3f ms per iteration  (%.0f it/s)  x %.3f per iteration\n", TY, KZ, XG, t2, t2 / 2, \
                   2e3 / t2, 2.0 * t1 / t2);                                                                                                              \
            fflush(stdout);                                                                                                                               \
        }
        TB(8, 8, 1) TB(8, 16, 1) TB(8, 8, 0) TB(8, 32, 1) TB(6, 8, 1) TB(12, 8, 1) TB(12, 16, 1)
    }
    printf("done\n");
    return 0;
}
