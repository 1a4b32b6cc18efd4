// kbench_gen.hip -- the general (finite dt) instantiation of k_fused3d over the whole grid: time + a checksum of the ten outputs, for A/B between two copies of stokes3d_kernels.hpp
// (hipcc ... -I <dir with the variant header> -I justrelax.jl_amd/csrc ...).  Derived from kbench_int.hip -- k_fused3d on the interior box of tiles: the shipped instantiation against the one that knows it is interior (INT; development tool).
// RECORD OF AN EXPERIMENT THAT WAS NOT SHIPPED (profiles/r04_interior_tiles_kernel.txt): building it needs a 17th template parameter `bool INT` on k_fused3d that states
//   __builtin_assume(i > 0 && i < nx - 1 && j > 0 && j < ny - 1 && kb > 0 && kb + KZ < nz) once and __builtin_assume(k > 0 && k < nz - 1) per plane; the tree does not carry it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I justrelax.jl_amd/csrc scripts/kbench_int.hip -o scripts/kbench_int
//   ./scripts/kbench_int [n=512] [reps=20]
// Every variant's ten output arrays are compared bit for bit with the shipped configuration's.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "jrx_internal.hpp"
#include "stokes3d_kernels.hpp"

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
__global__ void k_ndiff(const double *a, const double *b, i64 n, unsigned long long *out)
{
    unsigned long long m = 0;
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x)
        if (__double_as_longlong(a[t]) != __double_as_longlong(b[t])) m += 1;
    if (m) atomicAdd(out, m);
}
template <int NR, int NW, int NT>
struct StreamArgs { const double *r[NR > 0 ? NR : 1]; double *w[NW > 0 ? NW : 1]; i64 n; };
// pure streaming kernel with the stream mix of a sweep: NR arrays read, NW written, 8 B per lane, NT: non-temporal stores
template <int NR, int NW, int NT>
__global__ __launch_bounds__(256) void k_stream(StreamArgs<NR, NW, NT> a)
{
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n) return;
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < NR; q++) acc += a.r[q][t];
#pragma unroll
    for (int q = 0; q < NW; q++) {
        if (NT) __builtin_nontemporal_store(acc + q, a.w[q] + t);
        else a.w[q][t] = acc + q;
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



__global__ void k_cksum(const double *a, i64 n, unsigned long long *out)
{
    unsigned long long m = 0;
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x) m += (unsigned long long)__double_as_longlong(a[t]) * (unsigned long long)(2 * t + 1);
    if (m) atomicAdd(out, m);
}
int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 512, reps = argc > 2 ? atoi(argv[2]) : 20;
    const int nx = n, ny = n, nz = n;
    jrx_stokes3d_fields f;
    memset(&f, 0, sizeof(f));
    struct Ent { double **p; i64 n; double lo, hi; int expo; };
    const i64 nc = (i64)nx * ny * nz, nvx = (i64)(nx + 1) * (ny + 2) * (nz + 2), nvy = (i64)(nx + 2) * (ny + 1) * (nz + 2),
              nvz = (i64)(nx + 2) * (ny + 2) * (nz + 1), nxy = (i64)(nx + 1) * (ny + 1) * nz, nyz = (i64)nx * (ny + 1) * (nz + 1),
              nxz = (i64)(nx + 1) * ny * (nz + 1);
    std::vector<Ent> ents = {
        {&f.P, nc, -1, 1, 0}, {&f.Vx, nvx, -1, 1, 0}, {&f.Vy, nvy, -1, 1, 0}, {&f.Vz, nvz, -1, 1, 0},
        {&f.txx, nc, -1, 1, 0}, {&f.tyy, nc, -1, 1, 0}, {&f.tzz, nc, -1, 1, 0}, {&f.tyz, nyz, -1, 1, 0}, {&f.txz, nxz, -1, 1, 0}, {&f.txy, nxy, -1, 1, 0},
        {&f.eta, nc, -3, 0, 1}, {&f.fx, nc, -1, 1, 0}, {&f.fy, nc, -1, 1, 0}, {&f.fz, nc, -1, 1, 0},
        {&f.toxx, nc, -1, 1, 0}, {&f.toyy, nc, -1, 1, 0}, {&f.tozz, nc, -1, 1, 0}, {&f.toyz, nyz, -1, 1, 0}, {&f.toxz, nxz, -1, 1, 0}, {&f.toxy, nxy, -1, 1, 0},
        {&f.P0, nc, -1, 1, 0}, {&f.Q, nc, -0.1, 0.1, 0}, {&f.K, nc, 1, 2, 0}, {&f.G, nc, 1, 2, 0}};
    unsigned seed = 1;
    for (auto &e : ents) {
        CK(hipMalloc(e.p, e.n * sizeof(double)));
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, *e.p, e.n, seed++, e.lo, e.hi, e.expo);
    }
    double *etatau;
    CK(hipMalloc(&etatau, nc * sizeof(double)));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, etatau, nc, 99u, 0.5, 1.5, 0);
    SweepArgs a;
    a.f = f; a.etatau = etatau; a._dx = 51.2; a._dy = 49.0; a._dz = 47.5; a.dt = 0.37; a.r = 0.7; a.theta_dtau = 191.3; a.eta_dtau = 0.0119;
    a.L = make_lay(nx, ny, nz);
    a.i0 = a.j0 = a.k0 = 0;
    Out10 dst;
    const i64 dn[10] = {nc, nc, nc, nc, nyz, nxz, nxy, nvx, nvy, nvz};
    double **dp[10] = {&dst.P, &dst.txx, &dst.tyy, &dst.tzz, &dst.tyz, &dst.txz, &dst.txy, &dst.Vx, &dst.Vy, &dst.Vz};
    for (int q = 0; q < 10; q++) { CK(hipMalloc(dp[q], dn[q] * sizeof(double))); CK(hipMemset(*dp[q], 0, dn[q] * sizeof(double))); }
    unsigned long long *d_cnt;
    CK(hipMalloc(&d_cnt, 8));
    CK(hipDeviceSynchronize());
    FusedBC bc;
    memset(&bc, 0, sizeof(bc));
    bc.fsL = bc.fsF = bc.fsK0 = bc.fsR = bc.fsBk = bc.fsK1 = 1;
    Timer T;
    constexpr int TX = 64, TY = 4, KZ = 8;
    const int ntx = (nx + TX - 3) / (TX - 2), nty = (ny + TY - 2) / (TY - 1), ntz = (nz + KZ - 1) / KZ;
    SweepArgs b = a; b.o = dst;
    const bool visc = argc > 3 && atoi(argv[3]) != 0;
    if (visc) b.dt = INFINITY;
    // split: tiles [0, ntx-1) with 64-lane rows, the thin last tile column as 32-lane tiles (30 stress columns) with a column offset
    const int rem = nx - (ntx - 1) * (TX - 2);
    constexpr int SX = 32, SY = 8;
    const int nsx = (rem + SX - 3) / (SX - 2), nsy = (ny + SY - 2) / (SY - 1);
    SweepArgs bs = b; bs.i0 = (ntx - 1) * (TX - 2);
    printf("n=%d %s: %d x %d x %d tiles of 64 x 4; last tile column holds %d of 62 columns -> %d x %d x %d tiles of 32 x 8\n", n, visc ? "viscous form" : "general form", ntx, nty, ntz, rem, nsx, nsy, ntz);
    auto cks = [&]() {
        CK(hipMemset(d_cnt, 0, 8));
        for (int q = 0; q < 10; q++) hipLaunchKernelGGL(k_cksum, dim3(4096), dim3(256), 0, 0, *dp[q], dn[q], d_cnt);
        unsigned long long c;
        CK(hipMemcpy(&c, d_cnt, 8, hipMemcpyDeviceToHost));
        return c;
    };
    for (int rep = 0; rep < 3; rep++) {
        for (int q = 0; q < 10; q++) CK(hipMemset(*dp[q], 0, dn[q] * sizeof(double)));
        double ms;
        if (visc) ms = T.run(reps, [&] { hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, 4, 1, false, 1, false, true, 3, 1, 0, true, true, true>), dim3(ntx * nty * ntz), dim3(TX * TY), 0, 0, b, bc, ntx, nty, 0, 0, 0); });
        else ms = T.run(reps, [&] { hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, 4, 1, true, 1, false, true, 3, 1>), dim3(ntx * nty * ntz), dim3(TX * TY), 0, 0, b, bc, ntx, nty, 0, 0, 0); });
        const unsigned long long c0 = cks();
        for (int q = 0; q < 10; q++) CK(hipMemset(*dp[q], 0, dn[q] * sizeof(double)));
        double ms2;
        if (visc) ms2 = T.run(reps, [&] {
            hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, 4, 1, false, 1, false, true, 3, 1, 0, true, true, true>), dim3((ntx - 1) * nty * ntz), dim3(TX * TY), 0, 0, b, bc, ntx - 1, nty, 0, 0, 0);
            hipLaunchKernelGGL((k_fused3d<SX, SY, KZ, 4, 1, false, 1, false, true, 3, 1, 0, true, true, true>), dim3(nsx * nsy * ntz), dim3(SX * SY), 0, 0, bs, bc, nsx, nsy, 0, 0, 0); });
        else ms2 = T.run(reps, [&] {
            hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, 4, 1, true, 1, false, true, 3, 1>), dim3((ntx - 1) * nty * ntz), dim3(TX * TY), 0, 0, b, bc, ntx - 1, nty, 0, 0, 0);
            hipLaunchKernelGGL((k_fused3d<SX, SY, KZ, 4, 1, true, 1, false, true, 3, 1>), dim3(nsx * nsy * ntz), dim3(SX * SY), 0, 0, bs, bc, nsx, nsy, 0, 0, 0); });
        const unsigned long long c1 = cks();
        printf("  one launch %.3f ms | main + strip %.3f ms (x %.3f) | checksums %016llx %016llx %s\n", ms, ms2, ms / ms2, c0, c1, c0 == c1 ? "equal" : "DIFFERENT");
        fflush(stdout);
    }
    return 0;
}
