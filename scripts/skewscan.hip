// skewscan.hip -- does the relative placement of the equally sized field arrays in HBM matter?  (development tool)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I justrelax.jl_amd/csrc scripts/skewscan.hip -o scripts/skewscan
// One pool allocation; the 35 field arrays are carved out of it back to back with an extra `skew` bytes between consecutive
// arrays, for a list of skews, twice, in the same process (same physical pages), and the shipped sweeps are timed for each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "jrx_internal.hpp"
#include "stokes3d_kernels.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_fill(double *p, i64 n, unsigned seed, double lo, double hi, int expo)
{
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x) {
        unsigned long long x = (unsigned long long)t * 6364136223846793005ULL + seed * 1442695040888963407ULL + 1013904223ULL;
        x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33;
        double u = (double)(x >> 11) * (1.0 / 9007199254740992.0);
        double v = lo + (hi - lo) * u;
        p[t] = expo ? pow(10.0, v) : v;
    }
}

struct SArgs { const double *r[21]; double *w[7]; i64 n; };
__global__ __launch_bounds__(256) void k_stream28(SArgs a)
{
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n) return;
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < 21; q++) acc += a.r[q][t];
#pragma unroll
    for (int q = 0; q < 7; q++) a.w[q][t] = acc + q;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 512;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    int nx = n, ny = n, nz = n;
    if (getenv("NXYZ")) sscanf(getenv("NXYZ"), "%d,%d,%d", &nx, &ny, &nz);     // non-cubic shapes, e.g. NXYZ=256,256,2048
    const double cells = (double)nx * ny * nz;
    jrx_stokes3d_fields f;
    memset(&f, 0, sizeof(f));
    struct Ent { double **p; i64 n; double lo, hi; int expo; };
    const i64 nc = (i64)nx * ny * nz, nvx = (i64)(nx + 1) * (ny + 2) * (nz + 2), nvy = (i64)(nx + 2) * (ny + 1) * (nz + 2),
              nvz = (i64)(nx + 2) * (ny + 2) * (nz + 1), nxy = (i64)(nx + 1) * (ny + 1) * nz, nyz = (i64)nx * (ny + 1) * (nz + 1),
              nxz = (i64)(nx + 1) * ny * (nz + 1);
    double *etatau = nullptr;
    std::vector<Ent> ents = {
        {&f.P, nc, -1, 1, 0}, {&f.P0, nc, -1, 1, 0}, {&f.Q, nc, -0.1, 0.1, 0},
        {&f.Vx, nvx, -1, 1, 0}, {&f.Vy, nvy, -1, 1, 0}, {&f.Vz, nvz, -1, 1, 0},
        {&f.txx, nc, -1, 1, 0}, {&f.tyy, nc, -1, 1, 0}, {&f.tzz, nc, -1, 1, 0}, {&f.tyz, nyz, -1, 1, 0}, {&f.txz, nxz, -1, 1, 0}, {&f.txy, nxy, -1, 1, 0},
        {&f.toxx, nc, -1, 1, 0}, {&f.toyy, nc, -1, 1, 0}, {&f.tozz, nc, -1, 1, 0}, {&f.toyz, nyz, -1, 1, 0}, {&f.toxz, nxz, -1, 1, 0}, {&f.toxy, nxy, -1, 1, 0},
        {&f.eta, nc, -3, 0, 1}, {&f.K, nc, 1, 3, 0}, {&f.G, nc, 1, 2, 0},
        {&f.fx, nc, -1, 1, 0}, {&f.fy, nc, -1, 1, 0}, {&f.fz, nc, -1, 1, 0}, {&etatau, nc, 0.5, 1.5, 0}};
    // scratch set for the fused kernel's outputs
    double *S[10];
    const i64 sn[10] = {nc, nc, nc, nc, nyz, nxz, nxy, nvx, nvy, nvz};
    for (int q = 0; q < 10; q++) ents.push_back({&S[q], sn[q], -1, 1, 0});
    const i64 maxskew = 4 << 20;
    i64 total = 0;
    for (auto &e : ents) total += e.n * 8 + 256;
    char *pool;
    CK(hipMalloc(&pool, (size_t)(total + (i64)ents.size() * maxskew)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto fn) {
        fn(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; r++) fn();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return (double)ms / reps;
    };
    printf("%10s %12s %12s %12s %12s   (ms; n=%d)\n", "skew[B]", "stream21R7W", "stress_zb", "velocity_zb", "fused", n);
    // negative "skew" codes: -1 = one hipMalloc per array (allocation order = array order), -2 = same, arrays allocated in reverse order,
    // -3 = one hipMalloc per array with a 1.5 MiB dummy allocation between consecutive arrays
    std::vector<void *> owned;
    const std::vector<i64> skews = argc > 3 ? std::vector<i64>{0, -1, 0, -2, -3, -1, 0} : std::vector<i64>{0, 256, 4096 + 256, 65536 + 4096 + 256, (1 << 20) + 65536 + 4096 + 256, 0};
    for (i64 skew : skews) {
        i64 off = 0;
        unsigned seed = 1;
        for (void *q : owned) CK(hipFree(q));
        owned.clear();
        if (skew < 0) {
            const int ne = (int)ents.size();
            for (int q = 0; q < ne; q++) {
                auto &e = ents[skew == -2 ? ne - 1 - q : q];
                void *m;
                CK(hipMalloc(&m, (size_t)e.n * 8));
                owned.push_back(m);
                *e.p = (double *)m;
                if (skew == -3) { void *d; CK(hipMalloc(&d, 3 << 19)); owned.push_back(d); }
            }
        }
        for (auto &e : ents) {
            if (skew >= 0) *e.p = (double *)(pool + off);
            hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, *e.p, e.n, seed++, e.lo, e.hi, e.expo);
            off += ((e.n * 8 + 255) / 256) * 256 + (skew > 0 ? skew : 0);
        }
        CK(hipDeviceSynchronize());
        SweepArgs a;
        a.f = f; a.etatau = etatau; a._dx = 51.2; a._dy = 49.0; a._dz = 47.5; a.dt = 0.25; a.r = 0.7; a.theta_dtau = 191.3; a.eta_dtau = 0.0119;
        a.L = make_lay(nx, ny, nz);
        a.i0 = a.j0 = a.k0 = 0; a.i1 = nx; a.j1 = ny; a.k1 = nz;
        a.o = Out10{f.P, f.txx, f.tyy, f.tzz, f.tyz, f.txz, f.txy, f.Vx, f.Vy, f.Vz};
        SArgs sa;
        const double *rd[21] = {f.Vx, f.Vy, f.Vz, f.P, f.P0, f.Q, f.eta, f.K, f.G, f.txx, f.tyy, f.tzz, f.tyz, f.txz, f.txy, f.toxx, f.toyy, f.tozz, f.toyz, f.toxz, f.toxy};
        for (int q = 0; q < 21; q++) sa.r[q] = rd[q];
        for (int q = 0; q < 7; q++) sa.w[q] = S[q];
        sa.n = nc;
        const double t_stream = timeit([&] { hipLaunchKernelGGL(k_stream28, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, 0, sa); });
        double t_stress, t_vel;
        if (nx >= 512) {
            TileMap tm = make_tilemap(nx, ny, nz, 512, 1, 4);
            t_stress = timeit([&] { hipLaunchKernelGGL((k_stress3d_zb<false, 512, 1, 4, 4, false, 8>), dim3(tm.per * 8), dim3(512), 0, 0, a, tm); });
            t_vel = timeit([&] { hipLaunchKernelGGL((k_velocity3d_zb<false, 512, 1, 4, 4, 8>), dim3(tm.per * 8), dim3(512), 0, 0, a, tm); });
        } else {
            TileMap tm = make_tilemap(nx, ny, nz, 256, 1, 8);
            t_stress = timeit([&] { hipLaunchKernelGGL((k_stress3d_zb<false, 256, 1, 8, 4, false, 8>), dim3(tm.per * 8), dim3(256), 0, 0, a, tm); });
            t_vel = timeit([&] { hipLaunchKernelGGL((k_velocity3d_zb<false, 256, 1, 8, 4, 8>), dim3(tm.per * 8), dim3(256), 0, 0, a, tm); });
        }
        SweepArgs b = a;
        b.o = Out10{S[0], S[1], S[2], S[3], S[4], S[5], S[6], S[7], S[8], S[9]};
        FusedBC bc; memset(&bc, 0, sizeof(bc)); bc.fsL = bc.fsF = bc.fsK0 = 1;
        const int ntx = (nx + 62) / 63, nty = (ny + 2) / 3, ntz = (nz + 15) / 16;
        const double t_fused = timeit([&] { hipLaunchKernelGGL((k_fused3d<64, 4, 16, 2, 1, false, 8>), dim3(ntx * nty * ntz), dim3(256), 0, 0, b, bc, ntx, nty); });
        const double sc = 134217728.0 / cells;      // normalise to the cell count of 512^3
        printf("%10lld %12.3f %12.3f %12.3f %12.3f   [%dx%dx%d, per 512^3 cells]\n", (long long)skew, t_stream * sc, t_stress * sc, t_vel * sc, t_fused * sc, nx, ny, nz);
        fflush(stdout);
    }
    return 0;
}
