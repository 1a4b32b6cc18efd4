// kbench.hip -- kernel micro-benchmark / A-B harness for the 3D Stokes sweeps (development tool).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I justrelax.jl_amd/csrc scripts/kbench.hip -o scripts/kbench
//   ./scripts/kbench [n=512] [reps=10]
// Every variant is checked bit-for-bit against the parity-tested v1 kernels on the same inputs,
// then timed with hipEvents; plus pure streaming kernels (R read + W write streams) that give the
// practical HBM ceiling for this many concurrent streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include "jrx_internal.hpp"
#include "stokes3d_kernels.hpp"
#include "fused_ws.hpp"
namespace {
#include "fused_exp.hpp"
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_fill(double *p, i64 n, unsigned seed, double lo, double hi, int expo)
{
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x) {
        unsigned long long x = (unsigned long long)t * 6364136223846793005ULL + seed * 1442695040888963407ULL + 1013904223ULL;
        x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
        double u = (double)(x >> 11) * (1.0 / 9007199254740992.0);
        double v = lo + (hi - lo) * u;
        p[t] = expo ? pow(10.0, v) : v;
    }
}

__global__ void k_maxdiff(const double *a, const double *b, i64 n, unsigned long long *out)
{
    unsigned long long m = 0;
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x) {
        unsigned long long x = __double_as_longlong(a[t]), y = __double_as_longlong(b[t]);
        if (x != y) m += 1;
    }
    if (m) atomicAdd(out, m);
}

template <int NR, int NW, int VEC>
struct StreamArgs { const double *r[NR > 0 ? NR : 1]; double *w[NW > 0 ? NW : 1]; i64 n; };

template <int NR, int NW, int VEC>
__global__ __launch_bounds__(256) void k_stream(StreamArgs<NR, NW, VEC> a)
{
    const i64 t = ((i64)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (t + VEC > a.n) return;
    double acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) acc[v] = 0.0;
#pragma unroll
    for (int q = 0; q < NR; q++) {
        if (VEC == 2) {
            double2 x = *reinterpret_cast<const double2 *>(a.r[q] + t);
            acc[0] += x.x; acc[1] += x.y;
        } else {
            acc[0] += a.r[q][t];
        }
    }
#pragma unroll
    for (int q = 0; q < NW; q++) {
        if (VEC == 2) *reinterpret_cast<double2 *>(a.w[q] + t) = make_double2(acc[0] + q, acc[1] + q);
        else a.w[q][t] = acc[0] + q;
    }
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    template <class F> double run(int reps, F f)
    {
        f();
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
    const int n = argc > 1 ? atoi(argv[1]) : 512;
    const int reps = argc > 2 ? atoi(argv[2]) : 10;
    const int nx = n, ny = n, nz = n;
    const double cells = (double)nx * ny * nz;
    printf("kbench n=%d reps=%d\n", n, reps);

    jrx_stokes3d_fields f;
    memset(&f, 0, sizeof(f));
    struct Ent { double **p; i64 n; double lo, hi; int expo; };
    const i64 nc = (i64)nx * ny * nz, nvx = (i64)(nx + 1) * (ny + 2) * (nz + 2), nvy = (i64)(nx + 2) * (ny + 1) * (nz + 2),
              nvz = (i64)(nx + 2) * (ny + 2) * (nz + 1), nxy = (i64)(nx + 1) * (ny + 1) * nz, nyz = (i64)nx * (ny + 1) * (nz + 1),
              nxz = (i64)(nx + 1) * ny * (nz + 1);
    std::vector<Ent> ents = {
        {&f.P, nc, -1, 1, 0}, {&f.P0, nc, -1, 1, 0}, {&f.divV, nc, 0, 0, 0}, {&f.Q, nc, -0.1, 0.1, 0},
        {&f.Vx, nvx, -1, 1, 0}, {&f.Vy, nvy, -1, 1, 0}, {&f.Vz, nvz, -1, 1, 0},
        {&f.txx, nc, -1, 1, 0}, {&f.tyy, nc, -1, 1, 0}, {&f.tzz, nc, -1, 1, 0}, {&f.tyz, nyz, -1, 1, 0}, {&f.txz, nxz, -1, 1, 0}, {&f.txy, nxy, -1, 1, 0},
        {&f.toxx, nc, -1, 1, 0}, {&f.toyy, nc, -1, 1, 0}, {&f.tozz, nc, -1, 1, 0}, {&f.toyz, nyz, -1, 1, 0}, {&f.toxz, nxz, -1, 1, 0}, {&f.toxy, nxy, -1, 1, 0},
        {&f.exx, nc, 0, 0, 0}, {&f.eyy, nc, 0, 0, 0}, {&f.ezz, nc, 0, 0, 0}, {&f.eyz, nyz, 0, 0, 0}, {&f.exz, nxz, 0, 0, 0}, {&f.exy, nxy, 0, 0, 0},
        {&f.eta, nc, -3, 0, 1}, {&f.K, nc, 1, 3, 0}, {&f.G, nc, 1, 2, 0},
        {&f.fx, nc, -1, 1, 0}, {&f.fy, nc, -1, 1, 0}, {&f.fz, nc, -1, 1, 0},
        {&f.RP, nc, 0, 0, 0}, {&f.Rx, nc, 0, 0, 0}, {&f.Ry, nc, 0, 0, 0}, {&f.Rz, nc, 0, 0, 0}};
    unsigned seed = 1;
    // KB_SKEW: byte offset added per array (array q starts q*skew bytes into its allocation) to decorrelate the
    // HBM channel/bank mapping of equally sized arrays
    const i64 skew = getenv("KB_SKEW") ? atoll(getenv("KB_SKEW")) : 0;
    int qidx = 0;
    for (auto &e : ents) {
        char *base;
        CK(hipMalloc(&base, e.n * sizeof(double) + 64 * skew));
        *e.p = (double *)(base + (qidx++) * skew);
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, *e.p, e.n, seed++, e.lo, e.hi, e.expo);
    }
    double *etatau;
    CK(hipMalloc(&etatau, nc * sizeof(double)));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, etatau, nc, 99u, 0.5, 1.5, 0);
    CK(hipDeviceSynchronize());

    jrx_stokes3d_params p;
    memset(&p, 0, sizeof(p));
    p.nx = nx; p.ny = ny; p.nz = nz; p._dx = 51.2; p._dy = 49.0; p._dz = 47.5; p.dt = 0.25; p.r = 0.7; p.theta_dtau = 191.3; p.eta_dtau = 0.0119;
    SweepArgs a;
    a.f = f; a.etatau = etatau; a._dx = p._dx; a._dy = p._dy; a._dz = p._dz; a.dt = p.dt; a.r = p.r; a.theta_dtau = p.theta_dtau; a.eta_dtau = p.eta_dtau;
    a.L = make_lay(nx, ny, nz);
    a.i0 = a.j0 = a.k0 = 0;
    a.o = Out10{f.P, f.txx, f.tyy, f.tzz, f.tyz, f.txz, f.txy, f.Vx, f.Vy, f.Vz};

    // state snapshots for A/B equality
    struct St { double **p; i64 n; double *bak, *ref; };
    std::vector<St> sA = {{&f.P, nc}, {&f.txx, nc}, {&f.tyy, nc}, {&f.tzz, nc}, {&f.tyz, nyz}, {&f.txz, nxz}, {&f.txy, nxy}};
    std::vector<St> sB = {{&f.Vx, nvx}, {&f.Vy, nvy}, {&f.Vz, nvz}};
    for (auto *v : {&sA, &sB})
        for (auto &s : *v) {
            CK(hipMalloc(&s.bak, s.n * sizeof(double)));
            CK(hipMalloc(&s.ref, s.n * sizeof(double)));
            CK(hipMemcpy(s.bak, *s.p, s.n * sizeof(double), hipMemcpyDeviceToDevice));
        }
    unsigned long long *d_cnt;
    CK(hipMalloc(&d_cnt, 8));
    auto restore = [&](std::vector<St> &v) { for (auto &s : v) CK(hipMemcpy(*s.p, s.bak, s.n * sizeof(double), hipMemcpyDeviceToDevice)); };
    auto saveref = [&](std::vector<St> &v) { for (auto &s : v) CK(hipMemcpy(s.ref, *s.p, s.n * sizeof(double), hipMemcpyDeviceToDevice)); };
    auto ndiff = [&](std::vector<St> &v) {
        unsigned long long tot = 0;
        for (auto &s : v) {
            CK(hipMemset(d_cnt, 0, 8));
            hipLaunchKernelGGL(k_maxdiff, dim3(4096), dim3(256), 0, 0, *s.p, s.ref, s.n, d_cnt);
            unsigned long long c;
            CK(hipMemcpy(&c, d_cnt, 8, hipMemcpyDeviceToHost));
            tot += c;
        }
        return tot;
    };
    Timer T;
    auto report = [&](const char *name, double ms, double bytes_per_cell, unsigned long long nd) {
        printf("%-44s %8.3f ms  %7.1f GB/s(alg)  frac8T %.3f  mismatches %llu\n", name, ms, bytes_per_cell * cells / (ms * 1e-3) / 1e9,
               bytes_per_cell * cells / (ms * 1e-3) / 1e9 / 8000.0, nd);
        fflush(stdout);
    };

    // ---------------- streaming ceilings
    {
        std::vector<double *> pool;
        for (auto &e : ents) if (e.n >= nc) pool.push_back(*e.p);
#define STREAM(NR, NW, VEC)                                                                                         \
    {                                                                                                               \
        StreamArgs<NR, NW, VEC> sa;                                                                                 \
        for (int q = 0; q < NR; q++) sa.r[q] = pool[q];                                                             \
        for (int q = 0; q < NW; q++) sa.w[q] = pool[NR + q];                                                        \
        sa.n = nc;                                                                                                  \
        dim3 g((unsigned)((nc / VEC + 255) / 256));                                                                 \
        double ms = T.run(reps, [&] { hipLaunchKernelGGL((k_stream<NR, NW, VEC>), g, dim3(256), 0, 0, sa); });      \
        char nm[64];                                                                                                \
        snprintf(nm, 64, "stream %dR+%dW x %dB/lane", NR, NW, VEC * 8);                                             \
        report(nm, ms, (NR + NW) * 8.0, 0);                                                                         \
    }
        STREAM(1, 1, 1) STREAM(1, 1, 2) STREAM(8, 2, 1) STREAM(8, 2, 2) STREAM(21, 7, 1) STREAM(21, 7, 2) STREAM(14, 3, 1) STREAM(14, 3, 2)
    }
    // re-fill (streams clobbered the pool)
    seed = 1;
    for (auto &e : ents) hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, *e.p, e.n, seed++, e.lo, e.hi, e.expo);
    CK(hipDeviceSynchronize());
    for (auto *v : {&sA, &sB}) for (auto &s : *v) CK(hipMemcpy(s.bak, *s.p, s.n * sizeof(double), hipMemcpyDeviceToDevice));

    // ---------------- stress sweep
    auto stress_v1 = [&] {
        SweepArgs b = a; b.i1 = nx + 1; b.j1 = ny + 1; b.k1 = nz + 1;
        dim3 g((unsigned)(((i64)(nx + 1) * (ny + 1) + 255) / 256), nz + 1);
        hipLaunchKernelGGL(k_stress3d<false>, g, dim3(256), 0, 0, b);
    };
    stress_v1(); CK(hipDeviceSynchronize()); saveref(sA);
    report("stress v1 (flat xy, 1 node/thread)", T.run(reps, stress_v1), 224.0, 0);
#define STRESS_ZB(TX, TY, KZ, MW, ED, XM) STRESS_ZB2(TX, TY, KZ, MW, ED, XM, false)
#define STRESS_ZB2(TX, TY, KZ, MW, ED, XM, SH)                                                                              \
    {                                                                                                               \
        TileMap tm = make_tilemap(nx, ny, nz, TX, TY, KZ);                                                          \
        auto fn = [&] {                                                                                             \
            hipLaunchKernelGGL((k_stress3d_zb<false, TX, TY, KZ, MW, ED, XM, SH>), dim3(tm.per * 8), dim3(TX * TY), 0, 0, a, tm); \
            if (!ED) {                                                                                              \
                SweepArgs b = a;                                                                                    \
                b.i0 = nx; b.i1 = nx + 1; b.j0 = 0; b.j1 = ny + 1; b.k0 = 0; b.k1 = nz + 1;                         \
                hipLaunchKernelGGL(k_stress3d<false>, dim3((ny + 1 + 255) / 256, nz + 1), dim3(256), 0, 0, b);      \
                b.i0 = 0; b.i1 = nx; b.j0 = ny; b.j1 = ny + 1;                                                      \
                hipLaunchKernelGGL(k_stress3d<false>, dim3((nx + 255) / 256, nz + 1), dim3(256), 0, 0, b);          \
                b.j0 = 0; b.j1 = ny; b.k0 = nz; b.k1 = nz + 1;                                                      \
                hipLaunchKernelGGL(k_stress3d<false>, dim3((unsigned)(((i64)nx * ny + 255) / 256), 1), dim3(256), 0, 0, b); \
            }                                                                                                       \
        };                                                                                                          \
        restore(sA); fn(); CK(hipDeviceSynchronize());                                                              \
        unsigned long long nd = ndiff(sA);                                                                          \
        char nm[64];                                                                                                \
        snprintf(nm, 64, "stress zb %dx%dx%d minw%d e%d xcd%d shf%d", TX, TY, KZ, MW, (int)ED, (int)XM, (int)SH);                            \
        report(nm, T.run(reps, fn), 224.0, nd);                                                                     \
    }
    STRESS_ZB(512, 1, 4, 4, false, 0) STRESS_ZB(512, 1, 4, 4, false, 8) STRESS_ZB2(512, 1, 4, 4, false, 8, true) STRESS_ZB2(256, 1, 8, 4, false, 8, true) STRESS_ZB(512, 1, 4, 4, false, 8) STRESS_ZB2(512, 1, 4, 4, false, 8, true)  STRESS_ZB(512, 1, 4, 4, false, 4) STRESS_ZB(512, 1, 4, 4, false, 16) STRESS_ZB(512, 1, 4, 4, false, 2)
    STRESS_ZB(512, 1, 8, 4, false, 8) STRESS_ZB(512, 1, 16, 4, false, 8) STRESS_ZB(256, 1, 8, 4, false, 0) STRESS_ZB(256, 1, 8, 4, false, 8) STRESS_ZB(256, 1, 16, 4, false, 8)
    STRESS_ZB(128, 2, 8, 4, false, 8) STRESS_ZB(128, 2, 16, 4, false, 4) STRESS_ZB(512, 1, 32, 4, false, 8)
    restore(sA);

    // ---------------- velocity sweep
    auto vel_v1 = [&] {
        SweepArgs b = a; b.i1 = nx; b.j1 = ny; b.k1 = nz;
        dim3 g((unsigned)(((i64)nx * ny + 255) / 256), nz);
        hipLaunchKernelGGL(k_velocity3d<false>, g, dim3(256), 0, 0, b);
    };
    vel_v1(); CK(hipDeviceSynchronize()); saveref(sB);
    report("velocity v1 (flat xy, 1 cell/thread)", T.run(reps, vel_v1), 136.0, 0);
#define VEL_ZB(TX, TY, KZ, MW, XM) VEL_ZB2(TX, TY, KZ, MW, XM, false)
#define VEL_ZB2(TX, TY, KZ, MW, XM, SH)                                                                                     \
    {                                                                                                               \
        TileMap tm = make_tilemap(nx, ny, nz, TX, TY, KZ);                                                          \
        auto fn = [&] {                                                                                             \
            SweepArgs b = a; b.i1 = nx; b.j1 = ny; b.k1 = nz;                                                       \
            hipLaunchKernelGGL((k_velocity3d_zb<false, TX, TY, KZ, MW, XM, SH>), dim3(tm.per * 8), dim3(TX * TY), 0, 0, b, tm); \
        };                                                                                                          \
        restore(sB); fn(); CK(hipDeviceSynchronize());                                                              \
        unsigned long long nd = ndiff(sB);                                                                          \
        char nm[64];                                                                                                \
        snprintf(nm, 64, "velocity zb %dx%dx%d minw%d xcd%d shf%d", TX, TY, KZ, MW, (int)XM, (int)SH);                                           \
        report(nm, T.run(reps, fn), 136.0, nd);                                                                     \
    }
    VEL_ZB(512, 1, 4, 4, 0) VEL_ZB(512, 1, 4, 4, 8) VEL_ZB2(512, 1, 4, 4, 8, true) VEL_ZB2(256, 1, 8, 4, 8, true) VEL_ZB(512, 1, 4, 4, 8) VEL_ZB2(512, 1, 4, 4, 8, true)  VEL_ZB(512, 1, 4, 4, 4) VEL_ZB(512, 1, 4, 4, 16) VEL_ZB(512, 1, 8, 4, 8) VEL_ZB(512, 1, 16, 4, 8)
    VEL_ZB(256, 1, 8, 4, 0) VEL_ZB(256, 1, 8, 4, 8) VEL_ZB(256, 1, 16, 4, 8) VEL_ZB(128, 2, 8, 4, 8) VEL_ZB(512, 1, 32, 4, 8)
    // ---------------- fused iteration kernel (timing only; bit-exactness is covered by tests/test_gpu_stokes3d.py)
    {
        Out10 dst;
        double **dp[10] = {&dst.P, &dst.txx, &dst.tyy, &dst.tzz, &dst.tyz, &dst.txz, &dst.txy, &dst.Vx, &dst.Vy, &dst.Vz};
        const i64 dn[10] = {nc, nc, nc, nc, nyz, nxz, nxy, nvx, nvy, nvz};
        for (int q = 0; q < 10; q++) CK(hipMalloc(dp[q], dn[q] * sizeof(double) + 4096));   // slack: the ablation kernels shift stores by up to 64 B
        restore(sA); restore(sB);
        SweepArgs b = a; b.o = dst;
        FusedBC bc; memset(&bc, 0, sizeof(bc)); bc.fsL = bc.fsF = bc.fsK0 = 1;
#define FUSED(TX, TY, KZ, MW)                                                                                       \
    {                                                                                                               \
        const int ntx = (nx + TX - 2) / (TX - 1), nty = (ny + TY - 2) / (TY - 1), ntz = (nz + KZ - 1) / KZ;           \
        auto fn = [&] { hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, MW>), dim3(ntx * nty * ntz), dim3(TX * TY), 0, 0, b, bc, ntx, nty); }; \
        char nm[64];                                                                                                \
        snprintf(nm, 64, "fused %dx%dx%d minw%d", TX, TY, KZ, MW);                                                  \
        report(nm, T.run(reps, fn), 360.0, 0);                                                                      \
    }
        FUSED(64, 4, 16, 2)
#define FUSED2(TX, TY, KZ, MW, OV, LR, XG, LA)                                                                      \
    {                                                                                                               \
        const int ntx = (nx + TX - OV - 1) / (TX - OV), nty = (ny + TY - 2) / (TY - 1), ntz = (nz + KZ - 1) / KZ;     \
        auto fn = [&] { hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, MW, OV, LR, XG, LA>), dim3(ntx * nty * ntz), dim3(TX * TY), 0, 0, b, bc, ntx, nty); }; \
        char nm[80];                                                                                                \
        snprintf(nm, 80, "fused %dx%dx%d minw%d ovx%d lowreg%d xg%d latea%d", TX, TY, KZ, MW, OV, (int)LR, XG, (int)LA); \
        report(nm, T.run(reps, fn), 360.0, 0);                                                                      \
    }
        FUSED2(64, 4, 16, 2, 1, false, 8, false)
#define FEXP(EX)                                                                                                    \
    {                                                                                                               \
        const int ntx = (nx + 62) / 63, nty = (ny + 2) / 3, ntz = (nz + 15) / 16;                                     \
        auto fn = [&] { hipLaunchKernelGGL((k_fused3d_exp<64, 4, 16, 2, EX>), dim3(ntx * nty * ntz), dim3(256), 0, 0, b, bc, ntx, nty); }; \
        char nm[64];                                                                                                \
        snprintf(nm, 64, "fused 64x4x16 xg8 ablation EXP=%d", EX);                                                  \
        report(nm, T.run(reps, fn), 360.0, 0);                                                                      \
    }
        FEXP(0)
        {
            const int ntx = (nx + 61) / 62, nty = (ny + 2) / 3, ntz = (nz + 15) / 16;
            auto fn = [&] { hipLaunchKernelGGL((k_fused3d<64, 4, 16, 2, 1, false, 8, false, true>), dim3(ntx * nty * ntz), dim3(256), 0, 0, b, bc, ntx, nty); };
            report("fused 64x4x16 xg8 SHFL (lane shuffles for x-neighbours)", T.run(reps, fn), 360.0, 0);
            const int nty8 = (ny + 6) / 7;
            auto fn8 = [&] { hipLaunchKernelGGL((k_fused3d<64, 8, 16, 2, 1, false, 8, false, true>), dim3(ntx * nty8 * ntz), dim3(512), 0, 0, b, bc, ntx, nty8); };
            report("fused 64x8x16 xg8 SHFL", T.run(reps, fn8), 360.0, 0);
            auto fy = [&] { hipLaunchKernelGGL((k_fused3d<64, 4, 16, 2, 1, false, 8, false, true, 1>), dim3(ntx * nty * ntz), dim3(256), 0, 0, b, bc, ntx, nty); };
            report("fused 64x4x16 xg8 SHFL+YLDS", T.run(reps, fy), 360.0, 0);
            auto fy8 = [&] { hipLaunchKernelGGL((k_fused3d<64, 8, 16, 2, 1, false, 8, false, true, 1>), dim3(ntx * nty8 * ntz), dim3(512), 0, 0, b, bc, ntx, nty8); };
            report("fused 64x8x16 xg8 SHFL+YLDS", T.run(reps, fy8), 360.0, 0);
            const int nty6 = (ny + 4) / 5;
            auto fy6 = [&] { hipLaunchKernelGGL((k_fused3d<64, 6, 16, 2, 1, false, 8, false, true, 1>), dim3(ntx * nty6 * ntz), dim3(384), 0, 0, b, bc, ntx, nty6); };
            report("fused 64x6x16 xg8 SHFL+YLDS", T.run(reps, fy6), 360.0, 0);
            auto fy2 = [&] { hipLaunchKernelGGL((k_fused3d<64, 4, 16, 2, 1, false, 8, false, true, 2>), dim3(ntx * nty * ntz), dim3(256), 0, 0, b, bc, ntx, nty); };
            report("fused 64x4x16 xg8 SHFL+YLDS2 (stress operands queued last)", T.run(reps, fy2), 360.0, 0);
            const int ntz32 = (nz + 31) / 32, ntz24 = (nz + 23) / 24, ntz8 = (nz + 7) / 8;
            auto fz32 = [&] { hipLaunchKernelGGL((k_fused3d<64, 4, 32, 2, 1, false, 8, false, true, 1>), dim3(ntx * nty * ntz32), dim3(256), 0, 0, b, bc, ntx, nty); };
            report("fused 64x4x32 xg8 SHFL+YLDS", T.run(reps, fz32), 360.0, 0);
            auto fz24 = [&] { hipLaunchKernelGGL((k_fused3d<64, 4, 24, 2, 1, false, 8, false, true, 1>), dim3(ntx * nty * ntz24), dim3(256), 0, 0, b, bc, ntx, nty); };
            report("fused 64x4x24 xg8 SHFL+YLDS", T.run(reps, fz24), 360.0, 0);
            auto fz8 = [&] { hipLaunchKernelGGL((k_fused3d<64, 4, 8, 2, 1, false, 8, false, true, 1>), dim3(ntx * nty * ntz8), dim3(256), 0, 0, b, bc, ntx, nty); };
            report("fused 64x4x8 xg8 SHFL+YLDS", T.run(reps, fz8), 360.0, 0);
            auto fx4 = [&] { hipLaunchKernelGGL((k_fused3d<64, 4, 16, 2, 1, false, 4, false, true, 1>), dim3(ntx * nty * ntz), dim3(256), 0, 0, b, bc, ntx, nty); };
            report("fused 64x4x16 xg4 SHFL+YLDS", T.run(reps, fx4), 360.0, 0);
            auto fx16 = [&] { hipLaunchKernelGGL((k_fused3d<64, 4, 16, 2, 1, false, 16, false, true, 1>), dim3(ntx * nty * ntz), dim3(256), 0, 0, b, bc, ntx, nty); };
            report("fused 64x4x16 xg16 SHFL+YLDS", T.run(reps, fx16), 360.0, 0);
            auto fnt1 = [&] { hipLaunchKernelGGL((k_fused3d<64, 4, 16, 2, 1, false, 8, false, true, 1, 1>), dim3(ntx * nty * ntz), dim3(256), 0, 0, b, bc, ntx, nty); };
            report("fused 64x4x16 xg8 SHFL+YLDS nt-stores", T.run(reps, fnt1), 360.0, 0);
            auto fnt2 = [&] { hipLaunchKernelGGL((k_fused3d<64, 4, 16, 2, 1, false, 8, false, true, 1, 2>), dim3(ntx * nty * ntz), dim3(256), 0, 0, b, bc, ntx, nty); };
            report("fused 64x4x16 xg8 SHFL+YLDS nt-loads(streaming arrays)", T.run(reps, fnt2), 360.0, 0);
            auto fnt3 = [&] { hipLaunchKernelGGL((k_fused3d<64, 4, 16, 2, 1, false, 8, false, true, 1, 3>), dim3(ntx * nty * ntz), dim3(256), 0, 0, b, bc, ntx, nty); };
            report("fused 64x4x16 xg8 SHFL+YLDS nt-both", T.run(reps, fnt3), 360.0, 0);
            const int nty5 = (ny + 3) / 4, nty3 = (ny + 1) / 2;
            auto f5 = [&] { hipLaunchKernelGGL((k_fused3d<64, 5, 16, 2, 1, false, 8, false, true, 1>), dim3(ntx * nty5 * ntz), dim3(320), 0, 0, b, bc, ntx, nty5); };
            report("fused 64x5x16 xg8 SHFL+YLDS", T.run(reps, f5), 360.0, 0);
            auto f3 = [&] { hipLaunchKernelGGL((k_fused3d<64, 3, 16, 2, 1, false, 8, false, true, 1>), dim3(ntx * nty3 * ntz), dim3(192), 0, 0, b, bc, ntx, nty3); };
            report("fused 64x3x16 xg8 SHFL+YLDS", T.run(reps, f3), 360.0, 0);
            auto g4 = [&] { hipLaunchKernelGGL((k_fused3d<64, 4, 16, 4, 1, false, 8, false, true, 1, 1>), dim3(ntx * nty * ntz), dim3(256), 0, 0, b, bc, ntx, nty); };
            report("fused 64x4x16 minw4 YLDS nt-stores", T.run(reps, g4), 360.0, 0);
            auto g6 = [&] { hipLaunchKernelGGL((k_fused3d<64, 6, 16, 4, 1, false, 8, false, true, 1, 1>), dim3(ntx * nty6 * ntz), dim3(384), 0, 0, b, bc, ntx, nty6); };
            report("fused 64x6x16 minw4 YLDS nt-stores", T.run(reps, g6), 360.0, 0);
            auto g8 = [&] { hipLaunchKernelGGL((k_fused3d<64, 8, 16, 4, 1, false, 8, false, true, 1, 1>), dim3(ntx * nty8 * ntz), dim3(512), 0, 0, b, bc, ntx, nty8); };
            report("fused 64x8x16 minw4 YLDS nt-stores", T.run(reps, g8), 360.0, 0);
            auto g88 = [&] { hipLaunchKernelGGL((k_fused3d<64, 8, 8, 4, 1, false, 8, false, true, 1, 1>), dim3(ntx * nty8 * ntz8), dim3(512), 0, 0, b, bc, ntx, nty8); };
            report("fused 64x8x8 minw4 YLDS nt-stores", T.run(reps, g88), 360.0, 0);
            const int nty12 = (ny + 10) / 11;
            auto h12 = [&] { hipLaunchKernelGGL((k_fused3d<64, 12, 16, 1, 1, false, 8, false, true, 1, 1>), dim3(ntx * nty12 * ntz), dim3(768), 0, 0, b, bc, ntx, nty12); };
            report("fused 64x12x16 minw1 YLDS nt-stores", T.run(reps, h12), 360.0, 0);
            auto h128 = [&] { hipLaunchKernelGGL((k_fused3d<64, 12, 8, 1, 1, false, 8, false, true, 1, 1>), dim3(ntx * nty12 * ntz8), dim3(768), 0, 0, b, bc, ntx, nty12); };
            report("fused 64x12x8 minw1 YLDS nt-stores", T.run(reps, h128), 360.0, 0);
            auto h1232 = [&] { hipLaunchKernelGGL((k_fused3d<64, 12, 32, 1, 1, false, 8, false, true, 1, 1>), dim3(ntx * nty12 * ntz32), dim3(768), 0, 0, b, bc, ntx, nty12); };
            report("fused 64x12x32 minw1 YLDS nt-stores", T.run(reps, h1232), 360.0, 0);
#define FV(TY_, MW_, LR_, YL_, NTY_, NM_)                                                                                                    \
            {                                                                                                                                    \
                auto f_ = [&] { hipLaunchKernelGGL((k_fused3d<64, TY_, 16, MW_, 1, LR_, 8, false, true, YL_, 1>), dim3(ntx * NTY_ * ntz), dim3(64 * TY_), 0, 0, b, bc, ntx, NTY_); }; \
                report(NM_, T.run(reps, f_), 360.0, 0);                                                                                          \
            }
            FV(4, 2, true, 1, nty, "fused 64x4 lowreg ylds1 minw2")
            FV(4, 2, false, 3, nty, "fused 64x4 ylds3(late stress operands) minw2")
            FV(4, 2, true, 3, nty, "fused 64x4 lowreg ylds3 minw2")
            FV(4, 4, true, 3, nty, "fused 64x4 lowreg ylds3 minw4")
            FV(8, 4, true, 3, nty8, "fused 64x8 lowreg ylds3 minw4")
            FV(8, 2, true, 3, nty8, "fused 64x8 lowreg ylds3 minw2")
#define FVK(KZ_, XG_, NM_)                                                                                                               \
            {                                                                                                                                    \
                const int ntzk = (nz + KZ_ - 1) / KZ_;                                                                                           \
                auto f_ = [&] { hipLaunchKernelGGL((k_fused3d<64, 4, KZ_, 4, 1, true, XG_, false, true, 3, 1>), dim3(ntx * nty * ntzk), dim3(256), 0, 0, b, bc, ntx, nty); }; \
                report(NM_, T.run(reps, f_), 360.0, 0);                                                                                          \
            }
            FVK(16, 8, "shipped: 64x4x16 xg8 lowreg ylds3 minw4 nt")
            FVK(8, 8, "64x4x8 xg8 lowreg ylds3 minw4 nt")
            FVK(12, 8, "64x4x12 xg8 lowreg ylds3 minw4 nt")
            FVK(24, 8, "64x4x24 xg8 lowreg ylds3 minw4 nt")
            FVK(32, 8, "64x4x32 xg8 lowreg ylds3 minw4 nt")
            FVK(16, 4, "64x4x16 xg4 lowreg ylds3 minw4 nt")
            FVK(16, 16, "64x4x16 xg16 lowreg ylds3 minw4 nt")
            FVK(16, 32, "64x4x16 xg32 lowreg ylds3 minw4 nt")
            FVK(8, 4, "64x4x8 xg4 lowreg ylds3 minw4 nt")
            FVK(8, 2, "64x4x8 xg2 lowreg ylds3 minw4 nt")
            FVK(8, 16, "64x4x8 xg16 lowreg ylds3 minw4 nt")
            FVK(12, 4, "64x4x12 xg4 lowreg ylds3 minw4 nt")
            FVK(6, 4, "64x4x6 xg4 lowreg ylds3 minw4 nt")
            FVK(16, 2, "64x4x16 xg2 lowreg ylds3 minw4 nt")
            FVK(8, 1, "64x4x8 xg1 lowreg ylds3 minw4 nt")
            FVK(6, 1, "64x4x6 xg1 lowreg ylds3 minw4 nt")
            FVK(12, 1, "64x4x12 xg1 lowreg ylds3 minw4 nt")
            FVK(16, 1, "64x4x16 xg1 lowreg ylds3 minw4 nt")
            FVK(4, 2, "64x4x4 xg2 lowreg ylds3 minw4 nt")
            FVK(6, 2, "64x4x6 xg2 lowreg ylds3 minw4 nt")
            FVK(10, 2, "64x4x10 xg2 lowreg ylds3 minw4 nt")
            FVK(12, 2, "64x4x12 xg2 lowreg ylds3 minw4 nt")
            FVK(8, 3, "64x4x8 xg3 lowreg ylds3 minw4 nt")
            FVK(4, 1, "64x4x4 xg1 lowreg ylds3 minw4 nt")
            FVK(5, 1, "64x4x5 xg1 lowreg ylds3 minw4 nt")
            FVK(7, 1, "64x4x7 xg1 lowreg ylds3 minw4 nt")
            FVK(6, 0, "64x4x6 xg0(plain order) lowreg ylds3 minw4 nt")
            FVK(8, 0, "64x4x8 xg0(plain order) lowreg ylds3 minw4 nt")
            FVK(16, 0, "64x4x16 xg0(plain order) lowreg ylds3 minw4 nt")
#define FVK8(KZ_, XG_, NM_)                                                                                                              \
            {                                                                                                                                    \
                const int ntzk = (nz + KZ_ - 1) / KZ_;                                                                                           \
                auto f_ = [&] { hipLaunchKernelGGL((k_fused3d<64, 8, KZ_, 4, 1, true, XG_, false, true, 3, 1>), dim3(ntx * nty8 * ntzk), dim3(512), 0, 0, b, bc, ntx, nty8); }; \
                report(NM_, T.run(reps, f_), 360.0, 0);                                                                                          \
            }
            FVK8(8, 1, "64x8x8 xg1 lowreg ylds3 minw4 nt")
            FVK8(6, 1, "64x8x6 xg1 lowreg ylds3 minw4 nt")
            FVK8(8, 2, "64x8x8 xg2 lowreg ylds3 minw4 nt")
            FVK8(12, 1, "64x8x12 xg1 lowreg ylds3 minw4 nt")
            {
                const int ntzk = (nz + 7) / 8;
                auto f1 = [&] { hipLaunchKernelGGL((k_fused3d<64, 4, 8, 2, 1, false, 1, false, true, 1, 1>), dim3(ntx * nty * ntzk), dim3(256), 0, 0, b, bc, ntx, nty); };
                report("64x4x8 xg1 ylds1 minw2 (145 VGPRs, carried planes) nt", T.run(reps, f1), 360.0, 0);
                auto f2 = [&] { hipLaunchKernelGGL((k_fused3d<64, 4, 8, 3, 1, true, 1, false, true, 3, 1>), dim3(ntx * nty * ntzk), dim3(256), 0, 0, b, bc, ntx, nty); };
                report("64x4x8 xg1 lowreg ylds3 minw3 nt", T.run(reps, f2), 360.0, 0);
                auto f3 = [&] { hipLaunchKernelGGL((k_fused3d<64, 4, 8, 4, 1, true, 1, false, true, 3, 0>), dim3(ntx * nty * ntzk), dim3(256), 0, 0, b, bc, ntx, nty); };
                report("64x4x8 xg1 lowreg ylds3 minw4 (temporal stores)", T.run(reps, f3), 360.0, 0);
                auto f4 = [&] { hipLaunchKernelGGL((k_fused3d<64, 4, 8, 4, 1, true, 1, false, true, 1, 1>), dim3(ntx * nty * ntzk), dim3(256), 0, 0, b, bc, ntx, nty); };
                report("64x4x8 xg1 lowreg ylds1 minw4 nt", T.run(reps, f4), 360.0, 0);
            }
            FVK(8, 1, "64x4x8 xg1 (shipped now)")
            FVK(6, 1, "64x4x6 xg1 again")
            FVK(8, 2, "64x4x8 xg2 again")
            FVK(16, 8, "shipped again")
            report("fused 64x4x16 xg8 SHFL+YLDS (again)", T.run(reps, fy), 360.0, 0);
            report("fused 64x4x16 xg8 SHFL (again)", T.run(reps, fn), 360.0, 0);
        }
        FEXP(0)
    }
    printf("done\n");
    return 0;
}
