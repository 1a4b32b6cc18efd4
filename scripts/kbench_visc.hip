// kbench_visc.hip -- A/B harness for the viscous-limit form of k_fused3d (development tool): tile shapes, chunk depths, XCD banding.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I justrelax.jl_amd/csrc scripts/kbench_visc.hip -o scripts/kbench_visc
//   ./scripts/kbench_visc [n=512] [reps=20]
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

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 512, reps = argc > 2 ? atoi(argv[2]) : 20;
    const int nx = argc > 3 ? atoi(argv[3]) : n, ny = n, nz = n;      // optional third argument: nx alone (ragged last tile experiments)
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
    // the viscous-limit form never touches these: leave them NULL so that a stray load faults
    double *etatau;
    CK(hipMalloc(&etatau, nc * sizeof(double)));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, etatau, nc, 99u, 0.5, 1.5, 0);
    SweepArgs a;
    a.f = f; a.etatau = etatau; a._dx = 51.2; a._dy = 49.0; a._dz = 47.5; a.dt = INFINITY; a.r = 0.7; a.theta_dtau = 191.3; a.eta_dtau = 0.0119;
    a.L = make_lay(nx, ny, nz);
    a.i0 = a.j0 = a.k0 = 0;
    Out10 dst, ref;
    const i64 dn[10] = {nc, nc, nc, nc, nyz, nxz, nxy, nvx, nvy, nvz};
    double **dp[10] = {&dst.P, &dst.txx, &dst.tyy, &dst.tzz, &dst.tyz, &dst.txz, &dst.txy, &dst.Vx, &dst.Vy, &dst.Vz};
    double **rp[10] = {&ref.P, &ref.txx, &ref.tyy, &ref.tzz, &ref.tyz, &ref.txz, &ref.txy, &ref.Vx, &ref.Vy, &ref.Vz};
    for (int q = 0; q < 10; q++) {
        CK(hipMalloc(dp[q], dn[q] * sizeof(double)));
        CK(hipMalloc(rp[q], dn[q] * sizeof(double)));
        CK(hipMemset(*dp[q], 0, dn[q] * sizeof(double)));
        CK(hipMemset(*rp[q], 0, dn[q] * sizeof(double)));
    }
    unsigned long long *d_cnt;
    CK(hipMalloc(&d_cnt, 8));
    CK(hipDeviceSynchronize());
    FusedBC bc;
    memset(&bc, 0, sizeof(bc));
    bc.fsL = bc.fsF = bc.fsK0 = 1;
    Timer T;
    printf("kbench_visc nx=%d n=%d reps=%d   (200 B/cell needed, 280 B/cell = the two sweeps without the operands of the zero factors)\n", nx, n, reps);
    {   // streaming ceilings for the stream mixes of the two forms (the written arrays are the ten outputs: they are rewritten by every variant below)
        std::vector<const double *> rd;
        for (auto &e : ents) if (e.n >= nc) rd.push_back(*e.p);
        for (int q = 0; q < 10; q++) rd.push_back(*rp[q]);
        rd.push_back(etatau);
#define STREAM(NR, NW, NT)                                                                                           \
        {                                                                                                            \
            StreamArgs<NR, NW, NT> sa;                                                                               \
            for (int q = 0; q < NR; q++) sa.r[q] = rd[q % rd.size()];                                                \
            for (int q = 0; q < NW; q++) sa.w[q] = *dp[q];                                                           \
            sa.n = nc;                                                                                               \
            const double ms = T.run(reps, [&] { hipLaunchKernelGGL((k_stream<NR, NW, NT>), dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, 0, sa); }); \
            printf("stream %2dR + %2dW nt%d                      %8.3f ms  %6.0f GB/s\n", NR, NW, NT, ms, (NR + NW) * 8.0 * cells / (ms * 1e-3) / 1e9); \
            fflush(stdout);                                                                                          \
        }
        STREAM(15, 10, 1) STREAM(15, 10, 0) STREAM(25, 10, 1) STREAM(25, 10, 0) STREAM(14, 3, 1) STREAM(21, 7, 1) STREAM(25, 0, 0) STREAM(15, 0, 0) STREAM(1, 10, 1) STREAM(1, 10, 0) STREAM(1, 1, 1)
    }
    for (int q = 0; q < 10; q++) CK(hipMemset(*dp[q], 0, dn[q] * sizeof(double)));     // the streams wrote into them; no variant writes the outer shell of V
    bool have_ref = false;
    auto finish = [&](const char *name, double ms) {
        unsigned long long tot = 0;
        if (have_ref)
            for (int q = 0; q < 10; q++) {
                CK(hipMemset(d_cnt, 0, 8));
                hipLaunchKernelGGL(k_ndiff, dim3(4096), dim3(256), 0, 0, *dp[q], *rp[q], dn[q], d_cnt);
                unsigned long long c;
                CK(hipMemcpy(&c, d_cnt, 8, hipMemcpyDeviceToHost));
                tot += c;
            }
        printf("%-40s %8.3f ms  needed %6.0f GB/s  frac(280 B) %.3f  mismatches %llu\n", name, ms, 200.0 * cells / (ms * 1e-3) / 1e9, 280.0 * cells / (ms * 1e-3) / 1e9 / 8000.0, tot);
        fflush(stdout);
    };
#define V(TX, TY, KZ, MW, LR, XG, YL, NT)                                                                                            \
    {                                                                                                                                \
        SweepArgs b = a; b.o = have_ref ? dst : ref;                                                                                 \
        const int ntx = (nx + TX - 3) / (TX - 2), nty = (ny + TY - 2) / (TY - 1), ntz = (nz + KZ - 1) / KZ;                          \
        auto fn = [&] { hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, MW, 1, LR, XG, false, true, YL, NT, 0, true>), dim3(ntx * nty * ntz), dim3(TX * TY), 0, 0, b, bc, ntx, nty, 0, 0, 0); }; \
        const double ms = T.run(reps, fn);                                                                                           \
        finish(#TX "x" #TY "x" #KZ " minw" #MW " lowreg=" #LR " xg" #XG " ylds" #YL " nt" #NT, ms);                                  \
        have_ref = true;                                                                                                             \
    }
    V(64, 4, 8, 4, false, 1, 3, 1)      // shipped (reference for the comparisons)
    if (argc > 3) { V(64, 4, 8, 4, false, 1, 3, 1) V(32, 8, 8, 4, false, 1, 3, 1) printf("done\n"); return 0; }
    if (n < 200) {
        V(64, 4, 4, 4, false, 1, 3, 1)
        V(64, 4, 2, 4, false, 1, 3, 1)
        V(64, 4, 1, 4, false, 1, 3, 1)
        V(64, 4, 4, 4, false, 0, 3, 1)
        V(64, 4, 2, 4, false, 0, 3, 1)
        V(64, 2, 4, 4, false, 1, 3, 1)
        V(64, 2, 2, 4, false, 1, 3, 1)
        V(64, 2, 8, 4, false, 1, 3, 1)
        V(64, 3, 4, 4, false, 1, 3, 1)
        V(64, 4, 8, 4, false, 1, 3, 1)
        printf("done\n");
        return 0;
    }
    V(64, 4, 8, 4, true, 1, 3, 1)
    V(64, 4, 8, 4, false, 1, 3, 1)
    V(64, 4, 8, 4, true, 1, 2, 1)
    V(64, 4, 8, 4, true, 1, 1, 1)
    V(64, 4, 8, 4, true, 1, 3, 3)
    V(64, 4, 8, 4, true, 1, 3, 0)
    V(64, 4, 16, 4, true, 1, 3, 1)
    V(64, 4, 12, 4, true, 1, 3, 1)
    V(64, 4, 16, 4, true, 8, 3, 1)
    V(64, 4, 8, 4, true, 2, 3, 1)
    V(64, 4, 8, 4, true, 8, 3, 1)
    V(64, 6, 8, 4, true, 1, 3, 1)
    V(64, 6, 8, 3, true, 1, 3, 1)
    V(64, 6, 16, 4, true, 1, 3, 1)
    V(64, 8, 8, 4, true, 1, 3, 1)
    V(64, 8, 8, 2, true, 1, 3, 1)
    V(64, 8, 16, 4, true, 1, 3, 1)
    V(64, 8, 8, 4, true, 2, 3, 1)
    V(64, 8, 8, 4, false, 1, 3, 1)
    V(64, 12, 8, 2, true, 1, 3, 1)
    V(64, 16, 8, 2, true, 1, 3, 1)
    V(32, 8, 8, 4, true, 1, 3, 1)
    V(32, 16, 8, 4, true, 1, 3, 1)
    V(64, 4, 8, 4, true, 1, 3, 1)
    {   // two iterations as a wavefront of z slabs: iteration m on slab s, then iteration m+1 on slab s-1, so that what iteration m wrote is re-read while it may still be in the
        // 256 MiB Infinity Cache (timing experiment only: the boundary-layer launches between the iterations are left out; A -> B -> A ping-pong on the harness' arrays)
        constexpr int TX = 64, TY = 4, KZ = 8;
        const int ntx = (nx + TX - 3) / (TX - 2), nty = (ny + TY - 2) / (TY - 1), ntz = (nz + KZ - 1) / KZ;
        SweepArgs ab = a; ab.o = dst;                          // iteration m: harness inputs -> dst
        SweepArgs ba = a;                                      // iteration m+1: dst -> ref (the state arrays of a.f are read-only inputs of the other variants)
        ba.f.P = dst.P; ba.f.txx = dst.txx; ba.f.tyy = dst.tyy; ba.f.tzz = dst.tzz; ba.f.tyz = dst.tyz; ba.f.txz = dst.txz; ba.f.txy = dst.txy;
        ba.f.Vx = dst.Vx; ba.f.Vy = dst.Vy; ba.f.Vz = dst.Vz; ba.o = ref;
        auto full2 = [&] {
            hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, 4, 1, false, 1, false, true, 3, 1, 0, true>), dim3(ntx * nty * ntz), dim3(TX * TY), 0, 0, ab, bc, ntx, nty, 0, 0, 0);
            hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, 4, 1, false, 1, false, true, 3, 1, 0, true>), dim3(ntx * nty * ntz), dim3(TX * TY), 0, 0, ba, bc, ntx, nty, 0, 0, 0);
        };
        printf("two iterations, two full launches          %8.3f ms per iteration\n", T.run(reps, full2) / 2);
        for (int D : {1, 2, 4, 8, 16}) {
            auto wave = [&] {
                for (int s0 = 0; s0 < ntz + D; s0 += D) {
                    const int d1 = s0 < ntz ? (ntz - s0 < D ? ntz - s0 : D) : 0;
                    if (d1 > 0) hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, 4, 1, false, 1, false, true, 3, 1, 0, true>), dim3(ntx * nty * d1), dim3(TX * TY), 0, 0, ab, bc, ntx, nty, 0, 0, s0);
                    const int t0 = s0 - D;
                    if (t0 >= 0) {
                        const int d2 = ntz - t0 < D ? ntz - t0 : D;
                        hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, 4, 1, false, 1, false, true, 3, 1, 0, true>), dim3(ntx * nty * d2), dim3(TX * TY), 0, 0, ba, bc, ntx, nty, 0, 0, t0);
                    }
                }
            };
            printf("two iterations, wavefront of %2d-chunk slabs %8.3f ms per iteration\n", D, T.run(reps, wave) / 2);
            fflush(stdout);
        }
    }
    printf("done\n");
    return 0;
}
