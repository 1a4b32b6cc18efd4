// kbench_x2.hip -- validation + timing of the two-iterations-per-launch kernel (scripts/fused_x2.hpp) against two iterations of the shipped path
// (k_fused3d<..., VISC> + the boundary-layer launch k_stress3d_boxes with the flow_bcs! rules), bit for bit on all ten state arrays.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I include -I justrelax.jl_amd/csrc -I scripts scripts/kbench_x2.hip -o scripts/kbench_x2
//   ./scripts/kbench_x2 [nx=512] [ny=nx] [nz=nx] [reps=10] [bc: 0 free slip | 1 no slip | 2 none (prescribed) | 3 mixed]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "jrx_internal.hpp"
#include "stokes3d_kernels.hpp"
#include "fused_x2.hpp"

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
__global__ void k_ndiff(const double *a, const double *b, i64 n, unsigned long long *out, long long *first)
{
    unsigned long long m = 0;
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x)
        if (__double_as_longlong(a[t]) != __double_as_longlong(b[t])) { m += 1; atomicMin((unsigned long long *)first, (unsigned long long)t); }
    if (m) atomicAdd(out, m);
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

static Out10 alloc_set(const i64 dn[10])
{
    Out10 o;
    double **dp[10] = {&o.P, &o.txx, &o.tyy, &o.tzz, &o.tyz, &o.txz, &o.txy, &o.Vx, &o.Vy, &o.Vz};
    for (int q = 0; q < 10; q++) CK(hipMalloc(dp[q], dn[q] * sizeof(double)));
    return o;
}
static void use_state(SweepArgs &a, const Out10 &s)
{
    a.f.P = s.P; a.f.txx = s.txx; a.f.tyy = s.tyy; a.f.tzz = s.tzz; a.f.tyz = s.tyz; a.f.txz = s.txz; a.f.txy = s.txy; a.f.Vx = s.Vx; a.f.Vy = s.Vy; a.f.Vz = s.Vz;
}

int main(int argc, char **argv)
{
    const int nx = argc > 1 ? atoi(argv[1]) : 512, ny = argc > 2 ? atoi(argv[2]) : nx, nz = argc > 3 ? atoi(argv[3]) : nx;
    const int reps = argc > 4 ? atoi(argv[4]) : 10, bck = argc > 5 ? atoi(argv[5]) : 0;
    const double cells = (double)nx * ny * nz;
    const i64 nc = (i64)nx * ny * nz, nvx = (i64)(nx + 1) * (ny + 2) * (nz + 2), nvy = (i64)(nx + 2) * (ny + 1) * (nz + 2),
              nvz = (i64)(nx + 2) * (ny + 2) * (nz + 1), nxy = (i64)(nx + 1) * (ny + 1) * nz, nyz = (i64)nx * (ny + 1) * (nz + 1),
              nxz = (i64)(nx + 1) * ny * (nz + 1);
    const i64 dn[10] = {nc, nc, nc, nc, nyz, nxz, nxy, nvx, nvy, nvz};
    Out10 A = alloc_set(dn), B = alloc_set(dn), C = alloc_set(dn), D = alloc_set(dn);
    double *pa[10] = {A.P, A.txx, A.tyy, A.tzz, A.tyz, A.txz, A.txy, A.Vx, A.Vy, A.Vz};
    double *pb[10] = {B.P, B.txx, B.tyy, B.tzz, B.tyz, B.txz, B.txy, B.Vx, B.Vy, B.Vz};
    double *pc[10] = {C.P, C.txx, C.tyy, C.tzz, C.tyz, C.txz, C.txy, C.Vx, C.Vy, C.Vz};
    double *pd[10] = {D.P, D.txx, D.tyy, D.tzz, D.tyz, D.txz, D.txy, D.Vx, D.Vy, D.Vz};
    const char *names[10] = {"P", "txx", "tyy", "tzz", "tyz", "txz", "txy", "Vx", "Vy", "Vz"};
    for (int q = 0; q < 10; q++) {
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, pa[q], dn[q], 1u + q, -1.0, 1.0, 0);
        CK(hipMemset(pb[q], 0, dn[q] * 8)); CK(hipMemset(pc[q], 0, dn[q] * 8)); CK(hipMemset(pd[q], 0, dn[q] * 8));
    }
    CK(hipDeviceSynchronize());
    // the shells of V (boundary planes with the prescribed normal velocities, ghost planes) are never written by the fused pipeline: every set starts with A's
    for (int q = 7; q < 10; q++) { CK(hipMemcpy(pb[q], pa[q], dn[q] * 8, hipMemcpyDeviceToDevice)); CK(hipMemcpy(pc[q], pa[q], dn[q] * 8, hipMemcpyDeviceToDevice)); CK(hipMemcpy(pd[q], pa[q], dn[q] * 8, hipMemcpyDeviceToDevice)); }
    jrx_stokes3d_fields f;
    memset(&f, 0, sizeof(f));
    double *eta, *fx, *fy, *fz, *etatau;
    for (double **p : {&eta, &fx, &fy, &fz, &etatau}) CK(hipMalloc(p, nc * 8));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, eta, nc, 31u, -3.0, 0.0, 1);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, fx, nc, 32u, -1.0, 1.0, 0);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, fy, nc, 33u, -1.0, 1.0, 0);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, fz, nc, 34u, -1.0, 1.0, 0);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, etatau, nc, 99u, 0.5, 1.5, 0);
    f.eta = eta; f.fx = fx; f.fy = fy; f.fz = fz;
    SweepArgs a;
    a.f = f; a.etatau = etatau; a._dx = 51.2; a._dy = 49.0; a._dz = 47.5; a.dt = INFINITY; a.r = 0.7; a.theta_dtau = 191.3; a.eta_dtau = 0.0119;
    a.L = make_lay(nx, ny, nz);
    a.i0 = a.j0 = a.k0 = 0; a.i1 = a.j1 = a.k1 = 0;
    FusedBC bc;
    memset(&bc, 0, sizeof(bc));
    GhostRule gr;
    // t[face]: 0 none, 1 free slip, 2 no slip; faces x-lo, x-hi, y-lo, y-hi, z-lo, z-hi
    int t6[6] = {0, 0, 0, 0, 0, 0};
    if (bck == 0) for (int q = 0; q < 6; q++) t6[q] = 1;
    if (bck == 1) for (int q = 0; q < 6; q++) t6[q] = 2;
    if (bck == 3) { t6[0] = 1; t6[1] = 2; t6[2] = 2; t6[3] = 0; t6[4] = 0; t6[5] = 1; }
    for (int q = 0; q < 6; q++) gr.t[q] = t6[q];
    bc.fsL = t6[0] == 1; bc.nsL = t6[0] == 2; bc.fsR = t6[1] == 1; bc.nsR = t6[1] == 2; bc.fsF = t6[2] == 1; bc.nsF = t6[2] == 2; bc.fsBk = t6[3] == 1; bc.nsBk = t6[3] == 2;
    bc.fsK0 = t6[4] == 1; bc.nsK0 = t6[4] == 2; bc.fsK1 = t6[5] == 1; bc.nsK1 = t6[5] == 2;
    if (bck == 1 || bck == 3) {
        // flow_bcs! has run at least once in the library: the normal planes of no-slip faces are zero in every set
        // (the fused pipeline relies on it only through the rules, which return 0 themselves)
    }
    StressBoxes SB = {};
    {
        const int planes[3][6] = {{nx, nx + 1, 0, ny + 1, 0, nz + 1}, {0, nx, ny, ny + 1, 0, nz + 1}, {0, nx, 0, ny, nz, nz + 1}};
        int tot = 0;
        for (int q = 0; q < 3; q++) {
            const int *b = planes[q];
            const i64 plane = (i64)(b[1] - b[0]) * (b[3] - b[2]);
            for (int c = 0; c < 6; c++) SB.box[SB.n][c] = b[c];
            SB.per_plane[SB.n] = (int)((plane + 255) / 256);
            SB.start[SB.n] = tot;
            tot += SB.per_plane[SB.n] * (b[5] - b[4]);
            SB.n++;
        }
        SB.start[SB.n] = tot;
    }
    auto shipped_iter = [&](const Out10 &src, const Out10 &dst) {
        constexpr int TX = 64, TY = 4, KZ = 8;
        const int ntx = (nx + TX - 3) / (TX - 2), nty = (ny + TY - 2) / (TY - 1), ntz = (nz + KZ - 1) / KZ;
        SweepArgs s = a;
        use_state(s, src);
        s.o = dst;
        hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, 4, 1, false, 1, false, true, 3, 1, 0, true>), dim3(ntx * nty * ntz), dim3(TX * TY), 0, 0, s, bc, ntx, nty, 0, 0, 0);
        SweepArgs e = s;
        e.f.Vx = dst.Vx; e.f.Vy = dst.Vy; e.f.Vz = dst.Vz;
        hipLaunchKernelGGL((k_stress3d_boxes<false, true, true>), dim3((unsigned)SB.start[SB.n]), dim3(256), 0, 0, e, SB, gr);
    };
    auto hif_iter = [&](const Out10 &src, const Out10 &dst) {
        constexpr int TX = 64, TY = 4, KZ = 8;
        const int ntx = (nx + TX - 3) / (TX - 2), nty = (ny + TY - 2) / (TY - 1), ntz = (nz + KZ - 1) / KZ;
        SweepArgs s = a;
        use_state(s, src);
        s.o = dst;
        hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, 4, 1, false, 1, false, true, 3, 1, 0, true, true>), dim3(ntx * nty * ntz), dim3(TX * TY), 0, 0, s, bc, ntx, nty, 0, 0, 0);
    };
    unsigned long long *d_cnt;
    long long *d_first;
    CK(hipMalloc(&d_cnt, 8)); CK(hipMalloc(&d_first, 8));
    auto compare = [&](double *const *x, double *const *y, const char *what) {
        unsigned long long tot = 0;
        for (int q = 0; q < 10; q++) {
            CK(hipMemset(d_cnt, 0, 8)); CK(hipMemset(d_first, 0x7f, 8));
            hipLaunchKernelGGL(k_ndiff, dim3(4096), dim3(256), 0, 0, x[q], y[q], dn[q], d_cnt, d_first);
            unsigned long long c; long long fi;
            CK(hipMemcpy(&c, d_cnt, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&fi, d_first, 8, hipMemcpyDeviceToHost));
            if (c) {
                // decode the first mismatching index with the array's own extents
                const int e1[10] = {nx, nx, nx, nx, nx, nx + 1, nx + 1, nx + 1, nx + 2, nx + 2}, e2[10] = {ny, ny, ny, ny, ny + 1, ny, ny + 1, ny + 2, ny + 1, ny + 2};
                const long long ii = fi % e1[q], jj = (fi / e1[q]) % e2[q], kk = fi / ((long long)e1[q] * e2[q]);
                printf("   %s: %-3s %llu mismatches, first at (%lld, %lld, %lld)\n", what, names[q], c, ii, jj, kk);
            }
            tot += c;
        }
        printf("%s: %llu mismatching values in total\n", what, tot);
        return tot;
    };
    Timer T;
    printf("kbench_x2 %d x %d x %d reps=%d bc=%d\n", nx, ny, nz, reps, bck);
    shipped_iter(A, B); shipped_iter(B, C);
    CK(hipDeviceSynchronize());
    {   // the folded high-face form (k_fused3d<..., HIF>) against kernel + boundary-layer launch
        hif_iter(A, D);
        CK(hipDeviceSynchronize());
        compare(pd, pb, "k_fused3d<HIF> vs k_fused3d + boxes (one iteration)");
        for (int q = 0; q < 7; q++) CK(hipMemset(pd[q], 0, dn[q] * 8));
        for (int q = 7; q < 10; q++) CK(hipMemcpy(pd[q], pa[q], dn[q] * 8, hipMemcpyDeviceToDevice));
    }
    const double t1 = T.run(reps, [&] { shipped_iter(A, B); });
    const double t1h = T.run(reps, [&] { hif_iter(A, B); });
    printf("shipped: k_fused3d + boundary layers   %8.3f ms per iteration (%.0f it/s);  k_fused3d<HIF> alone %8.3f ms (%.0f it/s)\n", t1, 1e3 / t1, t1h, 1e3 / t1h);
    shipped_iter(A, B);
#define X2(TY, KZ, XG)                                                                                                                         \
    {                                                                                                                                          \
        constexpr int TX = 64;                                                                                                                 \
        const int ntx = (nx + TX - 5) / (TX - 4), nty = (ny + TY - 4) / (TY - 3), ntz = (nz + KZ - 1) / KZ;                                    \
        SweepArgs s = a;                                                                                                                       \
        use_state(s, A);                                                                                                                       \
        s.o = D;                                                                                                                               \
        for (int q = 0; q < 7; q++) CK(hipMemset(pd[q], 0, dn[q] * 8));                                                                         \
        auto fn = [&] { hipLaunchKernelGGL((k_fused3d_x2<TX, TY, KZ, XG>), dim3(ntx * nty * ntz), dim3(TX * TY), 0, 0, s, bc, ntx, nty); };    \
        fn();                                                                                                                                  \
        CK(hipDeviceSynchronize());                                                                                                            \
        char nm[128];                                                                                                                          \
        snprintf(nm, sizeof nm, "k_fused3d_x2<64,%d,%d,xg%d> vs two shipped iterations", TY, KZ, XG);                                         \
        compare(pd, pc, nm);                                                                                                                   \
        const double t2 = T.run(reps, fn);                                                                                                     \
        printf("k_fused3d_x2<64,%d,%d,xg%d>: %8.3f ms per launch = %8.3f ms per iteration (%.0f it/s)  x %.3f per iteration vs k_fused3d + boxes, x %.3f vs k_fused3d<HIF>\n", TY, KZ, XG, t2, \
               t2 / 2, 2e3 / t2, 2.0 * t1 / t2, 2.0 * t1h / t2);                                                                              \
        fflush(stdout);                                                                                                                        \
    }
    X2(12, 16, 1)
    if (reps > 1) { X2(12, 8, 1) X2(12, 32, 1) X2(12, 16, 0) X2(8, 16, 1) X2(10, 16, 1) }
    printf("done %g cells\n", cells);
    return 0;
}
