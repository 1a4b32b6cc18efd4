// kbench_general.hip -- (derived from kbench_place.hip) tile order / shape variants of the GENERAL form of k_fused3d (finite dt: every operand and body-force array loaded, 360 B/cell of SURVEY 8d)
// kbench_place.hip -- the headline form of k_fused3d (VISC, HIF, VFOLD, NOF = 2) under three physical backings of EVERY array (argv[3]: 0 hipMalloc, 2 physically contiguous,
//   1 shuffled 2 MiB chunks through hipMemCreate / hipMemMap): which tile order / tile shape is least sensitive to the placement? (round 5)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I justrelax.jl_amd/csrc scripts/kbench_place.hip -o scripts/kbench_place
//   ./scripts/kbench_place [n=512] [reps=20] [backing=0]
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


static int g_mode = 0;
static void dalloc(void **p, size_t bytes)
{
    if (g_mode == 2) { CK(hipExtMallocWithFlags(p, bytes, hipDeviceMallocContiguous)); return; }
    if (g_mode == 1) {
        static std::vector<hipMemGenericAllocationHandle_t> spare;
        static unsigned long long rng = 0x9E3779B97F4A7C15ull;
        const size_t chunk = (size_t)2 << 20, nch = (bytes + chunk - 1) / chunk;
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        while (spare.size() < nch + 4096) { hipMemGenericAllocationHandle_t hd; CK(hipMemCreate(&hd, chunk, &prop, 0)); spare.push_back(hd); }
        for (size_t i = spare.size(); i > 1; i--) { rng = rng * 6364136223846793005ull + 1442695040888963407ull; std::swap(spare[i - 1], spare[(rng >> 17) % i]); }
        CK(hipMemAddressReserve(p, nch * chunk, chunk, nullptr, 0));
        for (size_t c = 0; c < nch; c++) { CK(hipMemMap((char *)*p + c * chunk, chunk, 0, spare.back(), 0)); spare.pop_back(); }
        hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(*p, nch * chunk, &acc, 1));
        return;
    }
    CK(hipMalloc(p, bytes));
}
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
    g_mode = argc > 3 ? atoi(argv[3]) : 0;
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
        {&f.eta, nc, -3, 0, 1}, {&f.fx, nc, -1, 1, 0}, {&f.fy, nc, -1, 1, 0}, {&f.fz, nc, -1, 1, 0},
        {&f.P0, nc, -1, 1, 0}, {&f.Q, nc, -0.1, 0.1, 0}, {&f.K, nc, 1, 3, 0}, {&f.G, nc, 0.5, 1.5, 0}, {&f.toxx, nc, -1, 1, 0}, {&f.toyy, nc, -1, 1, 0}, {&f.tozz, nc, -1, 1, 0},
        {&f.toyz, nyz, -1, 1, 0}, {&f.toxz, nxz, -1, 1, 0}, {&f.toxy, nxy, -1, 1, 0}};
    unsigned seed = 1;
    for (auto &e : ents) {
        dalloc((void **)e.p, e.n * sizeof(double));
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, *e.p, e.n, seed++, e.lo, e.hi, e.expo);
    }
    // the viscous-limit form never touches these: leave them NULL so that a stray load faults
    double *etatau;
    dalloc((void **)&etatau, nc * sizeof(double));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, etatau, nc, 99u, 0.5, 1.5, 0);
    SweepArgs a;
    a.f = f; a.etatau = etatau; a._dx = 51.2; a._dy = 49.0; a._dz = 47.5; a.dt = 0.25; a.r = 0.7; a.theta_dtau = 191.3; a.eta_dtau = 0.0119;
    a.L = make_lay(nx, ny, nz);
    a.i0 = a.j0 = a.k0 = 0;
    Out10 dst, ref;
    const i64 dn[10] = {nc, nc, nc, nc, nyz, nxz, nxy, nvx, nvy, nvz};
    double **dp[10] = {&dst.P, &dst.txx, &dst.tyy, &dst.tzz, &dst.tyz, &dst.txz, &dst.txy, &dst.Vx, &dst.Vy, &dst.Vz};
    double **rp[10] = {&ref.P, &ref.txx, &ref.tyy, &ref.tzz, &ref.tyz, &ref.txz, &ref.txy, &ref.Vx, &ref.Vy, &ref.Vz};
    for (int q = 0; q < 10; q++) {
        dalloc((void **)dp[q], dn[q] * sizeof(double));
        dalloc((void **)rp[q], dn[q] * sizeof(double));
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
    printf("kbench_place n=%d reps=%d backing=%d (0 hipMalloc, 1 shuffled 2 MiB chunks, 2 contiguous)   headline form: 176 B/cell needed, 256 B/cell priced\n", n, reps, g_mode);
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
        printf("%-40s %8.3f ms  needed %6.0f GB/s  frac(360 B) %.3f  mismatches %llu\n", name, ms, 280.0 * cells / (ms * 1e-3) / 1e9, 360.0 * cells / (ms * 1e-3) / 1e9 / 8000.0, tot);
        fflush(stdout);
    };
#define V(TX, TY, KZ, MW, XG)                                                                                                        \
    {                                                                                                                                \
        SweepArgs b = a; b.o = have_ref ? dst : ref;                                                                                 \
        const int ntx = (nx + TX - 3) / (TX - 2), nty = (ny + TY - 2) / (TY - 1), ntz = (nz + KZ - 1) / KZ;                          \
        auto fn = [&] { hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, MW, 1, true, XG, false, true, 3, 1, 0>), dim3(ntx * nty * ntz), dim3(TX * TY), 0, 0, b, bc, ntx, nty, 0, 0, 0); }; \
        const double ms = T.run(reps, fn);                                                                                           \
        finish(#TX "x" #TY "x" #KZ " minw" #MW " xg" #XG, ms);                                                                       \
        have_ref = true;                                                                                                             \
    }
    for (int pass = 0; pass < 2; pass++) {
    V(64, 4, 8, 4, 1)      // shipped (reference for the comparisons)
    V(64, 4, 8, 4, 2)
    V(64, 4, 8, 4, 4)
    V(64, 4, 8, 4, 8)
    V(64, 4, 16, 4, 1)
    V(64, 4, 16, 4, 4)
    V(64, 8, 8, 4, 1)
    V(64, 8, 8, 4, 4)
    V(64, 8, 16, 4, 4)
    V(64, 4, 8, 4, 1)
    }
    printf("done\n");
    return 0;
}
