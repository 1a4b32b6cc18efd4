// kbench_chunks.hip -- is "a good placement" a property of the physical chunks one by one?  The 22 arrays of the headline kernel (512^3) each sit on ONE chunk of the virtual-memory API
// (1,040 MiB); a pool of spare chunks of the same size is cycled under ONE array at a time (the other 21 stay), and the kernel is timed for each.  If a chunk that is slow under txx
// is also slow under Vx, the chunks have a quality of their own and an allocator could rank them one by one; if not, only whole placements can be compared.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I justrelax.jl_amd/csrc scripts/kbench_chunks.hip -o scripts/kbench_chunks ; ./scripts/kbench_chunks [spares=40]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "jrx_internal.hpp"
#include "stokes3d_kernels.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
__global__ void k_fill(double *p, i64 n)
{
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x) {
        unsigned long long x = (unsigned long long)t * 6364136223846793005ULL + 1442695040888963407ULL;
        x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33;
        p[t] = 0.5 + (double)(x >> 11) * (1.0 / 9007199254740992.0);
    }
}
__global__ __launch_bounds__(256) void k_read(const double2 *__restrict__ s, i64 n2, double *out)
{
    double acc = 0.0;
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n2; t += (i64)gridDim.x * blockDim.x) { const double2 v = s[t]; acc += v.x + v.y; }
    if (acc == 12345.678) out[0] = acc;
}
static hipMemAllocationProp prop = {};
static hipMemAccessDesc acc = {};
static void flush() { void *t = nullptr; CK(hipHostMalloc(&t, 4096, hipHostMallocDefault)); CK(hipHostFree(t)); }
static hipEvent_t g_e0, g_e1;
template <int KZ, int MW, int XG>
static double time_variant(const SweepArgs &a, const FusedBC &bc, int ntx, int nty, int nz, int reps)
{
    const int ntz = (nz + KZ - 1) / KZ;
    auto go = [&] { hipLaunchKernelGGL((k_fused3d<64, 8, KZ, MW, 1, false, XG, false, true, 3, 1, 0, true, true, true, false, 2>), dim3(ntx * nty * ntz), dim3(512), 0, 0, a, bc, ntx, nty, 0, 0, 0); };
    go();
    CK(hipEventRecord(g_e0, 0));
    for (int r = 0; r < reps; r++) go();
    CK(hipEventRecord(g_e1, 0));
    CK(hipEventSynchronize(g_e1));
    float ms;
    CK(hipEventElapsedTime(&ms, g_e0, g_e1));
    return (double)ms / reps;
}
int main(int argc, char **argv)
{
    const int spares = argc > 1 ? atoi(argv[1]) : 40;
    const int n = 512, nx = n, ny = n, nz = n;
    constexpr int TX = 64, TY = 8, KZ = 8;
    const int ntx = (nx + TX - 3) / (TX - 2), nty = (ny + TY - 2) / (TY - 1), ntz = (nz + KZ - 1) / KZ;
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t CH = (size_t)1040 << 20;
    const int NA = 22, NC = NA + spares;
    std::vector<hipMemGenericAllocationHandle_t> ch(NC);
    for (auto &c : ch) CK(hipMemCreate(&c, CH, &prop, 0));
    std::vector<void *> va(NA);
    std::vector<int> at(NA);              // which chunk backs array k
    void *tmp = nullptr;
    CK(hipMemAddressReserve(&tmp, CH, 0, nullptr, 0));
    for (int c = NA; c < NC; c++) {       // the spare chunks are filled through a range of their own
        CK(hipMemMap(tmp, CH, 0, ch[c], 0)); CK(hipMemSetAccess(tmp, CH, &acc, 1)); flush();
        hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, (double *)tmp, (i64)(CH / 8));
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(tmp, CH));
    }
    for (int k = 0; k < NA; k++) {
        CK(hipMemAddressReserve(&va[k], CH, 0, nullptr, 0));
        CK(hipMemMap(va[k], CH, 0, ch[k], 0)); CK(hipMemSetAccess(va[k], CH, &acc, 1));
        at[k] = k;
    }
    flush();
    for (int k = 0; k < NA; k++) hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, (double *)va[k], (i64)(CH / 8));
    CK(hipDeviceSynchronize());
    jrx_stokes3d_fields f;
    memset(&f, 0, sizeof(f));
    double *etatau;
    Out10 dst;
    double **slot[22] = {&f.P, &f.Vx, &f.Vy, &f.Vz, &f.txx, &f.tyy, &f.tzz, &f.tyz, &f.txz, &f.txy, &f.eta, &etatau, &dst.P, &dst.txx, &dst.tyy, &dst.tzz, &dst.tyz, &dst.txz, &dst.txy, &dst.Vx, &dst.Vy, &dst.Vz};
    const char *names[22] = {"P", "Vx", "Vy", "Vz", "txx", "tyy", "tzz", "tyz", "txz", "txy", "eta", "etatau", "o.P", "o.txx", "o.tyy", "o.tzz", "o.tyz", "o.txz", "o.txy", "o.Vx", "o.Vy", "o.Vz"};
    for (int k = 0; k < NA; k++) *slot[k] = (double *)va[k];
    SweepArgs a;
    a.f = f; a.etatau = etatau; a._dx = 51.2; a._dy = 49.0; a._dz = 47.5; a.dt = INFINITY; a.r = 0.7; a.theta_dtau = 191.3; a.eta_dtau = 0.0119;
    a.L = make_lay(nx, ny, nz);
    a.i0 = a.j0 = a.k0 = 0;
    a.o = dst;
    FusedBC bc;
    memset(&bc, 0, sizeof(bc));
    bc.fsL = bc.fsF = bc.fsK0 = 1;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto go = [&] { hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, 2, 1, false, 4, false, true, 3, 1, 0, true, true, true, false, 2>), dim3(ntx * nty * ntz), dim3(TX * TY), 0, 0, a, bc, ntx, nty, 0, 0, 0); };
    auto timeit = [&](int reps) { go(); CK(hipEventRecord(e0, 0)); for (int r = 0; r < reps; r++) go(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return (double)ms / reps; };
    auto put = [&](int k, int c) { CK(hipDeviceSynchronize()); CK(hipMemUnmap(va[k], CH)); CK(hipMemMap(va[k], CH, 0, ch[c], 0)); CK(hipMemSetAccess(va[k], CH, &acc, 1)); flush(); at[k] = c; };
    double *vout; CK(hipMalloc(&vout, 8));
    auto readbw = [&](int k) { hipLaunchKernelGGL(k_read, dim3(8192), dim3(256), 0, 0, (const double2 *)va[k], (i64)1 << 26, vout); CK(hipEventRecord(e0, 0));
                               for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_read, dim3(8192), dim3(256), 0, 0, (const double2 *)va[k], (i64)1 << 26, vout);
                               CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return 3.0 * (double)((size_t)1 << 30) / (ms * 1e-3) / 1e9; };
    printf("# 512^3, 22 arrays on one 1,040 MiB chunk each, %d spare chunks; reference %.3f %.3f ms\n", spares, timeit(4), timeit(4));
    const int probes[4] = {4, 1, 13, 19};          // txx (read), Vx (read), o.txx (written), o.Vx (written)
    std::vector<std::vector<double>> T(4, std::vector<double>(spares));
    std::vector<double> bw(spares);
    for (int q = 0; q < 4; q++) {
        const int k = probes[q], home = at[k];
        for (int c = 0; c < spares; c++) {
            put(k, NA + c);
            T[q][c] = timeit(3);
            if (q == 0) bw[c] = readbw(k);
        }
        put(k, home);
        printf("back on its own chunk: %.3f ms\n", timeit(4));
    }
    printf("# spare chunk: ms with it under %s, %s, %s, %s; plain read of the chunk GB/s\n", names[probes[0]], names[probes[1]], names[probes[2]], names[probes[3]]);
    for (int c = 0; c < spares; c++) printf("chunk %2d: %.3f %.3f %.3f %.3f   %.0f\n", c, T[0][c], T[1][c], T[2][c], T[3][c], bw[c]);
    auto corr = [&](const std::vector<double> &x, const std::vector<double> &y) {
        double mx = 0, my = 0; for (int i = 0; i < spares; i++) { mx += x[i]; my += y[i]; } mx /= spares; my /= spares;
        double sxy = 0, sxx = 0, syy = 0; for (int i = 0; i < spares; i++) { sxy += (x[i] - mx) * (y[i] - my); sxx += (x[i] - mx) * (x[i] - mx); syy += (y[i] - my) * (y[i] - my); }
        return sxy / sqrt(sxx * syy + 1e-300); };
    for (int q = 0; q < 4; q++) { double lo = 1e9, hi = 0; for (double t : T[q]) { lo = std::min(lo, t); hi = std::max(hi, t); } printf("under %-6s: min %.3f max %.3f ms\n", names[probes[q]], lo, hi); }
    printf("correlation over the chunks: txx~Vx %.2f  txx~o.txx %.2f  txx~o.Vx %.2f  Vx~o.txx %.2f  o.txx~o.Vx %.2f  txx~read bandwidth %.2f\n", corr(T[0], T[1]), corr(T[0], T[2]), corr(T[0], T[3]), corr(T[1], T[2]), corr(T[2], T[3]), corr(T[0], bw));
    // ---- is the quality additive?  Every chunk of the pool (the 22 in use, too) is timed under txx; then the arrays are put on the 22 best / the 22 worst / 22 random chunks.
    {
        std::vector<double> q(NC, 0.0);
        const int R = 4;                                   // txx
        for (int c = 0; c < NC; c++) {
            // chunk c may be in use by another array: that array takes the chunk txx holds now (which array holds which chunk does not matter, section 11 of the record)
            int holder = -1;
            for (int k = 0; k < NA; k++) if (at[k] == c) holder = k;
            if (holder == R) { q[c] = timeit(6); continue; }
            const int mine = at[R];
            if (holder >= 0) {
                CK(hipDeviceSynchronize());
                CK(hipMemUnmap(va[holder], CH)); CK(hipMemUnmap(va[R], CH));
                CK(hipMemMap(va[holder], CH, 0, ch[mine], 0)); CK(hipMemSetAccess(va[holder], CH, &acc, 1));
                CK(hipMemMap(va[R], CH, 0, ch[c], 0)); CK(hipMemSetAccess(va[R], CH, &acc, 1));
                flush();
                at[holder] = mine; at[R] = c;
            } else put(R, c);
            q[c] = timeit(6);
        }
        std::vector<int> order(NC);
        for (int c = 0; c < NC; c++) order[c] = c;
        std::sort(order.begin(), order.end(), [&](int a_, int b_) { return q[a_] < q[b_]; });
        printf("# every chunk under txx: fastest %.3f ms, median %.3f, slowest %.3f\n", q[order[0]], q[order[NC / 2]], q[order[NC - 1]]);
        auto assemble = [&](const std::vector<int> &pick) {
            CK(hipDeviceSynchronize());
            for (int k = 0; k < NA; k++) CK(hipMemUnmap(va[k], CH));
            for (int k = 0; k < NA; k++) { CK(hipMemMap(va[k], CH, 0, ch[pick[k]], 0)); CK(hipMemSetAccess(va[k], CH, &acc, 1)); at[k] = pick[k]; }
            flush();
            return timeit(8);
        };
        std::vector<int> best(order.begin(), order.begin() + NA), worst(order.end() - NA, order.end()), mid(order.begin() + (NC - NA) / 2, order.begin() + (NC - NA) / 2 + NA);
        printf("arrays on the 22 fastest chunks: %.3f ms\n", assemble(best));
        printf("arrays on the 22 slowest chunks: %.3f ms\n", assemble(worst));
        printf("arrays on the 22 middle chunks:  %.3f ms\n", assemble(mid));
        std::vector<int> rev(best.rbegin(), best.rend());
        printf("the 22 fastest again, dealt in the opposite order: %.3f ms\n", assemble(rev));
        printf("arrays on the 22 fastest chunks: %.3f ms\n", assemble(best));
        // unlike chunks on purpose: evenly spaced through the ranking / through the order of creation, and random picks
        const int stp = NC / NA;
        for (int off = 0; off < std::min(stp, 3); off++) {
            std::vector<int> a1, a2;
            for (int k = 0; k < NA; k++) { a1.push_back(order[off + k * stp]); a2.push_back(off + k * stp); }
            printf("every %d-th chunk of the ranking (from %d): %.3f ms     every %d-th chunk in the order of creation (from %d): %.3f ms\n", stp, off, assemble(a1), stp, off, assemble(a2));
        }
        unsigned long long sd = 88172645463325252ull;
        for (int r = 0; r < 4; r++) {
            std::vector<int> all(NC); for (int c = 0; c < NC; c++) all[c] = c;
            for (int i = NC - 1; i > 0; i--) { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; std::swap(all[i], all[sd % (i + 1)]); }
            printf("22 random chunks: %.3f ms\n", assemble(std::vector<int>(all.begin(), all.begin() + NA)));
        }
        { std::vector<int> a0; for (int k = 0; k < NA; k++) a0.push_back(k); printf("the first 22 chunks in the order of creation (as at the start): %.3f ms\n", assemble(a0)); }
        // kernel variants on a GOOD placement (22 random chunks of the pool): chunk depth, XCD band width, waves per SIMD -- the library runs <64, 8, 12 (8 below nz = 384), MINW 4, XG 4>
        CK(hipEventCreate(&g_e0)); CK(hipEventCreate(&g_e1));
        for (int r = 0; r < 2; r++) {
            std::vector<int> all(NC); for (int c = 0; c < NC; c++) all[c] = c;
            for (int i = NC - 1; i > 0; i--) { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; std::swap(all[i], all[sd % (i + 1)]); }
            printf("22 random chunks: %.3f ms (the harness's own instantiation, MINW 2)\n", assemble(std::vector<int>(all.begin(), all.begin() + NA)));
            for (int q = 0; q < 2; q++) {
                printf("  MINW 4, XG 4: KZ 8 %.3f  KZ 10 %.3f  KZ 12 %.3f  KZ 14 %.3f  KZ 16 %.3f |", time_variant<8, 4, 4>(a, bc, ntx, nty, nz, 6), time_variant<10, 4, 4>(a, bc, ntx, nty, nz, 6),
                       time_variant<12, 4, 4>(a, bc, ntx, nty, nz, 6), time_variant<14, 4, 4>(a, bc, ntx, nty, nz, 6), time_variant<16, 4, 4>(a, bc, ntx, nty, nz, 6));
                printf(" KZ 12: XG 1 %.3f  XG 2 %.3f  XG 4 %.3f  XG 8 %.3f  XG 0 %.3f | MINW 2 %.3f  MINW 3 %.3f\n", time_variant<12, 4, 1>(a, bc, ntx, nty, nz, 6), time_variant<12, 4, 2>(a, bc, ntx, nty, nz, 6),
                       time_variant<12, 4, 4>(a, bc, ntx, nty, nz, 6), time_variant<12, 4, 8>(a, bc, ntx, nty, nz, 6), time_variant<12, 4, 0>(a, bc, ntx, nty, nz, 6), time_variant<12, 2, 4>(a, bc, ntx, nty, nz, 6),
                       time_variant<12, 3, 4>(a, bc, ntx, nty, nz, 6));
            }
        }
    }
    return 0;
}