// kbench_loop.hip -- the headline kernel in a loop, one line per batch with a wall-clock stamp: to be read beside a clock / power log of the same seconds (scripts/gpu_r05_n.sh):
// is the spread of the kernel's rate between processes and within them the device's clocks (node power budget, other GPUs of the node busy) rather than memory placement?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I justrelax.jl_amd/csrc scripts/kbench_loop.hip -o scripts/kbench_loop ; ./scripts/kbench_loop [n=512] [seconds=30]
#include <hip/hip_runtime.h>
#include <chrono>
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
// a pure streaming kernel (copy) for comparison: does it slow down in the same seconds?
__global__ __launch_bounds__(256) void k_copy(double *__restrict__ d, const double *__restrict__ s, i64 n)
{
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x) d[t] = s[t];
}
// read-only sweep of a buffer (16 B per lane): over 128 MiB it is served by the 256 MiB Infinity Cache after the first pass, over 4 GiB by HBM
__global__ __launch_bounds__(256) void k_read(const double2 *__restrict__ s, i64 n2, double *out)
{
    double acc = 0.0;
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n2; t += (i64)gridDim.x * blockDim.x) { const double2 v = s[t]; acc += v.x + v.y; }
    if (acc == 12345.678) out[0] = acc;
}
// dependent loads through a 256 MiB table (a full-period LCG permutation of 2^26 indices): nanoseconds per hop of ONE lane = the idle latency of a load that misses every cache
__global__ void k_perm(unsigned *p, unsigned n) { for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) p[t] = (1664525u * t + 1013904223u) & (n - 1u); }
__global__ void k_chase(const unsigned *__restrict__ p, int hops, unsigned *out)
{
    unsigned i = 12345u;
    for (int h = 0; h < hops; h++) i = __builtin_nontemporal_load(p + i);
    out[0] = i;
}
// the same chase by many independent lanes (one per wave, 8 waves per CU): latency under a moderate load
__global__ __launch_bounds__(64) void k_chase_many(const unsigned *__restrict__ p, int hops, unsigned *out)
{
    unsigned i = (blockIdx.x * 2654435761u) & ((1u << 26) - 1u);
    if (threadIdx.x == 0) { for (int h = 0; h < hops; h++) i = __builtin_nontemporal_load(p + i); out[blockIdx.x] = i; }
}
// where does workgroup b run?  XCC_ID (hwreg 20, bits 3:0) per block: the fused kernel's tile order assumes that blocks are dealt round-robin to the eight XCDs (block b on XCD b % 8)
__global__ __launch_bounds__(256) void k_where(unsigned *out, int spin)
{
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    double a = threadIdx.x;
    for (int i = 0; i < spin; i++) a = fma(a, 1.0000001, 0.5);        // keep the block resident for a while so that the launch fills the chip like the real kernel
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid + (a == 1.5 ? 1u : 0u); }
}
// a pure fp64 VALU kernel: no memory traffic
__global__ __launch_bounds__(256) void k_valu(double *out, int iters)
{
    double a = threadIdx.x * 1e-3, b = 1.000001, c = 0.5;
    for (int i = 0; i < iters; i++) { a = fma(a, b, c); c = fma(c, b, a); b = fma(b, 0.999999, 1e-9); }
    if (a + b + c == 12345.678) out[0] = a;
}
int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 512;
    const double seconds = argc > 2 ? atof(argv[2]) : 30.0;
    const int smooth = argc > 3 ? atoi(argv[3]) : 0;      // 1: every array a constant (the operands of a smooth model: few mantissa bits toggle), 0: random operands
    const int nx = n, ny = n, nz = n;
    jrx_stokes3d_fields f;
    memset(&f, 0, sizeof(f));
    struct Ent { double **p; i64 n; double lo, hi; int expo; };
    const i64 nc = (i64)nx * ny * nz, nvx = (i64)(nx + 1) * (ny + 2) * (nz + 2), nvy = (i64)(nx + 2) * (ny + 1) * (nz + 2),
              nvz = (i64)(nx + 2) * (ny + 2) * (nz + 1), nxy = (i64)(nx + 1) * (ny + 1) * nz, nyz = (i64)nx * (ny + 1) * (nz + 1), nxz = (i64)(nx + 1) * ny * (nz + 1);
    double *etatau;
    Out10 dst;
    std::vector<Ent> ents = {{&f.P, nc, -1, 1, 0}, {&f.Vx, nvx, -1, 1, 0}, {&f.Vy, nvy, -1, 1, 0}, {&f.Vz, nvz, -1, 1, 0}, {&f.txx, nc, -1, 1, 0}, {&f.tyy, nc, -1, 1, 0}, {&f.tzz, nc, -1, 1, 0},
                             {&f.tyz, nyz, -1, 1, 0}, {&f.txz, nxz, -1, 1, 0}, {&f.txy, nxy, -1, 1, 0}, {&f.eta, nc, -3, 0, 1}, {&etatau, nc, 0.5, 1.5, 0},
                             {&dst.P, nc, 0, 0, 0}, {&dst.txx, nc, 0, 0, 0}, {&dst.tyy, nc, 0, 0, 0}, {&dst.tzz, nc, 0, 0, 0}, {&dst.tyz, nyz, 0, 0, 0}, {&dst.txz, nxz, 0, 0, 0}, {&dst.txy, nxy, 0, 0, 0},
                             {&dst.Vx, nvx, 0, 0, 0}, {&dst.Vy, nvy, 0, 0, 0}, {&dst.Vz, nvz, 0, 0, 0}};
    unsigned seed = 1;
    for (auto &e : ents) { CK(hipMalloc(e.p, e.n * 8)); hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, *e.p, e.n, seed++, smooth ? (e.expo ? 0.0 : 0.25 * (e.lo + e.hi) + 0.125) : e.lo, smooth ? (e.expo ? 0.0 : 0.25 * (e.lo + e.hi) + 0.125) : e.hi, e.expo); }
    double *vout; CK(hipMalloc(&vout, 8));
    unsigned *perm, *cout;
    CK(hipMalloc(&perm, (size_t)4 << 26)); CK(hipMalloc(&cout, 4 * 4096));
    hipLaunchKernelGGL(k_perm, dim3(4096), dim3(256), 0, 0, perm, 1u << 26);
    CK(hipDeviceSynchronize());
    SweepArgs a;
    a.f = f; a.etatau = etatau; a._dx = 51.2; a._dy = 49.0; a._dz = 47.5; a.dt = INFINITY; a.r = 0.7; a.theta_dtau = 191.3; a.eta_dtau = 0.0119;
    a.L = make_lay(nx, ny, nz);
    a.i0 = a.j0 = a.k0 = 0;
    a.o = dst;
    FusedBC bc;
    memset(&bc, 0, sizeof(bc));
    bc.fsL = bc.fsF = bc.fsK0 = 1;
    hipEvent_t e0, e1, e2, e3, e4, e5, e6, e7, e8; for (hipEvent_t *e : {&e0, &e1, &e2, &e3, &e4, &e5, &e6, &e7, &e8}) CK(hipEventCreate(e));
    constexpr int TX = 64, TY = 8, KZ = 8;
    const int ntx = (nx + TX - 3) / (TX - 2), nty = (ny + TY - 2) / (TY - 1), ntz = (nz + KZ - 1) / KZ;
    {
        hipDeviceProp_t pr;
        CK(hipGetDeviceProperties(&pr, 0));
        printf("# device: %s, %d CUs, clock %d kHz, memory clock %d kHz, bus %d bit, L2 %d B, %zu B of memory, pci %04x:%02x:%02x\n", pr.gcnArchName, pr.multiProcessorCount, pr.clockRate, pr.memoryClockRate,
               pr.memoryBusWidth, pr.l2CacheSize, pr.totalGlobalMem, pr.pciDomainID, pr.pciBusID, pr.pciDeviceID);
    }
    {
        const int nb = 16384;
        unsigned *w, *hw = (unsigned *)malloc(8 * nb);
        CK(hipMalloc(&w, 8 * nb));
        for (int spin : {0, 20000}) {
            hipLaunchKernelGGL(k_where, dim3(nb), dim3(256), 0, 0, w, spin);
            CK(hipMemcpy(hw, w, 8 * nb, hipMemcpyDeviceToHost));
            int rr = 0, hist[16] = {}, first_bad = -1;
            const unsigned x0 = hw[0];
            for (int b = 0; b < nb; b++) { const unsigned x = hw[2 * b] & 15u; hist[x]++; if (x == ((x0 + b) & 7u)) rr++; else if (first_bad < 0) first_bad = b; }
            printf("# k_where (%d blocks of 256 threads, spin %d): block b on XCC (x0 + b) %% 8 for %d blocks (x0 = %u, first exception at b = %d); blocks per XCC:", nb, spin, rr, x0, first_bad);
            for (int x = 0; x < 8; x++) printf(" %d", hist[x]);
            printf("\n");
            if (spin) {
                std::vector<char> seen(1 << 16, 0);
                int ncu = 0;
                for (int b = 0; b < nb; b++) { const unsigned key = ((hw[2 * b] & 15u) << 8) | ((hw[2 * b + 1] >> 8) & 0xffu); if (!seen[key]) { seen[key] = 1; ncu++; } }
                printf("# distinct (XCC, SE / SH / CU) places the blocks ran on: %d\n", ncu);
            }
        }
        CK(hipFree(w)); free(hw);
    }
    const auto t00 = std::chrono::system_clock::now();
    printf("# operands: %s\n", smooth ? "constants" : "random"); printf("# unix_time  k_fused3d<64,8,8> ms   copy 1 GiB ms (GB/s)   fp64 VALU kernel ms   re-read of 128 MiB GB/s   read of 1 GiB GB/s   ns per dependent load (one lane)   (2,048 lanes)\n");
    while (std::chrono::duration<double>(std::chrono::system_clock::now() - t00).count() < seconds) {
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 20; r++)
            hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, 2, 1, false, 4, false, true, 3, 1, 0, true, true, true, false, 2>), dim3(ntx * nty * ntz), dim3(TX * TY), 0, 0, a, bc, ntx, nty, 0, 0, 0);
        CK(hipEventRecord(e1, 0));
        for (int r = 0; r < 10; r++) hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, dst.P, f.P, nc);
        CK(hipEventRecord(e2, 0));
        hipLaunchKernelGGL(k_valu, dim3(256 * 16), dim3(256), 0, 0, vout, 20000);
        CK(hipEventRecord(e3, 0));
        for (int r = 0; r < 21; r++) { if (r == 1) CK(hipEventRecord(e4, 0)); hipLaunchKernelGGL(k_read, dim3(8192), dim3(256), 0, 0, (const double2 *)f.P, (i64)(128 << 20) / 16, vout); }
        CK(hipEventRecord(e5, 0));
        for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k_read, dim3(8192), dim3(256), 0, 0, (const double2 *)f.Vx, (i64)1 << 26, vout);       // 2 x 1 GiB of another array
        CK(hipEventRecord(e6, 0));
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, perm, 4000, cout);
        CK(hipEventRecord(e7, 0));
        hipLaunchKernelGGL(k_chase_many, dim3(2048), dim3(64), 0, 0, perm, 2000, cout);
        CK(hipEventRecord(e8, 0));
        CK(hipEventSynchronize(e8));
        float m6, m7;
        CK(hipEventElapsedTime(&m6, e6, e7)); CK(hipEventElapsedTime(&m7, e7, e8));
        float m4, m5;
        CK(hipEventElapsedTime(&m4, e4, e5)); CK(hipEventElapsedTime(&m5, e5, e6));
        float m1, m2, m3;
        CK(hipEventElapsedTime(&m1, e0, e1)); CK(hipEventElapsedTime(&m2, e1, e2)); CK(hipEventElapsedTime(&m3, e2, e3));
        const double now = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
        printf("%.2f  %.3f  %.3f (%.0f)  %.3f  %.0f  %.0f  %.0f  %.0f\n", now, m1 / 20, m2 / 10, 2.0 * nc * 8 / (m2 / 10 * 1e-3) / 1e9, m3, 20.0 * (128 << 20) / (m4 * 1e-3) / 1e9, 2.0 * (double)((i64)1 << 30) / (m5 * 1e-3) / 1e9, m6 * 1e6 / 4000.0, m7 * 1e6 / 2000.0);
        fflush(stdout);
    }
    return 0;
}
