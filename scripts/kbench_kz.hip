// kbench_kz.hip -- the headline kernel (64 x 8 tile) with chunk depths KZ = 8 (shipped), 12, 16, 32 on the same arrays of one process (a spread placement), alternating
// The block is physically contiguous where the runtime grants it (hipDeviceMallocContiguous), so offsets in the block are offsets in physical memory.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I justrelax.jl_amd/csrc scripts/kbench_offsets.hip -o scripts/kbench_offsets ; ./scripts/kbench_offsets [n=256] [block_gib=8] [contiguous=1] [brief=0]
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
__global__ __launch_bounds__(256) void k_copy(double2 *__restrict__ d, const double2 *__restrict__ s, i64 n2)
{
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n2; t += (i64)gridDim.x * blockDim.x) d[t] = s[t];
}
static int nx, ny, nz, ntx, nty, ntz;
static hipEvent_t e0, e1;
static char *blk;
static size_t blk_bytes;
static std::vector<i64> cnt;       // elements of the 22 arrays, in the order of `slot`
template <int KZ, int MW = 4, int XG = 4>
static double time_kz(const std::vector<size_t> &off, int reps)
{
    jrx_stokes3d_fields f;
    memset(&f, 0, sizeof(f));
    double *etatau;
    Out10 dst;
    double **slot[22] = {&f.P, &f.Vx, &f.Vy, &f.Vz, &f.txx, &f.tyy, &f.tzz, &f.tyz, &f.txz, &f.txy, &f.eta, &etatau, &dst.P, &dst.txx, &dst.tyy, &dst.tzz, &dst.tyz, &dst.txz, &dst.txy, &dst.Vx, &dst.Vy, &dst.Vz};
    for (int k = 0; k < 22; k++) *slot[k] = (double *)(blk + off[k]);
    SweepArgs a;
    a.f = f; a.etatau = etatau; a._dx = 51.2; a._dy = 49.0; a._dz = 47.5; a.dt = INFINITY; a.r = 0.7; a.theta_dtau = 191.3; a.eta_dtau = 0.0119;
    a.L = make_lay(nx, ny, nz);
    a.i0 = a.j0 = a.k0 = 0;
    a.o = dst;
    FusedBC bc;
    memset(&bc, 0, sizeof(bc));
    bc.fsL = bc.fsF = bc.fsK0 = 1;
    constexpr int TX = 64, TY = 8;
    const int ntz_ = (nz + KZ - 1) / KZ;
    auto go = [&] { hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, MW, 1, false, XG, false, true, 3, 1, 0, true, true, true, false, 2>), dim3(ntx * nty * ntz_), dim3(TX * TY), 0, 0, a, bc, ntx, nty, 0, 0, 0); };
    go();
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; r++) go();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}
int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 512;
    nx = ny = nz = n;
    constexpr int TX = 64, TY = 8;
    ntx = (nx + TX - 3) / (TX - 2); nty = (ny + TY - 2) / (TY - 1); ntz = 0;
    const i64 nc = (i64)nx * ny * nz, nvx = (i64)(nx + 1) * (ny + 2) * (nz + 2), nvy = (i64)(nx + 2) * (ny + 1) * (nz + 2), nvz = (i64)(nx + 2) * (ny + 2) * (nz + 1),
              nxy = (i64)(nx + 1) * (ny + 1) * nz, nyz = (i64)nx * (ny + 1) * (nz + 1), nxz = (i64)(nx + 1) * ny * (nz + 1);
    cnt = {nc, nvx, nvy, nvz, nc, nc, nc, nyz, nxz, nxy, nc, nc, nc, nc, nc, nc, nyz, nxz, nxy, nvx, nvy, nvz};
    const size_t MiB = (size_t)1 << 20;
    const size_t S = ((size_t)(*std::max_element(cnt.begin(), cnt.end())) * 8 + 2 * MiB - 1) / (2 * MiB) * (2 * MiB);
    // 22 separate allocations with 1.5 GiB of unused memory between them: a spread placement
    std::vector<size_t> off(22);
    std::vector<void *> keep;
    char *base = nullptr;
    for (int k = 0; k < 22; k++) {
        void *p = nullptr, *b = nullptr;
        CK(hipMalloc(&p, S)); CK(hipMalloc(&b, (size_t)1536 * MiB)); keep.push_back(b);
        if (k == 0) base = (char *)p;
        off[k] = (size_t)((char *)p - base);            // offsets relative to the first array (may wrap: size_t arithmetic on pointers of one address space)
        hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, (double *)p, (i64)(S / 8));
    }
    blk = base; blk_bytes = ~(size_t)0;
    CK(hipDeviceSynchronize());
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("# n %d, 64 x 8 tiles; k_fused3d<64, 8, KZ, MINW, .., XG, ..>: ms per launch, three rounds (the library runs MINW 4, XG 4)\n", n);
    for (int r = 0; r < 3; r++) {
        printf("KZ 8: %.3f   KZ 10: %.3f   KZ 12: %.3f   KZ 14: %.3f   KZ 16: %.3f   KZ 8: %.3f\n", time_kz<8>(off, 8), time_kz<10>(off, 8), time_kz<12>(off, 8), time_kz<14>(off, 8), time_kz<16>(off, 8), time_kz<8>(off, 8));
        printf("KZ 12 with XG 1: %.3f   XG 2: %.3f   XG 4: %.3f   XG 8: %.3f   XG 0 (plain block order): %.3f   MINW 2 (XG 4): %.3f   MINW 3: %.3f\n", time_kz<12, 4, 1>(off, 8), time_kz<12, 4, 2>(off, 8), time_kz<12, 4, 4>(off, 8),
               time_kz<12, 4, 8>(off, 8), time_kz<12, 4, 0>(off, 8), time_kz<12, 2, 4>(off, 8), time_kz<12, 3, 4>(off, 8));
    }
    return 0;
}
