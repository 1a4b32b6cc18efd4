// kbench_offsets.hip -- the headline kernel on 22 arrays carved out of ONE block of device memory at chosen offsets: which relation between the arrays' addresses makes a launch slow?
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
static double time_at(const std::vector<size_t> &off, int reps)
{
    for (int k = 0; k < 22; k++) if (off[k] + (size_t)cnt[k] * 8 > blk_bytes) return -1.0;
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
    constexpr int TX = 64, TY = 8, KZ = 8;
    auto go = [&] { hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, 2, 1, false, 4, false, true, 3, 1, 0, true, true, true, false, 2>), dim3(ntx * nty * ntz), dim3(TX * TY), 0, 0, a, bc, ntx, nty, 0, 0, 0); };
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
    const int n = argc > 1 ? atoi(argv[1]) : 256;
    const size_t gib = argc > 2 ? (size_t)atoi(argv[2]) : 8;
    const int contiguous = argc > 3 ? atoi(argv[3]) : 1;
    nx = ny = nz = n;
    constexpr int TX = 64, TY = 8, KZ = 8;
    ntx = (nx + TX - 3) / (TX - 2); nty = (ny + TY - 2) / (TY - 1); ntz = (nz + KZ - 1) / KZ;
    const i64 nc = (i64)nx * ny * nz, nvx = (i64)(nx + 1) * (ny + 2) * (nz + 2), nvy = (i64)(nx + 2) * (ny + 1) * (nz + 2), nvz = (i64)(nx + 2) * (ny + 2) * (nz + 1),
              nxy = (i64)(nx + 1) * (ny + 1) * nz, nyz = (i64)nx * (ny + 1) * (nz + 1), nxz = (i64)(nx + 1) * ny * (nz + 1);
    cnt = {nc, nvx, nvy, nvz, nc, nc, nc, nyz, nxz, nxy, nc, nc, nc, nc, nc, nc, nyz, nxz, nxy, nvx, nvy, nvz};
    blk_bytes = gib << 30;
    void *p = nullptr;
    int got_contiguous = 0;
    if (contiguous && hipExtMallocWithFlags(&p, blk_bytes, hipDeviceMallocContiguous) == hipSuccess) got_contiguous = 1;
    else { (void)hipGetLastError(); CK(hipMalloc(&p, blk_bytes)); }
    blk = (char *)p;
    hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, (double *)blk, (i64)(blk_bytes / 8));
    CK(hipDeviceSynchronize());
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t MiB = (size_t)1 << 20, KiB = 1024;
    const size_t S = ((size_t)(*std::max_element(cnt.begin(), cnt.end())) * 8 + 2 * MiB - 1) / (2 * MiB) * (2 * MiB);
    printf("# n %d: block of %zu GiB, %s; largest array %.1f MiB, slot %zu MiB; %d x %d x %d tiles\n", n, gib, got_contiguous ? "physically contiguous" : "plain hipMalloc", *std::max_element(cnt.begin(), cnt.end()) * 8.0 / MiB, S / MiB, ntx, nty, ntz);
    const int reps = n >= 512 ? 6 : 20;
    {   // plain bandwidth inside the block: read of 1 GiB at several places, copy of 1 GiB between places
        double *vout; CK(hipMalloc(&vout, 8));
        printf("## (0) read of 1 GiB at offset X of the block (GB/s):");
        for (size_t X = 0; X + ((size_t)1 << 30) <= blk_bytes; X += std::max(blk_bytes / 8, (size_t)1 << 30)) {
            hipLaunchKernelGGL(k_read, dim3(8192), dim3(256), 0, 0, (const double2 *)(blk + X), (i64)1 << 26, vout);
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < 4; r++) hipLaunchKernelGGL(k_read, dim3(8192), dim3(256), 0, 0, (const double2 *)(blk + X), (i64)1 << 26, vout);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf(" %.0f", 4.0 * (double)((size_t)1 << 30) / (ms * 1e-3) / 1e9);
        }
        printf("\n## (0) copy of 1 GiB from offset 0 to offset X (GB/s read + written):");
        for (size_t X = (size_t)1 << 30; X + ((size_t)1 << 30) <= blk_bytes; X += std::max(blk_bytes / 8, (size_t)1 << 30)) {
            hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, (double2 *)(blk + X), (const double2 *)blk, (i64)1 << 26);
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < 4; r++) hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, (double2 *)(blk + X), (const double2 *)blk, (i64)1 << 26);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf(" %.0f", 8.0 * (double)((size_t)1 << 30) / (ms * 1e-3) / 1e9);
        }
        printf("\n");
    }
    const int brief = argc > 4 ? atoi(argv[4]) : 0;
    auto uniform = [&](size_t base, size_t stride) { std::vector<size_t> o(22); for (int k = 0; k < 22; k++) o[k] = base + k * stride; return o; };
    printf("## (1) array k at k * (slot + d)\n");
    std::vector<size_t> ds = brief ? std::vector<size_t>{0, 2 * MiB, 37 * MiB} : std::vector<size_t>{0, 256, 4 * KiB, 64 * KiB, 128 * KiB, 256 * KiB, 512 * KiB, 768 * KiB, MiB, 1280 * KiB, 1536 * KiB, 2 * MiB, 3 * MiB, 4 * MiB, 6 * MiB, 8 * MiB, 12 * MiB, 16 * MiB, 24 * MiB, 32 * MiB, 48 * MiB, 64 * MiB, 96 * MiB, 128 * MiB, 192 * MiB, 256 * MiB};
    for (size_t d : ds) { const double t = time_at(uniform(0, S + d), reps); if (t > 0) printf("d %9.3f MiB: %.3f ms\n", (double)d / MiB, t); }
    printf("## (2) packed (d = 0), the whole set moved by B\n");
    for (size_t B = 0; B + 22 * S <= blk_bytes; B += (blk_bytes - 22 * S) / (brief ? 6 : 40) / (2 * MiB) * (2 * MiB) + 2 * MiB) printf("B %8.1f MiB: %.3f ms\n", (double)B / MiB, time_at(uniform(B, S), reps));
    printf("## (3) packed, ONE array moved to the end of the block (slot 22 / 23)\n");
    const char *names[22] = {"P", "Vx", "Vy", "Vz", "txx", "tyy", "tzz", "tyz", "txz", "txy", "eta", "etatau", "o.P", "o.txx", "o.tyy", "o.tzz", "o.tyz", "o.txz", "o.txy", "o.Vx", "o.Vy", "o.Vz"};
    { const double t0 = time_at(uniform(0, S), reps); printf("reference %.3f ms\n", t0); }
    for (int k = 0; k < (brief ? 0 : 22); k++) {
        auto o = uniform(0, S);
        o[k] = 22 * S; const double ta = time_at(o, reps);
        o[k] = 23 * S + 37 * MiB; const double tb = time_at(o, reps);
        printf("%-7s at slot 22: %.3f ms   at slot 23 + 37 MiB: %.3f ms\n", names[k], ta, tb);
    }
    printf("## (4) random permutations of the 22 slots (packed) and random 2 MiB-aligned offsets anywhere in the block\n");
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int r = 0; r < 12; r++) {
        std::vector<size_t> o = uniform(0, S);
        for (int i = 21; i > 0; i--) std::swap(o[i], o[rnd() % (i + 1)]);
        printf("permutation %2d: %.3f ms\n", r, time_at(o, reps));
    }
    const size_t slots = blk_bytes / S;
    for (int r = 0; r < 12; r++) {
        std::vector<size_t> all(slots), o(22);
        for (size_t i = 0; i < slots; i++) all[i] = i;
        for (int k = 0; k < 22; k++) { const size_t j = k + rnd() % (slots - k); std::swap(all[k], all[j]); o[k] = all[k] * S; }
        printf("random slots %2d: %.3f ms\n", r, time_at(o, reps));
    }
    return 0;
}
