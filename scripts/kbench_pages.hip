// kbench_pages.hip -- the headline kernel (512^3) on arrays built page by page (2 MiB chunks of the virtual-memory API) out of ONE pool of chunks created in sequence: which
// ORDER of the pool's chunks under an array is fast?  (A physically contiguous array is the slowest placement there is, randomly shuffled 2 MiB chunks are a good one; is there a
// better one than random?)  The pool's creation order is taken for physical order -- an assumption the "identity" row tests: it should then behave like contiguous memory.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I justrelax.jl_amd/csrc scripts/kbench_pages.hip -o scripts/kbench_pages ; ./scripts/kbench_pages
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
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
static hipMemAllocationProp prop = {};
static hipMemAccessDesc acc = {};
static void flush() { void *t = nullptr; CK(hipHostMalloc(&t, 4096, hipHostMallocDefault)); CK(hipHostFree(t)); }
int main(int argc, char **argv)
{
    const int n = 512, nx = n, ny = n, nz = n;
    constexpr int TX = 64, TY = 8, KZ = 8;
    const int ntx = (nx + TX - 3) / (TX - 2), nty = (ny + TY - 2) / (TY - 1), ntz = (nz + KZ - 1) / KZ;
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t PG = (size_t)2 << 20;
    const int NA = 22, NP = 520;                       // pages per array (1,040 MiB: the largest array is 1,034 MiB)
    const int M = argc > 1 ? atoi(argv[1]) : 16384;    // pages in the pool (32 GiB)
    std::vector<hipMemGenericAllocationHandle_t> pool(M);
    for (auto &c : pool) CK(hipMemCreate(&c, PG, &prop, 0));
    std::vector<void *> va(NA);
    for (int k = 0; k < NA; k++) CK(hipMemAddressReserve(&va[k], NP * PG, 0, nullptr, 0));
    {   // benign values in every page of the pool (through the first array's range, 520 pages at a time)
        for (int b = 0; b < M; b += NP) {
            const int cnt = std::min(NP, M - b);
            for (int j = 0; j < cnt; j++) CK(hipMemMap((char *)va[0] + j * PG, PG, 0, pool[b + j], 0));
            CK(hipMemSetAccess(va[0], cnt * PG, &acc, 1)); flush();
            hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, (double *)va[0], (i64)(cnt * PG / 8));
            CK(hipDeviceSynchronize());
            CK(hipMemUnmap(va[0], cnt * PG));
        }
    }
    jrx_stokes3d_fields f;
    memset(&f, 0, sizeof(f));
    double *etatau;
    Out10 dst;
    double **slot[22] = {&f.P, &f.Vx, &f.Vy, &f.Vz, &f.txx, &f.tyy, &f.tzz, &f.tyz, &f.txz, &f.txy, &f.eta, &etatau, &dst.P, &dst.txx, &dst.tyy, &dst.tzz, &dst.tyz, &dst.txz, &dst.txy, &dst.Vx, &dst.Vy, &dst.Vz};
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
    bool mapped = false;
    auto apply = [&](const char *name, const std::function<long(int, int)> &g) {
        CK(hipDeviceSynchronize());
        if (mapped) for (int k = 0; k < NA; k++) CK(hipMemUnmap(va[k], NP * PG));
        std::vector<char> used(M, 0);
        for (int k = 0; k < NA; k++) {
            for (int j = 0; j < NP; j++) {
                const long c = g(k, j);
                if (c < 0 || c >= M || used[c]) { printf("%s: bad map (array %d page %d -> %ld)\n", name, k, j, c); exit(1); }
                used[c] = 1;
                CK(hipMemMap((char *)va[k] + (size_t)j * PG, PG, 0, pool[c], 0));
            }
            CK(hipMemSetAccess(va[k], NP * PG, &acc, 1));
        }
        flush();
        mapped = true;
        const double t1 = timeit(4), t2 = timeit(4);
        printf("%-58s %.3f %.3f ms\n", name, t1, t2);
        fflush(stdout);
    };
    const long NT = (long)NA * NP;      // 11,440 pages in use
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    apply("identity: array k = pages k*520 .. k*520+519", [&](int k, int j) { return (long)k * NP + j; });
    {
        std::vector<long> perm(NT); for (long i = 0; i < NT; i++) perm[i] = i;
        for (long i = NT - 1; i > 0; i--) std::swap(perm[i], perm[rnd() % (i + 1)]);
        apply("random over the first 11,440 pages", [&](int k, int j) { return perm[(long)k * NP + j]; });
        std::vector<long> big(M); for (long i = 0; i < M; i++) big[i] = i;
        for (long i = M - 1; i > 0; i--) std::swap(big[i], big[rnd() % (i + 1)]);
        apply("random over the whole pool", [&](int k, int j) { return big[(long)k * NP + j]; });
    }
    apply("arrays interleaved page by page: page j of array k = j*22 + k", [&](int k, int j) { return (long)j * NA + k; });
    apply("reversed inside each array", [&](int k, int j) { return (long)k * NP + (NP - 1 - j); });
    for (int st : {3, 7, 9, 11, 17, 33, 63, 129, 257}) {
        char nm[96]; snprintf(nm, sizeof nm, "inside each array page j -> (j * %d) mod 520", st);
        apply(nm, [&](int k, int j) { return (long)k * NP + ((long)j * st) % NP; });
    }
    apply("inside each array page j -> (j mod 8) * 65 + j / 8", [&](int k, int j) { return (long)k * NP + (j % 8) * 65 + j / 8; });
    apply("identity, arrays spread over the pool (array k at k * (M / 22))", [&](int k, int j) { return (long)k * (M / NA) + j; });
    for (int st : {3, 17, 257}) {
        char nm[96]; snprintf(nm, sizeof nm, "over all 11,440 pages: global page i -> (i * %d) mod 11,440", st == 3 ? 3 : st == 17 ? 17 : 257);
        apply(nm, [&](int k, int j) { return (((long)k * NP + j) * st) % NT; });
    }
    apply("identity again", [&](int k, int j) { return (long)k * NP + j; });
    return 0;
}
