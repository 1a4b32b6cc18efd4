// kbench_va.hip -- the headline form of k_fused3d as a function of the VIRTUAL layout of its 25 arrays (round 5: re-rolling the physical chunks under fixed virtual addresses changes
//   nothing, new virtual addresses do -- profiles/r05_placement.txt).  One process: every array owns physical chunks (hipMemCreate) once; per configuration the arrays are mapped at
//   chosen offsets of ONE reserved virtual range, the kernel is timed, and everything is unmapped again (no copies: the data stays in the chunks).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I justrelax.jl_amd/csrc scripts/kbench_va.hip -o scripts/kbench_va
//   ./scripts/kbench_va [n=512] [reps=6] [chunk_mib=64] [what=0: strides | 1: random offsets | 2: base shifts | 3: all]
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

struct Arr { double **slot; i64 n; double lo, hi; int expo; std::vector<hipMemGenericAllocationHandle_t> chunks; size_t mapped = 0; size_t off = 0; };

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 512, reps = argc > 2 ? atoi(argv[2]) : 6, chunk_mib = argc > 3 ? atoi(argv[3]) : 64, what = argc > 4 ? atoi(argv[4]) : 3;
    const int nx = n, ny = n, nz = n;
    const double cells = (double)nx * ny * nz;
    jrx_stokes3d_fields f;
    memset(&f, 0, sizeof(f));
    const i64 nc = (i64)nx * ny * nz, nvx = (i64)(nx + 1) * (ny + 2) * (nz + 2), nvy = (i64)(nx + 2) * (ny + 1) * (nz + 2),
              nvz = (i64)(nx + 2) * (ny + 2) * (nz + 1), nxy = (i64)(nx + 1) * (ny + 1) * nz, nyz = (i64)nx * (ny + 1) * (nz + 1), nxz = (i64)(nx + 1) * ny * (nz + 1);
    double *etatau = nullptr;
    Out10 dst;
    std::vector<Arr> A = {
        {&f.P, nc, -1, 1, 0}, {&f.Vx, nvx, -1, 1, 0}, {&f.Vy, nvy, -1, 1, 0}, {&f.Vz, nvz, -1, 1, 0},
        {&f.txx, nc, -1, 1, 0}, {&f.tyy, nc, -1, 1, 0}, {&f.tzz, nc, -1, 1, 0}, {&f.tyz, nyz, -1, 1, 0}, {&f.txz, nxz, -1, 1, 0}, {&f.txy, nxy, -1, 1, 0},
        {&f.eta, nc, -3, 0, 1}, {&etatau, nc, 0.5, 1.5, 0},
        {&dst.P, nc, 0, 0, 0}, {&dst.txx, nc, 0, 0, 0}, {&dst.tyy, nc, 0, 0, 0}, {&dst.tzz, nc, 0, 0, 0}, {&dst.tyz, nyz, 0, 0, 0}, {&dst.txz, nxz, 0, 0, 0}, {&dst.txy, nxy, 0, 0, 0},
        {&dst.Vx, nvx, 0, 0, 0}, {&dst.Vy, nvy, 0, 0, 0}, {&dst.Vz, nvz, 0, 0, 0}};
    const int NA = (int)A.size();
    const size_t M2 = (size_t)2 << 20;
    const size_t arena_bytes = (size_t)1 << 40;      // 1 TiB of address space
    char *arena = nullptr;
    CK(hipMemAddressReserve((void **)&arena, arena_bytes, (size_t)1 << 30, nullptr, 0));
    Timer T;
    for (int chunk_mib_i : (what == 4 ? std::vector<int>{1024, 2, 64, 16, 256, 1024, 2, 64} : (what == 5 ? std::vector<int>{2, 64} : std::vector<int>{chunk_mib}))) {
    const size_t chunk = (size_t)chunk_mib_i << 20;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    size_t slot_max = 0;
    for (auto &a : A) {
        const size_t nch = ((size_t)a.n * 8 + chunk - 1) / chunk;
        a.chunks.resize(nch);
        for (auto &hd : a.chunks) CK(hipMemCreate(&hd, chunk, &prop, 0));
        a.mapped = nch * chunk;
        slot_max = a.mapped > slot_max ? a.mapped : slot_max;
    }
    auto map_all = [&]() {
        for (auto &a : A) {
            char *va = arena + a.off;
            for (size_t c = 0; c < a.chunks.size(); c++) CK(hipMemMap(va + c * chunk, chunk, 0, a.chunks[c], 0));
            CK(hipMemSetAccess(va, a.mapped, &acc, 1));
            *a.slot = (double *)va;
        }
    };
    auto unmap_all = [&]() { CK(hipDeviceSynchronize()); for (auto &a : A) CK(hipMemUnmap(arena + a.off, a.mapped)); };
    // first layout: plain packing; fill the data once (it lives in the chunks)
    { size_t at = 0; for (auto &a : A) { a.off = at; at += a.mapped; } }
    map_all();
    unsigned seed = 1;
    for (auto &a : A) {
        if (a.lo == 0 && a.hi == 0) CK(hipMemset(*a.slot, 0, a.n * 8));
        else hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, *a.slot, a.n, seed++, a.lo, a.hi, a.expo);
    }
    CK(hipDeviceSynchronize());
    SweepArgs a0;
    a0._dx = 51.2; a0._dy = 49.0; a0._dz = 47.5; a0.dt = INFINITY; a0.r = 0.7; a0.theta_dtau = 191.3; a0.eta_dtau = 0.0119;
    a0.L = make_lay(nx, ny, nz);
    a0.i0 = a0.j0 = a0.k0 = 0;
    FusedBC bc;
    memset(&bc, 0, sizeof(bc));
    bc.fsL = bc.fsF = bc.fsK0 = 1;
    printf("kbench_va n=%d reps=%d chunk=%d MiB: %d arrays, arena at %p\n", n, reps, chunk_mib_i, NA, (void *)arena);
    auto time_cfg = [&](const char *desc) {
        SweepArgs b = a0; b.f = f; b.etatau = etatau; b.o = dst;
        double ms[2];
        {
            constexpr int TX = 64, TY = 4, KZ = 8;
            const int ntx = (nx + TX - 3) / (TX - 2), nty = (ny + TY - 2) / (TY - 1), ntz = (nz + KZ - 1) / KZ;
            ms[0] = T.run(reps, [&] { hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, 4, 1, false, 1, false, true, 3, 1, 0, true, true, true, false, 2>), dim3(ntx * nty * ntz), dim3(TX * TY), 0, 0, b, bc, ntx, nty, 0, 0, 0); });
        }
        {
            constexpr int TX = 64, TY = 8, KZ = 8;
            const int ntx = (nx + TX - 3) / (TX - 2), nty = (ny + TY - 2) / (TY - 1), ntz = (nz + KZ - 1) / KZ;
            ms[1] = T.run(reps, [&] { hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, 2, 1, false, 4, false, true, 3, 1, 0, true, true, true, false, 2>), dim3(ntx * nty * ntz), dim3(TX * TY), 0, 0, b, bc, ntx, nty, 0, 0, 0); });
        }
        printf("%-44s 64x4 %7.3f ms   64x8 %7.3f ms\n", desc, ms[0], ms[1]);
        fflush(stdout);
    };
    time_cfg("packed (slots = own size)");
    char desc[256];
    const size_t slot = (slot_max + ((size_t)1 << 30) - 1) >> 30 << 30;       // a power-of-two-ish slot that holds every array (2 GiB at 512^3)
    auto layout_stride = [&](size_t base, size_t stride) { for (int q = 0; q < NA; q++) A[q].off = base + (size_t)q * stride; };
    if (what == 0 || what == 3) {
        // (1) equal strides: slot + g; the residues of the array bases modulo powers of two are q * g
        for (size_t gm : {0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 24, 30, 32, 34, 48, 62, 64, 66, 96, 126, 128, 130, 192, 254, 256, 258, 384, 510, 512, 514, 766, 1022}) {
            unmap_all();
            layout_stride(0, slot + gm * ((size_t)1 << 20));
            map_all();
            snprintf(desc, sizeof desc, "stride %zu GiB + %4zu MiB", slot >> 30, (size_t)gm);
            time_cfg(desc);
        }
    }
    unsigned long long rng = 0x1234567ull;
    auto rnd = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (size_t)(rng >> 33); };
    if (what == 1 || what == 3) {
        // (2) random offsets inside the slots (multiples of 2 MiB, up to 1 GiB): the distribution, and the raw material for a model
        for (int t = 0; t < 40; t++) {
            unmap_all();
            size_t o[64];
            for (int q = 0; q < NA; q++) { o[q] = (rnd() % 512) * M2; A[q].off = (size_t)q * (slot + ((size_t)1 << 30)) + o[q]; }
            map_all();
            int w = snprintf(desc, sizeof desc, "rnd");
            for (int q = 0; q < NA && w < 200; q++) w += snprintf(desc + w, sizeof desc - w, " %zu", o[q] / M2);
            printf("%s\n", desc);
            time_cfg("  ^ random offsets (units of 2 MiB)");
        }
    }
    if (what == 2 || what == 3) {
        // (3) one relative layout (stride slot + 34 MiB), shifted as a whole
        for (size_t sh : {0, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536}) {
            unmap_all();
            layout_stride(sh * ((size_t)1 << 20), slot + ((size_t)34 << 20));
            map_all();
            snprintf(desc, sizeof desc, "stride %zu GiB + 34 MiB, base + %zu MiB", slot >> 30, (size_t)sh);
            time_cfg(desc);
        }
    }
    if (what == 5) {
        // is "packed" fast because of the layout or because it is measured first?  packed again after strided layouts, and packed variants
        auto pack = [&](size_t round_to, size_t gap, bool reverse) {
            size_t at = 0;
            for (int q = 0; q < NA; q++) { Arr &a = A[reverse ? NA - 1 - q : q]; a.off = at; at += (a.mapped + round_to - 1) / round_to * round_to + gap; }
        };
        unmap_all(); layout_stride(0, slot); map_all(); time_cfg("stride 2 GiB");
        unmap_all(); pack(chunk, 0, false); map_all(); time_cfg("packed again");
        unmap_all(); layout_stride(0, slot + ((size_t)34 << 20)); map_all(); time_cfg("stride 2 GiB + 34 MiB");
        unmap_all(); pack(chunk, 0, false); map_all(); time_cfg("packed a third time");
        unmap_all(); pack(chunk, M2, false); map_all(); time_cfg("packed, 2 MiB gaps");
        unmap_all(); pack(chunk, 0, true); map_all(); time_cfg("packed in reverse order");
        unmap_all(); pack((size_t)64 << 20, 0, false); map_all(); time_cfg("packed, sizes rounded to 64 MiB");
        unmap_all(); pack((size_t)512 << 20, 0, false); map_all(); time_cfg("packed, sizes rounded to 512 MiB");
        unmap_all(); pack((size_t)1 << 30, 0, false); map_all(); time_cfg("packed, sizes rounded to 1 GiB");
        unmap_all(); pack(chunk, (size_t)1 << 30, false); map_all(); time_cfg("packed, 1 GiB gaps");
        unmap_all(); pack(chunk, ((size_t)1 << 30) + ((size_t)6 << 20), false); map_all(); time_cfg("packed, 1 GiB + 6 MiB gaps");
        unmap_all(); pack(chunk, 0, false); map_all(); time_cfg("packed a fourth time");
    }
    if (what == 4) {
        for (size_t gm : {0, 2, 34}) {
            unmap_all();
            layout_stride(0, slot + gm * ((size_t)1 << 20));
            map_all();
            snprintf(desc, sizeof desc, "chunk %4d MiB, stride %zu GiB + %2zu MiB", chunk_mib_i, slot >> 30, (size_t)gm);
            time_cfg(desc);
        }
    }
    unmap_all();
    for (auto &a : A) { for (auto hd : a.chunks) CK(hipMemRelease(hd)); a.chunks.clear(); }
    }
    printf("done\n");
    return 0;
}
