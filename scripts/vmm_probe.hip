// vmm_probe.hip -- does the PHYSICAL placement of 22 arrays of 1 GiB change the rate of a kernel that streams them in lockstep (12 read, 10 written: the stream mix of the headline form of k_fused3d)?
//   A  hipMalloc per array                      B  hipExtMallocWithFlags(hipDeviceMallocContiguous) per array
//   C  virtual memory management: one VA range per array, physical chunks of `gran` bytes created in round-robin order over the arrays (chunk c of array a is the (c * 22 + a)-th allocation),
//      so that physically consecutive chunks belong to different arrays              D  like C, but all chunks of array 0 first, then array 1, ... (array-major)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/vmm_probe.hip -o scripts/vmm_probe && ./scripts/vmm_probe [reps] [1 = also the VMM modes C, D]
// Record of an experiment (profiles/r04_alloc_stagger.txt); not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
constexpr int NR = 12, NW = 10, NA = NR + NW;
struct Args { const double *r[NR]; double *w[NW]; long long n; };
__global__ __launch_bounds__(256) void k_stream(Args a)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < a.n; t += (long long)gridDim.x * blockDim.x) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < NR; q++) s += a.r[q][t];
#pragma unroll
        for (int q = 0; q < NW; q++) __builtin_nontemporal_store(s + (double)q, &a.w[q][t]);
    }
}
// z-marching variant: a block owns a 64 x 4 column tile of a 512^3 grid and walks the planes of an 8-plane chunk (the access pattern of the fused kernel, without its halos)
__global__ __launch_bounds__(256) void k_march(Args a)
{
    const int n = 512, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int tile = blockIdx.x, ntx = n / 64, nty = n / 4;
    const int tix = tile % ntx, tiy = (tile / ntx) % nty, tiz = tile / (ntx * nty);
    long long t = (long long)(tix * 64 + tx) + (long long)n * (tiy * 4 + ty) + (long long)n * n * (tiz * 8);
    for (int k = 0; k < 8; k++, t += (long long)n * n) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < NR; q++) s += a.r[q][t];
#pragma unroll
        for (int q = 0; q < NW; q++) __builtin_nontemporal_store(s + (double)q, &a.w[q][t]);
    }
}
static double run(const Args &a, int reps, bool march)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto go = [&]() { if (march) hipLaunchKernelGGL(k_march, dim3(8 * 128 * 64), dim3(256), 0, 0, a); else hipLaunchKernelGGL(k_stream, dim3(256 * 16), dim3(256), 0, 0, a); };
    go(); go(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) go();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}
int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    const bool vmm = argc > 2 && atoi(argv[2]) != 0;      // modes C, D: creating and mapping the 11,264 chunks did not finish within ten minutes on the box it was tried on
    const size_t bytes = (size_t)1 << 30;
    const long long n = (long long)(bytes / 8);
    for (int pass = 0; pass < 2; pass++) {
    for (char mode : {'A', 'B', 'C', 'D'}) {
        if (mode >= 'C' && !vmm) continue;
        void *p[NA] = {};
        std::vector<hipMemGenericAllocationHandle_t> handles;
        size_t gran = 0;
        if (mode == 'A') { for (int q = 0; q < NA; q++) CK(hipMalloc(&p[q], bytes)); }
        else if (mode == 'B') { for (int q = 0; q < NA; q++) CK(hipExtMallocWithFlags(&p[q], bytes, hipDeviceMallocContiguous)); }
        else {
            hipMemAllocationProp prop = {};
            prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
            CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
            const size_t nchunk = bytes / gran;
            for (int q = 0; q < NA; q++) CK(hipMemAddressReserve(&p[q], bytes, 0, nullptr, 0));
            handles.resize(nchunk * NA);
            hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
            if (mode == 'C') { for (size_t c = 0; c < nchunk; c++) for (int q = 0; q < NA; q++) CK(hipMemCreate(&handles[c * NA + q], gran, &prop, 0)); }
            else { for (int q = 0; q < NA; q++) for (size_t c = 0; c < nchunk; c++) CK(hipMemCreate(&handles[c * NA + q], gran, &prop, 0)); }
            for (size_t c = 0; c < nchunk; c++) for (int q = 0; q < NA; q++) CK(hipMemMap((char *)p[q] + c * gran, gran, 0, handles[c * NA + q], 0));
            for (int q = 0; q < NA; q++) CK(hipMemSetAccess(p[q], bytes, &acc, 1));
        }
        Args a; a.n = n;
        for (int q = 0; q < NR; q++) { a.r[q] = (const double *)p[q]; CK(hipMemset(p[q], 0, bytes)); }
        for (int q = 0; q < NW; q++) a.w[q] = (double *)p[NR + q];
        const double ms_s = run(a, reps, false), ms_m = run(a, reps, true);
        printf("pass %d mode %c%s: lockstep stream %.3f ms = %.2f TB/s    z-marching tiles %.3f ms = %.2f TB/s\n", pass, mode, mode >= 'C' ? (gran == ((size_t)2 << 20) ? " (2 MiB chunks)" : " (chunks)") : "",
               ms_s, NA * (double)bytes / ms_s / 1e9, ms_m, NA * (double)bytes / ms_m / 1e9);
        fflush(stdout);
        if (mode <= 'B') { for (int q = 0; q < NA; q++) CK(hipFree(p[q])); }
        else {
            for (int q = 0; q < NA; q++) { CK(hipMemUnmap(p[q], bytes)); CK(hipMemAddressFree(p[q], bytes)); }
            for (auto &hd : handles) CK(hipMemRelease(hd));
        }
    }
    }
    return 0;
}
