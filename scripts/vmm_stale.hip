// vmm_stale.hip -- after hipMemUnmap + hipMemMap of OTHER chunks at the same address, where do the first shader accesses go?  (the finding behind jrx_tuning_field_reroll's "contents
// undefined", csrc/fieldpool.hip.)  For each candidate "flush" F:  map chunks A at va, fill them with 1 (kernel); unmap; map chunks B at va; F; a kernel writes 2 through va; then B is
// mapped at a range never used before and read there: words holding 2 = writes that landed in B; A is read the same way: words holding 2 = writes that went through a stale translation.
// build: hipcc --offload-arch=gfx950 -O2 scripts/vmm_stale.hip -o scripts/vmm_stale
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void k_fill(unsigned *p, size_t n, unsigned v) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }
__global__ void k_count(const unsigned *p, size_t n, unsigned v, unsigned long long *out)
{
    unsigned long long c = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += p[i] == v;
    atomicAdd(out, c);
}
static hipMemAllocationProp prop = {};
static hipMemAccessDesc acc = {};
static const size_t CH = (size_t)2 << 20;
struct Chunks { std::vector<hipMemGenericAllocationHandle_t> h; };
static Chunks create(int n) { Chunks c; c.h.resize(n); for (auto &x : c.h) CK(hipMemCreate(&x, CH, &prop, 0)); return c; }
static void map_at(void *va, const Chunks &c) { for (size_t q = 0; q < c.h.size(); q++) CK(hipMemMap((char *)va + q * CH, CH, 0, c.h[q], 0)); CK(hipMemSetAccess(va, c.h.size() * CH, &acc, 1)); }
static unsigned long long count_in(const Chunks &c, unsigned v, unsigned long long *d_out)
{
    void *fresh = nullptr;                                    // a range never used before (never freed either: the runtime would hand it out again)
    CK(hipMemAddressReserve(&fresh, c.h.size() * CH, 0, nullptr, 0));
    map_at(fresh, c);
    CK(hipMemset(d_out, 0, 8));
    k_count<<<1024, 256>>>((const unsigned *)fresh, c.h.size() * CH / 4, v, d_out);
    unsigned long long r = 0;
    CK(hipMemcpy(&r, d_out, 8, hipMemcpyDeviceToHost));
    CK(hipMemUnmap(fresh, c.h.size() * CH));
    return r;
}
int main(int argc, char **argv)
{
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    unsigned long long *d_out; CK(hipMalloc(&d_out, 8));
    const char *names[] = {"nothing", "hipDeviceSynchronize", "usleep 20 ms", "hipMalloc + hipFree 1 MiB", "hipHostMalloc + hipHostFree 1 MiB", "hipMemcpy D2H 8 B from another buffer",
                           "hipMemcpy H2D 4 KiB into the range", "hipMemset of the range", "an empty kernel + sync", "hipStreamCreate + destroy", "a first kernel write (discarded) + sync"};
    for (int nch : {1, 6, 64}) {
        const size_t words = nch * CH / 4;
        printf("range of %d chunks of 2 MiB\n", nch);
        for (int F = 0; F < 11; F++) {
            for (int rep = 0; rep < 2; rep++) {
                void *va = nullptr;
                CK(hipMemAddressReserve(&va, nch * CH, 0, nullptr, 0));
                Chunks A = create(nch), B = create(nch);
                map_at(va, A);
                k_fill<<<1024, 256>>>((unsigned *)va, words, 1u);
                CK(hipDeviceSynchronize());
                CK(hipMemUnmap(va, nch * CH));
                map_at(va, B);
                void *t = nullptr; unsigned long long hv = 0; hipStream_t s;
                switch (F) {
                case 1: CK(hipDeviceSynchronize()); break;
                case 2: usleep(20000); break;
                case 3: CK(hipMalloc(&t, 1 << 20)); CK(hipFree(t)); break;
                case 4: CK(hipHostMalloc(&t, 1 << 20)); CK(hipHostFree(t)); break;
                case 5: CK(hipMemcpy(&hv, d_out, 8, hipMemcpyDeviceToHost)); break;
                case 6: { std::vector<char> z(4096); CK(hipMemcpy(va, z.data(), 4096, hipMemcpyHostToDevice)); } break;
                case 7: CK(hipMemset(va, 0, nch * CH)); CK(hipDeviceSynchronize()); break;
                case 8: k_fill<<<1, 64>>>((unsigned *)d_out, 0, 0u); CK(hipDeviceSynchronize()); break;
                case 9: CK(hipStreamCreate(&s)); CK(hipStreamDestroy(s)); break;
                case 10: k_fill<<<1024, 256>>>((unsigned *)va, words, 3u); CK(hipDeviceSynchronize()); break;
                default: break;
                }
                k_fill<<<1024, 256>>>((unsigned *)va, words, 2u);
                CK(hipDeviceSynchronize());
                CK(hipMemUnmap(va, nch * CH));
                const unsigned long long inB = count_in(B, 2u, d_out), inA = count_in(A, 2u, d_out);
                printf("  %-42s run %d: %5.1f %% of the writes landed in the new chunks, %5.1f %% in the old ones\n", names[F], rep, 100.0 * inB / words, 100.0 * inA / words);
                for (auto x : A.h) CK(hipMemRelease(x));
                for (auto x : B.h) CK(hipMemRelease(x));
                // va is kept reserved (not freed): a later reservation must not be handed the same address
            }
        }
    }
    {   // may one chunk set be mapped at two ranges at once?  (a second, never-used range would let the pool CHECK where its writes went)
        const int nch = 3;
        void *v1 = nullptr, *v2 = nullptr;
        CK(hipMemAddressReserve(&v1, nch * CH, 0, nullptr, 0)); CK(hipMemAddressReserve(&v2, nch * CH, 0, nullptr, 0));
        Chunks A = create(nch);
        map_at(v1, A);
        hipError_t e = hipSuccess;
        for (size_t q = 0; q < A.h.size() && e == hipSuccess; q++) e = hipMemMap((char *)v2 + q * CH, CH, 0, A.h[q], 0);
        if (e == hipSuccess) e = hipMemSetAccess(v2, nch * CH, &acc, 1);
        if (e != hipSuccess) { (void)hipGetLastError(); printf("a second mapping of the same chunks: refused (%s)\n", hipGetErrorString(e)); }
        else {
            k_fill<<<1024, 256>>>((unsigned *)v1, nch * CH / 4, 77u);
            CK(hipDeviceSynchronize());
            CK(hipMemset(d_out, 0, 8));
            k_count<<<1024, 256>>>((const unsigned *)v2, nch * CH / 4, 77u, d_out);
            unsigned long long r = 0;
            CK(hipMemcpy(&r, d_out, 8, hipMemcpyDeviceToHost));
            printf("a second mapping of the same chunks: allowed; %.1f %% of what was written through the first range is read through the second\n", 100.0 * r / (nch * CH / 4));
            CK(hipMemUnmap(v2, nch * CH));
        }
        CK(hipMemUnmap(v1, nch * CH));
        for (auto x : A.h) CK(hipMemRelease(x));
    }
    // ---- how reliable is a flush?  300 re-mappings each, 6 chunks of 2 MiB, the range and both chunk sets reused every time (as a pool does)
    {
        const int nch = 6;
        const size_t words = nch * CH / 4;
        void *va = nullptr;
        CK(hipMemAddressReserve(&va, nch * CH, 0, nullptr, 0));
        Chunks A = create(nch), B = create(nch);
        static char hostbuf[1 << 16];
        const char *fn[] = {"hipHostMalloc + hipHostFree 4 KiB", "hipHostRegister + hipHostUnregister 64 KiB", "hipStreamCreate + hipStreamDestroy", "hipHostMalloc + hipHostFree of a size that changes every time", "hipMalloc + hipFree 256 MiB", "hipHostMalloc + hipHostFree 4 KiB with other host allocations coming and going"};
        std::vector<void *> others;
        hipStream_t old_stream; CK(hipStreamCreateWithFlags(&old_stream, hipStreamNonBlocking));
        k_fill<<<64, 256, 0, old_stream>>>((unsigned *)d_out, 0, 0u); CK(hipDeviceSynchronize());
        for (int F = 0; F < 6; F++) {
            int bad = 0;
            unsigned long long lost = 0;
            for (int it = 0; it < 300; it++) {
                const Chunks &X = (it & 1) ? B : A, &Y = (it & 1) ? A : B;        // X is mapped and filled, then Y takes its place
                map_at(va, X);
                if (it == 0) { void *t0 = nullptr; CK(hipHostMalloc(&t0, 4096)); CK(hipHostFree(t0)); }
                k_fill<<<1024, 256>>>((unsigned *)va, words, 1u);
                CK(hipDeviceSynchronize());
                CK(hipMemUnmap(va, nch * CH));
                map_at(va, Y);
                void *t = nullptr; hipStream_t st;
                switch (F) {
                case 0: CK(hipHostMalloc(&t, 4096)); CK(hipHostFree(t)); break;
                case 1: CK(hipHostRegister(hostbuf, sizeof hostbuf, hipHostRegisterDefault)); CK(hipHostUnregister(hostbuf)); break;
                case 2: CK(hipStreamCreate(&st)); CK(hipStreamDestroy(st)); break;
                case 3: CK(hipHostMalloc(&t, 4096 * (1 + it % 37))); CK(hipHostFree(t)); break;
                case 4: CK(hipMalloc(&t, (size_t)256 << 20)); CK(hipFree(t)); break;
                default: {      // a busy process: pinned buffers of many sizes are allocated and freed around the flush
                    for (int q = 0; q < 3; q++) { void *o = nullptr; CK(hipHostMalloc(&o, 4096 << ((it + q) % 9))); others.push_back(o); }
                    CK(hipHostMalloc(&t, 4096)); CK(hipHostFree(t));
                    while (others.size() > 8) { CK(hipHostFree(others.front())); others.erase(others.begin()); }
                } break;
                }
                // the writing kernel runs on a stream that existed before the re-mapping (as the library's own stream does), a large grid so that every XCD takes part
                hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, old_stream, (unsigned *)va, words, 2u + (unsigned)it);
                CK(hipDeviceSynchronize());
                CK(hipMemset(d_out, 0, 8));
                k_count<<<1024, 256>>>((const unsigned *)va, words, 2u + (unsigned)it, d_out);      // read back through the same range, as a later kernel of the application would
                unsigned long long r = 0;
                CK(hipMemcpy(&r, d_out, 8, hipMemcpyDeviceToHost));
                CK(hipMemUnmap(va, nch * CH));
                // and where did the words really go?  Y through a fresh range
                const unsigned long long inY = count_in(Y, 2u + (unsigned)it, d_out);
                if (inY != words) { bad++; lost += words - inY; }
                (void)r;
            }
            printf("%-62s 300 re-mappings: %d with writes that did not reach the new chunks (%llu words in all)\n", fn[F], bad, lost);
        }
    }
    return 0;
}