// fieldpool.hip -- device memory for the state arrays, owned by the library (jrx_field_alloc / jrx_field_free).
//
// Replaces the array constructor the backend owns in the reference: StokesArrays(::Type{AMDGPUBackend}, ni) -> @zeros(ni...) -> ROCArray
// (src/ext/AMDGPU/3D.jl:46-48, src/types/constructors/stokes.jl:279-303).  Why the library wants a say in it: the 512^3 kernels run at one of
// two rates for a process's lifetime depending on how the driver happened to back the arrays physically (profiles/r04_alloc_stagger.txt;
// contiguous backing = the slow rate).  hipMalloc gives nobody a say; the virtual-memory-management API does: an array is one reserved
// virtual range onto which physical chunks are mapped in an order the pool chooses.
//
// Placement kinds ("field_placement"):
//   0  hipMalloc (what a ROCArray / torch tensor gets)
//   1  chunks: hipMemCreate handles of "field_chunk_mib" MiB, created in batches, handed to the arrays in shuffled order (a fixed LCG: the same
//      sequence of requests gives the same chunk order), mapped with hipMemMap, one hipMemSetAccess per array
//   2  physically contiguous (hipDeviceMallocContiguous): the reproducer of the slow rate, for A/B runs only
// Host-only code; every entry point requires the handle's device to be current.
#include "jrx_internal.hpp"
#include <algorithm>
#include <chrono>
#include <map>
#include <vector>

struct jrx_field_pool {
    struct Alloc { size_t bytes = 0, mapped = 0; int kind = 0; bool in_arena = false; std::vector<hipMemGenericAllocationHandle_t> chunks; size_t chunk = 0; };
    std::map<void *, Alloc> live;
    std::vector<hipMemGenericAllocationHandle_t> spare;      // created, unmapped chunks (all of `spare_chunk` bytes)
    size_t spare_chunk = 0;
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    double create_ms = 0, map_ms = 0;
    int64_t chunks_created = 0, bytes_live = 0, rerolls = 0;
    // the arena: ONE reserved virtual range in which the chunk-backed arrays are placed one behind the other, `gap` bytes apart -- the rate of the large kernels turned out to
    // depend on the arrays' VIRTUAL addresses (re-rolling the physical chunks under fixed addresses changes nothing, new addresses do: profiles/r05_placement.txt), and this is
    // what makes them a choice instead of a draw
    char *arena = nullptr;
    size_t arena_bytes = 0, arena_used = 0;
    std::multimap<size_t, void *> arena_free;     // released sub-ranges by size, reused for arrays of exactly that size
};

namespace {
using Clock = std::chrono::steady_clock;
double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }
uint64_t next_rng(uint64_t &s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return s >> 17; }

jrx_field_pool *pool_of(jrx_handle *h)
{
    if (!h->pool) h->pool = new jrx_field_pool();
    return h->pool;
}

void release_spare(jrx_field_pool *P)
{
    for (auto hd : P->spare) (void)hipMemRelease(hd);
    P->spare.clear();
    P->spare_chunk = 0;
}

jrx_status alloc_chunks(jrx_handle *h, jrx_field_pool *P, size_t bytes, void **out)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = h->device;
    size_t gran = 0;
    JRX_HIP(h, hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (gran == 0) gran = (size_t)2 << 20;
    size_t chunk = (size_t)(h->field_chunk_mib > 0 ? h->field_chunk_mib : 64) << 20;
    chunk = (chunk + gran - 1) / gran * gran;
    if (P->spare_chunk != chunk) release_spare(P);
    P->spare_chunk = chunk;
    const size_t nch = (bytes + chunk - 1) / chunk;
    // a batch of new chunks: at least what this array needs, and at least "field_batch_mib" MiB, so that the shuffle mixes the chunks of several arrays
    if (P->spare.size() < nch) {
        const size_t batch_min = ((size_t)(h->field_batch_mib > 0 ? h->field_batch_mib : 0) << 20) / chunk;
        const size_t want = std::max(nch - P->spare.size(), batch_min);
        const auto t0 = Clock::now();
        for (size_t c = 0; c < want; c++) {
            hipMemGenericAllocationHandle_t hd;
            const hipError_t e = hipMemCreate(&hd, chunk, &prop, 0);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                if (P->spare.size() >= nch) break;             // the batch was a wish; the array itself is covered
                return jrx_fail(h, JRX_ERR_HIP, "jrx_field_alloc: hipMemCreate(%zu MiB) -> %s after %lld chunks", chunk >> 20, hipGetErrorString(e), (long long)P->chunks_created);
            }
            P->spare.push_back(hd);
            P->chunks_created++;
        }
        P->create_ms += ms_since(t0);
    }
    // Fisher-Yates over the spare list, then the array takes the tail
    if (h->field_shuffle)
        for (size_t i = P->spare.size(); i > 1; i--) std::swap(P->spare[i - 1], P->spare[next_rng(P->rng) % i]);
    const auto t1 = Clock::now();
    void *va = nullptr;
    size_t align = (size_t)(h->field_va_align_mib > 0 ? h->field_va_align_mib : 0) << 20;
    if (align < gran) align = gran;
    bool in_arena = false;
    if (h->field_arena_gib > 0) {
        if (!P->arena) {
            void *a = nullptr;
            const size_t want = (size_t)h->field_arena_gib << 30;
            if (hipMemAddressReserve(&a, want, (size_t)1 << 30, nullptr, 0) == hipSuccess) { P->arena = (char *)a; P->arena_bytes = want; P->arena_used = 0; }
            else (void)hipGetLastError();
        }
        if (P->arena) {
            auto it = P->arena_free.find(nch * chunk);
            if (it != P->arena_free.end()) { va = it->second; P->arena_free.erase(it); in_arena = true; }
            else {
                const size_t gap = ((size_t)(h->field_va_gap_mib > 0 ? h->field_va_gap_mib : 0) << 20) / gran * gran;
                size_t at = (P->arena_used + align - 1) / align * align;
                if (at + nch * chunk <= P->arena_bytes) { va = P->arena + at; P->arena_used = at + nch * chunk + gap; in_arena = true; }
            }
        }
    }
    if (!in_arena) JRX_HIP(h, hipMemAddressReserve(&va, nch * chunk, align, nullptr, 0));
    jrx_field_pool::Alloc A;
    A.bytes = bytes; A.kind = 1; A.chunk = chunk; A.in_arena = in_arena;
    for (size_t c = 0; c < nch; c++) {
        hipMemGenericAllocationHandle_t hd = P->spare.back();
        const hipError_t e = hipMemMap((char *)va + c * chunk, chunk, 0, hd, 0);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (A.mapped) (void)hipMemUnmap(va, A.mapped);
            for (auto x : A.chunks) P->spare.push_back(x);
            if (in_arena) P->arena_free.insert({nch * chunk, va}); else (void)hipMemAddressFree(va, nch * chunk);
            return jrx_fail(h, JRX_ERR_HIP, "jrx_field_alloc: hipMemMap -> %s", hipGetErrorString(e));
        }
        P->spare.pop_back();
        A.chunks.push_back(hd);
        A.mapped += chunk;
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const hipError_t e = hipMemSetAccess(va, nch * chunk, &acc, 1);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipMemUnmap(va, A.mapped);
        for (auto x : A.chunks) P->spare.push_back(x);
        if (in_arena) P->arena_free.insert({nch * chunk, va}); else (void)hipMemAddressFree(va, nch * chunk);
        return jrx_fail(h, JRX_ERR_HIP, "jrx_field_alloc: hipMemSetAccess -> %s", hipGetErrorString(e));
    }
    P->map_ms += ms_since(t1);
    P->live[va] = std::move(A);
    *out = va;
    return JRX_OK;
}
// Give one chunk-backed array new physical backing IN PLACE: its virtual range, and therefore every pointer the caller and the library hold, stays as it is.  New chunks (spare ones
// first, in shuffled order; freshly created ones if the spare list is short) are mapped at a temporary range, the contents are copied, then the array's range is re-mapped onto them
// and the old chunks join the spare list.  The device must be idle as far as this array is concerned (the caller's contract; the function synchronises the device itself).
jrx_status reroll_one(jrx_handle *h, jrx_field_pool *P, void *va, jrx_field_pool::Alloc &A)
{
    if (A.kind != 1) return JRX_OK;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = h->device;
    const size_t chunk = A.chunk, nch = A.chunks.size();
    if (P->spare_chunk != chunk) { release_spare(P); P->spare_chunk = chunk; }
    const auto t0 = Clock::now();
    while (P->spare.size() < nch) {
        hipMemGenericAllocationHandle_t hd;
        const hipError_t e = hipMemCreate(&hd, chunk, &prop, 0);
        if (e != hipSuccess) { (void)hipGetLastError(); return jrx_fail(h, JRX_ERR_HIP, "jrx_field_reroll: hipMemCreate(%zu MiB) -> %s", chunk >> 20, hipGetErrorString(e)); }
        P->spare.push_back(hd);
        P->chunks_created++;
    }
    P->create_ms += ms_since(t0);
    if (h->field_shuffle)
        for (size_t i = P->spare.size(); i > 1; i--) std::swap(P->spare[i - 1], P->spare[next_rng(P->rng) % i]);
    const auto t1 = Clock::now();
    std::vector<hipMemGenericAllocationHandle_t> fresh(P->spare.end() - (long)nch, P->spare.end());
    P->spare.resize(P->spare.size() - nch);
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    void *tmp = nullptr;
    auto give_back = [&] { for (auto hd : fresh) P->spare.push_back(hd); };
    hipError_t e = hipMemAddressReserve(&tmp, A.mapped, chunk < ((size_t)2 << 20) ? ((size_t)2 << 20) : 0, nullptr, 0);
    if (e != hipSuccess) { (void)hipGetLastError(); give_back(); return jrx_fail(h, JRX_ERR_HIP, "jrx_field_reroll: hipMemAddressReserve -> %s", hipGetErrorString(e)); }
    size_t mapped = 0;
    for (size_t c = 0; c < nch && e == hipSuccess; c++) { e = hipMemMap((char *)tmp + c * chunk, chunk, 0, fresh[c], 0); if (e == hipSuccess) mapped += chunk; }
    if (e == hipSuccess) e = hipMemSetAccess(tmp, A.mapped, &acc, 1);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(tmp, va, A.bytes, hipMemcpyDeviceToDevice);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (mapped) (void)hipMemUnmap(tmp, mapped);
    (void)hipMemAddressFree(tmp, A.mapped);
    if (e != hipSuccess) { (void)hipGetLastError(); give_back(); return jrx_fail(h, JRX_ERR_HIP, "jrx_field_reroll: staging the new chunks -> %s", hipGetErrorString(e)); }
    // the switch: from here on a failure leaves the array without backing, which is reported as such
    e = hipMemUnmap(va, A.mapped);
    for (size_t c = 0; c < nch && e == hipSuccess; c++) e = hipMemMap((char *)va + c * chunk, chunk, 0, fresh[c], 0);
    if (e == hipSuccess) e = hipMemSetAccess(va, A.mapped, &acc, 1);
    if (e != hipSuccess) { (void)hipGetLastError(); return jrx_fail(h, JRX_ERR_HIP, "jrx_field_reroll: re-mapping the array at %p -> %s (the array has lost its backing)", va, hipGetErrorString(e)); }
    for (auto hd : A.chunks) P->spare.push_back(hd);
    A.chunks = fresh;
    P->map_ms += ms_since(t1);
    P->rerolls++;
    return JRX_OK;
}
}   // namespace

// internal: every large library-owned array (second state sets, ητ) comes from the same place as the caller's
jrx_status jrx_dev_alloc(jrx_handle *h, size_t bytes, void **out)
{
    *out = nullptr;
    if (bytes == 0) bytes = 8;
    jrx_field_pool *P = pool_of(h);
    // small arrays never matter for the placement and would waste a chunk each
    const int kind = (h->field_placement == 1 && bytes < ((size_t)8 << 20)) ? 0 : h->field_placement;
    if (kind == 1) {
        JRX_TRY(alloc_chunks(h, P, bytes, out));
    } else {
        void *p = nullptr;
        if (kind == 2) {
            if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocContiguous) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
        }
        if (!p) JRX_HIP(h, hipMalloc(&p, bytes));
        jrx_field_pool::Alloc A;
        A.bytes = bytes; A.kind = kind;
        P->live[p] = std::move(A);
        *out = p;
    }
    P->bytes_live += (int64_t)bytes;
    return JRX_OK;
}

jrx_status jrx_dev_free(jrx_handle *h, void *p)
{
    if (!p) return JRX_OK;
    jrx_field_pool *P = pool_of(h);
    auto it = P->live.find(p);
    if (it == P->live.end()) return jrx_fail(h, JRX_ERR_ARG, "jrx_field_free: %p was not allocated by jrx_field_alloc on this handle", p);
    jrx_field_pool::Alloc &A = it->second;
    P->bytes_live -= (int64_t)A.bytes;
    if (A.kind == 1) {
        // nothing of this handle may still be using the range
        JRX_HIP(h, hipDeviceSynchronize());
        JRX_HIP(h, hipMemUnmap(p, A.mapped));
        if (A.in_arena) P->arena_free.insert({A.mapped, p});
        else JRX_HIP(h, hipMemAddressFree(p, A.mapped));
        if (A.chunk == P->spare_chunk) for (auto hd : A.chunks) P->spare.push_back(hd);
        else for (auto hd : A.chunks) (void)hipMemRelease(hd);
    } else {
        JRX_HIP(h, hipFree(p));
    }
    P->live.erase(it);
    return JRX_OK;
}

void jrx_pool_destroy(jrx_handle *h)
{
    jrx_field_pool *P = h->pool;
    if (!P) return;
    (void)hipDeviceSynchronize();
    for (auto &kv : P->live) {
        if (kv.second.kind == 1) {
            (void)hipMemUnmap(kv.first, kv.second.mapped);
            if (!kv.second.in_arena) (void)hipMemAddressFree(kv.first, kv.second.mapped);
            for (auto hd : kv.second.chunks) (void)hipMemRelease(hd);
        } else {
            (void)hipFree(kv.first);
        }
    }
    release_spare(P);
    if (P->arena) (void)hipMemAddressFree(P->arena, P->arena_bytes);
    delete P;
    h->pool = nullptr;
}

extern "C" {

jrx_status jrx_field_alloc(jrx_handle *h, int64_t count, double **out)
{
    if (!h) return JRX_ERR_ARG;
    if (!out) return jrx_fail(h, JRX_ERR_ARG, "jrx_field_alloc: out is NULL");
    if (count < 0) return jrx_fail(h, JRX_ERR_ARG, "jrx_field_alloc: count %lld < 0", (long long)count);
    JRX_TRY(jrx_check_device(h));
    return jrx_dev_alloc(h, (size_t)count * sizeof(double), (void **)out);
}

jrx_status jrx_field_free(jrx_handle *h, double *p)
{
    if (!h) return JRX_ERR_ARG;
    JRX_TRY(jrx_check_device(h));
    return jrx_dev_free(h, p);
}

jrx_status jrx_field_reroll(jrx_handle *h, double *p)
{
    if (!h) return JRX_ERR_ARG;
    JRX_TRY(jrx_check_device(h));
    jrx_field_pool *P = pool_of(h);
    if (p) {
        auto it = P->live.find(p);
        if (it == P->live.end()) return jrx_fail(h, JRX_ERR_ARG, "jrx_field_reroll: %p was not allocated by jrx_field_alloc on this handle", (void *)p);
        return reroll_one(h, P, it->first, it->second);
    }
    for (auto &kv : P->live) JRX_TRY(reroll_one(h, P, kv.first, kv.second));
    return JRX_OK;
}

jrx_status jrx_field_list(jrx_handle *h, int64_t cap, double **ptrs, int64_t *bytes, int64_t *count)
{
    if (!h) return JRX_ERR_ARG;
    if (!count) return jrx_fail(h, JRX_ERR_ARG, "jrx_field_list: count is NULL");
    jrx_field_pool *P = pool_of(h);
    int64_t n = 0;
    for (auto &kv : P->live) {
        if (n < cap && ptrs) ptrs[n] = (double *)kv.first;
        if (n < cap && bytes) bytes[n] = kv.second.kind == 1 ? (int64_t)kv.second.bytes : -(int64_t)kv.second.bytes;       // negative: not chunk-backed (cannot be re-rolled)
        n++;
    }
    *count = n;
    return JRX_OK;
}

jrx_status jrx_field_trim(jrx_handle *h)
{
    if (!h) return JRX_ERR_ARG;
    JRX_TRY(jrx_check_device(h));
    if (h->pool) release_spare(h->pool);
    return JRX_OK;
}

jrx_status jrx_field_stats(jrx_handle *h, int64_t out[6])
{
    if (!h) return JRX_ERR_ARG;
    if (!out) return jrx_fail(h, JRX_ERR_ARG, "jrx_field_stats: out is NULL");
    jrx_field_pool *P = pool_of(h);
    out[0] = (int64_t)P->live.size();
    out[1] = P->bytes_live;
    out[2] = P->chunks_created;
    out[3] = (int64_t)P->spare.size();
    out[4] = (int64_t)(P->create_ms * 1e3);      // microseconds spent in hipMemCreate
    out[5] = (int64_t)(P->map_ms * 1e3);         // microseconds spent reserving, mapping and setting access
    return JRX_OK;
}

}   // extern "C"
