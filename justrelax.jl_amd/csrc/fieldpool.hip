// fieldpool.hip -- device memory for the state arrays, owned by the library (jrx_field_alloc / jrx_field_free).
//
// Replaces the array constructor the backend owns in the reference: StokesArrays(::Type{AMDGPUBackend}, ni) -> @zeros(ni...) -> ROCArray
// (src/ext/AMDGPU/3D.jl:46-48, src/types/constructors/stokes.jl:279-303).  Why the library wants a say in it: the same launch of the 512^3 kernel takes
// 4.7 .. 6.9 ms depending on which physical pages its arrays got (profiles/r05_placement_search.txt) -- same traffic, same plain bandwidth, nothing but a
// run tells.  hipMalloc gives nobody a say; the virtual-memory-management API does: an array is one reserved virtual range onto which physical chunks are
// mapped, and the chunks under it can be exchanged IN PLACE (jrx_tuning_field_reroll / _undo / _keep), which is what the placement search (jrx_field_tune) does.
//
// Placement kinds ("field_placement"):
//   0  hipMalloc (what a ROCArray / torch tensor gets)
//   1  chunks: hipMemCreate handles of "field_chunk_mib" MiB, created in batches, handed to the arrays in shuffled order (a fixed LCG: the same
//      sequence of requests gives the same chunk order), mapped with hipMemMap, one hipMemSetAccess per array
//   2  physically contiguous (hipDeviceMallocContiguous): the slowest placement there is, for A/B runs only
// Host-only code; every entry point requires the handle's device to be current.
#include "jrx_internal.hpp"
#include "jrx_tuning.h"
#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>
#include <vector>

struct jrx_field_pool {
    struct Alloc { size_t bytes = 0, mapped = 0, skew = 0; int kind = 0; bool in_arena = false, cold = false;   // cold: left alone by whole-set re-rolls (jrx_pool_mark_cold)
                   std::vector<hipMemGenericAllocationHandle_t> chunks, prev; size_t chunk = 0; };   // prev: the chunks before the last re-roll (jrx_tuning_field_undo)
    std::map<void *, Alloc> live;
    std::map<size_t, std::vector<hipMemGenericAllocationHandle_t>> spare;      // created, unmapped chunks by their size
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    double create_ms = 0, map_ms = 0;
    int64_t chunks_created = 0, bytes_live = 0, rerolls = 0, large_allocs = 0, lost_remaps = 0;
    std::vector<void *> ballast;                  // "field_ballast_mib": allocations nobody uses, made behind every large array so that the arrays spread over the device's memory
    void *stage = nullptr;                        // jrx_tuning_field_reroll: the contents of the array being re-mapped
    size_t stage_bytes = 0;
    // the arena ("field_arena_gib", an experiment knob): ONE reserved virtual range in which the chunk-backed arrays are placed one behind the other, `gap` bytes apart
    char *arena = nullptr;
    size_t arena_bytes = 0, arena_used = 0;
    std::multimap<size_t, void *> arena_free;     // released sub-ranges by size, reused for arrays of exactly that size
};

namespace {
// jrx_tuning_field_reroll moves the contents with a kernel (the range is ordinary device memory for a kernel on the null stream)
__global__ void k_pool_copy(double *__restrict__ dst, const double *__restrict__ src, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// remap_with checks where its writes went (see there): words of `a` (the new chunks through a range of their own) that differ from `b` (the staged contents)
__global__ void k_pool_diff(const unsigned long long *__restrict__ a, const unsigned long long *__restrict__ b, size_t n, unsigned long long *out)
{
    unsigned long long c = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
    if (c) atomicAdd(out, c);
}

// alloc_chunks checks a fresh mapping at a range that was used before (see there): one word per 2 MiB page is stamped through the range and looked for through a second mapping
__global__ void k_pool_stamp(unsigned long long *p, size_t pages, size_t stride_words, unsigned long long nonce)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < pages; i += (size_t)gridDim.x * blockDim.x) p[i * stride_words] = nonce + i;
}
__global__ void k_pool_stamped(const unsigned long long *p, size_t pages, size_t stride_words, unsigned long long nonce, unsigned long long *out)
{
    unsigned long long c = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < pages; i += (size_t)gridDim.x * blockDim.x) c += p[i * stride_words] != nonce + i;
    if (c) atomicAdd(out, c);
}

// STALE TRANSLATIONS.  On this ROCm release (7.2, gfx950) hipMemUnmap + hipMemMap of OTHER chunks at an address that was mapped before leaves the shaders with the OLD translation:
// scripts/vmm_stale.hip -- 100 % of the words a kernel writes through the re-mapped range land in the old chunks, for ranges of 2 .. 128 MiB, whether or not a hipDeviceSynchronize, a
// 20 ms sleep, a hipMalloc / hipFree, a copy, a memset or a complete earlier kernel lies in between; after a hipHostMalloc + hipHostFree (or a hipStreamCreate + destroy) 100 % land in
// the new ones.  (Both go through the driver's map / queue path, which flushes the translation caches of the process; the plain re-mapping evidently does not.)  So every mapping this
// file makes at an address that may have been mapped before -- a re-roll, a range of the arena handed out again, a reservation the runtime hands out again -- is followed by that flush.
// VIRTUAL RANGES ARE NEVER GIVEN BACK to the runtime.  A range that hipMemAddressFree has returned can come back from hipMalloc / hipExtMallocWithFlags, and the full GPU suite then
// died now and then inside a later hipFree (a segmentation fault in the runtime; gpurun_out/r05y) or produced wrong bits -- the runtime's own books and the translations of such an
// address are not to be trusted on this release.  Released ranges are parked in a process-wide list by size and handed to the next array of that size of ANY handle; address space is
// the one thing there is plenty of (47 bits against the few hundred GiB a process ever reserves here).
std::mutex g_va_mu;
std::multimap<size_t, void *> g_va_free;
hipError_t va_reserve(void **va, size_t bytes, size_t align)
{
    {
        std::lock_guard<std::mutex> lk(g_va_mu);
        auto r = g_va_free.equal_range(bytes);
        for (auto it = r.first; it != r.second; ++it)
            if ((uintptr_t)it->second % align == 0) { *va = it->second; g_va_free.erase(it); return hipSuccess; }
    }
    return hipMemAddressReserve(va, bytes, align, nullptr, 0);
}
void va_release(void *va, size_t bytes)
{
    std::lock_guard<std::mutex> lk(g_va_mu);
    g_va_free.insert({bytes, va});
}

hipError_t flush_translations()
{
    // both of the two things that were seen to flush (300 of 300 re-mappings each, scripts/vmm_stale.hip): neither is a documented contract, and a lost flush is silent corruption
    void *t = nullptr;
    hipError_t e = hipHostMalloc(&t, 4096, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostFree(t);
    hipStream_t s = nullptr;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamDestroy(s);
    return e;
}

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
    for (auto &kv : P->spare)
        for (auto hd : kv.second) (void)hipMemRelease(hd);
    P->spare.clear();
}

jrx_status alloc_chunks(jrx_handle *h, jrx_field_pool *P, size_t bytes, size_t skew, void **out)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = h->device;
    size_t gran = 0;
    JRX_HIP(h, hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (gran == 0) gran = (size_t)2 << 20;
    // "field_chunk_mib" = 0: the whole array is ONE chunk (one hipMemCreate of its size: as contiguous as the driver makes it)
    size_t chunk = h->field_chunk_mib > 0 ? (size_t)h->field_chunk_mib << 20 : bytes + skew;
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t nch = (bytes + skew + chunk - 1) / chunk;
    auto &sp = P->spare[chunk];
    // a batch of new chunks: at least what this array needs, and at least "field_batch_mib" MiB, so that the shuffle mixes the chunks of several arrays
    if (sp.size() < nch) {
        const size_t batch_min = ((size_t)(h->field_batch_mib > 0 ? h->field_batch_mib : 0) << 20) / chunk;
        const size_t want = std::max(nch - sp.size(), batch_min);
        const auto t0 = Clock::now();
        for (size_t c = 0; c < want; c++) {
            hipMemGenericAllocationHandle_t hd;
            const hipError_t e = hipMemCreate(&hd, chunk, &prop, 0);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                if (sp.size() >= nch) break;             // the batch was a wish; the array itself is covered
                return jrx_fail(h, JRX_ERR_HIP, "jrx_field_alloc: hipMemCreate(%zu MiB) -> %s after %lld chunks", chunk >> 20, hipGetErrorString(e), (long long)P->chunks_created);
            }
            sp.push_back(hd);
            P->chunks_created++;
        }
        P->create_ms += ms_since(t0);
    }
    // Fisher-Yates over the spare list, then the array takes the tail
    if (h->field_shuffle)
        for (size_t i = sp.size(); i > 1; i--) std::swap(sp[i - 1], sp[next_rng(P->rng) % i]);
    const auto t1 = Clock::now();
    void *va = nullptr;
    size_t align = (size_t)(h->field_va_align_mib > 0 ? h->field_va_align_mib : 0) << 20;
    if (align < gran) align = gran;
    bool in_arena = false;
    if (h->field_arena_gib > 0) {
        if (!P->arena) {
            void *a = nullptr;
            const size_t want = (size_t)h->field_arena_gib << 30;
            if (va_reserve(&a, want, (size_t)1 << 30) == hipSuccess) { P->arena = (char *)a; P->arena_bytes = want; P->arena_used = 0; }
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
    if (!in_arena) JRX_HIP(h, va_reserve(&va, nch * chunk, align));
    jrx_field_pool::Alloc A;
    A.bytes = bytes; A.kind = 1; A.chunk = chunk; A.in_arena = in_arena; A.skew = skew;
    for (size_t c = 0; c < nch; c++) {
        hipMemGenericAllocationHandle_t hd = sp.back();
        const hipError_t e = hipMemMap((char *)va + c * chunk, chunk, 0, hd, 0);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (A.mapped) (void)hipMemUnmap(va, A.mapped);
            for (auto x : A.chunks) sp.push_back(x);
            if (in_arena) P->arena_free.insert({nch * chunk, va}); else va_release(va, nch * chunk);
            return jrx_fail(h, JRX_ERR_HIP, "jrx_field_alloc: hipMemMap -> %s", hipGetErrorString(e));
        }
        sp.pop_back();
        A.chunks.push_back(hd);
        A.mapped += chunk;
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    hipError_t e = hipMemSetAccess(va, nch * chunk, &acc, 1);
    if (e == hipSuccess) e = flush_translations();       // the range may have been mapped before (arena sub-range, a reservation handed out again)
    // ... and whether the flush took is checked (it is a side effect, not a contract; what a lost one costs: the caller's first writes go to chunks that belong to somebody else).  One
    // word per 2 MiB page is stamped through the range and looked for through a second mapping of the same chunks; not there: flush again, up to four times.
    if (e == hipSuccess) {
        void *alias = nullptr;
        unsigned long long *cnt = nullptr, bad = 1;
        const size_t pages = nch * chunk / ((size_t)2 << 20), strw = ((size_t)2 << 20) / 8;
        hipError_t ea = va_reserve(&alias, nch * chunk, gran);
        for (size_t c = 0; c < nch && ea == hipSuccess; c++) ea = hipMemMap((char *)alias + c * chunk, chunk, 0, A.chunks[c], 0);
        if (ea == hipSuccess) ea = hipMemSetAccess(alias, nch * chunk, &acc, 1);
        if (ea == hipSuccess) ea = hipMalloc((void **)&cnt, 8);
        for (int attempt = 0; attempt < 4 && ea == hipSuccess && bad; attempt++) {
            if (attempt) { h->stat_field_reflushes++; ea = flush_translations(); if (ea != hipSuccess) break; }
            const unsigned long long nonce = 0x9E3779B97F4A7C15ull * (unsigned long long)(P->chunks_created + 7 * attempt + 1) + (unsigned long long)(uintptr_t)va;
            hipLaunchKernelGGL(k_pool_stamp, dim3(64), dim3(256), 0, 0, (unsigned long long *)va, pages, strw, nonce);
            ea = hipMemset(cnt, 0, 8);
            if (ea != hipSuccess) break;
            hipLaunchKernelGGL(k_pool_stamped, dim3(64), dim3(256), 0, 0, (const unsigned long long *)alias, pages, strw, nonce, cnt);
            ea = hipMemcpy(&bad, cnt, 8, hipMemcpyDeviceToHost);
        }
        if (cnt) (void)hipFree(cnt);
        if (alias) { (void)hipMemUnmap(alias, nch * chunk); va_release(alias, nch * chunk); }
        if (ea != hipSuccess) { (void)hipGetLastError(); e = ea; }
        else if (bad) e = hipErrorUnknown;
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipMemUnmap(va, A.mapped);
        for (auto x : A.chunks) sp.push_back(x);
        if (in_arena) P->arena_free.insert({nch * chunk, va}); else va_release(va, nch * chunk);
        return jrx_fail(h, JRX_ERR_HIP, "jrx_field_alloc: hipMemSetAccess / flush / check of the new mapping -> %s", hipGetErrorString(e));
    }
    P->map_ms += ms_since(t1);
    P->live[(char *)va + skew] = std::move(A);
    *out = (char *)va + skew;
    return JRX_OK;
}
// re-map the array at `key` onto `fresh` (as many chunks as it has now), contents carried over; on failure `fresh` stays with the caller
jrx_status remap_with(jrx_handle *h, jrx_field_pool *P, void *key, jrx_field_pool::Alloc &A, const std::vector<hipMemGenericAllocationHandle_t> &fresh)
{
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = h->device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    // The contents travel through a plain hipMalloc buffer of the pool: out of the old chunks before the range is re-mapped, into the new ones after the translations have been flushed.
    hipError_t e = hipSuccess;
    const size_t chunk = A.chunk, nch = fresh.size();
    void *va = (char *)key - A.skew;              // the mapped range starts `skew` bytes before the array
    const size_t nw = (A.bytes + 7) / 8;          // the mapped range is a whole number of chunks
    if (P->stage_bytes < nw * 8) {
        if (P->stage) (void)hipFree(P->stage);
        P->stage = nullptr; P->stage_bytes = 0;
        e = hipMalloc(&P->stage, nw * 8);
        if (e == hipSuccess) P->stage_bytes = nw * 8;
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) { hipLaunchKernelGGL(k_pool_copy, dim3(4096), dim3(256), 0, 0, (double *)P->stage, (const double *)key, nw); e = hipDeviceSynchronize(); }
    if (e != hipSuccess) { (void)hipGetLastError(); return jrx_fail(h, JRX_ERR_HIP, "jrx_tuning_field_reroll: staging %zu bytes -> %s", A.bytes, hipGetErrorString(e)); }
    e = hipMemUnmap(va, A.mapped);
    for (size_t c = 0; c < nch && e == hipSuccess; c++) e = hipMemMap((char *)va + c * chunk, chunk, 0, fresh[c], 0);
    if (e == hipSuccess) e = hipMemSetAccess(va, A.mapped, &acc, 1);
    if (e == hipSuccess) e = flush_translations();
    if (e != hipSuccess) { (void)hipGetLastError(); return jrx_fail(h, JRX_ERR_HIP, "jrx_tuning_field_reroll: re-mapping the array at %p -> %s (the array has lost its backing)", va, hipGetErrorString(e)); }
    // DID THE COPY LAND?  The flush above is a side effect, not a contract; a lost one is silent corruption (the copy goes to the old chunks, the array then shows whatever the new ones
    // held -- NaNs of an earlier test, in the GPU suite).  So the new chunks are mapped a second time at a range that has never been used -- no translation of it can be stale -- and
    // compared with the staged contents there.  Not landed: flush again, copy again, up to four times; then the old chunks (which hold the contents either way: a stale copy wrote
    // the same values back into them) are put back under the array and the re-roll fails.
    void *alias = nullptr;
    unsigned long long *cnt = nullptr, bad = 1;
    // (the second range comes from the parked list when there is one of that size: a stale translation of IT makes the comparison fail, never pass, and the next round -- behind
    // another flush -- sees the truth)
    hipError_t ea = va_reserve(&alias, A.mapped, (size_t)2 << 20);
    for (size_t c = 0; c < nch && ea == hipSuccess; c++) ea = hipMemMap((char *)alias + c * chunk, chunk, 0, fresh[c], 0);
    if (ea == hipSuccess) ea = hipMemSetAccess(alias, A.mapped, &acc, 1);
    if (ea == hipSuccess) ea = hipMalloc((void **)&cnt, 8);
    for (int attempt = 0; attempt < 4 && ea == hipSuccess && bad; attempt++) {
        if (attempt) { P->lost_remaps++; h->stat_field_reflushes++; ea = flush_translations(); if (ea != hipSuccess) break; }
        hipLaunchKernelGGL(k_pool_copy, dim3(4096), dim3(256), 0, 0, (double *)key, (const double *)P->stage, nw);
        ea = hipMemset(cnt, 0, 8);
        if (ea != hipSuccess) break;
        hipLaunchKernelGGL(k_pool_diff, dim3(4096), dim3(256), 0, 0, (const unsigned long long *)((char *)alias + A.skew), (const unsigned long long *)P->stage, nw, cnt);
        ea = hipMemcpy(&bad, cnt, 8, hipMemcpyDeviceToHost);
    }
    if (cnt) (void)hipFree(cnt);
    if (alias) { (void)hipMemUnmap(alias, A.mapped); va_release(alias, A.mapped); }
    if (ea != hipSuccess || bad) {
        (void)hipGetLastError();
        // back onto the chunks the array had
        hipError_t eb = hipDeviceSynchronize();
        if (eb == hipSuccess) eb = hipMemUnmap(va, A.mapped);
        for (size_t c = 0; c < A.chunks.size() && eb == hipSuccess; c++) eb = hipMemMap((char *)va + c * chunk, chunk, 0, A.chunks[c], 0);
        if (eb == hipSuccess) eb = hipMemSetAccess(va, A.mapped, &acc, 1);
        if (eb == hipSuccess) eb = flush_translations();
        (void)hipGetLastError();
        if (ea != hipSuccess) return jrx_fail(h, JRX_ERR_HIP, "jrx_tuning_field_reroll: checking the re-mapped array at %p -> %s%s", va, hipGetErrorString(ea), eb == hipSuccess ? " (the array is back on its chunks)" : " (and the array has lost its backing)");
        return jrx_fail(h, JRX_ERR_HIP, "jrx_tuning_field_reroll: %llu of %zu words written through the re-mapped range %p did not reach its new chunks after four flushes%s", bad, nw, va,
                        eb == hipSuccess ? " (the array is back on its old chunks)" : " (and the array has lost its backing)");
    }
    return JRX_OK;
}
// Give one chunk-backed array new physical backing IN PLACE: its virtual range, and therefore every pointer the caller and the library hold, stays as it is; its contents are carried over.  New chunks: spare ones first, in shuffled order; freshly created ones if the spare list is short; the old chunks join the spare list.  An experiment
// primitive (include/jrx_tuning.h), not part of the drop-in ABI.
jrx_status reroll_one(jrx_handle *h, jrx_field_pool *P, void *key, jrx_field_pool::Alloc &A)
{
    if (A.kind != 1) return JRX_OK;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = h->device;
    const size_t chunk = A.chunk, nch = A.chunks.size();
    auto &sp = P->spare[chunk];
    const auto t0 = Clock::now();
    while (sp.size() < nch) {
        hipMemGenericAllocationHandle_t hd;
        const hipError_t e = hipMemCreate(&hd, chunk, &prop, 0);
        if (e != hipSuccess) { (void)hipGetLastError(); return jrx_fail(h, JRX_ERR_HIP, "jrx_tuning_field_reroll: hipMemCreate(%zu MiB) -> %s", chunk >> 20, hipGetErrorString(e)); }
        sp.push_back(hd);
        P->chunks_created++;
    }
    P->create_ms += ms_since(t0);
    if (h->field_shuffle)
        for (size_t i = sp.size(); i > 1; i--) std::swap(sp[i - 1], sp[next_rng(P->rng) % i]);
    const auto t1 = Clock::now();
    std::vector<hipMemGenericAllocationHandle_t> fresh(sp.end() - (long)nch, sp.end());
    sp.resize(sp.size() - nch);
    { const jrx_status st = remap_with(h, P, key, A, fresh); if (st != JRX_OK) { for (auto hd : fresh) sp.push_back(hd); return st; } }
    for (auto hd : A.prev) sp.push_back(hd);       // an earlier re-roll is thereby kept
    A.prev = std::move(A.chunks);
    A.chunks = fresh;
    P->map_ms += ms_since(t1);
    P->rerolls++;
    return JRX_OK;
}
// back onto the chunks the array had before its last re-roll (nothing to do if there was none, or if it has been kept since)
jrx_status undo_one(jrx_handle *h, jrx_field_pool *P, void *key, jrx_field_pool::Alloc &A)
{
    if (A.kind != 1 || A.prev.empty()) return JRX_OK;
    JRX_TRY(remap_with(h, P, key, A, A.prev));
    for (auto hd : A.chunks) P->spare[A.chunk].push_back(hd);
    A.chunks = std::move(A.prev);
    A.prev.clear();
    return JRX_OK;
}
void keep_one(jrx_field_pool *P, jrx_field_pool::Alloc &A)
{
    for (auto hd : A.prev) P->spare[A.chunk].push_back(hd);
    A.prev.clear();
}
}   // namespace

// internal: arrays a placement search need not move (the kernel it times does not touch them): whole-set re-rolls skip them.  Pointers the pool does not know are ignored.
void jrx_pool_mark_cold(jrx_handle *h, const double *const *ptrs, int n, bool cold)
{
    jrx_field_pool *P = pool_of(h);
    for (int i = 0; i < n; i++) {
        if (!ptrs[i]) continue;
        auto it = P->live.find((void *)ptrs[i]);
        if (it != P->live.end()) it->second.cold = cold;
    }
}

// internal: every large library-owned array (second state sets, ητ) comes from the same place as the caller's
jrx_status jrx_dev_alloc(jrx_handle *h, size_t bytes, void **out)
{
    *out = nullptr;
    if (bytes == 0) bytes = 8;
    jrx_field_pool *P = pool_of(h);
    // small arrays never matter for the placement and would waste a chunk each
    const int kind = (h->field_placement == 1 && bytes < ((size_t)8 << 20)) ? 0 : h->field_placement;
    // "field_skew_bytes": the k-th large array starts (k mod "field_skew_mod") * skew bytes into its allocation, so that element i of different arrays does not sit at the same
    // offset of a 2 MiB page (all large allocations are 2 MiB-aligned otherwise); multiples of 256 B keep every alignment the kernels rely on
    size_t skew = 0;
    if (h->field_skew_bytes > 0 && bytes >= ((size_t)8 << 20)) {
        const int mod = h->field_skew_mod > 0 ? h->field_skew_mod : 32;
        skew = (size_t)(P->large_allocs++ % mod) * ((size_t)h->field_skew_bytes / 256 * 256);
    }
    if (kind == 1) {
        JRX_TRY(alloc_chunks(h, P, bytes, skew, out));
    } else {
        void *p = nullptr;
        if (kind == 2) {
            if (hipExtMallocWithFlags(&p, bytes + skew, hipDeviceMallocContiguous) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
        }
        if (!p) JRX_HIP(h, hipMalloc(&p, bytes + skew));
        jrx_field_pool::Alloc A;
        A.bytes = bytes; A.kind = kind; A.skew = skew;
        P->live[(char *)p + skew] = std::move(A);
        *out = (char *)p + skew;
    }
    if (h->field_ballast_mib > 0 && bytes >= ((size_t)8 << 20)) {
        void *b = nullptr;
        if (hipMalloc(&b, (size_t)h->field_ballast_mib << 20) == hipSuccess) P->ballast.push_back(b); else (void)hipGetLastError();
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
        void *va = (char *)p - A.skew;
        JRX_HIP(h, hipMemUnmap(va, A.mapped));
        if (A.in_arena) P->arena_free.insert({A.mapped, va});
        else va_release(va, A.mapped);
        for (auto hd : A.prev) A.chunks.push_back(hd);
        for (auto hd : A.chunks) P->spare[A.chunk].push_back(hd);
    } else {
        JRX_HIP(h, hipFree((char *)p - A.skew));
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
            void *va = (char *)kv.first - kv.second.skew;
            (void)hipMemUnmap(va, kv.second.mapped);
            if (!kv.second.in_arena) va_release(va, kv.second.mapped);
            for (auto hd : kv.second.chunks) (void)hipMemRelease(hd);
            for (auto hd : kv.second.prev) (void)hipMemRelease(hd);
        } else {
            (void)hipFree((char *)kv.first - kv.second.skew);
        }
    }
    release_spare(P);
    for (void *b : P->ballast) (void)hipFree(b);
    if (P->stage) (void)hipFree(P->stage);
    if (P->arena) va_release(P->arena, P->arena_bytes);
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

jrx_status jrx_tuning_field_reroll(jrx_handle *h, double *p)
{
    if (!h) return JRX_ERR_ARG;
    JRX_TRY(jrx_check_device(h));
    jrx_field_pool *P = pool_of(h);
    if (p) {
        auto it = P->live.find(p);
        if (it == P->live.end()) return jrx_fail(h, JRX_ERR_ARG, "jrx_tuning_field_reroll: %p was not allocated by jrx_field_alloc on this handle", (void *)p);
        return reroll_one(h, P, it->first, it->second);
    }
    for (auto &kv : P->live) if (!kv.second.cold) JRX_TRY(reroll_one(h, P, kv.first, kv.second));
    return JRX_OK;
}

// One draw of the placement search: fresh chunks from the driver for every chunk-backed array (the spare list is emptied first: chunks of earlier draws would only be dealt again),
// with `ballast` bytes of throw-away chunks created in between, one piece before each array -- what the driver hands out next depends on what is held, so the pieces push the arrays
// apart and every draw to other places of the device's memory (pages a launch touches at the same time are best far apart: profiles/r05_placement_search.txt, sections 3 and 8).
static jrx_status draw_spread(jrx_handle *h, size_t ballast)
{
    jrx_field_pool *P = pool_of(h);
    release_spare(P);
    size_t n = 0;
    for (auto &kv : P->live) n += kv.second.kind == 1 && !kv.second.cold;
    if (n == 0) return JRX_OK;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = h->device;
    const size_t piece = ballast / n / ((size_t)2 << 20) * ((size_t)2 << 20);
    std::vector<hipMemGenericAllocationHandle_t> held;
    jrx_status st = JRX_OK;
    for (auto &kv : P->live) {
        if (kv.second.kind != 1 || kv.second.cold) continue;
        if (piece) {
            hipMemGenericAllocationHandle_t b;
            if (hipMemCreate(&b, piece, &prop, 0) == hipSuccess) held.push_back(b); else (void)hipGetLastError();
        }
        st = reroll_one(h, P, kv.first, kv.second);
        if (st != JRX_OK) break;
    }
    for (auto b : held) (void)hipMemRelease(b);
    return st;
}

jrx_status jrx_tuning_field_undo(jrx_handle *h, double *p)
{
    if (!h) return JRX_ERR_ARG;
    JRX_TRY(jrx_check_device(h));
    jrx_field_pool *P = pool_of(h);
    if (p) {
        auto it = P->live.find(p);
        if (it == P->live.end()) return jrx_fail(h, JRX_ERR_ARG, "jrx_tuning_field_undo: %p was not allocated by jrx_field_alloc on this handle", (void *)p);
        return undo_one(h, P, it->first, it->second);
    }
    for (auto &kv : P->live) JRX_TRY(undo_one(h, P, kv.first, kv.second));
    return JRX_OK;
}

jrx_status jrx_tuning_field_keep(jrx_handle *h, double *p)
{
    if (!h) return JRX_ERR_ARG;
    jrx_field_pool *P = pool_of(h);
    if (p) {
        auto it = P->live.find(p);
        if (it == P->live.end()) return jrx_fail(h, JRX_ERR_ARG, "jrx_tuning_field_keep: %p was not allocated by jrx_field_alloc on this handle", (void *)p);
        keep_one(P, it->second);
        return JRX_OK;
    }
    for (auto &kv : P->live) keep_one(P, kv.second);
    return JRX_OK;
}

// The placement search (include/jrx.h): draw, let the caller's probe time whatever it is going to run, keep the draw if it is the fastest so far, undo it otherwise.
jrx_status jrx_field_tune(jrx_handle *h, int32_t draws, jrx_probe_fn probe, void *ctx, double *ms, int32_t *kept)
{
    if (!h) return JRX_ERR_ARG;
    JRX_TRY(jrx_check_device(h));
    if (draws < 0 || draws > 64) return jrx_fail(h, JRX_ERR_ARG, "jrx_field_tune: draws = %d (0 .. 64)", (int)draws);
    if (!probe || !ms) return jrx_fail(h, JRX_ERR_ARG, "jrx_field_tune: probe / ms is NULL");
    for (int d = 0; d <= draws + 1; d++) ms[d] = -1.0;
    // with neighbours every rank makes the same number of probes (they may exchange halos), whatever its own draws come to: a rank that cannot draw (no room) says so to all
    auto agree = [&](bool ok, bool *all) -> jrx_status {
        double v = ok ? 0.0 : 1.0;
        if (jrx_comm_active(h)) JRX_TRY(jrx_allreduce_host(h, &v, 1, 1));
        *all = v == 0.0;
        return JRX_OK;
    };
    auto run = [&](double *out) -> jrx_status {
        const double t = probe(ctx);
        if (!(t > 0.0)) return jrx_fail(h, JRX_ERR_ARG, "jrx_field_tune: the probe returned %g (it reports milliseconds, > 0)", t);
        *out = t;
        return JRX_OK;
    };
    double best = 0.0;
    JRX_TRY(run(&best));
    ms[0] = best;
    int nk = 0;
    // THE POOL.  A placement is good when the chunks its arrays use at the same time lie far apart in the device's memory (profiles/r05_placement_search.txt, sections 3, 8, 12: 22 random
    // chunks out of a pool that spans most of the memory gave 4.81 - 4.85 ms in sixteen of sixteen picks where compact sets gave 4.9 - 6.2).  So where every array that takes part is
    // ONE chunk of one common size ("field_chunk_mib" = that size), the spare list is first filled with chunks for "field_pool_pct" % of the free memory: the draws below then deal
    // random chunks of that pool, and what is not used goes back to the driver at the end.
    if (draws > 0 && h->field_pool_pct > 0) {
        jrx_field_pool *P = pool_of(h);
        size_t S = 0, m = 0;
        bool uniform = true;
        for (auto &kv : P->live) {
            const jrx_field_pool::Alloc &A = kv.second;
            if (A.kind != 1 || A.cold) continue;
            if (A.chunks.size() != 1 || (S && A.chunk != S)) { uniform = false; break; }
            S = A.chunk; m++;
        }
        size_t free_b = 0, total_b = 0;
        if (uniform && m > 0 && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > ((size_t)6 << 30)) {
            const size_t pct = (size_t)(h->field_pool_pct > 90 ? 90 : h->field_pool_pct);
            size_t want = (free_b - ((size_t)6 << 30)) / 100 * pct / S;
            if (want > 16 * m) want = 16 * m;
            hipMemAllocationProp prop = {};
            prop.type = hipMemAllocationTypePinned;
            prop.location.type = hipMemLocationTypeDevice;
            prop.location.id = h->device;
            auto &sp = P->spare[S];
            const auto t0 = Clock::now();
            while (sp.size() < want) {
                hipMemGenericAllocationHandle_t hd;
                if (hipMemCreate(&hd, S, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
                sp.push_back(hd);
                P->chunks_created++;
            }
            P->create_ms += ms_since(t0);
        }
    }
    for (int d = 0; d < draws; d++) {
        // test switch "field_test_fail_draw" = k: this rank's k-th draw fails as if there were no room (tests/test_gpu_two_blocks.py: the ranks must stop together)
        // tuning switch "field_spread_draws" (off: measured to find nothing better -- kernel 4.76 - 4.79 ms after 8 draws either way -- at five times the cost, 21 - 25 s against
        // 2 - 4 s, profiles/r05_placement_search.txt section 9): fresh chunks from the driver for every draw, a different share of the free memory held back meanwhile
        size_t ballast = 0;
        {
            static const double share[8] = {0.0, 0.5, 0.25, 0.75, 0.125, 0.625, 0.375, 0.875};
            size_t free_b = 0, total_b = 0, need = 0;
            for (auto &kv : pool_of(h)->live) if (kv.second.kind == 1) need += kv.second.mapped;
            if (h->field_spread_draws == 1 && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > need + ((size_t)4 << 30))
                ballast = (size_t)((double)(free_b - need - ((size_t)4 << 30)) * share[d % 8]);
        }
        const jrx_status st = h->field_test_fail_draw == d + 1 ? JRX_ERR_HIP : h->field_spread_draws ? draw_spread(h, ballast) : jrx_tuning_field_reroll(h, nullptr);
        bool all = false;
        JRX_TRY(agree(st == JRX_OK, &all));
        if (!all) {                                    // some rank could not make the draw: everybody goes back to what it had and the search ends
            JRX_TRY(jrx_tuning_field_undo(h, nullptr));
            break;
        }
        double t = 0.0;
        const jrx_status sp = run(&t);
        if (sp != JRX_OK) { (void)jrx_tuning_field_undo(h, nullptr); return sp; }
        ms[d + 1] = t;
        if (t < best * 0.997) { best = t; nk++; JRX_TRY(jrx_tuning_field_keep(h, nullptr)); }
        else JRX_TRY(jrx_tuning_field_undo(h, nullptr));
    }
    JRX_TRY(jrx_field_trim(h));                        // the chunks of the draws that lost go back to the driver
    JRX_TRY(run(&ms[draws + 1]));
    if (kept) *kept = nk;
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
    if (h->pool) {
        release_spare(h->pool);
        if (h->pool->stage) { (void)hipFree(h->pool->stage); h->pool->stage = nullptr; h->pool->stage_bytes = 0; }
    }
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
    out[3] = 0;
    for (auto &kv : P->spare) out[3] += (int64_t)kv.second.size();
    out[4] = (int64_t)(P->create_ms * 1e3);      // microseconds spent in hipMemCreate
    out[5] = (int64_t)(P->map_ms * 1e3);         // microseconds spent reserving, mapping and setting access
    return JRX_OK;
}

}   // extern "C"
