// fieldpool.hip -- device memory for the state arrays, owned by the library (jrx_field_alloc / jrx_field_free).
//
// Replaces the array constructor the backend owns in the reference: StokesArrays(::Type{AMDGPUBackend}, ni) -> @zeros(ni...) -> ROCArray
// (src/ext/AMDGPU/3D.jl:46-48, src/types/constructors/stokes.jl:279-303).  Why the library wants a say in it: the same launch of the 512^3 kernel takes
// 4.7 .. 6.9 ms depending on which physical pages its arrays got (profiles/r05_placement_search.txt) -- same traffic, same plain bandwidth.  What was measured to
// work without timing anything (section 13 there): every large array ONE physical chunk, the chunks picked AT RANDOM out of a pool that spans most of the device's
// memory (4.81 - 4.85 ms in sixteen of sixteen picks, where the first chunks the driver hands out give 4.95 / 5.64 ms).  That is what "field_placement" = 1 does,
// at allocation time:
//
//   * a large request of a chunk size of which no array is live fills a pool of hipMemCreate chunks of that size for "field_pool_pct" % of the free memory;
//   * every array takes random chunks of the pool and maps them ONCE, at a virtual range that has NEVER been used before (a fresh hipMemAddressReserve);
//   * a freed array's range is unmapped and retired for good (never handed out again, never given back to the runtime); its chunks return to the pool;
//   * jrx_field_trim releases the chunks nobody uses (call it when the arrays of a run have been made); arrays made later -- while arrays of the run are live -- get chunks
//     created on the spot (no second pool: a solve! behind the trim does not spend seconds, and 70 % of the memory, on one).
//
// Nothing here ever maps anything at an address that has been mapped before.  Round 5's in-place re-mapping (hipMemUnmap + hipMemMap under a live array, the
// placement search on top of it) is gone from the library: on ROCm 7.2 the shaders keep the OLD translation of a re-mapped address until some unrelated driver
// call happens to flush it (scripts/vmm_stale.hip), the flush is a side effect and not a contract, and on the round-5 driver's box a search left NaNs behind
// (DESIGN.md section 3).
//
// Placement kinds ("field_placement"):
//   0  hipMalloc (what a ROCArray / torch tensor gets)
//   1  pool-dealt chunks (above); arrays below 8 MiB come from hipMalloc
//   2  physically contiguous (hipDeviceMallocContiguous): the slowest placement there is, for A/B runs only
// Host-only code; every entry point requires the handle's device to be current.
#include "jrx_internal.hpp"
#include "jrx_tuning.h"
#include <algorithm>
#include <chrono>
#include <map>
#include <set>
#include <vector>

struct jrx_field_pool {
    struct Alloc { size_t bytes = 0, mapped = 0, chunk = 0; int kind = 0; std::vector<hipMemGenericAllocationHandle_t> chunks; };
    std::map<void *, Alloc> live;
    std::map<size_t, std::vector<hipMemGenericAllocationHandle_t>> spare;      // created, unmapped chunks by their size: the pool
    std::set<size_t> pooled;                      // chunk sizes that have (had) a pool: their freed chunks stay in `spare` until jrx_field_trim
    std::vector<std::pair<void *, size_t>> retired;   // virtual ranges of freed arrays: never used again
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    double create_ms = 0, map_ms = 0;
    int64_t chunks_created = 0, bytes_live = 0;
};

namespace {
using Clock = std::chrono::steady_clock;
double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }
uint64_t next_rng(uint64_t &s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return s >> 17; }

constexpr size_t kLarge = (size_t)8 << 20;        // arrays below this never matter for the placement and would waste a chunk each
constexpr size_t kPoolChunkMin = (size_t)128 << 20;   // chunk sizes below this get no pool (tests and experiments with small chunks)
constexpr size_t kHeadroom = (size_t)6 << 30;     // free memory the pool never takes

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

hipMemAllocationProp device_prop(const jrx_handle *h)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = h->device;
    return prop;
}

jrx_status alloc_chunks(jrx_handle *h, jrx_field_pool *P, size_t bytes, void **out)
{
    const hipMemAllocationProp prop = device_prop(h);
    size_t gran = 0;
    JRX_HIP(h, hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (gran == 0) gran = (size_t)2 << 20;
    // "field_chunk_mib" = 0: the whole array is ONE chunk of its own size (no pool: the sizes differ)
    size_t chunk = h->field_chunk_mib > 0 ? (size_t)h->field_chunk_mib << 20 : bytes;
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t nch = (bytes + chunk - 1) / chunk;
    auto &sp = P->spare[chunk];
    const auto t0 = Clock::now();
    // the pool: when no array of this chunk size is live (the first array of a run), chunks for "field_pool_pct" % of what is free now (less a headroom)
    size_t want = sp.size() < nch ? nch - sp.size() : 0;
    if (h->field_chunk_mib > 0 && chunk >= kPoolChunkMin && h->field_pool_pct > 0) {
        bool any_live = false;
        for (auto &kv : P->live) if (kv.second.kind == 1 && kv.second.chunk == chunk) { any_live = true; break; }
        if (!any_live) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > kHeadroom) {
                const size_t pct = (size_t)std::min(h->field_pool_pct, 90);
                const size_t pool = (free_b - kHeadroom) / 100 * pct / chunk;
                if (pool > sp.size()) want = std::max(want, pool - sp.size());
            } else (void)hipGetLastError();
        }
        P->pooled.insert(chunk);
    }
    for (size_t c = 0; c < want; c++) {
        hipMemGenericAllocationHandle_t hd;
        const hipError_t e = hipMemCreate(&hd, chunk, &prop, 0);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (sp.size() >= nch) break;                 // the pool was a wish; the array itself is covered
            return jrx_fail(h, JRX_ERR_HIP, "jrx_field_alloc: hipMemCreate(%zu MiB) -> %s after %lld chunks", chunk >> 20, hipGetErrorString(e), (long long)P->chunks_created);
        }
        sp.push_back(hd);
        P->chunks_created++;
    }
    P->create_ms += ms_since(t0);
    const auto t1 = Clock::now();
    // a virtual range nobody has used before
    void *va = nullptr;
    JRX_HIP(h, hipMemAddressReserve(&va, nch * chunk, gran, nullptr, 0));
    jrx_field_pool::Alloc A;
    A.bytes = bytes; A.kind = 1; A.chunk = chunk;
    hipError_t e = hipSuccess;
    for (size_t c = 0; c < nch && e == hipSuccess; c++) {
        // a random chunk of the pool ("field_shuffle" = 0: in the order of creation, A/B of the dealing itself)
        const size_t pick = h->field_shuffle ? (size_t)(next_rng(P->rng) % sp.size()) : 0;
        const hipMemGenericAllocationHandle_t hd = sp[pick];
        e = hipMemMap((char *)va + c * chunk, chunk, 0, hd, 0);
        if (e != hipSuccess) break;
        if (h->field_shuffle) { sp[pick] = sp.back(); sp.pop_back(); } else sp.erase(sp.begin());
        A.chunks.push_back(hd);
        A.mapped += chunk;
    }
    if (e == hipSuccess) {
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(va, nch * chunk, &acc, 1);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (A.mapped) (void)hipMemUnmap(va, A.mapped);
        for (auto x : A.chunks) sp.push_back(x);
        P->retired.push_back({va, nch * chunk});
        return jrx_fail(h, JRX_ERR_HIP, "jrx_field_alloc: mapping %zu chunks of %zu MiB -> %s", nch, chunk >> 20, hipGetErrorString(e));
    }
    P->map_ms += ms_since(t1);
    P->live[va] = std::move(A);
    *out = va;
    return JRX_OK;
}
}   // namespace

// internal: every large library-owned array (second state sets, ητ) comes from the same place as the caller's
jrx_status jrx_dev_alloc(jrx_handle *h, size_t bytes, void **out, int tag)
{
    *out = nullptr;
    if (bytes == 0) bytes = 8;
    jrx_field_pool *P = pool_of(h);
    const int kind = (h->field_placement == 1 && bytes < kLarge) ? 0 : h->field_placement;
    if (kind == 1) {
        JRX_TRY(alloc_chunks(h, P, bytes, out));
    } else {
        void *p = nullptr;
        if (kind == 2) {
            if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocContiguous) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
        }
        if (!p) {
            hipError_t e = hipMalloc(&p, bytes);
            if (e != hipSuccess && P->spare.size()) {        // the pool's unused chunks may be what is in the way
                (void)hipGetLastError();
                release_spare(P);
                e = hipMalloc(&p, bytes);
            }
            JRX_HIP(h, e);
        }
        jrx_field_pool::Alloc A;
        A.bytes = bytes; A.kind = kind;
        P->live[p] = std::move(A);
        *out = p;
    }
    // test switch "scratch_poison" (bit mask by `tag`): what an array holds before its first use must not matter (every entry a kernel reads has been written before) -- every byte 0xFF (NaNs)
    // (hipMemset of device memory returns before the fill has run and the handle's streams do not wait for the null stream: without the synchronisation the fill would land
    // on top of whatever the first kernels on those streams have written by then -- round 6 took exactly that for reads of unwritten memory in the coupled pipelines)
    if (h->scratch_poison & tag) { JRX_HIP(h, hipMemset(*out, 0xFF, bytes)); JRX_HIP(h, hipDeviceSynchronize()); }
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
        // nothing of this device may still be using the range (one synchronisation per freed array: a binding's finalizers pay it)
        JRX_HIP(h, hipDeviceSynchronize());
        JRX_HIP(h, hipMemUnmap(p, A.mapped));
        P->retired.push_back({p, A.mapped});             // the range is never mapped again
        if (P->pooled.count(A.chunk)) for (auto hd : A.chunks) P->spare[A.chunk].push_back(hd);
        else for (auto hd : A.chunks) (void)hipMemRelease(hd);      // no pool of that size: the memory goes back to the driver now
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
            for (auto hd : kv.second.chunks) (void)hipMemRelease(hd);
        } else {
            (void)hipFree(kv.first);
        }
    }
    release_spare(P);
    // the retired ranges stay reserved for the life of the process: address space is the one thing there is plenty of (47 bits against the few hundred GiB a
    // process ever reserves here), and a range that hipMemAddressFree has returned was seen to come back from hipMalloc with the runtime's books in disorder (round 5)
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

jrx_status jrx_field_list(jrx_handle *h, int64_t cap, double **ptrs, int64_t *bytes, int64_t *count)
{
    if (!h) return JRX_ERR_ARG;
    if (!count) return jrx_fail(h, JRX_ERR_ARG, "jrx_field_list: count is NULL");
    jrx_field_pool *P = pool_of(h);
    int64_t n = 0;
    for (auto &kv : P->live) {
        if (n < cap && ptrs) ptrs[n] = (double *)kv.first;
        if (n < cap && bytes) bytes[n] = kv.second.kind == 1 ? (int64_t)kv.second.bytes : -(int64_t)kv.second.bytes;       // negative: not chunk-backed
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
    out[3] = 0;
    for (auto &kv : P->spare) out[3] += (int64_t)kv.second.size();
    out[4] = (int64_t)(P->create_ms * 1e3);      // microseconds spent in hipMemCreate
    out[5] = (int64_t)(P->map_ms * 1e3);         // microseconds spent reserving, mapping and setting access
    return JRX_OK;
}

}   // extern "C"
