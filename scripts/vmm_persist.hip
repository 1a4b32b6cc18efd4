// vmm_persist.hip -- does the content of a hipMemCreate handle survive hipMemUnmap / hipMemMap at another address?  (jrx_field_reroll stages the new chunks at a temporary range.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main()
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t sz = (size_t)8 << 20;
    hipMemGenericAllocationHandle_t hd;
    CK(hipMemCreate(&hd, sz, &prop, 0));
    void *a = nullptr, *b = nullptr;
    CK(hipMemAddressReserve(&a, sz, 0, nullptr, 0));
    CK(hipMemMap(a, sz, 0, hd, 0)); CK(hipMemSetAccess(a, sz, &acc, 1));
    std::vector<unsigned> h(sz / 4), g(sz / 4);
    for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned)(i * 2654435761u);
    CK(hipMemcpy(a, h.data(), sz, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    CK(hipMemUnmap(a, sz));
    CK(hipMemAddressReserve(&b, sz, 0, nullptr, 0));
    CK(hipMemMap(b, sz, 0, hd, 0)); CK(hipMemSetAccess(b, sz, &acc, 1));
    CK(hipMemcpy(g.data(), b, sz, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < h.size(); i++) bad += h[i] != g[i];
    printf("unmap + map at another address: %zu of %zu words differ (a = %p, b = %p)\n", bad, h.size(), a, b);
    // D2D copy between two mapped ranges, then switch
    void *c = nullptr;
    hipMemGenericAllocationHandle_t hd2;
    CK(hipMemCreate(&hd2, sz, &prop, 0));
    CK(hipMemAddressReserve(&c, sz, 0, nullptr, 0));
    CK(hipMemMap(c, sz, 0, hd2, 0)); CK(hipMemSetAccess(c, sz, &acc, 1));
    CK(hipMemcpy(c, b, sz, hipMemcpyDeviceToDevice));
    CK(hipDeviceSynchronize());
    CK(hipMemUnmap(c, sz));
    CK(hipMemUnmap(b, sz));
    CK(hipMemMap(b, sz, 0, hd2, 0)); CK(hipMemSetAccess(b, sz, &acc, 1));
    CK(hipMemcpy(g.data(), b, sz, hipMemcpyDeviceToHost));
    bad = 0;
    for (size_t i = 0; i < h.size(); i++) bad += h[i] != g[i];
    printf("copy to a second handle staged elsewhere, then mapped in place: %zu words differ\n", bad);
    {   // several 2 MiB chunks per range, unmapped in ONE call, as jrx_field_reroll does
        const size_t ch = (size_t)2 << 20; const int n = 5;
        hipMemGenericAllocationHandle_t o[n], f[n];
        void *va = nullptr, *tmp = nullptr;
        CK(hipMemAddressReserve(&va, n * ch, 0, nullptr, 0));
        for (int q = 0; q < n; q++) { CK(hipMemCreate(&o[q], ch, &prop, 0)); CK(hipMemMap((char *)va + q * ch, ch, 0, o[q], 0)); }
        CK(hipMemSetAccess(va, n * ch, &acc, 1));
        std::vector<unsigned> hh(n * ch / 4), gg(n * ch / 4);
        for (size_t i = 0; i < hh.size(); i++) hh[i] = (unsigned)(i * 40503u + 7u);
        CK(hipMemcpy(va, hh.data(), n * ch, hipMemcpyHostToDevice));
        CK(hipMemAddressReserve(&tmp, n * ch, 0, nullptr, 0));
        for (int q = 0; q < n; q++) { CK(hipMemCreate(&f[q], ch, &prop, 0)); CK(hipMemMap((char *)tmp + q * ch, ch, 0, f[q], 0)); }
        CK(hipMemSetAccess(tmp, n * ch, &acc, 1));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(tmp, va, n * ch - 4096, hipMemcpyDeviceToDevice));
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(tmp, n * ch)); CK(hipMemAddressFree(tmp, n * ch));
        CK(hipMemUnmap(va, n * ch));
        for (int q = 0; q < n; q++) CK(hipMemMap((char *)va + q * ch, ch, 0, f[q], 0));
        CK(hipMemSetAccess(va, n * ch, &acc, 1));
        CK(hipMemcpy(gg.data(), va, n * ch, hipMemcpyDeviceToHost));
        size_t bad2 = 0;
        for (size_t i = 0; i < hh.size() - 1024; i++) bad2 += hh[i] != gg[i];
        printf("five 2 MiB chunks, one unmap call, re-mapped onto the staged chunks: %zu words differ\n", bad2);
    }
    return 0;
}
