#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
// each lane fetches one double (per-lane address) straight into LDS as two dwords; after the barrier everybody reads lane l's value
__global__ void k(const double *in, double *out, int n)
{
    __shared__ unsigned int sh[2][128];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int idx = (blockIdx.x * 64 + lane) * 3 % n;      // scattered per-lane source
    if (w == 0) {
        const char *g = (const char *)(in + idx);
        __builtin_amdgcn_global_load_lds((glb_void *)g, (lds_void *)&sh[0][0], 4, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void *)(g + 4), (lds_void *)&sh[0][64], 4, 0, 0);
    }
    __syncthreads();
    const unsigned lo = sh[0][lane], hi = sh[0][64 + lane];
    out[blockIdx.x * 128 + threadIdx.x] = __hiloint2double((int)hi, (int)lo) * (w + 1);
}
int main()
{
    const int n = 1 << 16, nb = 64;
    std::vector<double> h(n);
    for (int i = 0; i < n; i++) h[i] = 1.0 + i * 0.001;
    double *d, *o;
    hipMalloc(&d, n * 8); hipMalloc(&o, nb * 128 * 8);
    hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(nb), dim3(128), 0, 0, d, o, n);
    std::vector<double> r(nb * 128);
    hipMemcpy(r.data(), o, nb * 128 * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < nb; b++)
        for (int t = 0; t < 128; t++) {
            const int lane = t & 63, w = t >> 6;
            const double want = h[(b * 64 + lane) * 3 % n] * (w + 1);
            if (r[b * 128 + t] != want) bad++;
        }
    printf("bad = %d\n", bad);
    return bad != 0;
}
