/* Plain-C caller of the drop-in boundary: include/jrx.h must be a valid C99 header and the host-only entry points must be callable
 * from C without a GPU.  Built and run by tests/test_host_abi.py (gcc, links libjrx_hip.so). */
#include <stdio.h>
#include <string.h>
#include "jrx.h"

int main(void)
{
    const int64_t n[3] = {512, 512, 512};
    const int32_t dims_in[3] = {0, 0, 0}, periods[3] = {0, 0, 0};
    jrx_cart c;
    memset(&c, 0, sizeof(c));
    if (jrx_cart_create(5, 8, n, dims_in, periods, &c) != JRX_OK) return 1;
    printf("dims %d %d %d coords %d %d %d\n", c.dims[0], c.dims[1], c.dims[2], c.coords[0], c.coords[1], c.coords[2]);
    printf("neighbors %d %d %d %d %d %d\n", c.neighbor[0][0], c.neighbor[0][1], c.neighbor[1][0], c.neighbor[1][1], c.neighbor[2][0], c.neighbor[2][1]);
    int64_t sl, sr, rl, rr;
    /* Vx of a 512-cell block: 513 planes in x, overlap 3 -> sends planes 2 and 510, receives into 0 and 512 */
    if (jrx_halo_planes(512, 513, &sl, &sr, &rl, &rr) != JRX_OK) return 2;
    printf("halo planes %lld %lld %lld %lld\n", (long long)sl, (long long)sr, (long long)rl, (long long)rr);
    printf("nx_g %lld\n", (long long)jrx_n_global(512, 2, 0));
    return 0;
}
