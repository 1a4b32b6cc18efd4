// x2_tu.hip -- the one- and two-iterations-per-launch kernels as a translation unit of their own, so that the SAME sources can be compiled twice with different floating-point flags and
// timed in one process (scripts/kbench_x2t.hip):
//   exact:      -ffp-contract=off -fno-fast-math                  (the library's flags: the oracle's bits)            -DX2_SUFFIX=exact
//   tolerance:  -ffp-contract=fast -fapprox-func                  (fused multiply-adds, v_rcp_f64 + Newton divisions) -DX2_SUFFIX=tol
// SweepArgs / FusedBC live in an anonymous namespace of stokes3d_kernels.hpp: they cross the TU boundary as bytes.
#include <hip/hip_runtime.h>
#include "jrx_internal.hpp"
#include "stokes3d_kernels.hpp"
#include "fused_x2.hpp"
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
extern "C" size_t CAT(x2_sizeof_args_, X2_SUFFIX)(void) { return sizeof(SweepArgs); }
extern "C" size_t CAT(x2_sizeof_bc_, X2_SUFFIX)(void) { return sizeof(FusedBC); }
// kind: 1 = k_fused3d<64,4,8,VISC,HIF,VFOLD> (one iteration, the shipped one-launch form with body-force loads), 18 = the same with the 64 x 8 tile,
//       2 = k_fused3d_x2<64,TY,KZ> (two iterations), ty / kz as given
extern "C" int CAT(x2_launch_, X2_SUFFIX)(int kind, int ty, int kz, const void *args, const void *bcp, int nx, int ny, int nz)
{
    SweepArgs a;
    FusedBC bc;
    memcpy((void *)&a, args, sizeof(a));
    memcpy((void *)&bc, bcp, sizeof(bc));
    if (kind == 1) {
        constexpr int TX = 64, TY = 4, KZ = 8;
        const int ntx = (nx + TX - 3) / (TX - 2), nty = (ny + TY - 2) / (TY - 1), ntz = (nz + KZ - 1) / KZ;
        hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, 4, 1, false, 1, false, true, 3, 1, 0, true, true, true>), dim3(ntx * nty * ntz), dim3(TX * TY), 0, 0, a, bc, ntx, nty, 0, 0, 0);
        return 0;
    }
    if (kind == 18) {
        constexpr int TX = 64, TY = 8, KZ = 8;
        const int ntx = (nx + TX - 3) / (TX - 2), nty = (ny + TY - 2) / (TY - 1), ntz = (nz + KZ - 1) / KZ;
        hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, 2, 1, false, 4, false, true, 3, 1, 0, true, true, true>), dim3(ntx * nty * ntz), dim3(TX * TY), 0, 0, a, bc, ntx, nty, 0, 0, 0);
        return 0;
    }
#define X2(TY_, KZ_)                                                                                                                    \
    if (kind == 2 && ty == TY_ && kz == KZ_) {                                                                                          \
        constexpr int TX = 64;                                                                                                          \
        const int ntx = (nx + TX - 5) / (TX - 4), nty = (ny + TY_ - 4) / (TY_ - 3), ntz = (nz + KZ_ - 1) / KZ_;                         \
        hipLaunchKernelGGL((k_fused3d_x2<TX, TY_, KZ_, 1>), dim3(ntx * nty * ntz), dim3(TX * TY_), 0, 0, a, bc, ntx, nty);              \
        return 0;                                                                                                                       \
    }
    X2(12, 16) X2(12, 32) X2(8, 16) X2(8, 32)
    return 1;
}
