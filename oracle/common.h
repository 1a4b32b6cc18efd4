/* oracle/common.h -- TEST INFRASTRUCTURE ONLY (see jrx_oracle.h). Index helpers shared by the
 * restated kernels.  0-based C indices; the reference is 1-based, so reference A[i+1,j,k] at
 * 1-based (i,j,k) is A3(A,n1,n2,i+1,j,k) at the same 0-based (i,j,k) shifted consistently. */
#ifndef ORC_COMMON_H
#define ORC_COMMON_H
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#define IDX2(n1, i, j) ((size_t)(i) + (size_t)(n1) * (size_t)(j))
#define IDX3(n1, n2, i, j, k) ((size_t)(i) + (size_t)(n1) * ((size_t)(j) + (size_t)(n2) * (size_t)(k)))

static inline int64_t clampi(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline double inv(double x) { return 1.0 / x; }

/* face bits -- must match include/jrx.h JRX_FACE_* */
enum { F_LEFT = 1, F_RIGHT = 2, F_FRONT = 4, F_BACK = 8, F_TOP = 16, F_BOT = 32 };

/* src/rheology/StressUpdate.jl:70 */
static inline double compute_dtau_r(double theta_dtau, double eta, double _Gdt)
{
    return inv(theta_dtau + fma(eta, _Gdt, 1.0));
}
/* src/stokes/StressKernels.jl:2-5 */
/* compute_stress_increment(τ, τ_o, η, Δε, _G, dτ_r, dt) -- StressKernels.jl:18-21 (strain-increment form) */
static inline double stress_increment_dt(double t, double to, double eta, double de, double _G, double dtau_r, double dt)
{
    return dtau_r * fma(2.0 * eta, de, fma(-(t - to) * eta, _G, -t * dt));
}
static inline double stress_increment(double t, double to, double eta, double e, double _Gdt, double dtau_r)
{
    return dtau_r * fma(2.0 * eta, e, fma(-(t - to) * eta, _Gdt, -t));
}
#endif
