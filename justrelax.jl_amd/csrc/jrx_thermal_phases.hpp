// jrx_thermal_phases.hpp -- device helpers of the phase-ratio form of the PT heat-diffusion kernels
// (heatdiffusion_PT!(...; kwargs = (phase = phase_ratios, ...)), src/thermal_diffusion/DiffusionPT_solver.jl:181-305):
// conductivity from the face ratios (DiffusionPT_kernels.jl:366-440), ρCp / radioactive heat from the centre ratios
// (:553-601, :631-668; DiffusionPT_GeoParams.jl:97-175), pseudo-transient coefficients per iteration
// (DiffusionPT_coefficients.jl:123-136).  fn_ratio with args returns a pure phase alone (phases.jl:17-30), without args it is
// the plain weighted sum (:6-15).
#pragma once
#include "jrx_internal.hpp"

struct NoPh {};                                        // array-coefficient / single-rheology forms: nothing extra in the kernel arguments
struct TPh { jrx_thermal_phases m; jrx_thermal_phase_fields f; };
// TPhN<N>: the same with the number of phases as a compile-time constant (1..4): the ratios of a cell / face are loaded in one batch and the phase loops unroll (with the
// run-time count every iteration of every loop is a load the next instruction waits for)
template <int NPH> struct TPhN : TPh {};
template <class P> struct is_tph { static constexpr bool value = false; };
template <> struct is_tph<TPh> { static constexpr bool value = true; };
template <int NPH> struct is_tph<TPhN<NPH>> { static constexpr bool value = true; };
template <class P> struct tph_np { static constexpr int value = 0; };
template <int NPH> struct tph_np<TPhN<NPH>> { static constexpr int value = NPH; };
template <class P> __device__ __forceinline__ int tph_nph(const P &ph)
{
    if constexpr (tph_np<P>::value > 0) return tph_np<P>::value;
    else return ph.m.nphase;
}

__device__ __forceinline__ double tph_density(const jrx_thermal_phases &m, int q, double T, double P)
{
    switch (m.rho_kind[q]) {
    case 1: return m.rho0[q] * (1.0 - m.alpha[q] * (T - m.T0[q]) + m.beta[q] * (P - m.P0[q]));
    case 2: return m.rho0[q] * (1.0 - m.alpha[q] * (T - m.T0[q]));
    case 3: return m.rho0[q] * exp(m.beta[q] * (P - m.P0[q]));
    default: return m.rho0[q];
    }
}
// the ratios of one cell / face: a register copy when the count is a constant (one batch of loads), the array itself otherwise
template <int NPH> struct TphRatios {
    double v[NPH > 0 ? NPH : 1];
    const double *p;
    __device__ __forceinline__ TphRatios(const double *__restrict__ r) : p(r)
    {
        if constexpr (NPH > 0) {
#pragma unroll
            for (int q = 0; q < NPH; q++) v[q] = r[q];
        }
    }
    __device__ __forceinline__ double operator[](int q) const { if constexpr (NPH > 0) return v[q]; else return p[q]; }
};
template <int NPH = 0>
__device__ __forceinline__ double tph_rhoCp(const jrx_thermal_phases &m, const double *__restrict__ rp, double T, double P)
{
    const TphRatios<NPH> r(rp);
    const int np = NPH > 0 ? NPH : m.nphase;
    double x = 0.0;
#pragma unroll
    for (int q = 0; q < np; q++) {
        const double rq = r[q];
        if (rq == 1.0) return (m.Cp[q] * tph_density(m, q, T, P)) * rq;
        x += (rq == 0.0) ? 0.0 : (m.Cp[q] * tph_density(m, q, T, P)) * rq;
    }
    return x;
}
template <int NPH = 0>
__device__ __forceinline__ double tph_cond(const jrx_thermal_phases &m, const double *__restrict__ rp)
{
    const TphRatios<NPH> r(rp);
    const int np = NPH > 0 ? NPH : m.nphase;
    double x = 0.0;
#pragma unroll
    for (int q = 0; q < np; q++) {
        const double rq = r[q];
        if (rq == 1.0) return m.k[q] * rq;
        x += (rq == 0.0) ? 0.0 : m.k[q] * rq;
    }
    return x;
}
template <int NPH = 0>
__device__ __forceinline__ double tph_Hr(const jrx_thermal_phases &m, const double *__restrict__ rp)
{
    const TphRatios<NPH> r(rp);
    const int np = NPH > 0 ? NPH : m.nphase;
    double x = 0.0;
#pragma unroll
    for (int q = 0; q < np; q++) x += (r[q] == 0.0) ? 0.0 : m.Hr[q] * r[q];
    return x;
}
template <int NPH = 0>
__device__ __forceinline__ void tph_pt_coeffs(const jrx_thermal_phases &m, const double *__restrict__ r, double T, double P, double _dt, double &thetar_dtau, double &dtau_rho)
{
    const double pi = 3.14159265358979323846;
    const double rcp = tph_rhoCp<NPH>(m, r, T, P);
    const double _K = 1.0 / tph_cond<NPH>(m, r);
    const double _Re = 1.0 / (pi + sqrt(pi * pi + rcp * (m.max_lxyz * m.max_lxyz) * _K * _dt));
    thetar_dtau = m.max_lxyz / m.Vpdtau * _Re;
    dtau_rho = m.Vpdtau * m.max_lxyz * _K * _Re;
}

inline jrx_status tph_check(jrx_handle *h, const jrx_thermal_phases *ph, const jrx_thermal_phase_fields *pf, bool three)
{
    if (!ph || !pf) return jrx_fail(h, JRX_ERR_ARG, "phase-ratio thermal form: null phases / phase fields");
    if (ph->nphase < 1 || ph->nphase > JRX_MAXPHASE) return jrx_fail(h, JRX_ERR_ARG, "phase-ratio thermal form: nphase out of range");
    if (!pf->P || !pf->phase_c || !pf->phase_qx || !pf->phase_qy || (three && !pf->phase_qz))
        return jrx_fail(h, JRX_ERR_ARG, "phase-ratio thermal form: args.P and the centre / face phase ratios are required");
    if (!(ph->max_lxyz > 0.0) || !(ph->Vpdtau > 0.0)) return jrx_fail(h, JRX_ERR_ARG, "phase-ratio thermal form: max_lxyz and Vpdtau must be positive");
    return JRX_OK;
}

// thermal2d.hip: enqueue update_pt_thermal_arrays! on stream s (2D: nz = 1, ndim = 2)
jrx_status jrx_enqueue_pt_thermal_arrays(jrx_handle *h, hipStream_t s, double *th, double *dr, const double *T, int nx, int ny, int nz, int ndim, double _dt, const TPh &ph);
