// jrx_material.hpp -- per-phase material laws of the rheology table (jrx_rheology): density, strain softening, creep viscosity.
// The reference delegates these to GeoParams.jl (compute_density, softening_C / softening_ϕ, compute_viscosity_τII; call sites
// rheology/BuoyancyForces.jl:37-60, rheology/StressUpdate.jl:305-381, rheology/Viscosity.jl:142-167); the forms are stated in include/jrx.h.
#pragma once
#include "jrx_internal.hpp"

__device__ __forceinline__ double mat_density(const jrx_rheology &rh, int q, double T, double P)
{
    switch (rh.rho_kind[q]) {
    case 1: return rh.rho0[q] * (1.0 - rh.alpha[q] * (T - rh.T0[q]) + rh.beta[q] * (P - rh.P0[q]));
    case 2: return rh.rho0[q] * (1.0 - rh.alpha[q] * (T - rh.T0[q]));
    case 3: return rh.rho0[q] * exp(rh.beta[q] * (P - rh.P0[q]));
    default: return rh.rho0[q];
    }
}
// fn_ratio(compute_density, rheology, ratio, args) -- src/phases/phases.jl:17-30
__device__ __forceinline__ double mat_density_ratio(const jrx_rheology &rh, const double *r, double T, double P)
{
    double x = 0.0;
    for (int q = 0; q < rh.nphase; q++) {
        const double rq = r[q];
        if (rq == 1.0) return mat_density(rh, q, T, P) * rq;
        x += (rq == 0.0) ? 0.0 : mat_density(rh, q, T, P) * rq;
    }
    return x;
}
static inline bool mat_density_is_constant(const jrx_rheology *rh)
{
    for (int q = 0; q < rh->nphase; q++)
        if (rh->rho_kind[q] != 0) return false;
    return true;
}
static inline bool mat_has_softening(const jrx_rheology *rh)
{
    for (int q = 0; q < rh->nphase; q++)
        if (rh->softC_kind[q] != 0 || rh->softphi_kind[q] != 0) return true;
    return false;
}

__device__ __forceinline__ double mat_soften(int kind, double a, double b, double c, double d, double EII, double v0)
{
    if (kind == 1) {
        if (EII >= d) return a;
        if (EII <= c) return b;
        return b + (a - b) / (d - c) * (EII - c);
    }
    if (kind == 2) return a - 0.5 * b * erfc(-(EII - c) / d);
    return v0;
}
__device__ __forceinline__ double mat_cohesion(const jrx_rheology &rh, int q, double EII)
{
    return mat_soften(rh.softC_kind[q], rh.softC_a[q], rh.softC_b[q], rh.softC_c[q], rh.softC_d[q], EII, rh.C[q]);
}
__device__ __forceinline__ void mat_friction(const jrx_rheology &rh, int q, double EII, double &sinphi, double &cosphi)
{
    if (rh.softphi_kind[q] == 0) { sinphi = rh.sinphi[q]; cosphi = rh.cosphi[q]; return; }
    const double phi = mat_soften(rh.softphi_kind[q], rh.softphi_a[q], rh.softphi_b[q], rh.softphi_c[q], rh.softphi_d[q], EII, rh.phi_deg[q]);
    const double rad = phi * (3.14159265358979323846 / 180.0);
    sinphi = sin(rad); cosphi = cos(rad);
}
__device__ __forceinline__ double mat_creep_viscosity(const jrx_rheology &rh, int q, double T, double P)
{
    if (rh.visc_kind[q] == 1) {
        const double e = rh.eta[q] * exp((rh.Ea[q] + P * rh.Va[q]) / (rh.Rgas[q] * T) - rh.Ea[q] / (rh.Rgas[q] * rh.Tref[q]));
        return fmin(fmax(e, rh.visc_lo[q]), rh.visc_hi[q]);
    }
    return rh.eta[q];
}
// fn_viscosity(rheology[q].CompositeRheology[1], AII, args): AII is a stress invariant (compute_viscosity_τII, tau = true) or a strain-rate invariant
// (compute_viscosity_εII); only the power-law creep (visc_kind 2) reads it
__device__ __forceinline__ double mat_viscosity(const jrx_rheology &rh, int q, double AII, double T, double P, bool tau)
{
    if (rh.visc_kind[q] != 2) return mat_creep_viscosity(rh, q, T, P);
    const double n = rh.creep_n[q], H = rh.Ea[q] + P * rh.Va[q], RT = rh.Rgas[q] * T;
    if (tau) {
        const double eps = rh.creep_A[q] * pow(AII * rh.creep_FT[q], n) * exp(-H / RT) / rh.creep_FE[q];
        return 0.5 * AII / eps;
    }
    const double t = pow(rh.creep_A[q], -1.0 / n) * pow(AII * rh.creep_FE[q], 1.0 / n) * exp(H / (n * RT)) / rh.creep_FT[q];
    return 0.5 * t / AII;
}
// compute_phase_viscosity (rheology/Viscosity.jl:599-619): a phase above 0.999 alone, else the ratio-weighted harmonic mean
__device__ __forceinline__ double mat_phase_viscosity(const jrx_rheology &rh, const double *r, double AII, double T, double P, bool tau)
{
    for (int q = 0; q < rh.nphase; q++)
        if (r[q] > 0.999) return mat_viscosity(rh, q, AII, T, P, tau);
    double s = 0.0;
    for (int q = 0; q < rh.nphase; q++)
        if (r[q] != 0.0) s += (1.0 / mat_viscosity(rh, q, AII, T, P, tau)) * r[q];
    return 1.0 / s;
}
// the invariant the viscosity kernels form from (xx, yy, xy) (Viscosity.jl:394-404): eps() on the normal components of an all-zero tensor
__device__ __forceinline__ double mat_visc_invariant2(double xx, double yy, double xy)
{
    const double a0 = (xx == 0.0 && yy == 0.0 && xy == 0.0) ? 2.220446049250313e-16 : 0.0;
    const double x = a0 + xx, y = -a0 + yy;
    return sqrt(0.5 * (x * x + y * y) + xy * xy);
}
__host__ __device__ static inline bool mat_viscosity_reads_fields(const jrx_rheology *rh)
{
    for (int q = 0; q < rh->nphase; q++)
        if (rh->visc_kind[q] != 0) return true;
    return false;
}
__host__ __device__ static inline bool mat_viscosity_reads_invariant(const jrx_rheology *rh)
{
    for (int q = 0; q < rh->nphase; q++)
        if (rh->visc_kind[q] == 2) return true;
    return false;
}
