/* oracle/material.h -- TEST INFRASTRUCTURE ONLY (see jrx_oracle.h).
 * Per-phase material laws of the rheology table (orc_rheology), restated from the reference's call sites; the functions themselves
 * live in GeoParams.jl (>= 0.7.19, not vendored): every form below is ASSUMED from GeoParams' documented definitions unless a
 * reference test pins it (the PT_Density form with beta = 0 is pinned by test/test_diffusion3D.jl:143-151, see thermal3d.c). */
#ifndef ORC_MATERIAL_H
#define ORC_MATERIAL_H
#include <math.h>
#include "jrx_oracle.h"

/* compute_density(rheology[q], (; T, P)) */
static inline double mat_density(const orc_rheology *rh, int q, double T, double P)
{
    switch (rh->rho_kind[q]) {
    case 1: return rh->rho0[q] * (1.0 - rh->alpha[q] * (T - rh->T0[q]) + rh->beta[q] * (P - rh->P0[q]));   /* PT_Density */
    case 2: return rh->rho0[q] * (1.0 - rh->alpha[q] * (T - rh->T0[q]));                                   /* T_Density */
    case 3: return rh->rho0[q] * exp(rh->beta[q] * (P - rh->P0[q]));                                       /* Compressible_Density */
    default: return rh->rho0[q];                                                                           /* ConstantDensity */
    }
}
/* fn_ratio(compute_density, rheology, ratio, args) -- src/phases/phases.jl:17-30: a ratio of exactly one returns that phase alone */
static inline double mat_density_ratio(const orc_rheology *rh, const double *r, double T, double P)
{
    double x = 0.0;
    for (int q = 0; q < rh->nphase; q++) {
        const double rq = r[q];
        if (rq == 1.0) return mat_density(rh, q, T, P) * rq;
        x += (rq == 0.0) ? 0.0 : mat_density(rh, q, T, P) * rq;
    }
    return x;
}
/* isconstant(rheology): update_ρg! recomputes only when some phase's density depends on T or P (BuoyancyForces.jl:153-167) */
static inline int mat_density_is_constant(const orc_rheology *rh)
{
    for (int q = 0; q < rh->nphase; q++)
        if (rh->rho_kind[q] != 0) return 0;
    return 1;
}

/* softening law (value at the accumulated plastic strain EII; v0 = unsoftened value) */
static inline double mat_soften(int kind, double a, double b, double c, double d, double EII, double v0)
{
    if (kind == 1) {                       /* LinearSoftening((min = a, max = b), (lo = c, hi = d)) */
        if (EII >= d) return a;
        if (EII <= c) return b;
        return b + (a - b) / (d - c) * (EII - c);
    }
    if (kind == 2) return a - 0.5 * b * erfc(-(EII - c) / d);     /* NonLinearSoftening(ξ₀ = a, Δ = b, μ = c, σ = d) */
    return v0;
}
/* soften_cohesion / soften_friction_angle (rheology/StressUpdate.jl:305-381) */
static inline double mat_cohesion(const orc_rheology *rh, int q, double EII)
{
    return mat_soften(rh->softC_kind[q], rh->softC_a[q], rh->softC_b[q], rh->softC_c[q], rh->softC_d[q], EII, rh->C[q]);
}
static inline void mat_friction(const orc_rheology *rh, int q, double EII, double *sinphi, double *cosphi)
{
    if (rh->softphi_kind[q] == 0) { *sinphi = rh->sinphi[q]; *cosphi = rh->cosphi[q]; return; }
    const double phi = mat_soften(rh->softphi_kind[q], rh->softphi_a[q], rh->softphi_b[q], rh->softphi_c[q], rh->softphi_d[q], EII, rh->phi_deg[q]);
    const double rad = phi * (3.14159265358979323846 / 180.0);      /* sincosd */
    *sinphi = sin(rad); *cosphi = cos(rad);
}
/* viscosity of the creep element at dt = Inf (compute_viscosity_τII of CompositeRheology((creep, el[, pl])): the elastic strain rate vanishes) */
static inline double mat_creep_viscosity(const orc_rheology *rh, int q, double T, double P)
{
    if (rh->visc_kind[q] == 1) {
        const double e = rh->eta[q] * exp((rh->Ea[q] + P * rh->Va[q]) / (rh->Rgas[q] * T) - rh->Ea[q] / (rh->Rgas[q] * rh->Tref[q]));
        return fmin(fmax(e, rh->visc_lo[q]), rh->visc_hi[q]);
    }
    return rh->eta[q];
}
/* fn_viscosity(rheology[q].CompositeRheology[1], AII, args): compute_viscosity_τII (tau) or compute_viscosity_εII; only the power-law creep reads AII */
static inline double mat_viscosity(const orc_rheology *rh, int q, double AII, double T, double P, int tau)
{
    if (rh->visc_kind[q] != 2) return mat_creep_viscosity(rh, q, T, P);
    const double n = rh->creep_n[q], H = rh->Ea[q] + P * rh->Va[q], RT = rh->Rgas[q] * T;
    if (tau) {
        const double eps = rh->creep_A[q] * pow(AII * rh->creep_FT[q], n) * exp(-H / RT) / rh->creep_FE[q];
        return 0.5 * AII / eps;
    }
    const double t = pow(rh->creep_A[q], -1.0 / n) * pow(AII * rh->creep_FE[q], 1.0 / n) * exp(H / (n * RT)) / rh->creep_FT[q];
    return 0.5 * t / AII;
}
/* compute_phase_viscosity (rheology/Viscosity.jl:599-619) */
static inline double mat_phase_viscosity(const orc_rheology *rh, const double *r, double AII, double T, double P, int tau)
{
    for (int q = 0; q < rh->nphase; q++)
        if (r[q] > 0.999) return mat_viscosity(rh, q, AII, T, P, tau);
    double s = 0.0;
    for (int q = 0; q < rh->nphase; q++)
        if (r[q] != 0.0) s += (1.0 / mat_viscosity(rh, q, AII, T, P, tau)) * r[q];
    return 1.0 / s;
}
/* AII of compute_viscosity_kernel! (Viscosity.jl:394-404): eps() on the normal components of an all-zero tensor */
static inline double mat_visc_invariant2(double xx, double yy, double xy)
{
    const double a0 = (xx == 0.0 && yy == 0.0 && xy == 0.0) ? 2.220446049250313e-16 : 0.0;
    const double x = a0 + xx, y = -a0 + yy;
    return sqrt(0.5 * (x * x + y * y) + xy * xy);
}
static inline int mat_viscosity_reads_fields(const orc_rheology *rh)
{
    for (int q = 0; q < rh->nphase; q++)
        if (rh->visc_kind[q] != 0) return 1;
    return 0;
}
#endif
