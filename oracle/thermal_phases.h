/* oracle/thermal_phases.h -- TEST INFRASTRUCTURE ONLY (see jrx_oracle.h).
 * Phase-ratio form of the PT heat-diffusion kernels: heatdiffusion_PT!(thermal, pt_thermal, bc, rheology, args, dt, grid; kwargs = (phase = phase_ratios, ...))
 * -- src/thermal_diffusion/DiffusionPT_solver.jl:181-305 with phase !== nothing: update_pt_thermal_arrays! every iteration
 * (DiffusionPT_coefficients.jl:123-136), conductivity from the face phase ratios (DiffusionPT_kernels.jl:366-440), ρCp and radioactive
 * heat from the centre ratios (:553-601, :631-668; DiffusionPT_GeoParams.jl:97-175).  GeoParams forms assumed: ConstantConductivity k,
 * ConstantHeatCapacity Cp, ConstantRadioactiveHeat H_r, densities as in material.h.  fn_ratio with args returns a pure phase alone
 * (phases.jl:17-30); without args it is the plain weighted sum (:6-15). */
#ifndef ORC_THERMAL_PHASES_H
#define ORC_THERMAL_PHASES_H
#include <math.h>
#include "jrx_oracle.h"

static inline double tph_density(const orc_thermal_phases *ph, int q, double T, double P)
{
    switch (ph->rho_kind[q]) {
    case 1: return ph->rho0[q] * (1.0 - ph->alpha[q] * (T - ph->T0[q]) + ph->beta[q] * (P - ph->P0[q]));
    case 2: return ph->rho0[q] * (1.0 - ph->alpha[q] * (T - ph->T0[q]));
    case 3: return ph->rho0[q] * exp(ph->beta[q] * (P - ph->P0[q]));
    default: return ph->rho0[q];
    }
}
/* compute_ρCp(rheology, phase_ratios, args) = fn_ratio(compute_ρCp, ...): Σ r_q Cp_q ρ_q(T, P) */
static inline double tph_rhoCp(const orc_thermal_phases *ph, const double *r, double T, double P)
{
    double x = 0.0;
    for (int q = 0; q < ph->nphase; q++) {
        const double rq = r[q];
        if (rq == 1.0) return (ph->Cp[q] * tph_density(ph, q, T, P)) * rq;
        x += (rq == 0.0) ? 0.0 : (ph->Cp[q] * tph_density(ph, q, T, P)) * rq;
    }
    return x;
}
/* fn_ratio(compute_conductivity, rheology, phase_ij, args_ij) */
static inline double tph_cond(const orc_thermal_phases *ph, const double *r)
{
    double x = 0.0;
    for (int q = 0; q < ph->nphase; q++) {
        const double rq = r[q];
        if (rq == 1.0) return ph->k[q] * rq;
        x += (rq == 0.0) ? 0.0 : ph->k[q] * rq;
    }
    return x;
}
/* compute_radioactive_heating(rheology, phase::SArray) = fn_ratio(compute_radioactive_heat, rheology, phase) (no args) */
static inline double tph_Hr(const orc_thermal_phases *ph, const double *r)
{
    double x = 0.0;
    for (int q = 0; q < ph->nphase; q++) x += (r[q] == 0.0) ? 0.0 : ph->Hr[q] * r[q];
    return x;
}
/* _compute_pt_thermal_arrays! (DiffusionPT_coefficients.jl:123-136) */
static inline void tph_pt_coeffs(const orc_thermal_phases *ph, const double *r, double T, double P, double _dt, double *thetar_dtau, double *dtau_rho)
{
    const double rcp = tph_rhoCp(ph, r, T, P);
    const double _K = 1.0 / tph_cond(ph, r);
    const double _Re = 1.0 / (3.14159265358979323846 + sqrt(3.14159265358979323846 * 3.14159265358979323846 + rcp * (ph->max_lxyz * ph->max_lxyz) * _K * _dt));
    *thetar_dtau = ph->max_lxyz / ph->Vpdtau * _Re;
    *dtau_rho = ph->Vpdtau * ph->max_lxyz * _K * _Re;
}
#endif
