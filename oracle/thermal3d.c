/* oracle/thermal3d.c -- TEST INFRASTRUCTURE ONLY (see jrx_oracle.h).
 * CPU restatement of the 3D pseudo-transient heat-diffusion path of JustRelax.jl:
 * src/thermal_diffusion/DiffusionPT_solver.jl:34-149 (array-coefficient form) and :181-305 (rheology form, restricted to what
 * test/test_diffusion3D.jl evaluates: constant conductivity and Cp, PT_Density rho = rho0*(1 - alpha*(T - T0)) [ASSUMED form,
 * pinned in 2D by test_diffusion2D.jl], no phases), kernels DiffusionPT_kernels.jl:6-61 (compute_flux! 3D), :160-199
 * (update_T! 3D), :250-282 (check_res! 3D), update_ΔT! :670-673; thermal_bcs! 3D (BoundaryConditions.jl:46-54 with
 * constant_value.jl:15-33, free_slip.jl:86-103, periodic.jl:42-60).  3D thermal face names: bot <-> k = 1, top <-> k = end.
 * Pinned by the temperatures quoted in test/test_diffusion3D.jl:150-151 (tests/test_oracle_golden.py). */
#include "jrx_oracle.h"
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include <omp.h>

enum { XL = 0, XR = 1, YF = 2, YB = 3, ZT = 4, ZB = 5 };   /* left,right,front,back,top,bot */

#define T3(i, j, k) T[IDX3(nx + 2, ny + 2, i, j, k)]

/* One reference kernel per BC type; inside it the z faces, then the x faces, then the y faces (later writes win on the ghost
 * edges, which no kernel reads). */
void orc_thermal_bcs3d(double *T, const orc_thermal_params3d *p)
{
    const int64_t nx = p->nx, ny = p->ny, nz = p->nz, n1 = nx + 2, n2 = ny + 2, n3 = nz + 2;
    for (int step = 0; step < 3; step++) {
        const int32_t *on = step == 0 ? p->constant_value_on : (step == 1 ? p->no_flux : p->periodic);
        int any = 0;
        for (int f = 0; f < 6; f++) any |= on[f];
        if (!any) continue;
        const double *cv = p->constant_value;
        for (int64_t j = 0; j < n2; j++)
            for (int64_t i = 0; i < n1; i++) {
                if (on[ZB]) T3(i, j, 0) = step == 0 ? 2 * cv[ZB] - T3(i, j, 1) : (step == 1 ? T3(i, j, 1) : T3(i, j, n3 - 2));
                if (on[ZT]) T3(i, j, n3 - 1) = step == 0 ? 2 * cv[ZT] - T3(i, j, n3 - 2) : (step == 1 ? T3(i, j, n3 - 2) : T3(i, j, 1));
            }
        for (int64_t k = 0; k < n3; k++)
            for (int64_t j = 0; j < n2; j++) {
                if (on[XL]) T3(0, j, k) = step == 0 ? 2 * cv[XL] - T3(1, j, k) : (step == 1 ? T3(1, j, k) : T3(n1 - 2, j, k));
                if (on[XR]) T3(n1 - 1, j, k) = step == 0 ? 2 * cv[XR] - T3(n1 - 2, j, k) : (step == 1 ? T3(n1 - 2, j, k) : T3(1, j, k));
            }
        for (int64_t k = 0; k < n3; k++)
            for (int64_t i = 0; i < n1; i++) {
                if (on[YF]) T3(i, 0, k) = step == 0 ? 2 * cv[YF] - T3(i, 1, k) : (step == 1 ? T3(i, 1, k) : T3(i, n2 - 2, k));
                if (on[YB]) T3(i, n2 - 1, k) = step == 0 ? 2 * cv[YB] - T3(i, n2 - 2, k) : (step == 1 ? T3(i, n2 - 2, k) : T3(i, 1, k));
            }
    }
}

static inline double rhoCp3(const orc_thermal_params3d *p, double T)
{   /* DiffusionPT_GeoParams.jl:97-104 : compute_heatcapacity * compute_density */
    return p->Cp * (p->rho0 * (1.0 - p->alpha * (T - p->T0)));
}

#include "thermal_phases.h"
extern const orc_thermal_phases *g_tph;
extern const orc_thermal_phase_fields *g_tpf;

void orc_thermal3d_iteration(const orc_thermal3d *t, const orc_thermal_params3d *p)
{
    const int64_t nx = p->nx, ny = p->ny, nz = p->nz;
    const double _dx = p->_dx, _dy = p->_dy, _dz = p->_dz, _dt = inv(p->dt);
    double *T = t->T;
#define CC(A, i, j, k) (A)[IDX3(nx, ny, i, j, k)]
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < nz + 1; k++)
        for (int64_t j = 0; j < ny + 1; j++)
            for (int64_t i = 0; i < nx + 1; i++) {
                if (j < ny && k < nz) {   /* qTx (nx+1, ny, nz) */
                    const size_t q = IDX3(nx + 1, ny, i, j, k);
                    if (i == 0 && p->constant_flux_on[XL]) t->qTx[q] = p->constant_flux[XL];
                    else if (i == nx && p->constant_flux_on[XR]) t->qTx[q] = p->constant_flux[XR];
                    else {
                        const int64_t a = clampi(i - 1, 0, nx - 1), b = clampi(i, 0, nx - 1);
                        const double Kx = p->rheology_form == 2 ? (tph_cond(g_tph, g_tpf->phase_qx + g_tph->nphase * IDX3(nx + 1, ny, a, j, k)) + tph_cond(g_tph, g_tpf->phase_qx + g_tph->nphase * IDX3(nx + 1, ny, b, j, k))) * 0.5
                                        : p->rheology_form ? (p->k_const + p->k_const) * 0.5 : (CC(t->K, a, j, k) + CC(t->K, b, j, k)) * 0.5;
                        const double th = (CC(t->thetar_dtau, a, j, k) + CC(t->thetar_dtau, b, j, k)) * 0.5;
                        const double qx = -Kx * (T3(i + 1, j + 1, k + 1) - T3(i, j + 1, k + 1)) * _dx;
                        t->qTx2[q] = qx;
                        t->qTx[q] = (t->qTx[q] * th + qx) / (1.0 + th);
                    }
                }
                if (i < nx && k < nz) {   /* qTy (nx, ny+1, nz) */
                    const size_t q = IDX3(nx, ny + 1, i, j, k);
                    if (j == 0 && p->constant_flux_on[YF]) t->qTy[q] = p->constant_flux[YF];
                    else if (j == ny && p->constant_flux_on[YB]) t->qTy[q] = p->constant_flux[YB];
                    else {
                        const int64_t a = clampi(j - 1, 0, ny - 1), b = clampi(j, 0, ny - 1);
                        const double Ky = p->rheology_form == 2 ? (tph_cond(g_tph, g_tpf->phase_qy + g_tph->nphase * IDX3(nx, ny + 1, i, a, k)) + tph_cond(g_tph, g_tpf->phase_qy + g_tph->nphase * IDX3(nx, ny + 1, i, b, k))) * 0.5
                                        : p->rheology_form ? (p->k_const + p->k_const) * 0.5 : (CC(t->K, i, a, k) + CC(t->K, i, b, k)) * 0.5;
                        const double th = (CC(t->thetar_dtau, i, a, k) + CC(t->thetar_dtau, i, b, k)) * 0.5;
                        const double qy = -Ky * (T3(i + 1, j + 1, k + 1) - T3(i + 1, j, k + 1)) * _dy;
                        t->qTy2[q] = qy;
                        t->qTy[q] = (t->qTy[q] * th + qy) / (1.0 + th);
                    }
                }
                if (i < nx && j < ny) {   /* qTz (nx, ny, nz+1) */
                    const size_t q = IDX3(nx, ny, i, j, k);
                    if (k == 0 && p->constant_flux_on[ZB]) t->qTz[q] = p->constant_flux[ZB];
                    else if (k == nz && p->constant_flux_on[ZT]) t->qTz[q] = p->constant_flux[ZT];
                    else {
                        const int64_t a = clampi(k - 1, 0, nz - 1), b = clampi(k, 0, nz - 1);
                        const double Kz = p->rheology_form == 2 ? (tph_cond(g_tph, g_tpf->phase_qz + g_tph->nphase * IDX3(nx, ny, i, j, a)) + tph_cond(g_tph, g_tpf->phase_qz + g_tph->nphase * IDX3(nx, ny, i, j, b))) * 0.5
                                        : p->rheology_form ? (p->k_const + p->k_const) * 0.5 : (CC(t->K, i, j, a) + CC(t->K, i, j, b)) * 0.5;
                        const double th = (CC(t->thetar_dtau, i, j, a) + CC(t->thetar_dtau, i, j, b)) * 0.5;
                        const double qz = -Kz * (T3(i + 1, j + 1, k + 1) - T3(i + 1, j + 1, k)) * _dz;
                        t->qTz2[q] = qz;
                        t->qTz[q] = (t->qTz[q] * th + qz) / (1.0 + th);
                    }
                }
            }
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < nz; k++)
        for (int64_t j = 0; j < ny; j++)
            for (int64_t i = 0; i < nx; i++) {
                const size_t c = IDX3(nx, ny, i, j, k), I1 = IDX3(nx + 2, ny + 2, i + 1, j + 1, k + 1);
                const double Tc = T[I1];
                const double *rc = p->rheology_form == 2 ? g_tpf->phase_c + g_tph->nphase * c : NULL;
                const double rcp = rc ? tph_rhoCp(g_tph, rc, Tc, g_tpf->P[c]) : p->rheology_form ? rhoCp3(p, Tc) : t->rhoCp[c];
                const double dr = t->dtau_rho[c];
                const double divq = (t->qTx[IDX3(nx + 1, ny, i + 1, j, k)] - t->qTx[IDX3(nx + 1, ny, i, j, k)]) * _dx +
                                    (t->qTy[IDX3(nx, ny + 1, i, j + 1, k)] - t->qTy[IDX3(nx, ny + 1, i, j, k)]) * _dy +
                                    (t->qTz[IDX3(nx, ny, i, j, k + 1)] - t->qTz[IDX3(nx, ny, i, j, k)]) * _dz;
                const double adi = (p->rheology_form && t->adiabatic) ? t->adiabatic[c] * Tc : 0.0;        /* + adiabatic[i, j] * T[I1...] of the rheology forms */
                if (t->dirichlet_mask && t->dirichlet_mask[I1] != 0.0) {      /* isdirichlet -> apply_dirichlet!: A = inv(m) A + m B (mask/mask.jl:49-50) */
                    const double m = t->dirichlet_mask[I1], B = t->dirichlet_value ? t->dirichlet_value[I1] : p->dirichlet_const;
                    T[I1] = (1 - m) * Tc + m * B;
                } else if (rc) T[I1] = (dr * (-divq + t->Told[I1] * rcp * _dt + tph_Hr(g_tph, rc) + t->H[c] + t->shear_heating[c] + adi) + Tc) / (1.0 + dr * rcp * _dt);
                else if (p->rheology_form && t->adiabatic) T[I1] = (dr * (-divq + t->Told[I1] * rcp * _dt + t->H[c] + t->shear_heating[c] + adi) + Tc) / (1.0 + dr * rcp * _dt);
                else T[I1] = (dr * (-divq + t->Told[I1] * rcp * _dt + t->H[c] + t->shear_heating[c]) + Tc) / (1.0 + dr * rcp * _dt);
            }
    orc_thermal_bcs3d(T, p);
}

void orc_thermal3d_check_res(const orc_thermal3d *t, const orc_thermal_params3d *p)
{
    const int64_t nx = p->nx, ny = p->ny, nz = p->nz;
    const double _dx = p->_dx, _dy = p->_dy, _dz = p->_dz, _dt = inv(p->dt);
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < nz; k++)
        for (int64_t j = 0; j < ny; j++)
            for (int64_t i = 0; i < nx; i++) {
                const size_t c = IDX3(nx, ny, i, j, k), I1 = IDX3(nx + 2, ny + 2, i + 1, j + 1, k + 1);
                const double *rc = p->rheology_form == 2 ? g_tpf->phase_c + g_tph->nphase * c : NULL;
                const double rcp = rc ? tph_rhoCp(g_tph, rc, t->T[I1], g_tpf->P[c]) : p->rheology_form ? rhoCp3(p, t->T[I1]) : t->rhoCp[c];
                const double dq = (t->qTx2[IDX3(nx + 1, ny, i + 1, j, k)] - t->qTx2[IDX3(nx + 1, ny, i, j, k)]) * _dx +
                                  (t->qTy2[IDX3(nx, ny + 1, i, j + 1, k)] - t->qTy2[IDX3(nx, ny + 1, i, j, k)]) * _dy +
                                  (t->qTz2[IDX3(nx, ny, i, j, k + 1)] - t->qTz2[IDX3(nx, ny, i, j, k)]) * _dz;
                const double adi = (p->rheology_form && t->adiabatic) ? t->adiabatic[c] * t->T[I1] : 0.0;
                if (t->dirichlet_mask && t->dirichlet_mask[I1] != 0.0) t->ResT[c] = 0.0;        /* isNotDirichlet(dirichlet.mask, I1...) ? ... : zero(_T) */
                else if (rc) t->ResT[c] = -rcp * (t->T[I1] - t->Told[I1]) * _dt - dq + tph_Hr(g_tph, rc) + t->H[c] + t->shear_heating[c] + adi;
                else if (p->rheology_form && t->adiabatic) t->ResT[c] = -rcp * (t->T[I1] - t->Told[I1]) * _dt - dq + t->H[c] + t->shear_heating[c] + adi;
                else t->ResT[c] = -rcp * (t->T[I1] - t->Told[I1]) * _dt - dq + t->H[c] + t->shear_heating[c];
            }
}

int32_t orc_heatdiffusion_PT3d(const orc_thermal3d *t, const orc_thermal_params3d *p, int64_t *iter_out, double *norm_ResT, int64_t cap,
                               int64_t *nnorms)
{
    const int64_t nx = p->nx, ny = p->ny, nz = p->nz, n = nx * ny * nz;
    const size_t nT = (size_t)(nx + 2) * (ny + 2) * (nz + 2);
    const double sq = inv(sqrt((double)n));
    memcpy(t->Told, t->T, nT * sizeof(double));
    int64_t iter = 0, cnt = 0;
    double err = 2 * p->eps;
    while (err > p->eps && iter < p->iterMax) {
        if (p->rheology_form == 2)      /* update_pt_thermal_arrays! (DiffusionPT_solver.jl:233-234) */
            for (int64_t k = 0; k < nz; k++)
                for (int64_t j = 0; j < ny; j++)
                    for (int64_t i = 0; i < nx; i++) {
                        const size_t c = IDX3(nx, ny, i, j, k);
                        tph_pt_coeffs(g_tph, g_tpf->phase_c + g_tph->nphase * c, t->T[IDX3(nx + 2, ny + 2, i + 1, j + 1, k + 1)], g_tpf->P[c], inv(p->dt),
                                      &t->thetar_dtau[c], &t->dtau_rho[c]);
                    }
        orc_thermal3d_iteration(t, p);
        iter += 1;
        if (iter % p->nout == 0) {
            orc_thermal3d_check_res(t, p);
            double s = 0.0;
            for (int64_t c = 0; c < n; c++) s += t->ResT[c] * t->ResT[c];
            err = sqrt(s) * sq;
            if (cnt < cap) { norm_ResT[cnt] = err; iter_out[cnt] = iter; }
            cnt++;
        }
    }
    for (size_t c = 0; c < nT; c++) t->dT[c] = t->T[c] - t->Told[c];
    *nnorms = cnt < cap ? cnt : cap;
    return 0;
}
