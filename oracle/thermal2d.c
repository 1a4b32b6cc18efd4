/* oracle/thermal2d.c -- TEST INFRASTRUCTURE ONLY (see jrx_oracle.h).
 * CPU restatement of the 2D pseudo-transient heat-diffusion path of JustRelax.jl:
 * src/thermal_diffusion/DiffusionPT_solver.jl:34-149 (array-coefficient form) and :181-305
 * (rheology form, restricted to what test/test_diffusion2D.jl evaluates: constant conductivity,
 * constant Cp, GeoParams PT_Density rho = rho0*(1 - alpha*(T - T0)) [ASSUMED form: GeoParams is
 * not vendored in the reference], no phases, no radioactive / adiabatic / shear heating). */
#include "jrx_oracle.h"
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include <omp.h>

enum { TL = 0, TR = 1, TT = 2, TB = 3 }; /* left,right,top,bot ; 2D: bot <-> j=1, top <-> j=end */

#define TT_(i, j) T[IDX2(nx + 2, i, j)]

/* BoundaryConditions.jl:45-53 : constant_value -> no_flux -> periodic.
 * constant_value.jl:1-13, free_slip.jl:72-84, periodic.jl:1-13. */
void orc_thermal_bcs2d(double *T, const orc_thermal_params2d *p)
{
    const int64_t nx = p->nx, ny = p->ny, n1 = nx + 2, n2 = ny + 2;
    int any = 0;
    for (int f = 0; f < 4; f++) any |= p->constant_value_on[f];
    if (any) {
        for (int64_t i = 0; i < n1; i++) {
            if (p->constant_value_on[TB]) TT_(i, 0) = 2 * p->constant_value[TB] - TT_(i, 1);
            if (p->constant_value_on[TT]) TT_(i, n2 - 1) = 2 * p->constant_value[TT] - TT_(i, n2 - 2);
        }
        for (int64_t j = 0; j < n2; j++) {
            if (p->constant_value_on[TL]) TT_(0, j) = 2 * p->constant_value[TL] - TT_(1, j);
            if (p->constant_value_on[TR]) TT_(n1 - 1, j) = 2 * p->constant_value[TR] - TT_(n1 - 2, j);
        }
    }
    if (p->no_flux[0] | p->no_flux[1] | p->no_flux[2] | p->no_flux[3]) {
        for (int64_t i = 0; i < n1; i++) {
            if (p->no_flux[TB]) TT_(i, 0) = TT_(i, 1);
            if (p->no_flux[TT]) TT_(i, n2 - 1) = TT_(i, n2 - 2);
        }
        for (int64_t j = 0; j < n2; j++) {
            if (p->no_flux[TL]) TT_(0, j) = TT_(1, j);
            if (p->no_flux[TR]) TT_(n1 - 1, j) = TT_(n1 - 2, j);
        }
    }
    if (p->periodic[0] | p->periodic[1] | p->periodic[2] | p->periodic[3]) {
        for (int64_t i = 0; i < n1; i++) {
            if (p->periodic[TB]) TT_(i, 0) = TT_(i, n2 - 2);
            if (p->periodic[TT]) TT_(i, n2 - 1) = TT_(i, 1);
        }
        for (int64_t j = 0; j < n2; j++) {
            if (p->periodic[TL]) TT_(0, j) = TT_(n1 - 2, j);
            if (p->periodic[TR]) TT_(n1 - 1, j) = TT_(1, j);
        }
    }
}

#include "thermal_phases.h"
const orc_thermal_phases *g_tph = NULL;              /* phase-ratio form (rheology_form = 2): set by orc_thermal_set_phases */
const orc_thermal_phase_fields *g_tpf = NULL;
void orc_thermal_set_phases(const orc_thermal_phases *ph, const orc_thermal_phase_fields *pf) { g_tph = ph; g_tpf = pf; }

void orc_adiabatic_heating(double *A, const double *P, const double *P0, int64_t n, const orc_thermal_phases *ph, const double *phase_c, double _dt)
{
    for (int64_t c = 0; c < n; c++) {
        double al = 0.0;
        for (int q = 0; q < ph->nphase; q++) {
            const double r = phase_c ? phase_c[(size_t)ph->nphase * c + q] : (q == 0 ? 1.0 : 0.0);
            const double aq = (ph->rho_kind[q] == 1 || ph->rho_kind[q] == 2) ? ph->alpha[q] : 0.0;
            al += (r == 0.0) ? 0.0 : aq * r;
        }
        A[c] = (P[c] - P0[c]) * al * _dt;
    }
}

static inline double rhoCp_rheology(const orc_thermal_params2d *p, double T)
{   /* DiffusionPT_GeoParams.jl:97-104 : compute_heatcapacity * compute_density */
    return p->Cp * (p->rho0 * (1.0 - p->alpha * (T - p->T0)));
}

/* compute_flux!  DiffusionPT_kernels.jl:327-364 (array K) / :366-440 (rheology, constant k)
 * then update_T! :519-551 / :553-601, thermal_bcs! ; launch boxes (nx+1,ny+1) and ni */
void orc_thermal2d_iteration(const orc_thermal2d *t, const orc_thermal_params2d *p)
{
    const int64_t nx = p->nx, ny = p->ny;
    const double _dx = p->_dx, _dy = p->_dy, _dt = inv(p->dt);
    double *T = t->T;
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < ny + 1; j++)
        for (int64_t i = 0; i < nx + 1; i++) {
            if (j < ny) { /* qTx (nx+1, ny) */
                size_t q = IDX2(nx + 1, i, j);
                if (i == 0 && p->constant_flux_on[TL]) t->qTx[q] = p->constant_flux[TL];
                else if (i == nx && p->constant_flux_on[TR]) t->qTx[q] = p->constant_flux[TR];
                else {
                    int64_t iL = clampi(i - 1, 0, nx - 1), iR = clampi(i, 0, nx - 1);
                    double Kx = p->rheology_form == 2 ? (tph_cond(g_tph, g_tpf->phase_qx + g_tph->nphase * IDX2(nx + 1, iL, j)) +
                                                         tph_cond(g_tph, g_tpf->phase_qx + g_tph->nphase * IDX2(nx + 1, iR, j))) * 0.5
                                : p->rheology_form ? (p->k_const + p->k_const) * 0.5
                                                   : (t->K[IDX2(nx, iL, j)] + t->K[IDX2(nx, iR, j)]) * 0.5;
                    double th = (t->thetar_dtau[IDX2(nx, iL, j)] + t->thetar_dtau[IDX2(nx, iR, j)]) * 0.5;
                    double qx = -Kx * (TT_(i + 1, j + 1) - TT_(i, j + 1)) * (p->inv_spacing[0] ? p->inv_spacing[0][clampi(i, 0, nx - 2)] : _dx);   /* @dx(_di_center, clamp(i, 1, nx - 1)) */
                    t->qTx2[q] = qx;
                    t->qTx[q] = (t->qTx[q] * th + qx) / (1.0 + th);
                }
            }
            if (i < nx) { /* qTy (nx, ny+1) */
                size_t q = IDX2(nx, i, j);
                if (j == 0 && p->constant_flux_on[TB]) t->qTy[q] = p->constant_flux[TB];
                else if (j == ny && p->constant_flux_on[TT]) t->qTy[q] = p->constant_flux[TT];
                else {
                    int64_t jB = clampi(j - 1, 0, ny - 1), jT = clampi(j, 0, ny - 1);
                    double Ky = p->rheology_form == 2 ? (tph_cond(g_tph, g_tpf->phase_qy + g_tph->nphase * IDX2(nx, i, jB)) +
                                                         tph_cond(g_tph, g_tpf->phase_qy + g_tph->nphase * IDX2(nx, i, jT))) * 0.5
                                : p->rheology_form ? (p->k_const + p->k_const) * 0.5
                                                   : (t->K[IDX2(nx, i, jB)] + t->K[IDX2(nx, i, jT)]) * 0.5;
                    double th = (t->thetar_dtau[IDX2(nx, i, jB)] + t->thetar_dtau[IDX2(nx, i, jT)]) * 0.5;
                    double qy = -Ky * (TT_(i + 1, j + 1) - TT_(i + 1, j)) * (p->inv_spacing[1] ? p->inv_spacing[1][clampi(j, 0, ny - 2)] : _dy);
                    t->qTy2[q] = qy;
                    t->qTy[q] = (t->qTy[q] * th + qy) / (1.0 + th);
                }
            }
        }
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < ny; j++)
        for (int64_t i = 0; i < nx; i++) {
            size_t c = IDX2(nx, i, j), I1 = IDX2(nx + 2, i + 1, j + 1);
            double Tij = T[I1];
            const double *rc = p->rheology_form == 2 ? g_tpf->phase_c + g_tph->nphase * c : NULL;
            double rcp = rc ? tph_rhoCp(g_tph, rc, Tij, g_tpf->P[c]) : p->rheology_form ? rhoCp_rheology(p, Tij) : t->rhoCp[c];
            double dr = t->dtau_rho[c];
            double divq = (t->qTx[IDX2(nx + 1, i + 1, j)] - t->qTx[IDX2(nx + 1, i, j)]) * (p->inv_spacing[2] ? p->inv_spacing[2][i] : _dx) +     /* _di.vertex (:579-580) */
                          (t->qTy[IDX2(nx, i, j + 1)] - t->qTy[IDX2(nx, i, j)]) * (p->inv_spacing[3] ? p->inv_spacing[3][j] : _dy);
            const double adi = (p->rheology_form && t->adiabatic) ? t->adiabatic[c] * Tij : 0.0;        /* + adiabatic[i, j] * T[I1...] of the rheology forms */
            if (t->dirichlet_mask && t->dirichlet_mask[I1] != 0.0) {      /* isdirichlet -> apply_dirichlet!: A = inv(m) A + m B (mask/mask.jl:49-50) */
                const double m = t->dirichlet_mask[I1], B = t->dirichlet_value ? t->dirichlet_value[I1] : p->dirichlet_const;
                T[I1] = (1 - m) * Tij + m * B;
            } else if (rc) T[I1] = (dr * (-divq + t->Told[I1] * rcp * _dt + tph_Hr(g_tph, rc) + t->H[c] + t->shear_heating[c] + adi) + Tij) / (1.0 + dr * rcp * _dt);
            else if (p->rheology_form && t->adiabatic) T[I1] = (dr * (-divq + t->Told[I1] * rcp * _dt + t->H[c] + t->shear_heating[c] + adi) + Tij) / (1.0 + dr * rcp * _dt);
            else T[I1] = (dr * (-divq + t->Told[I1] * rcp * _dt + t->H[c] + t->shear_heating[c]) + Tij) / (1.0 + dr * rcp * _dt);
        }
    orc_thermal_bcs2d(T, p);
}

/* check_res!  DiffusionPT_kernels.jl:603-629 / :631-668 */
void orc_thermal2d_check_res(const orc_thermal2d *t, const orc_thermal_params2d *p)
{
    const int64_t nx = p->nx, ny = p->ny;
    const double _dx = p->_dx, _dy = p->_dy, _dt = inv(p->dt);
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < ny; j++)
        for (int64_t i = 0; i < nx; i++) {
            size_t c = IDX2(nx, i, j), I1 = IDX2(nx + 2, i + 1, j + 1);
            const double *rc = p->rheology_form == 2 ? g_tpf->phase_c + g_tph->nphase * c : NULL;
            double rcp = rc ? tph_rhoCp(g_tph, rc, t->T[I1], g_tpf->P[c]) : p->rheology_form ? rhoCp_rheology(p, t->T[I1]) : t->rhoCp[c];
            const double dq = (t->qTx2[IDX2(nx + 1, i + 1, j)] - t->qTx2[IDX2(nx + 1, i, j)]) * (p->inv_spacing[2] ? p->inv_spacing[2][i] : _dx) +
                              (t->qTy2[IDX2(nx, i, j + 1)] - t->qTy2[IDX2(nx, i, j)]) * (p->inv_spacing[3] ? p->inv_spacing[3][j] : _dy);
            const double adi = (p->rheology_form && t->adiabatic) ? t->adiabatic[c] * t->T[I1] : 0.0;
            if (t->dirichlet_mask && t->dirichlet_mask[I1] != 0.0) t->ResT[c] = 0.0;        /* isNotDirichlet(dirichlet.mask, I1...) ? ... : zero(_T) */
            else if (rc) t->ResT[c] = -rcp * (t->T[I1] - t->Told[I1]) * _dt - dq + tph_Hr(g_tph, rc) + t->H[c] + t->shear_heating[c] + adi;
            else if (p->rheology_form && t->adiabatic) t->ResT[c] = -rcp * (t->T[I1] - t->Told[I1]) * _dt - dq + t->H[c] + t->shear_heating[c] + adi;
            else t->ResT[c] = -rcp * (t->T[I1] - t->Told[I1]) * _dt - dq + t->H[c] + t->shear_heating[c];
        }
}

/* DiffusionPT_solver.jl:34-149 / :181-305 */
int32_t orc_heatdiffusion_PT2d(const orc_thermal2d *t, const orc_thermal_params2d *p,
                               int64_t *iter_out, double *norm_ResT, int64_t cap, int64_t *nnorms)
{
    const int64_t nx = p->nx, ny = p->ny;
    const size_t nT = (size_t)(nx + 2) * (ny + 2);
    const double sq = inv(sqrt((double)(nx * ny)));
    memcpy(t->Told, t->T, nT * sizeof(double));   /* @copy thermal.Told thermal.T */
    int64_t iter = 0, cnt = 0;
    double err = 2 * p->eps;
    while (err > p->eps && iter < p->iterMax) {
        if (p->rheology_form == 2)      /* update_pt_thermal_arrays!(pt_thermal, phase, rheology, args, _dt) :233-234 ; T at Idx .+ 1 */
            for (int64_t j = 0; j < ny; j++)
                for (int64_t i = 0; i < nx; i++) {
                    const size_t c = IDX2(nx, i, j);
                    tph_pt_coeffs(g_tph, g_tpf->phase_c + g_tph->nphase * c, t->T[IDX2(nx + 2, i + 1, j + 1)], g_tpf->P[c], inv(p->dt),
                                  &t->thetar_dtau[c], &t->dtau_rho[c]);
                }
        orc_thermal2d_iteration(t, p);
        iter += 1;
        if (iter % p->nout == 0) {
            orc_thermal2d_check_res(t, p);
            double s = 0.0;
            for (int64_t c = 0; c < nx * ny; c++) s += t->ResT[c] * t->ResT[c];
            err = sqrt(s) * sq;
            if (cnt < cap) { norm_ResT[cnt] = err; iter_out[cnt] = iter; }
            cnt++;
        }
    }
    for (size_t c = 0; c < nT; c++) t->dT[c] = t->T[c] - t->Told[c];   /* update_ΔT! :670-673 */
    *nnorms = cnt < cap ? cnt : cap;
    return 0;
}
