/* oracle/stokes2d_vep.c -- TEST INFRASTRUCTURE ONLY (see jrx_oracle.h).
 * CPU restatement of the 2D multiphase visco-elasto-plastic PT Stokes driver of JustRelax.jl
 * (src/stokes/Stokes2D.jl:577-866) as test/test_shearband2D.jl exercises it.
 *
 * Third-party arithmetic (GeoParams.jl >= 0.7.19, not vendored) restated from the reference's call sites and docs
 * (docs/src/man/constitutive_equations.md:23,87,93) -- ASSUMED forms:
 *   second_invariant(xx,yy,xy)            = sqrt(0.5*(xx^2+yy^2) + xy^2)
 *   DruckerPrager_regularised: F          = tauII - cos(phi)*C - sin(phi)*P          (lambda-term handled by the caller)
 *                              dQ/dtau_ij = 0.5*tau_ij/tauII (shear slot already halved by StressUpdate.jl:448-452)
 *                              dQ/dP = -sin(psi),  dF/dP = -sin(phi)
 *   compute_viscosity_tauII(CompositeRheology(visc, el, pl)) with dt = Inf (Viscosity.jl:599-603): the linear viscosity
 *   second_invariant_staggered(xx,yy,(xy1..4)): selectable, see orc_vep_params2d
 * Pinned only by the regression scalars of test/test_shearband2D.jl:197-201 (tests/test_oracle_golden.py).
 *
 * The reference's stress kernel (StressKernels.jl:992-1144) reads neighbouring centre stresses for the vertex update
 * while other threads overwrite them in the same launch (a data race); here the vertex pass runs first on the old
 * centre stresses, then the centre pass (what a GPU launch does when all loads precede the stores). */
#include "jrx_oracle.h"
#include "common.h"
#include "material.h"
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#define C2(A, i, j) (A)[IDX2(nx, i, j)]
#define V2(A, i, j) (A)[IDX2(nx + 1, i, j)]

static inline double sinv2(double xx, double yy, double xy) { return sqrt(0.5 * (xx * xx + yy * yy) + xy * xy); }

static inline double sinv_stag(double xx, double yy, double a, double b, double c, double d, int mode)
{
    if (mode) return sqrt(0.5 * (xx * xx + yy * yy) + 0.25 * (a * a + b * b + c * c + d * d));
    double m = 0.25 * (a + b + c + d);
    return sqrt(0.5 * (xx * xx + yy * yy) + m * m);
}

/* fn_ratio (src/phases/phases.jl:6-15) */
static inline double ratio_avg(const double *val, const double *r, int n)
{
    double x = 0.0;
    for (int q = 0; q < n; q++) x += (r[q] == 0.0) ? 0.0 : val[q] * r[q];
    return x;
}

/* plastic_params_phase (StressUpdate.jl:152-176): ratio-weighted parameters; is_pl if any phase is plastic */
static inline void plastic_params(const orc_rheology *rh, const double *r, int *is_pl, double *eta_reg)
{
    *is_pl = 0; *eta_reg = 0.0;
    for (int q = 0; q < rh->nphase; q++) {
        if (rh->is_pl[q]) { *is_pl = 1; *eta_reg += rh->eta_vp[q] * r[q]; }
    }
}

/* compute_yieldfunction_phase (StressUpdate.jl:399-410): sum_i r_i F_i, non-plastic phases contribute tauII */
static inline double yield_F(const orc_rheology *rh, const double *r, double P, double tII, double EII)
{
    double F = 0.0;
    for (int q = 0; q < rh->nphase; q++) {
        if (r[q] == 0.0) continue;
        double Fq = tII;
        if (rh->is_pl[q]) {
            double sp, cp;
            mat_friction(rh, q, EII, &sp, &cp);                       /* softening_ϕ / softening_C at the EII keyword */
            Fq = tII - cp * mat_cohesion(rh, q, EII) - sp * P;
        }
        F += r[q] * Fq;
    }
    return F;
}

/* compute_plastic_gradients_phase (StressUpdate.jl:476-495) */
static inline void plastic_grad(const orc_rheology *rh, const double *r, const double t[3], double dQdt[3], double *dQdP, double *dFdP)
{
    dQdt[0] = dQdt[1] = dQdt[2] = 0.0; *dQdP = 0.0; *dFdP = 0.0;
    const double tII = sinv2(t[0], t[1], t[2]);
    for (int q = 0; q < rh->nphase; q++) {
        if (r[q] == 0.0 || !rh->is_pl[q]) continue;
        const double g0 = 0.5 * t[0] / tII, g1 = 0.5 * t[1] / tII, g2 = 0.5 * (t[2] / tII);
        dQdt[0] = fma(r[q], g0, dQdt[0]); dQdt[1] = fma(r[q], g1, dQdt[1]); dQdt[2] = fma(r[q], g2, dQdt[2]);
        *dQdP = fma(r[q], -rh->sinpsi[q], *dQdP);
        *dFdP = fma(r[q], -rh->sinphi[q], *dFdP);
    }
}

/* update_stresses_center_vertex_ps! 2D (StressKernels.jl:992-1144) */
void orc_vep2d_stress(const orc_vep2d *f, const double *theta, double *lam, double *lamv, const orc_rheology *rh,
                      const orc_vep_params2d *p)
{
    const int64_t nx = p->nx, ny = p->ny;
    const int np = rh->nphase;
    const double dt = p->dt, th = p->theta_dtau, rel = p->lambda_relaxation;
    /* strain_increment variant (StressKernels.jl:1147-1302): the increments Δε replace ε in the stress increment, which is written in its dt-multiplied
     * form (compute_stress_increment(τ, τ_o, η, Δε, _G, dτ_r, dt) :18-21, dτ_r = inv(θ_dτ dt + η _G + dt)); plastic terms carry the factor dt */
    const int si = p->strain_increment != 0;
    /* vertex pass */
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < ny + 1; j++)
        for (int64_t i = 0; i < nx + 1; i++) {
            const int64_t i0 = clampi(i - 1, 0, nx - 1), ic = clampi(i, 0, nx - 1), j0 = clampi(j - 1, 0, ny - 1), jc = clampi(j, 0, ny - 1);
#define AVC(A) (0.25 * (C2(A, i0, j0) + C2(A, ic, jc) + C2(A, i0, jc) + C2(A, ic, j0)))   /* av_clamped :1311-1313 */
            const double Pv = AVC(theta), exxv = AVC(f->exx), eyyv = AVC(f->eyy), txxv = AVC(f->txx), tyyv = AVC(f->tyy);
            const double toxxv = AVC(f->toxx), toyyv = AVC(f->toyy);
            const double *rv = f->phase_v + (size_t)np * IDX2(nx + 1, i, j);
            int is_pl; double eta_reg;
            plastic_params(rh, rv, &is_pl, &eta_reg);
            const double _Gdt = si ? inv(ratio_avg(rh->G, rv, np)) : inv(ratio_avg(rh->G, rv, np) * dt);      /* si: _Gv */
            const double Kv = ratio_avg(rh->Kb, rv, np);
            const double etav = 4.0 / (1.0 / C2(f->eta, i0, j0) + 1.0 / C2(f->eta, ic, jc) + 1.0 / C2(f->eta, i0, jc) + 1.0 / C2(f->eta, ic, j0));
            const double dtr = si ? inv(th * dt + etav * _Gdt + dt) : inv(th + etav * _Gdt + 1.0);
            const size_t v = IDX2(nx + 1, i, j);
            const double txy = f->txy[v];
            double dxx, dyy, dxy;
            if (si) {
                const double dexxv = AVC(f->dexx), deyyv = AVC(f->deyy);
                dxx = stress_increment_dt(txxv, toxxv, etav, dexxv, _Gdt, dtr, dt);
                dyy = stress_increment_dt(tyyv, toyyv, etav, deyyv, _Gdt, dtr, dt);
                dxy = stress_increment_dt(txy, f->toxy[v], etav, f->dexy[v], _Gdt, dtr, dt);
                (void)exxv; (void)eyyv;
            } else {
                dxx = stress_increment(txxv, toxxv, etav, exxv, _Gdt, dtr);
                dyy = stress_increment(tyyv, toyyv, etav, eyyv, _Gdt, dtr);
                dxy = stress_increment(txy, f->toxy[v], etav, f->exy[v], _Gdt, dtr);
            }
            const double tt[3] = {txxv + dxx, tyyv + dyy, txy + dxy};
            const double tIIv = sinv2(dxx + txxv, dyy + tyyv, dxy + txy);
            double dQdt[3], dQdP, dFdP;
            plastic_grad(rh, rv, tt, dQdt, &dQdP, &dFdP);
            const double vol = isinf(Kv) ? 0.0 : Kv * dt * dFdP * dQdP;
            const double EIIv = AVC(f->EII_pl);                        /* EIIv_ij = av_clamped(EII, Ic...) :1030 */
            const double F = yield_F(rh, rv, Pv, tIIv, EIIv);
            if (is_pl && tIIv != 0.0 && F > 0) {
                lamv[v] = fma(1.0 - rel, lamv[v], rel * (fmax(F, 0.0) / (si ? etav * dtr * dt + eta_reg + vol : etav * dtr + eta_reg + vol)));
                const double epl = lamv[v] * dQdt[2];
                f->txy[v] = txy + (si ? fma(-2.0 * etav * dt * epl, dtr, dxy) : fma(-2.0 * etav * epl, dtr, dxy));
                f->eplxy[v] = epl;
            } else {
                f->txy[v] = txy + dxy;
                f->eplxy[v] = 0.0;
            }
        }
    /* centre pass */
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < ny; j++)
        for (int64_t i = 0; i < nx; i++) {
            const size_t c = IDX2(nx, i, j);
            const double *rc = f->phase_c + (size_t)np * c;
            const double _Gdt = si ? inv(ratio_avg(rh->G, rc, np)) : inv(ratio_avg(rh->G, rc, np) * dt);
            int is_pl; double eta_reg;
            plastic_params(rh, rc, &is_pl, &eta_reg);
            const double K = ratio_avg(rh->Kb, rc, np);
            const double e = f->eta[c];
            const double dtr = si ? 1.0 / (th * dt + e * _Gdt + dt) : 1.0 / (th + e * _Gdt + 1.0);
            const double exyc = (V2(f->exy, i, j) + V2(f->exy, i + 1, j) + V2(f->exy, i, j + 1) + V2(f->exy, i + 1, j + 1)) / 4;   /* cache_tensors :208-222 */
            const double eij[3] = {f->exx[c], f->eyy[c], exyc};
            double tij[3] = {f->txx[c], f->tyy[c], f->txy_c[c]};
            const double toij[3] = {f->toxx[c], f->toyy[c], f->toxy_c[c]};
            double d[3];
            if (si) {      /* Δεij = (Δε.xx, Δε.yy, av_shear(Δε.xy)) -- cache_tensors :226-246 */
                const double dexyc = (V2(f->dexy, i, j) + V2(f->dexy, i + 1, j) + V2(f->dexy, i, j + 1) + V2(f->dexy, i + 1, j + 1)) / 4;
                const double deij[3] = {f->dexx[c], f->deyy[c], dexyc};
                for (int q = 0; q < 3; q++) d[q] = stress_increment_dt(tij[q], toij[q], e, deij[q], _Gdt, dtr, dt);
            } else
                for (int q = 0; q < 3; q++) d[q] = stress_increment(tij[q], toij[q], e, eij[q], _Gdt, dtr);
            double tII = sinv2(d[0] + tij[0], d[1] + tij[1], d[2] + tij[2]);
            const double tt[3] = {tij[0] + d[0], tij[1] + d[1], tij[2] + d[2]};
            double dQdt[3], dQdP, dFdP;
            plastic_grad(rh, rc, tt, dQdt, &dQdP, &dFdP);
            const double vol = isinf(K) ? 0.0 : K * dt * dFdP * dQdP;
            const double Pr = theta[c];
            const double F = yield_F(rh, rc, Pr, tII, f->EII_pl[c]);
            if (is_pl && tII != 0.0 && F > 0) {
                lam[c] = fma(1.0 - rel, lam[c], rel * (fmax(F, 0.0) / (si ? e * dtr * dt + eta_reg + vol : e * dtr + eta_reg + vol)));
                double epl[3];
                for (int q = 0; q < 3; q++) {
                    epl[q] = lam[c] * dQdt[q];
                    d[q] = si ? fma(-2.0 * e * dt * epl[q], dtr, d[q]) : fma(-2.0 * e * epl[q], dtr, d[q]);
                    tij[q] = d[q] + tij[q];
                }
                f->evol_pl[c] = -lam[c] * dQdP;
                f->txx[c] = tij[0]; f->tyy[c] = tij[1]; f->txy_c[c] = tij[2];
                f->eplxx[c] = epl[0]; f->eplyy[c] = epl[1];
                tII = sinv2(tij[0], tij[1], tij[2]);
            } else {
                f->evol_pl[c] = 0.0;
                f->txx[c] = d[0] + tij[0]; f->tyy[c] = d[1] + tij[1]; f->txy_c[c] = d[2] + tij[2];
                f->eplxx[c] = 0.0; f->eplyy[c] = 0.0;
            }
            f->tII[c] = tII;
            f->eta_vep[c] = tII * 0.5 * inv(sinv2(eij[0], eij[1], eij[2]));
            f->P[c] = Pr - (isinf(K) ? 0.0 : K * dt * lam[c] * dQdP);
        }
}

/* compute_τ_nonlinear! 2D: single phase (StressKernels.jl:266-307) and phases at cell centres (:310-351), with
 * _compute_τ_nonlinear! (rheology/StressUpdate.jl:2-57), compute_stress_increment_and_trial (:72-81), compute_dτ_pl
 * (:83-105), isyielding (:68), cache_tensors (:190-222), plastic_params_phase (:146-176; NoSoftening only).
 * Centre-only update.  As the reference's only caller passes them (Stokes2D.jl:442-458): τ = (xx, yy, xy_c),
 * τ_old = @tensor(τ_o) = (xx, yy, xy) whose shear member is the VERTEX array read at the centre's index [i,j],
 * ε_pl = (xx, yy, xy) whose shear member is likewise the vertex array written at [i,j].
 * PARITY UNPINNED in its plastic branch: the only test that reaches this kernel (test/test_WENO5.jl:226-291, through the single-phase
 * driver below) uses a non-plastic rheology and asserts convergence only. */
void orc_compute_tau_nonlinear2d(const orc_vep2d *f, double *theta, double *lam, const orc_rheology *rh, const orc_vep_params2d *p,
                                 int32_t multiphase)
{
    const int64_t nx = p->nx, ny = p->ny;
    const int np = rh->nphase;
    const double dt = p->dt, th = p->theta_dtau, one = 1.0;
    for (int64_t j = 0; j < ny; j++)
        for (int64_t i = 0; i < nx; i++) {
            const size_t c = IDX2(nx, i, j);
            const double *r = multiphase ? f->phase_c + (size_t)np * c : &one;
            const int n = multiphase ? np : 1;
            const double eta = f->eta[c];
            const double _Gdt = inv((multiphase ? ratio_avg(rh->G, r, n) : rh->G[0]) * dt);
            const double dtr = inv(th + fma(eta, _Gdt, 1.0));                       /* compute_dτ_r, StressUpdate.jl:70 */
            int is_pl = 0;
            double C = 0.0, sinphi = 0.0, cosphi = 0.0, sinpsi = 0.0, eta_reg = 0.0;
            for (int q = 0; q < n; q++) {
                if (r[q] == 0.0 || !rh->is_pl[q]) continue;                         /* empty_args for absent / non-plastic phases */
                is_pl = 1;
                double sp, cp;
                mat_friction(rh, q, f->EII_pl[c], &sp, &cp);                          /* soften_friction_angle / soften_cohesion at EII[I...] */
                C += mat_cohesion(rh, q, f->EII_pl[c]) * r[q]; sinphi += sp * r[q]; cosphi += cp * r[q];
                sinpsi += rh->sinpsi[q] * r[q]; eta_reg += rh->eta_vp[q] * r[q];
            }
            const double K = multiphase ? ratio_avg(rh->Kb, r, n) : rh->Kb[0];
            const double volume = isinf(K) ? 0.0 : K * dt * sinphi * sinpsi;
            const double eij[3] = {f->exx[c], f->eyy[c],
                                   (V2(f->exy, i, j) + V2(f->exy, i + 1, j) + V2(f->exy, i, j + 1) + V2(f->exy, i + 1, j + 1)) / 4};
            const double tij[3] = {f->txx[c], f->tyy[c], f->txy_c[c]};
            const double toij[3] = {f->toxx[c], f->toyy[c], V2(f->toxy, i, j)};
            double d[3];
            for (int q = 0; q < 3; q++) d[q] = dtr * fma(2.0 * eta, eij[q], fma(-((tij[q] - toij[q])) * eta, _Gdt, -tij[q]));
            const double tII_trial = sinv2(tij[0] + d[0], tij[1] + d[1], tij[2] + d[2]);
            const double ty = fmax(C * cosphi + f->P[c] * sinphi, 0.0);
            double ldq[3] = {0.0, 0.0, 0.0};
            if (is_pl && tII_trial > ty) {
                const double F = tII_trial - ty;
                const double l = 0.5 * lam[c] + (1 - 0.5) * (F > 0.0 ? 1.0 : 0.0) * F * inv(eta * dtr + eta_reg + volume);
                const double l_tII = l * 0.5 * inv(tII_trial);
                double dpl[3];
                for (int q = 0; q < 3; q++) {
                    ldq[q] = (tij[q] + d[q]) * l_tII;
                    dpl[q] = fma(-dtr * 2.0, eta * ldq[q], d[q]);
                }
                for (int q = 0; q < 3; q++) d[q] = dpl[q];
                lam[c] = l;
            }
            f->eplxx[c] = isinf(ldq[0]) ? 0.0 : ldq[0];
            f->eplyy[c] = isinf(ldq[1]) ? 0.0 : ldq[1];
            V2(f->eplxy, i, j) = isinf(ldq[2]) ? 0.0 : ldq[2];
            f->txx[c] = tij[0] + d[0]; f->tyy[c] = tij[1] + d[1]; f->txy_c[c] = tij[2] + d[2];
            const double tII = sinv2(tij[0] + d[0], tij[1] + d[1], tij[2] + d[2]);
            f->tII[c] = tII;
            f->eta_vep[c] = tII * 0.5 * inv(sinv2(eij[0], eij[1], eij[2]));
            theta[c] = f->P[c] + (isinf(K) ? 0.0 : K * dt * lam[c] * sinpsi);
        }
}

/* center2vertex!(vertex, center) 2D (Interpolations.jl:101-114): inner vertices = mean of the 4 cells, then the edge
 * rows/columns copy their inner neighbour (rows first, then columns, so corners take the column copy) */
void orc_center2vertex2d(double *v, const double *c, int64_t nx, int64_t ny)
{
    for (int64_t j = 1; j < ny; j++)
        for (int64_t i = 1; i < nx; i++)
            V2(v, i, j) = 0.25 * (C2(c, i - 1, j - 1) + C2(c, i, j - 1) + C2(c, i - 1, j) + C2(c, i, j));
    for (int64_t j = 0; j <= ny; j++) { V2(v, 0, j) = V2(v, 1, j); V2(v, nx, j) = V2(v, nx - 1, j); }
    for (int64_t i = 0; i <= nx; i++) { V2(v, i, 0) = V2(v, i, 1); V2(v, i, ny) = V2(v, i, ny - 1); }
}

/* scalar entry points for the known-answer tests of test/test_Utils.jl:399-470 */
double orc_yieldfunction_phase(const orc_rheology *rh, const double *ratio, double P, double tII) { return yield_F(rh, ratio, P, tII, 0.0); }
void orc_plastic_gradients_phase2d(const orc_rheology *rh, const double *ratio, const double t[3], double dQdt[3], double *dQdP, double *dFdP)
{
    plastic_grad(rh, ratio, t, dQdt, dQdP, dFdP);
}
int32_t orc_isyielding(int32_t is_pl, double tII_trial, double ty) { return is_pl * (tII_trial > ty); }   /* StressUpdate.jl:68 */
/* compute_dτ_pl (StressUpdate.jl:83-105), N = 3; returns λ */
double orc_compute_dtau_pl(const double tij[3], const double dtij[3], double ty, double tII_trial, double eta, double lam0, double eta_reg,
                           double dtr, double volume, double dtau_pl[3], double ldq[3])
{
    const double F = tII_trial - ty;
    const double l = 0.5 * lam0 + (1 - 0.5) * (F > 0.0 ? 1.0 : 0.0) * F * inv(eta * dtr + eta_reg + volume);
    const double l_tII = l * 0.5 * inv(tII_trial);
    for (int q = 0; q < 3; q++) {
        ldq[q] = (tij[q] + dtij[q]) * l_tII;
        dtau_pl[q] = fma(-dtr * 2.0, eta * ldq[q], dtij[q]);
    }
    return l;
}

/* update_viscosity_τII! (rheology/Viscosity.jl:67-106,382-418): centre and vertex viscosities relaxed towards the
 * per-phase value; with LinearViscous + dt = Inf the composite viscosity is the linear one (ASSUMED, see header) */
static inline double phase_viscosity(const orc_rheology *rh, const double *r)
{
    for (int q = 0; q < rh->nphase; q++)
        if (r[q] > 0.999) return rh->eta[q];
    double s = 0.0;
    for (int q = 0; q < rh->nphase; q++)
        if (r[q] != 0.0) s += inv(rh->eta[q]) * r[q];
    return inv(s);
}
/* creep laws that read fields: compute_viscosity_kernel! with local_viscosity_args / local_viscosity_args_vertex (Viscosity.jl:382-418, 513-552);
 * centre: invariant of @stress_center / @strain_center, T at I .+ 1 of the ghosted thermal.T; vertex: (xx_v, yy_v, xy) with xx_v = yy_v = 0 (never written by
 * the PT solvers), args averaged over the clamped surrounding centres, T over its 2 x 2 nodes */
static void viscosity2d_fields(const orc_vep2d *f, const orc_rheology *rh, const orc_vep_params2d *p, double nu, int tau)
{
    const int64_t nx = p->nx, ny = p->ny;
    const int np = rh->nphase;
    for (int64_t j = 0; j < ny; j++)
        for (int64_t i = 0; i < nx; i++) {
            const size_t c = IDX2(nx, i, j);
            const double AII = tau ? mat_visc_invariant2(f->txx[c], f->tyy[c], f->txy_c[c]) : mat_visc_invariant2(f->exx[c], f->eyy[c], f->exy_c[c]);
            const double T = !f->T ? 0.0 : (p->T_ghosted ? f->T[IDX2(nx + 2, i + 1, j + 1)] : f->T[c]);
            double e = mat_phase_viscosity(rh, f->phase_c + (size_t)np * c, AII, T, f->P[c], tau);
            e = e * nu + f->eta[c] * (1.0 - nu);
            f->eta[c] = fmin(fmax(e, p->cutoff_lo), p->cutoff_hi);
        }
    if (!f->eta_v) return;
    for (int64_t j = 0; j <= ny; j++)
        for (int64_t i = 0; i <= nx; i++) {
            const size_t v = IDX2(nx + 1, i, j);
            const int64_t il = i > 0 ? i - 1 : 0, ir = i < nx ? i : nx - 1, jb = j > 0 ? j - 1 : 0, jt = j < ny ? j : ny - 1;
            const double AII = mat_visc_invariant2(0.0, 0.0, tau ? f->txy[v] : f->exy[v]);
            const double P = 0.25 * (f->P[IDX2(nx, il, jb)] + f->P[IDX2(nx, ir, jb)] + f->P[IDX2(nx, il, jt)] + f->P[IDX2(nx, ir, jt)]);
            double T = 0.0;
            if (f->T && p->T_ghosted)
                T = 0.25 * (f->T[IDX2(nx + 2, i, j)] + f->T[IDX2(nx + 2, i + 1, j)] + f->T[IDX2(nx + 2, i, j + 1)] + f->T[IDX2(nx + 2, i + 1, j + 1)]);
            else if (f->T) T = 0.25 * (f->T[IDX2(nx, il, jb)] + f->T[IDX2(nx, ir, jb)] + f->T[IDX2(nx, il, jt)] + f->T[IDX2(nx, ir, jt)]);
            double e = mat_phase_viscosity(rh, f->phase_v + (size_t)np * v, AII, T, P, tau);
            e = e * nu + f->eta_v[v] * (1.0 - nu);
            f->eta_v[v] = fmin(fmax(e, p->cutoff_lo), p->cutoff_hi);
        }
}
void orc_compute_viscosity2d(const orc_vep2d *f, const orc_rheology *rh, const orc_vep_params2d *p, double nu) { orc_compute_viscosity2d_form(f, rh, p, nu, 0); }
void orc_compute_viscosity2d_form(const orc_vep2d *f, const orc_rheology *rh, const orc_vep_params2d *p, double nu, int32_t tau)
{
    const int64_t nx = p->nx, ny = p->ny;
    const int np = rh->nphase;
    if (mat_viscosity_reads_fields(rh)) { viscosity2d_fields(f, rh, p, nu, tau); return; }
    for (int64_t c = 0; c < nx * ny; c++) {
        double e = phase_viscosity(rh, f->phase_c + (size_t)np * c);
        e = e * nu + f->eta[c] * (1.0 - nu);                        /* continuation_linear */
        f->eta[c] = fmin(fmax(e, p->cutoff_lo), p->cutoff_hi);
    }
    if (f->eta_v)
        for (int64_t v = 0; v < (nx + 1) * (ny + 1); v++) {
            double e = phase_viscosity(rh, f->phase_v + (size_t)np * v);
            e = e * nu + f->eta_v[v] * (1.0 - nu);
            f->eta_v[v] = fmin(fmax(e, p->cutoff_lo), p->cutoff_hi);
        }
}

/* tensor_invariant_kernel! 2D (StressKernels.jl:458-470) */
void orc_tensor_invariant2d(double *II, const double *xx, const double *yy, const double *xy, int64_t nx, int64_t ny, int32_t mode)
{
    for (int64_t j = 0; j < ny; j++)
        for (int64_t i = 0; i < nx; i++)
            II[IDX2(nx, i, j)] = sinv_stag(xx[IDX2(nx, i, j)], yy[IDX2(nx, i, j)], V2(xy, i, j), V2(xy, i + 1, j), V2(xy, i, j + 1), V2(xy, i + 1, j + 1), mode);
}

static void shear2center(double *c, const double *v, int64_t nx, int64_t ny)
{   /* Interpolations.jl:304-309 */
    if (!c || !v) return;
    for (int64_t j = 0; j < ny; j++)
        for (int64_t i = 0; i < nx; i++) c[IDX2(nx, i, j)] = 0.25 * (V2(v, i, j) + V2(v, i + 1, j) + V2(v, i, j + 1) + V2(v, i + 1, j + 1));
}

int32_t orc_stokes2d_vep_solve(const orc_vep2d *f, const orc_rheology *rh, const orc_vep_params2d *p, orc_result *res)
{
    const int64_t nx = p->nx, ny = p->ny;
    const size_t n = (size_t)nx * ny, nv = (size_t)(nx + 1) * (ny + 1);
    const int np = rh->nphase;
    double *etatau = malloc(n * 8), *theta = malloc(n * 8), *lam = calloc(n, 8), *lamv = calloc(nv, 8), *Kc = malloc(n * 8), *Gc = malloc(n * 8);
    memcpy(f->P0, f->P, n * 8);                      /* @copy stokes.P0 stokes.P :608 */
    orc_compute_maxloc2d(etatau, f->eta, nx, ny);
    memcpy(theta, f->P, n * 8);                      /* θ = deepcopy(stokes.P) :635 */
    memset(f->eplxx, 0, n * 8); memset(f->eplyy, 0, n * 8); memset(f->eplxy_c, 0, n * 8);   /* :641-643 */
    for (size_t c = 0; c < n; c++) { Kc[c] = ratio_avg(rh->Kb, f->phase_c + np * c, np); Gc[c] = ratio_avg(rh->G, f->phase_c + np * c, np); }

    /* views of this problem for the shared 2D kernels */
    orc_fields2d g;
    memset(&g, 0, sizeof(g));
    g.P = f->P; g.P0 = f->P0; g.divV = f->divV; g.Q = f->Q; g.Vx = f->Vx; g.Vy = f->Vy; g.Ux = f->Ux; g.Uy = f->Uy;
    g.txx = f->txx; g.tyy = f->tyy; g.txy = f->txy; g.exx = f->exx; g.eyy = f->eyy; g.exy = f->exy;
    g.eta = f->eta; g.fx = f->fx; g.fy = f->fy; g.RP = f->RP; g.Rx = f->Rx; g.Ry = f->Ry;
    orc_params2d q;
    memset(&q, 0, sizeof(q));
    q.nx = nx; q.ny = ny; q.nxg = p->nxg; q.nyg = p->nyg; q._dx = p->_dx; q._dy = p->_dy; q.dt = p->dt; q.r = p->r;
    for (int d_ = 0; d_ < 6; d_++) q.inv_spacing[d_] = p->inv_spacing[d_];
    q.theta_dtau = p->theta_dtau; q.eta_dtau = p->eta_dtau; q.free_slip = p->free_slip; q.no_slip = p->no_slip; q.periodic = p->periodic;

    /* compute_ρg!(ρg, phase_ratios, rheology, args) :646 -- scalar gravity fills the last component (BuoyancyForces.jl:69-70) */
    const int upd_rho = rh->has_density && !mat_density_is_constant(rh);
    if (rh->has_density)
        for (size_t c = 0; c < n; c++)      /* T_ghosted: args.T = thermal.T (nx+2, ny+2), read at the cell's own [i, j], unshifted (getindex_NamedTuple, BuoyancyForces.jl:52) */
                f->fy[c] = mat_density_ratio(rh, f->phase_c + np * c, !f->T ? 0.0 : (p->T_ghosted ? f->T[IDX2(nx + 2, c % (size_t)nx, c / (size_t)nx)] : f->T[c]), f->P[c]) * rh->gravity;
    /* displacement2velocity!(stokes, dt, flow_bcs) :647 -- V = U * inv(dt) for DisplacementBoundaryConditions only */
    if (p->displacement_bcs) {
        const double _dt = inv(p->dt);
        for (size_t c = 0; c < (size_t)(nx + 1) * (ny + 2); c++) f->Vx[c] = f->Ux[c] * _dt;
        for (size_t c = 0; c < (size_t)(nx + 2) * (ny + 1); c++) f->Vy[c] = f->Uy[c] * _dt;
    }
    const double fs_dt = p->free_surface ? p->dt : 0.0;      /* dt * free_surface with a Bool: Julia's false is a strong zero, Inf * false == 0.0 */

    double err_it1 = 1.0, err = 1.0;
    int64_t iter = 0, cont = 0;
    res->status = 0;
    while (iter <= p->iterMax) {
        if (p->iterMin < iter && ((err / err_it1) < p->eps_rel || err < p->eps_abs)) break;    /* :650-651 */
        orc_compute_maxloc2d(etatau, f->eta, nx, ny);
        { const int64_t e[3] = {nx, ny, 1}; orc_self_halo(etatau, e, e); }                               /* update_halo!(ητ) :655 */
        orc_compute_divV2d_sp(f->divV, f->Vx, f->Vy, nx, ny, p->_dx, p->_dy, p->inv_spacing);
        orc_compute_P3d(theta, f->P0, f->RP, f->divV, f->Q, etatau, Kc, Gc, (int64_t)n, p->dt, p->r, p->theta_dtau);   /* :663-676 */
        if (upd_rho)                                    /* update_ρg!(ρg, phase_ratios, rheology, args) :678 ; args.P is stokes.P */
            for (size_t c = 0; c < n; c++)      /* T_ghosted: args.T = thermal.T (nx+2, ny+2), read at the cell's own [i, j], unshifted (getindex_NamedTuple, BuoyancyForces.jl:52) */
                f->fy[c] = mat_density_ratio(rh, f->phase_c + np * c, !f->T ? 0.0 : (p->T_ghosted ? f->T[IDX2(nx + 2, c % (size_t)nx, c / (size_t)nx)] : f->T[c]), f->P[c]) * rh->gravity;
        if (p->strain_increment) {
            /* ∇U, Δε from the displacements (:659-661, :680-688), then ε = Δε * _dt (compute_strain_rate_from_increment!, VelocityKernels.jl:46-57) */
            orc_fields2d gu = g;
            gu.Vx = f->Ux; gu.Vy = f->Uy; gu.divV = f->divU; gu.exx = f->dexx; gu.eyy = f->deyy; gu.exy = f->dexy;
            orc_compute_divV2d(f->divU, f->Ux, f->Uy, nx, ny, p->_dx, p->_dy);
            orc_compute_strain_rate2d(&gu, &q);
            const double _dt = inv(p->dt);
            for (size_t c = 0; c < n; c++) { f->exx[c] = f->dexx[c] * _dt; f->eyy[c] = f->deyy[c] * _dt; }
            for (size_t v = 0; v < nv; v++) f->exy[v] = f->dexy[v] * _dt;
        } else orc_compute_strain_rate2d(&g, &q);
        orc_vep2d_stress(f, theta, lam, lamv, rh, p);
        { const int64_t e[3] = {nx + 1, ny + 1, 1}, nn[3] = {nx, ny, 1}; orc_self_halo(f->txy, e, nn); }    /* update_halo!(τ.xy) :757 */
        orc_compute_viscosity2d_form(f, rh, p, p->viscosity_relaxation, 1);         /* update_viscosity_τII! */
        orc_compute_V2d_fs(&g, etatau, &q, fs_dt);   /* free-surface form; with dt*free_surface = 0 it reduces to the plain one */
        orc_velocity2displacement2d(&g, &q);
        if (p->displacement_bcs) orc_flow_bcs2d(f->Ux, f->Uy, nx, ny, p->free_slip, p->no_slip, p->periodic);   /* flow_bcs! on @displacement */
        else orc_flow_bcs2d(f->Vx, f->Vy, nx, ny, p->free_slip, p->no_slip, p->periodic);
        {   /* update_halo!(@velocity(stokes)...) :784 */
            const int64_t nn[3] = {nx, ny, 1}, ex[3] = {nx + 1, ny + 2, 1}, ey[3] = {nx + 2, ny + 1, 1};
            orc_self_halo(f->Vx, ex, nn); orc_self_halo(f->Vy, ey, nn);
        }
        iter += 1;
        if (iter % p->nout == 0 && iter > 1) {
            orc_compute_Res2d_fs(&g, &q, fs_dt);
            double s[3];
            orc_residual_sumsq2d(&g, &q, s);
            const double nRx = sqrt(s[0]) / sqrt((double)((p->nxg - 2) * (p->nyg - 1)));
            const double nRy = sqrt(s[1]) / sqrt((double)((p->nxg - 1) * (p->nyg - 2)));
            const double nDV = sqrt(s[2]) / sqrt((double)(p->nxg * p->nyg));
            err = fmax(nRx, fmax(nRy, nDV));
            if (isnan(nRx) || isnan(nRy) || isnan(nDV)) err = NAN;
            if (cont < res->cap) { res->norm_Rx[cont] = nRx; res->norm_Ry[cont] = nRy; res->norm_divV[cont] = nDV; res->err_evo1[cont] = err; res->err_evo2[cont] = iter; }
            if (cont == 0) err_it1 = err;
            cont++;
            if (isnan(err)) { res->status = 1; break; }
        }
    }
    res->iter = iter;
    res->nchecks = cont < res->cap ? cont : res->cap;
    if (res->status == 0) {
        if (f->omega_xy)                                /* compute_vorticity! :831-833 */
            for (int64_t j = 0; j < ny + 1; j++)
                for (int64_t i = 0; i < nx + 1; i++)
                    V2(f->omega_xy, i, j) = 0.5 * ((-f->Vy[IDX2(nx + 2, i, j)] + f->Vy[IDX2(nx + 2, i + 1, j)]) * (p->inv_spacing[5] ? p->inv_spacing[5][i] : p->_dx) -
                                                   (-f->Vx[IDX2(nx + 1, i, j)] + f->Vx[IDX2(nx + 1, i, j + 1)]) * (p->inv_spacing[4] ? p->inv_spacing[4][j] : p->_dy));
        shear2center(f->exy_c, f->exy, nx, ny);
        shear2center(f->eplxy_c, f->eplxy, nx, ny);
        shear2center(f->dexy_c, f->dexy, nx, ny);
        for (int64_t j = 0; j < ny; j++)                /* accumulate_tensor! / accumulate_vol! :842-843 */
            for (int64_t i = 0; i < nx; i++) {
                const size_t c = IDX2(nx, i, j);
                f->EII_pl[c] += sinv_stag(f->eplxx[c], f->eplyy[c], V2(f->eplxy, i, j), V2(f->eplxy, i + 1, j), V2(f->eplxy, i, j + 1),
                                          V2(f->eplxy, i + 1, j + 1), p->staggered_invariant_mean_of_squares) * p->dt;
                f->EVol_pl[c] += p->dt * f->evol_pl[c];
            }
        memcpy(f->toxx, f->txx, n * 8); memcpy(f->toyy, f->tyy, n * 8); memcpy(f->toxy, f->txy, nv * 8);    /* multi_copy! :845-846 */
        memcpy(f->toxy_c, f->txy_c, n * 8);
    }
    free(etatau); free(theta); free(lam); free(lamv); free(Kc); free(Gc);
    return res->status;
}

/* test hook: the softening laws of material.h */
double orc_soften(int32_t kind, double a, double b, double c, double d, double EII, double v0) { return mat_soften(kind, a, b, c, d, EII, v0); }


/* ---------------------------------------------------------------------------------------------------------------------------------
 * 2D single-phase visco-elasto-plastic driver: solve!(stokes, pt_stokes, grid, flow_bcs, ρg, rheology::MaterialParams, args, dt, igg)
 * -- src/stokes/Stokes2D.jl:345-557, the only caller of compute_τ_nonlinear! + center2vertex!; reached by test/test_WENO5.jl:226-291.
 * The rheology is phase 0 of the table.  args = (; T, P = stokes.P): with T_ghosted the temperature is thermal.T (nx+2, ny+2) and is
 * indexed as the reference does -- compute_ρg! at [i, j] (getindex_NamedTuple(args, I...), BuoyancyForces.jl:17) but the viscosity at
 * [i+1, j+1] (local_viscosity_args, Viscosity.jl:513-523); otherwise T is cell-centred (nx, ny).
 * compute_viscosity! / compute_viscosity_τII! (Viscosity.jl:142-167) for creep laws without strain-rate dependence:
 * η <- clamp((1 - ν) η + ν η_creep(T, P), cutoff). */
static inline double T_at(const orc_vep2d *f, const orc_vep_params2d *p, int64_t i, int64_t j, int shift)
{
    if (!f->T) return 0.0;
    if (p->T_ghosted) return f->T[IDX2(p->nx + 2, i + shift, j + shift)];
    return f->T[IDX2(p->nx, i, j)];
}
/* a power-law creep takes its invariant from @strain(stokes) = (ε.xx, ε.yy, ε.xy[i, j]: the vertex array at the cell's index) whatever fn_viscosity is
 * (_compute_viscosity!(stokes, ν, args, rheology, cutoff, fn_viscosity), Viscosity.jl:136-167) */
static void visc_single(const orc_vep2d *f, const orc_rheology *rh, const orc_vep_params2d *p, double nu, int tau)
{
    const int64_t nx = p->nx, ny = p->ny;
    for (int64_t j = 0; j < ny; j++)
        for (int64_t i = 0; i < nx; i++) {
            const size_t c = IDX2(nx, i, j);
            const double AII = rh->visc_kind[0] == 2 ? mat_visc_invariant2(f->exx[c], f->eyy[c], f->exy[IDX2(nx + 1, i, j)]) : 0.0;
            const double en = mat_viscosity(rh, 0, AII, T_at(f, p, i, j, 1), f->P[c], tau);
            const double e = (1 - nu) * f->eta[c] + nu * en;                        /* continuation_linear, Utils.jl:662 */
            f->eta[c] = fmin(fmax(e, p->cutoff_lo), p->cutoff_hi);
        }
}
static void rhog_single(const orc_vep2d *f, const orc_rheology *rh, const orc_vep_params2d *p)
{
    const int64_t nx = p->nx, ny = p->ny;
    for (int64_t j = 0; j < ny; j++)
        for (int64_t i = 0; i < nx; i++) {
            const size_t c = IDX2(nx, i, j);
            f->fy[c] = mat_density(rh, 0, T_at(f, p, i, j, 0), f->P[c]) * rh->gravity;     /* compute_buoyancy(rheology, args_ijk) */
        }
}

int32_t orc_stokes2d_nonlinear_solve(const orc_vep2d *f, const orc_rheology *rh, const orc_vep_params2d *p, orc_result *res)
{
    const int64_t nx = p->nx, ny = p->ny;
    const size_t n = (size_t)nx * ny, nv = (size_t)(nx + 1) * (ny + 1);
    double *etatau = malloc(n * 8), *theta = calloc(n, 8), *lam = calloc(n, 8), *Kc = malloc(n * 8), *Gc = malloc(n * 8);
    orc_compute_maxloc2d(etatau, f->eta, nx, ny);                                   /* :372-375 */
    for (size_t c = 0; c < n; c++) { Kc[c] = rh->Kb[0]; Gc[c] = rh->G[0]; }         /* Kb = get_Kb(rheology); G = get_G(rheology) :378-379 */
    memset(f->eplxx, 0, n * 8); memset(f->eplyy, 0, n * 8); memset(f->eplxy_c, 0, n * 8);   /* :391-393 */
    const int upd_rho = rh->has_density && rh->rho_kind[0] != 0;
    if (rh->has_density) rhog_single(f, rh, p);                                    /* compute_ρg!(ρg[end], rheology, args) :406 */
    visc_single(f, rh, p, 1.0, 0);                                                     /* compute_viscosity!(stokes, args, rheology, cutoff) :407 */
    if (p->displacement_bcs) {                                                      /* displacement2velocity! :410 */
        const double _dt = inv(p->dt);
        for (size_t c = 0; c < (size_t)(nx + 1) * (ny + 2); c++) f->Vx[c] = f->Ux[c] * _dt;
        for (size_t c = 0; c < (size_t)(nx + 2) * (ny + 1); c++) f->Vy[c] = f->Uy[c] * _dt;
    }
    const double fs_dt = p->free_surface ? p->dt : 0.0;      /* dt * free_surface with a Bool: Julia's false is a strong zero, Inf * false == 0.0 */

    orc_fields2d g;
    memset(&g, 0, sizeof(g));
    g.P = f->P; g.P0 = f->P0; g.divV = f->divV; g.Q = f->Q; g.Vx = f->Vx; g.Vy = f->Vy; g.Ux = f->Ux; g.Uy = f->Uy;
    g.txx = f->txx; g.tyy = f->tyy; g.txy = f->txy; g.exx = f->exx; g.eyy = f->eyy; g.exy = f->exy;
    g.eta = f->eta; g.fx = f->fx; g.fy = f->fy; g.RP = f->RP; g.Rx = f->Rx; g.Ry = f->Ry;
    orc_params2d q;
    memset(&q, 0, sizeof(q));
    q.nx = nx; q.ny = ny; q.nxg = p->nxg; q.nyg = p->nyg; q._dx = p->_dx; q._dy = p->_dy; q.dt = p->dt; q.r = p->r;
    for (int d_ = 0; d_ < 6; d_++) q.inv_spacing[d_] = p->inv_spacing[d_];
    q.theta_dtau = p->theta_dtau; q.eta_dtau = p->eta_dtau; q.free_slip = p->free_slip; q.no_slip = p->no_slip; q.periodic = p->periodic;

    double err_it1 = 1.0, err = 1.0;
    int64_t iter = 0, cont = 0;
    res->status = 0;
    while (iter < 2 || (((err / err_it1) > p->eps_rel && err > p->eps_abs) && iter <= p->iterMax)) {      /* :412 */
        orc_compute_maxloc2d(etatau, f->eta, nx, ny);
        orc_compute_divV2d_sp(f->divV, f->Vx, f->Vy, nx, ny, p->_dx, p->_dy, p->inv_spacing);
        orc_compute_P3d(f->P, f->P0, f->RP, f->divV, f->Q, f->eta, Kc, Gc, (int64_t)n, p->dt, p->r, p->theta_dtau);   /* with η, in place :418-420 */
        if (upd_rho) rhog_single(f, rh, p);                                        /* update_ρg!(ρg[2], rheology, args) :422 */
        orc_compute_strain_rate2d(&g, &q);
        visc_single(f, rh, p, p->viscosity_relaxation, 1);                            /* compute_viscosity_τII! :433-435 */
        orc_compute_maxloc2d(etatau, f->eta, nx, ny);                              /* :437-438 */
        orc_compute_tau_nonlinear2d(f, theta, lam, rh, p, 0);                      /* :440-458 */
        orc_center2vertex2d(f->txy, f->txy_c, nx, ny);                             /* :459 */
        orc_compute_V2d_fs(&g, etatau, &q, fs_dt);                                 /* :463-474 */
        orc_velocity2displacement2d(&g, &q);
        if (p->displacement_bcs) orc_flow_bcs2d(f->Ux, f->Uy, nx, ny, p->free_slip, p->no_slip, p->periodic);
        else orc_flow_bcs2d(f->Vx, f->Vy, nx, ny, p->free_slip, p->no_slip, p->periodic);
        iter += 1;
        if (iter % p->nout == 0 && iter > 1) {
            orc_compute_Res2d_fs(&g, &q, fs_dt);
            double s[3];
            orc_residual_sumsq2d(&g, &q, s);
            const double nRx = sqrt(s[0]) / sqrt((double)((p->nxg - 2) * (p->nyg - 1)));
            const double nRy = sqrt(s[1]) / sqrt((double)((p->nxg - 1) * (p->nyg - 2)));
            const double nDV = sqrt(s[2]) / sqrt((double)(p->nxg * p->nyg));
            err = fmax(nRx, fmax(nRy, nDV));
            if (isnan(nRx) || isnan(nRy) || isnan(nDV)) err = NAN;
            if (cont < res->cap) { res->norm_Rx[cont] = nRx; res->norm_Ry[cont] = nRy; res->norm_divV[cont] = nDV; res->err_evo1[cont] = err; res->err_evo2[cont] = iter; }
            if (cont == 0) err_it1 = err;
            cont++;
            if (isnan(err)) { res->status = 1; break; }
        }
    }
    res->iter = iter;
    res->nchecks = cont < res->cap ? cont : res->cap;
    if (res->status == 0) {
        memcpy(f->P, theta, n * 8);                                                 /* stokes.P .= θ :523 */
        if (f->omega_xy)
            for (int64_t j = 0; j < ny + 1; j++)
                for (int64_t i = 0; i < nx + 1; i++)
                    V2(f->omega_xy, i, j) = 0.5 * ((-f->Vy[IDX2(nx + 2, i, j)] + f->Vy[IDX2(nx + 2, i + 1, j)]) * (p->inv_spacing[5] ? p->inv_spacing[5][i] : p->_dx) -
                                                   (-f->Vx[IDX2(nx + 1, i, j)] + f->Vx[IDX2(nx + 1, i, j + 1)]) * (p->inv_spacing[4] ? p->inv_spacing[4][j] : p->_dy));
        shear2center(f->exy_c, f->exy, nx, ny);
        shear2center(f->eplxy_c, f->eplxy, nx, ny);
        shear2center(f->dexy_c, f->dexy, nx, ny);
        for (int64_t j = 0; j < ny; j++)
            for (int64_t i = 0; i < nx; i++) {
                const size_t c = IDX2(nx, i, j);
                f->EII_pl[c] += sinv_stag(f->eplxx[c], f->eplyy[c], V2(f->eplxy, i, j), V2(f->eplxy, i + 1, j), V2(f->eplxy, i, j + 1),
                                          V2(f->eplxy, i + 1, j + 1), p->staggered_invariant_mean_of_squares) * p->dt;
                f->EVol_pl[c] += p->dt * f->evol_pl[c];
            }
        memcpy(f->toxx, f->txx, n * 8); memcpy(f->toyy, f->tyy, n * 8); memcpy(f->toxy, f->txy, nv * 8);
        memcpy(f->toxy_c, f->txy_c, n * 8);
    }
    free(etatau); free(theta); free(lam); free(Kc); free(Gc);
    return res->status;
}
