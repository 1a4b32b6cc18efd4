/* oracle/stokes3d_vep.c -- TEST INFRASTRUCTURE ONLY (see jrx_oracle.h).
 * CPU restatement of the 3D multiphase visco-elasto-plastic PT Stokes driver of JustRelax.jl
 * (src/stokes/Stokes3D.jl:447-668) and its stress kernel update_stresses_center_vertex_ps! 3D
 * (src/stokes/StressKernels.jl:604-989), as test/test_shearband3D_MPI.jl drives them.
 *
 * GeoParams forms as in stokes2d_vep.c (ASSUMED there, pinned by the 2D shear-band scalars):
 *   second_invariant(xx,yy,zz,yz,xz,xy) = sqrt(0.5*(xx^2+yy^2+zz^2) + yz^2+xz^2+xy^2)
 *   second_invariant_staggered: each shear slot enters as the mean of its 4 squared edge values
 *   DruckerPrager_regularised: F = tauII - cos(phi) C - sin(phi) P; dQ/dtau_ij = tau_ij/(2 tauII); dQ/dP = -sin(psi);
 *   dF/dP = -sin(phi); LinearViscous + dt = Inf: composite viscosity = the linear one.
 * PARITY UNPINNED in 3D: test/test_shearband3D_MPI.jl asserts nothing.  Checked here by (i) the 2D restatement it
 * mirrors, (ii) a 3D problem that is uniform in y reproducing the 2D kernel's plane-strain update (tests/).
 *
 * The reference launches all four updates (yz, xz, xy edges, centres) in one kernel in which edge updates read
 * neighbouring edge/centre stresses that other threads overwrite (a data race).  Here, and in the HIP kernels, every
 * update reads the values of the previous iteration: the new edge stresses go to temporaries that are committed after
 * the three edge passes, and the centre pass runs last. */
#include "jrx_oracle.h"
#include "common.h"
#include "material.h"
#include <stdlib.h>
#include <string.h>
#include <omp.h>

/* update_halo!(A) on a grid that is IGG-periodic in the flagged dimensions and held by ONE rank (the rank is its own neighbour):
 * per dimension x -> y -> z the left ghost plane takes the right send plane and vice versa (overlap ol_A = 2 + size_A - n).
 * Test hook for the multi-rank code path of the device drivers; off by default. */
static int g_self_periods[3] = {0, 0, 0};
void orc_set_self_halo(int px, int py, int pz) { g_self_periods[0] = px; g_self_periods[1] = py; g_self_periods[2] = pz; }
void orc_self_halo(double *A, const int64_t ext[3], const int64_t n[3])
{
    const int64_t s[3] = {1, ext[0], ext[0] * ext[1]};
    for (int d = 0; d < 3; d++) {
        if (!g_self_periods[d]) continue;
        const int64_t nA = ext[d], ol = 2 + (nA - n[d]);
        if (ol < 2 || nA < ol) continue;
        const int d1 = d == 0 ? 1 : 0, d2 = d == 2 ? 1 : 2;
        const int64_t sl = ol - 1, sr = nA - ol;
        for (int64_t v = 0; v < ext[d2]; v++)
            for (int64_t u = 0; u < ext[d1]; u++) {
                const int64_t base = u * s[d1] + v * s[d2];
                const double left_going = A[base + sl * s[d]], right_going = A[base + sr * s[d]];
                A[base + (nA - 1) * s[d]] = left_going;
                A[base] = right_going;
            }
    }
}

static inline double sinv3(const double t[6])
{
    return sqrt(0.5 * (t[0] * t[0] + t[1] * t[1] + t[2] * t[2]) + t[3] * t[3] + t[4] * t[4] + t[5] * t[5]);
}

static inline double ratio_avg(const double *val, const double *r, int n)
{   /* fn_ratio (src/phases/phases.jl:6-15) */
    double x = 0.0;
    for (int q = 0; q < n; q++) x += (r[q] == 0.0) ? 0.0 : val[q] * r[q];
    return x;
}
static inline void plastic_params(const orc_rheology *rh, const double *r, int *is_pl, double *eta_reg)
{   /* plastic_params_phase (StressUpdate.jl:152-176) */
    *is_pl = 0; *eta_reg = 0.0;
    for (int q = 0; q < rh->nphase; q++)
        if (rh->is_pl[q]) { *is_pl = 1; *eta_reg += rh->eta_vp[q] * r[q]; }
}
static inline double yield_F(const orc_rheology *rh, const double *r, double P, double tII, double EII)
{   /* compute_yieldfunction_phase (StressUpdate.jl:435-452); softening laws evaluated at the EII keyword */
    double F = 0.0;
    for (int q = 0; q < rh->nphase; q++) {
        if (r[q] == 0.0) continue;
        double Fq = tII;
        if (rh->is_pl[q]) {
            double sp, cp;
            mat_friction(rh, q, EII, &sp, &cp);
            Fq = tII - cp * mat_cohesion(rh, q, EII) - sp * P;
        }
        F += r[q] * Fq;
    }
    return F;
}
static inline void plastic_grad(const orc_rheology *rh, const double *r, const double t[6], double dQdt[6], double *dQdP, double *dFdP)
{   /* compute_plastic_gradients_phase (StressUpdate.jl:463-550); shear slots halved once (:466-472) */
    for (int q = 0; q < 6; q++) dQdt[q] = 0.0;
    *dQdP = 0.0; *dFdP = 0.0;
    const double tII = sinv3(t);
    for (int q = 0; q < rh->nphase; q++) {
        if (r[q] == 0.0 || !rh->is_pl[q]) continue;
        for (int s = 0; s < 3; s++) dQdt[s] = fma(r[q], 0.5 * t[s] / tII, dQdt[s]);
        for (int s = 3; s < 6; s++) dQdt[s] = fma(r[q], 0.5 * (t[s] / tII), dQdt[s]);
        *dQdP = fma(r[q], -rh->sinpsi[q], *dQdP);
        *dFdP = fma(r[q], -rh->sinphi[q], *dFdP);
    }
}

/* Stencil tables for the three edge families t = 0 (yz), 1 (xz), 2 (xy).  Entries select one of the clamped indices
 * {0: n-1 clamped, 1: n clamped, 2: n+1 clamped} per direction (clamped_indices, StressKernels.jl:604-616).
 * CEN: av_clamped_yz/xz/xy and harm_clamped_* (:618-640); OTH[t][s]: the average that brings edge family s to an edge
 * of family t (av_clamped_yz_y, _yz_z, _xz_x, _xz_z, _xy_x, _xy_y, :642-668), in the reference's summation order. */
static const int CEN[3][4][3] = {
    {{1, 0, 0}, {1, 1, 0}, {1, 0, 1}, {1, 1, 1}},
    {{0, 1, 0}, {1, 1, 0}, {0, 1, 1}, {1, 1, 1}},
    {{0, 0, 1}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}}};
static const int OTH[3][3][4][3] = {
    /* on yz */ {{{0}}, /* xz: av_clamped_yz_y */ {{1, 0, 1}, {2, 0, 1}, {1, 1, 1}, {2, 1, 1}}, /* xy: av_clamped_yz_z */ {{1, 1, 0}, {2, 1, 0}, {1, 1, 1}, {2, 1, 1}}},
    /* on xz */ {/* yz: av_clamped_xz_x */ {{0, 1, 1}, {1, 1, 1}, {1, 2, 1}, {0, 2, 1}}, {{0}}, /* xy: av_clamped_xz_z */ {{1, 1, 0}, {1, 2, 0}, {1, 1, 1}, {1, 2, 1}}},
    /* on xy */ {/* yz: av_clamped_xy_x */ {{0, 1, 1}, {1, 1, 1}, {0, 1, 2}, {1, 1, 2}}, /* xz: av_clamped_xy_y */ {{1, 0, 1}, {1, 1, 1}, {1, 0, 2}, {1, 1, 2}}, {{0}}}};

typedef struct { int64_t n1, n2, n3; } ext3;

static inline ext3 edge_ext(int t, int64_t nx, int64_t ny, int64_t nz)
{
    ext3 e = {nx + (t != 0), ny + (t != 1), nz + (t != 2)};
    return e;
}

/* update_stresses_center_vertex_ps! 3D (StressKernels.jl:671-989) */
void orc_vep3d_stress(const orc_vep3d *f, const double *theta, double *lam, double *const lamv[3], const orc_rheology *rh,
                      const orc_vep_params3d *p)
{
    const int64_t nx = p->nx, ny = p->ny, nz = p->nz;
    const int np = rh->nphase;
    const double dt = p->dt, th = p->theta_dtau, rel = p->lambda_relaxation;
    double *const tsh[3] = {f->tyz, f->txz, f->txy};
    double *const tosh[3] = {f->toyz, f->toxz, f->toxy};
    double *const esh[3] = {f->eyz, f->exz, f->exy};
    double *const eplsh[3] = {f->eplyz, f->eplxz, f->eplxy};
    const double *const phsh[3] = {f->phase_yz, f->phase_xz, f->phase_xy};
    const double *const en[3] = {f->exx, f->eyy, f->ezz};
    const double *const tn[3] = {f->txx, f->tyy, f->tzz};
    const double *const ton[3] = {f->toxx, f->toyy, f->tozz};
    double *tnew[3];
    for (int t = 0; t < 3; t++) {
        const ext3 E = edge_ext(t, nx, ny, nz);
        tnew[t] = malloc((size_t)E.n1 * E.n2 * E.n3 * sizeof(double));
    }
    for (int t = 0; t < 3; t++) {
        const ext3 E = edge_ext(t, nx, ny, nz);
#pragma omp parallel for schedule(static)
        for (int64_t k = 0; k < E.n3; k++)
            for (int64_t j = 0; j < E.n2; j++)
                for (int64_t i = 0; i < E.n1; i++) {
                    const int64_t ci[3] = {clampi(i - 1, 0, nx - 1), clampi(i, 0, nx - 1), clampi(i + 1, 0, nx - 1)};
                    const int64_t cj[3] = {clampi(j - 1, 0, ny - 1), clampi(j, 0, ny - 1), clampi(j + 1, 0, ny - 1)};
                    const int64_t ck[3] = {clampi(k - 1, 0, nz - 1), clampi(k, 0, nz - 1), clampi(k + 1, 0, nz - 1)};
                    size_t cidx[4];
                    for (int q = 0; q < 4; q++) cidx[q] = IDX3(nx, ny, ci[CEN[t][q][0]], cj[CEN[t][q][1]], ck[CEN[t][q][2]]);
#define AVC(A) (0.25 * ((A)[cidx[0]] + (A)[cidx[1]] + (A)[cidx[2]] + (A)[cidx[3]]))
                    const double etav = 4 / (1 / f->eta[cidx[0]] + 1 / f->eta[cidx[1]] + 1 / f->eta[cidx[2]] + 1 / f->eta[cidx[3]]);
                    const double Pv = AVC(theta);
                    const size_t v = IDX3(E.n1, E.n2, i, j, k);
                    double eij[6], tij[6], toij[6];
                    for (int s = 0; s < 3; s++) { eij[s] = AVC(en[s]); tij[s] = AVC(tn[s]); toij[s] = AVC(ton[s]); }
                    for (int s = 0; s < 3; s++) {
                        if (s == t) { eij[3 + s] = esh[s][v]; tij[3 + s] = tsh[s][v]; toij[3 + s] = tosh[s][v]; continue; }
                        const ext3 S = edge_ext(s, nx, ny, nz);
                        size_t o[4];
                        for (int q = 0; q < 4; q++) o[q] = IDX3(S.n1, S.n2, ci[OTH[t][s][q][0]], cj[OTH[t][s][q][1]], ck[OTH[t][s][q][2]]);
                        eij[3 + s] = 0.25 * (esh[s][o[0]] + esh[s][o[1]] + esh[s][o[2]] + esh[s][o[3]]);
                        tij[3 + s] = 0.25 * (tsh[s][o[0]] + tsh[s][o[1]] + tsh[s][o[2]] + tsh[s][o[3]]);
                        toij[3 + s] = 0.25 * (tosh[s][o[0]] + tosh[s][o[1]] + tosh[s][o[2]] + tosh[s][o[3]]);
                    }
                    const double *rv = phsh[t] + (size_t)np * v;
                    int is_pl; double eta_reg;
                    plastic_params(rh, rv, &is_pl, &eta_reg);
                    const double _Gdt = inv(ratio_avg(rh->G, rv, np) * dt);
                    const double Kv = ratio_avg(rh->Kb, rv, np);
                    const double dtr = inv(th + etav * _Gdt + 1.0);
                    double d[6], tt[6];
                    for (int s = 0; s < 6; s++) { d[s] = stress_increment(tij[s], toij[s], etav, eij[s], _Gdt, dtr); tt[s] = tij[s] + d[s]; }
                    const double tIIv = sinv3(tt);
                    double dQdt[6], dQdP, dFdP;
                    plastic_grad(rh, rv, tt, dQdt, &dQdP, &dFdP);
                    const double vol = isinf(Kv) ? 0.0 : Kv * dt * dFdP * dQdP;
                    const double F = yield_F(rh, rv, Pv, tIIv, AVC(f->EII_pl));      /* EIIv_ij = av_clamped_yz/xz/xy(EII, Ic...) :710,783,854 */
                    const int own = 3 + t;
                    if (is_pl && tIIv != 0.0 && F > 0) {
                        lamv[t][v] = (1.0 - rel) * lamv[t][v] + rel * (fmax(F, 0.0) / (etav * dtr + eta_reg + vol));
                        const double epl = lamv[t][v] * dQdt[own];
                        tnew[t][v] = tij[own] + fma(-(2.0 * etav * epl), dtr, d[own]);      /* @muladd dτ - 2η ε_pl dτ_r */
                        eplsh[t][v] = epl;
                    } else {
                        tnew[t][v] = tij[own] + d[own];
                        eplsh[t][v] = 0.0;
                    }
#undef AVC
                }
    }
    for (int t = 0; t < 3; t++) {
        const ext3 E = edge_ext(t, nx, ny, nz);
        memcpy(tsh[t], tnew[t], (size_t)E.n1 * E.n2 * E.n3 * sizeof(double));
        free(tnew[t]);
    }
    /* centre pass (:906-985) */
    double *const tc[6] = {f->txx, f->tyy, f->tzz, f->tyz_c, f->txz_c, f->txy_c};
    const double *const toc[6] = {f->toxx, f->toyy, f->tozz, f->toyz_c, f->toxz_c, f->toxy_c};
    double *const eplc[3] = {f->eplxx, f->eplyy, f->eplzz};
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < nz; k++)
        for (int64_t j = 0; j < ny; j++)
            for (int64_t i = 0; i < nx; i++) {
                const size_t c = IDX3(nx, ny, i, j, k);
                const double *rc = f->phase_c + (size_t)np * c;
                const double _Gdt = inv(ratio_avg(rh->G, rc, np) * dt);
                int is_pl; double eta_reg;
                plastic_params(rh, rc, &is_pl, &eta_reg);
                const double K = ratio_avg(rh->Kb, rc, np);
                const double e = f->eta[c];
                const double dtr = inv(th + e * _Gdt + 1.0);
                /* cache_tensors 3D (StressUpdate.jl:269-301): _av_yz/_av_xz/_av_xy = 0.25*mysum (MiniKernels.jl:116-121) */
                double eij[6] = {f->exx[c], f->eyy[c], f->ezz[c], 0, 0, 0};
                {
                    double s = 0.0;
                    for (int64_t kk = k; kk <= k + 1; kk++) for (int64_t jj = j; jj <= j + 1; jj++) s += f->eyz[IDX3(nx, ny + 1, i, jj, kk)];
                    eij[3] = 0.25 * s;
                    s = 0.0;
                    for (int64_t kk = k; kk <= k + 1; kk++) for (int64_t ii = i; ii <= i + 1; ii++) s += f->exz[IDX3(nx + 1, ny, ii, j, kk)];
                    eij[4] = 0.25 * s;
                    s = 0.0;
                    for (int64_t jj = j; jj <= j + 1; jj++) for (int64_t ii = i; ii <= i + 1; ii++) s += f->exy[IDX3(nx + 1, ny + 1, ii, jj, k)];
                    eij[5] = 0.25 * s;
                }
                double tij[6], toij[6], d[6], tt[6];
                for (int s = 0; s < 6; s++) { tij[s] = tc[s][c]; toij[s] = toc[s][c]; }
                for (int s = 0; s < 6; s++) {
                    d[s] = (-(tij[s] - toij[s]) * e * _Gdt - tij[s] + 2.0 * e * eij[s]) * dtr;         /* :926 (plain, no fma) */
                    tt[s] = tij[s] + d[s];
                }
                double tII;
                {
                    double dt6[6];
                    for (int s = 0; s < 6; s++) dt6[s] = d[s] + tij[s];
                    tII = sinv3(dt6);
                }
                double dQdt[6], dQdP, dFdP;
                plastic_grad(rh, rc, tt, dQdt, &dQdP, &dFdP);
                const double vol = isinf(K) ? 0.0 : K * dt * dFdP * dQdP;
                const double Pr = theta[c];
                const double F = yield_F(rh, rc, Pr, tII, f->EII_pl[c]);
                if (is_pl && tII != 0.0 && F > 0) {
                    lam[c] = (1.0 - rel) * lam[c] + rel * (fmax(F, 0.0) / (e * dtr + eta_reg + vol));
                    double epl[6];
                    for (int s = 0; s < 6; s++) { epl[s] = lam[c] * dQdt[s]; d[s] = d[s] - 2.0 * e * epl[s] * dtr; tij[s] = d[s] + tij[s]; }
                    f->evol_pl[c] = -lam[c] * dQdP;
                    for (int s = 0; s < 6; s++) tc[s][c] = tij[s];
                    for (int s = 0; s < 3; s++) eplc[s][c] = epl[s];
                    tII = sinv3(tij);
                } else {
                    f->evol_pl[c] = 0.0;
                    for (int s = 0; s < 6; s++) tc[s][c] = d[s] + tij[s];
                    for (int s = 0; s < 3; s++) eplc[s][c] = 0.0;
                }
                f->tII[c] = tII;
                f->eta_vep[c] = tII * 0.5 * inv(sinv3(eij));
                f->P[c] = Pr - (isinf(K) ? 0.0 : K * dt * lam[c] * dQdP);
            }
}

/* update_viscosity_τII! 3D (rheology/Viscosity.jl:67-106,282-300): centre viscosity relaxed towards the phase value */
/* creep laws that read fields (Viscosity.jl:455-503): invariant of @stress / @strain with the shear components gathered from the cell's four edges (mean of
 * squares, second_invariant_staggered), eps() on the normal components when those vanish; T at I .+ 1 of the ghosted thermal.T */
static void viscosity3d_fields(const orc_vep3d *f, const orc_rheology *rh, const orc_vep_params3d *p, double nu, int tau)
{
    const int64_t nx = p->nx, ny = p->ny, nz = p->nz;
    const double *xx = tau ? f->txx : f->exx, *yy = tau ? f->tyy : f->eyy, *zz = tau ? f->tzz : f->ezz;
    const double *yz = tau ? f->tyz : f->eyz, *xz = tau ? f->txz : f->exz, *xy = tau ? f->txy : f->exy;
    for (int64_t k = 0; k < nz; k++)
        for (int64_t j = 0; j < ny; j++)
            for (int64_t i = 0; i < nx; i++) {
                const size_t c = IDX3(nx, ny, i, j, k);
                const double a0 = (xx[c] == 0.0 && yy[c] == 0.0 && zz[c] == 0.0) ? 2.220446049250313e-16 : 0.0;
                const double x = xx[c] + a0, y = yy[c] + -a0 * 0.5, z = zz[c] + -a0 * 0.5;
                const double p0 = yz[IDX3(nx, ny + 1, i, j, k)], p1 = yz[IDX3(nx, ny + 1, i, j + 1, k)], p2 = yz[IDX3(nx, ny + 1, i, j, k + 1)], p3 = yz[IDX3(nx, ny + 1, i, j + 1, k + 1)];
                const double q0 = xz[IDX3(nx + 1, ny, i, j, k)], q1 = xz[IDX3(nx + 1, ny, i + 1, j, k)], q2 = xz[IDX3(nx + 1, ny, i, j, k + 1)], q3 = xz[IDX3(nx + 1, ny, i + 1, j, k + 1)];
                const double r0 = xy[IDX3(nx + 1, ny + 1, i, j, k)], r1 = xy[IDX3(nx + 1, ny + 1, i + 1, j, k)], r2 = xy[IDX3(nx + 1, ny + 1, i, j + 1, k)], r3 = xy[IDX3(nx + 1, ny + 1, i + 1, j + 1, k)];
                const double AII = sqrt(0.5 * (x * x + y * y + z * z) + 0.25 * (p0 * p0 + p1 * p1 + p2 * p2 + p3 * p3) + 0.25 * (q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3) +
                                        0.25 * (r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3));
                const double T = !f->T ? 0.0 : (p->T_ghosted ? f->T[IDX3(nx + 2, ny + 2, i + 1, j + 1, k + 1)] : f->T[c]);
                double e = mat_phase_viscosity(rh, f->phase_c + (size_t)rh->nphase * c, AII, T, f->P[c], tau);
                e = e * nu + f->eta[c] * (1.0 - nu);
                f->eta[c] = fmin(fmax(e, p->cutoff_lo), p->cutoff_hi);
            }
}
void orc_compute_viscosity3d(const orc_vep3d *f, const orc_rheology *rh, const orc_vep_params3d *p, double nu) { orc_compute_viscosity3d_form(f, rh, p, nu, 0); }
void orc_compute_viscosity3d_form(const orc_vep3d *f, const orc_rheology *rh, const orc_vep_params3d *p, double nu, int32_t tau)
{
    const int64_t n = p->nx * p->ny * p->nz;
    const int np = rh->nphase;
    if (mat_viscosity_reads_fields(rh)) { viscosity3d_fields(f, rh, p, nu, tau); return; }
    for (int64_t c = 0; c < n; c++) {
        const double *r = f->phase_c + (size_t)np * c;
        double e = 0.0;
        int pure = 0;
        for (int q = 0; q < np; q++)
            if (r[q] > 0.999) { e = rh->eta[q]; pure = 1; break; }
        if (!pure) {
            double s = 0.0;
            for (int q = 0; q < np; q++)
                if (r[q] != 0.0) s += inv(rh->eta[q]) * r[q];
            e = inv(s);
        }
        e = e * nu + f->eta[c] * (1.0 - nu);
        f->eta[c] = fmin(fmax(e, p->cutoff_lo), p->cutoff_hi);
    }
}

/* second_invariant_staggered on (xx, yy, zz, gather_yz, gather_xz, gather_xy) -- tensor_invariant_kernel! 3D
 * (StressKernels.jl:472-487) and accumulate_tensor_kernel! 3D (:394-408) */
static inline double sinv_stag3(const double *xx, const double *yy, const double *zz, const double *yz, const double *xz, const double *xy,
                                int64_t nx, int64_t ny, int64_t i, int64_t j, int64_t k)
{
    const size_t c = IDX3(nx, ny, i, j, k);
    double syz = 0.0, sxz = 0.0, sxy = 0.0;
    {   /* _gather_yz: center, front (j+1), top (k+1), [j+1,k+1]  (MiniKernels.jl:196-204) */
        const double a = yz[IDX3(nx, ny + 1, i, j, k)], b = yz[IDX3(nx, ny + 1, i, j + 1, k)], cc = yz[IDX3(nx, ny + 1, i, j, k + 1)], d = yz[IDX3(nx, ny + 1, i, j + 1, k + 1)];
        syz = 0.25 * (a * a + b * b + cc * cc + d * d);
    }
    {
        const double a = xz[IDX3(nx + 1, ny, i, j, k)], b = xz[IDX3(nx + 1, ny, i + 1, j, k)], cc = xz[IDX3(nx + 1, ny, i, j, k + 1)], d = xz[IDX3(nx + 1, ny, i + 1, j, k + 1)];
        sxz = 0.25 * (a * a + b * b + cc * cc + d * d);
    }
    {
        const double a = xy[IDX3(nx + 1, ny + 1, i, j, k)], b = xy[IDX3(nx + 1, ny + 1, i + 1, j, k)], cc = xy[IDX3(nx + 1, ny + 1, i, j + 1, k)], d = xy[IDX3(nx + 1, ny + 1, i + 1, j + 1, k)];
        sxy = 0.25 * (a * a + b * b + cc * cc + d * d);
    }
    return sqrt(0.5 * (xx[c] * xx[c] + yy[c] * yy[c] + zz[c] * zz[c]) + syz + sxz + sxy);
}

void orc_tensor_invariant3d(double *II, const double *xx, const double *yy, const double *zz, const double *yz, const double *xz, const double *xy,
                            int64_t nx, int64_t ny, int64_t nz)
{
    for (int64_t k = 0; k < nz; k++)
        for (int64_t j = 0; j < ny; j++)
            for (int64_t i = 0; i < nx; i++) II[IDX3(nx, ny, i, j, k)] = sinv_stag3(xx, yy, zz, yz, xz, xy, nx, ny, i, j, k);
}

/* args.T at cell c for the density laws: cell-centred (ni), or -- T_ghosted -- thermal.T (ni .+ 2) read at the cell's own [i, j, k] without a shift, as
 * getindex_NamedTuple(args, I...) does in compute_ρg! (BuoyancyForces.jl:52; miniapps/convection/RisingBlob3D/Blob3D.jl:355-356 passes thermal.T) */
static inline double T_of(const orc_vep3d *f, const orc_vep_params3d *p, size_t c)
{
    if (!f->T) return 0.0;
    if (!p->T_ghosted) return f->T[c];
    const size_t nx = (size_t)p->nx, ny = (size_t)p->ny, k = c / (nx * ny), j = (c - k * nx * ny) / nx, i = c - k * nx * ny - j * nx;
    return f->T[IDX3(nx + 2, ny + 2, i, j, k)];
}

/* shear2center_kernel! 3D (Interpolations.jl:314-323) */
void orc_shear2center3d(double *yz_c, double *xz_c, double *xy_c, const double *yz, const double *xz, const double *xy, int64_t nx, int64_t ny, int64_t nz)
{
    if (!yz_c || !xz_c || !xy_c || !yz || !xz || !xy) return;
    for (int64_t k = 0; k < nz; k++)
        for (int64_t j = 0; j < ny; j++)
            for (int64_t i = 0; i < nx; i++) {
                const size_t c = IDX3(nx, ny, i, j, k);
                yz_c[c] = 0.25 * (yz[IDX3(nx, ny + 1, i, j, k)] + yz[IDX3(nx, ny + 1, i, j + 1, k)] + yz[IDX3(nx, ny + 1, i, j, k + 1)] + yz[IDX3(nx, ny + 1, i, j + 1, k + 1)]);
                xz_c[c] = 0.25 * (xz[IDX3(nx + 1, ny, i, j, k)] + xz[IDX3(nx + 1, ny, i + 1, j, k)] + xz[IDX3(nx + 1, ny, i, j, k + 1)] + xz[IDX3(nx + 1, ny, i + 1, j, k + 1)]);
                xy_c[c] = 0.25 * (xy[IDX3(nx + 1, ny + 1, i, j, k)] + xy[IDX3(nx + 1, ny + 1, i + 1, j, k)] + xy[IDX3(nx + 1, ny + 1, i, j + 1, k)] + xy[IDX3(nx + 1, ny + 1, i + 1, j + 1, k)]);
            }
}

/* compute_vorticity!(ωyz, ωxz, ωxy, Vx, Vy, Vz, _di) as Stokes3D.jl:640-642 calls it
 * (stress_rotation_particles.jl:31-50): forward differences _d_ya(A) = (-A[i,j,k] + A[i,j+1,k])*_dy at the un-shifted
 * index of the velocity arrays (ghost layers included), kept as in the reference */
void orc_compute_vorticity3d(double *wyz, double *wxz, double *wxy, const double *Vx, const double *Vy, const double *Vz,
                             int64_t nx, int64_t ny, int64_t nz, double _dx, double _dy, double _dz)
{
#define VX(i, j, k) Vx[IDX3(nx + 1, ny + 2, i, j, k)]
#define VY(i, j, k) Vy[IDX3(nx + 2, ny + 1, i, j, k)]
#define VZ(i, j, k) Vz[IDX3(nx + 2, ny + 2, i, j, k)]
    for (int64_t k = 0; k < nz + 1; k++)
        for (int64_t j = 0; j < ny + 1; j++)
            for (int64_t i = 0; i < nx + 1; i++) {
                if (i < nx) wyz[IDX3(nx, ny + 1, i, j, k)] = 0.5 * ((-VZ(i, j, k) + VZ(i, j + 1, k)) * _dy - (-VY(i, j, k) + VY(i, j, k + 1)) * _dz);
                if (j < ny) wxz[IDX3(nx + 1, ny, i, j, k)] = 0.5 * ((-VX(i, j, k) + VX(i, j, k + 1)) * _dz - (-VZ(i, j, k) + VZ(i + 1, j, k)) * _dx);
                if (k < nz) wxy[IDX3(nx + 1, ny + 1, i, j, k)] = 0.5 * ((-VY(i, j, k) + VY(i + 1, j, k)) * _dx - (-VX(i, j, k) + VX(i, j + 1, k)) * _dy);
            }
#undef VX
#undef VY
#undef VZ
}

int32_t orc_stokes3d_vep_solve(const orc_vep3d *f, const orc_rheology *rh, const orc_vep_params3d *p, orc_result *res)
{
    const int64_t nx = p->nx, ny = p->ny, nz = p->nz;
    const size_t n = (size_t)nx * ny * nz;
    const size_t nyz = (size_t)nx * (ny + 1) * (nz + 1), nxz = (size_t)(nx + 1) * ny * (nz + 1), nxy = (size_t)(nx + 1) * (ny + 1) * nz;
    const int np = rh->nphase;
    double *etatau = malloc(n * 8), *theta = malloc(n * 8), *lam = calloc(n, 8), *Kc = malloc(n * 8), *Gc = malloc(n * 8);
    double *lamv[3] = {calloc(nyz, 8), calloc(nxz, 8), calloc(nxy, 8)};
    memcpy(f->P0, f->P, n * 8);                       /* @copy stokes.P0 stokes.P :494 */
    memcpy(theta, f->P, n * 8);                       /* θ = deepcopy(stokes.P) :495 */
    for (size_t c = 0; c < n; c++) { Kc[c] = ratio_avg(rh->Kb, f->phase_c + np * c, np); Gc[c] = ratio_avg(rh->G, f->phase_c + np * c, np); }
    /* compute_ρg!(ρg, phase_ratios, rheology, args) :505 -- the scalar gravity fills the last component (BuoyancyForces.jl:69-70) */
    const int upd_rho = rh->has_density && !mat_density_is_constant(rh);
    if (rh->has_density)
        for (size_t c = 0; c < n; c++) f->fz[c] = mat_density_ratio(rh, f->phase_c + np * c, T_of(f, p, c), f->P[c]) * rh->gravity;
    orc_compute_viscosity3d(f, rh, p, 1.0);           /* compute_viscosity!(stokes, phase_ratios, args, rheology, cutoff) :507 */
    if (p->displacement_bcs) {                        /* displacement2velocity!(stokes, dt, flow_bcs) :509 */
        const double _dt = inv(p->dt);
        for (size_t c = 0; c < (size_t)(nx + 1) * (ny + 2) * (nz + 2); c++) f->Vx[c] = f->Ux[c] * _dt;
        for (size_t c = 0; c < (size_t)(nx + 2) * (ny + 1) * (nz + 2); c++) f->Vy[c] = f->Uy[c] * _dt;
        for (size_t c = 0; c < (size_t)(nx + 2) * (ny + 2) * (nz + 1); c++) f->Vz[c] = f->Uz[c] * _dt;
    }

    orc_fields3d g;
    memset(&g, 0, sizeof(g));
    g.P = f->P; g.P0 = f->P0; g.divV = f->divV; g.Q = f->Q; g.Vx = f->Vx; g.Vy = f->Vy; g.Vz = f->Vz; g.Ux = f->Ux; g.Uy = f->Uy; g.Uz = f->Uz;
    g.txx = f->txx; g.tyy = f->tyy; g.tzz = f->tzz; g.tyz = f->tyz; g.txz = f->txz; g.txy = f->txy;
    g.exx = f->exx; g.eyy = f->eyy; g.ezz = f->ezz; g.eyz = f->eyz; g.exz = f->exz; g.exy = f->exy;
    g.eta = f->eta; g.fx = f->fx; g.fy = f->fy; g.fz = f->fz; g.RP = f->RP; g.Rx = f->Rx; g.Ry = f->Ry; g.Rz = f->Rz;
    orc_params3d q;
    memset(&q, 0, sizeof(q));
    q.nx = nx; q.ny = ny; q.nz = nz; q.nxg = p->nxg; q.nyg = p->nyg; q.nzg = p->nzg; q._dx = p->_dx; q._dy = p->_dy; q._dz = p->_dz;
    q.dt = p->dt; q.r = p->r; q.theta_dtau = p->theta_dtau; q.eta_dtau = p->eta_dtau;
    q.free_slip = p->free_slip; q.no_slip = p->no_slip; q.periodic = p->periodic;

    double err_it1 = 1.0, err = INFINITY;
    int64_t iter = 0, cont = 0;
    res->status = 0;
    while (iter < 2 || (((err / err_it1) > p->eps_rel && err > p->eps_abs) && iter <= p->iterMax)) {      /* :513 */
        orc_compute_maxloc3d(etatau, f->eta, nx, ny, nz);
        { const int64_t e[3] = {nx, ny, nz}; orc_self_halo(etatau, e, e); }                               /* update_halo!(ητ) :515 */
        orc_compute_divV3d(f->divV, f->Vx, f->Vy, f->Vz, nx, ny, nz, p->_dx, p->_dy, p->_dz);
        orc_compute_P3d(theta, f->P0, f->RP, f->divV, f->Q, etatau, Kc, Gc, (int64_t)n, p->dt, p->r, p->theta_dtau);   /* :520-533 */
        orc_compute_strain_rate3d(&g, &q);
        if (upd_rho)                                  /* update_ρg!(ρg, phase_ratios, rheology, args) :538 ; args.P is stokes.P */
            for (size_t c = 0; c < n; c++) f->fz[c] = mat_density_ratio(rh, f->phase_c + np * c, T_of(f, p, c), f->P[c]) * rh->gravity;
        orc_compute_viscosity3d_form(f, rh, p, p->viscosity_relaxation, 1);        /* update_viscosity_τII! */
        orc_vep3d_stress(f, theta, lam, lamv, rh, p);
        {   /* update_halo!(τ.yz), (τ.xz), (τ.xy) :578-580 */
            const int64_t nn[3] = {nx, ny, nz};
            const int64_t eyz[3] = {nx, ny + 1, nz + 1}, exz[3] = {nx + 1, ny, nz + 1}, exy[3] = {nx + 1, ny + 1, nz};
            orc_self_halo(f->tyz, eyz, nn); orc_self_halo(f->txz, exz, nn); orc_self_halo(f->txy, exy, nn);
        }
        orc_compute_V3d(&g, etatau, &q);
        orc_velocity2displacement3d(&g, &q);
        if (p->displacement_bcs) orc_flow_bcs3d(f->Ux, f->Uy, f->Uz, nx, ny, nz, p->free_slip, p->no_slip, p->periodic);   /* flow_bcs! on @displacement */
        else orc_flow_bcs3d(f->Vx, f->Vy, f->Vz, nx, ny, nz, p->free_slip, p->no_slip, p->periodic);
        {   /* update_halo!(@velocity(stokes)...) :596 */
            const int64_t nn[3] = {nx, ny, nz};
            const int64_t ex[3] = {nx + 1, ny + 2, nz + 2}, ey[3] = {nx + 2, ny + 1, nz + 2}, ez[3] = {nx + 2, ny + 2, nz + 1};
            orc_self_halo(f->Vx, ex, nn); orc_self_halo(f->Vy, ey, nn); orc_self_halo(f->Vz, ez, nn);
        }
        iter += 1;
        if (iter % p->nout == 0 && iter > 1) {
            double s[4];
            orc_residual_sumsq3d(&g, &q, s);
            const double den = (double)((p->nxg - 1) * (p->nyg - 1) * (p->nzg - 1));                      /* :607-612 */
            const double nRx = sqrt(s[0]) / den, nRy = sqrt(s[1]) / den, nRz = sqrt(s[2]) / den;
            const double nDV = sqrt(s[3]) / (double)n;                                                   /* norm_mpi(RP)/length(RP) :614 */
            err = fmax(fmax(nRx, nRy), fmax(nRz, nDV));
            if (isnan(nRx) || isnan(nRy) || isnan(nRz) || isnan(nDV)) err = NAN;
            if (cont < res->cap) {
                res->norm_Rx[cont] = nRx; res->norm_Ry[cont] = nRy; res->norm_Rz[cont] = nRz; res->norm_divV[cont] = nDV;
                res->err_evo1[cont] = err; res->err_evo2[cont] = iter;
            }
            if (cont == 0) err_it1 = err;
            cont++;
            if (isnan(err)) { res->status = 1; break; }
        }
    }
    res->iter = iter;
    res->nchecks = cont < res->cap ? cont : res->cap;
    if (res->status == 0) {
        if (f->omega_yz && f->omega_xz && f->omega_xy)
            orc_compute_vorticity3d(f->omega_yz, f->omega_xz, f->omega_xy, f->Vx, f->Vy, f->Vz, nx, ny, nz, p->_dx, p->_dy, p->_dz);
        orc_shear2center3d(f->eyz_c, f->exz_c, f->exy_c, f->eyz, f->exz, f->exy, nx, ny, nz);
        orc_shear2center3d(f->eplyz_c, f->eplxz_c, f->eplxy_c, f->eplyz, f->eplxz, f->eplxy, nx, ny, nz);
        orc_shear2center3d(f->deyz_c, f->dexz_c, f->dexy_c, f->deyz, f->dexz, f->dexy, nx, ny, nz);
        for (int64_t k = 0; k < nz; k++)              /* accumulate_tensor! / accumulate_vol! :654-655 */
            for (int64_t j = 0; j < ny; j++)
                for (int64_t i = 0; i < nx; i++) {
                    const size_t c = IDX3(nx, ny, i, j, k);
                    f->EII_pl[c] += sinv_stag3(f->eplxx, f->eplyy, f->eplzz, f->eplyz, f->eplxz, f->eplxy, nx, ny, i, j, k) * p->dt;
                    f->EVol_pl[c] += p->dt * f->evol_pl[c];
                }
        memcpy(f->toxx, f->txx, n * 8); memcpy(f->toyy, f->tyy, n * 8); memcpy(f->tozz, f->tzz, n * 8);      /* multi_copy! :657-658 */
        memcpy(f->toyz, f->tyz, nyz * 8); memcpy(f->toxz, f->txz, nxz * 8); memcpy(f->toxy, f->txy, nxy * 8);
        memcpy(f->toyz_c, f->tyz_c, n * 8); memcpy(f->toxz_c, f->txz_c, n * 8); memcpy(f->toxy_c, f->txy_c, n * 8);
    }
    free(etatau); free(theta); free(lam); free(Kc); free(Gc);
    for (int t = 0; t < 3; t++) free(lamv[t]);
    return res->status;
}
