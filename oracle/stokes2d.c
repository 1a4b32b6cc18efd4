/* oracle/stokes2d.c -- TEST INFRASTRUCTURE ONLY (see jrx_oracle.h).
 * CPU restatement of the 2D visco-elastic pseudo-transient Stokes path of JustRelax.jl
 * (src/stokes/Stokes2D.jl:181-325, the variant SolCx / SolKz / elastic build-up run). */
#include "jrx_oracle.h"
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include <omp.h>

/* extents: Vx (nx+1, ny+2)  Vy (nx+2, ny+1)  τxy/εxy (nx+1, ny+1)  Rx (nx-1, ny)  Ry (nx, ny-1) */
#define VX(i, j) Vx[IDX2(nx + 1, i, j)]
#define VY(i, j) Vy[IDX2(nx + 2, i, j)]
#define C(A, i, j) (A)[IDX2(nx, i, j)]
#define XY(A, i, j) (A)[IDX2(nx + 1, i, j)]

/* src/stokes/VelocityKernels.jl:3-6 + MiniKernels.jl:46-51,57-58 */
void orc_compute_divV2d(double *divV, const double *Vx, const double *Vy, int64_t nx, int64_t ny,
                        double _dx, double _dy)
{
    orc_compute_divV2d_sp(divV, Vx, Vy, nx, ny, _dx, _dy, NULL);
}
/* the same on a non-uniform grid: compute_∇V!(∇V, V, _di.vertex) (Stokes2D.jl:229); sp = the six inverse-spacing arrays of orc_params2d or NULL */
#define SPC(arr, idx, uni) ((arr) ? (arr)[idx] : (uni))
void orc_compute_divV2d_sp(double *divV, const double *Vx, const double *Vy, int64_t nx, int64_t ny, double _dx, double _dy, const double *const *sp)
{
    const double *vx = sp ? sp[0] : NULL, *vy = sp ? sp[1] : NULL;
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < ny; j++)
        for (int64_t i = 0; i < nx; i++)
            divV[IDX2(nx, i, j)] = (-VX(i, j + 1) + VX(i + 1, j + 1)) * SPC(vx, i, _dx) + (-VY(i + 1, j) + VY(i + 1, j + 1)) * SPC(vy, j, _dy);
}

/* src/stokes/VelocityKernels.jl:10-44 ; launch box ni.+1 */
void orc_compute_strain_rate2d(const orc_fields2d *f, const orc_params2d *p)
{
    const int64_t nx = p->nx, ny = p->ny;
    const double _dx = p->_dx, _dy = p->_dy;
    const double *const *sp = p->inv_spacing;       /* _di.vertex for the normal components, _di.velocity[1][2] / [2][1] for the shear one (VelocityKernels.jl:22,33-34) */
    const double *Vx = f->Vx, *Vy = f->Vy;
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < ny + 1; j++)
        for (int64_t i = 0; i < nx + 1; i++) {
            if (i < nx && j < ny) {
                double d3 = C(f->divV, i, j) * inv(3.0);
                C(f->exx, i, j) = (-VX(i, j + 1) + VX(i + 1, j + 1)) * SPC(sp[0], i, _dx) - d3;
                C(f->eyy, i, j) = (-VY(i + 1, j) + VY(i + 1, j + 1)) * SPC(sp[1], j, _dy) - d3;
            }
            XY(f->exy, i, j) = 0.5 * (SPC(sp[4], j, _dy) * (VX(i, j + 1) - VX(i, j)) + SPC(sp[5], i, _dx) * (VY(i + 1, j) - VY(i, j)));
        }
}

/* src/MiniKernels.jl:76-80 */
static inline double av_clamped2(const double *A, int64_t nx, int64_t ny, int64_t i, int64_t j)
{
    int64_t i0 = clampi(i - 1, 0, nx - 1), i1 = clampi(i, 0, nx - 1);
    int64_t j0 = clampi(j - 1, 0, ny - 1), j1 = clampi(j, 0, ny - 1);
    return 0.25 * (A[IDX2(nx, i0, j0)] + A[IDX2(nx, i1, j0)] + A[IDX2(nx, i0, j1)] + A[IDX2(nx, i1, j1)]);
}

/* src/stokes/StressKernels.jl:63-91 (visco-elastic form) ; launch box ni.+1 */
void orc_compute_tau2d(const orc_fields2d *f, const orc_params2d *p)
{
    const int64_t nx = p->nx, ny = p->ny;
    const double dt = p->dt, th = p->theta_dtau;
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < ny + 1; j++)
        for (int64_t i = 0; i < nx + 1; i++) {
            if (i < nx && j < ny) {
                size_t c = IDX2(nx, i, j);
                double _Gdt = inv(f->G[c] * dt);
                double e = f->eta[c];
                double dtr = compute_dtau_r(th, e, _Gdt);
                f->txx[c] += stress_increment(f->txx[c], f->toxx[c], e, f->exx[c], _Gdt, dtr);
                f->tyy[c] += stress_increment(f->tyy[c], f->toyy[c], e, f->eyy[c], _Gdt, dtr);
            }
            size_t v = IDX2(nx + 1, i, j);
            double e = av_clamped2(f->eta, nx, ny, i, j);
            double _Gdt = inv(av_clamped2(f->G, nx, ny, i, j) * dt);
            double dtr = compute_dtau_r(th, e, _Gdt);
            f->txy[v] += stress_increment(f->txy[v], f->toxy[v], e, f->exy[v], _Gdt, dtr);
        }
}

/* src/stokes/VelocityKernels.jl:108-131 */
void orc_compute_V2d(const orc_fields2d *f, const double *etatau, const orc_params2d *p) { orc_compute_V2d_fs(f, etatau, p, 0.0); }

/* free-surface form (VelocityKernels.jl:134-180): fs_dt = dt * free_surface; the vertical momentum gets Vy ∂(ρg_y)/∂y θ dt (θ = 1) */
void orc_compute_V2d_fs(const orc_fields2d *f, const double *etatau, const orc_params2d *p, double fs_dt)
{
    const int64_t nx = p->nx, ny = p->ny;
    const double edt = p->eta_dtau;
    const double *const *sp = p->inv_spacing;
    double *Vx = f->Vx, *Vy = f->Vy;
    const double *P = f->P;
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < ny; j++)
        for (int64_t i = 0; i < nx; i++) {
            if (i < nx - 1) { /* all((i,j) .< size(Vx) .- 1) : i <= nx-1, j <= ny (1-based) */
                const double _dx = SPC(sp[2], i, p->_dx), _dy = SPC(sp[1], j, p->_dy);      /* _dx_c, _dy_v (VelocityKernels.jl:115-116) */
                double r = -((-C(P, i, j) + C(P, i + 1, j)) * _dx) + (-C(f->txx, i, j) + C(f->txx, i + 1, j)) * _dx +
                           (-XY(f->txy, i + 1, j) + XY(f->txy, i + 1, j + 1)) * _dy -
                           (C(f->fx, i, j) + C(f->fx, i + 1, j)) * 0.5;
                VX(i + 1, j + 1) += r * edt / ((C(etatau, i, j) + C(etatau, i + 1, j)) * 0.5);
            }
            if (j < ny - 1) {
                const double _dx = SPC(sp[0], i, p->_dx), _dy = SPC(sp[3], j, p->_dy);      /* _dx_v, _dy_c (:123-124) */
                double r = -((-C(P, i, j) + C(P, i, j + 1)) * _dy) + (-C(f->tyy, i, j) + C(f->tyy, i, j + 1)) * _dy +
                           (-XY(f->txy, i, j + 1) + XY(f->txy, i + 1, j + 1)) * _dx -
                           (C(f->fy, i, j) + C(f->fy, i, j + 1)) * 0.5;
                if (fs_dt != 0.0) {
                    const int64_t jN = j + 1 < ny - 1 ? j + 1 : ny - 1;            /* j_N = min(j + 1, ny) */
                    const double drg = (C(f->fy, i, jN) - C(f->fy, i, j)) * _dy;
                    r += VY(i + 1, j + 1) * drg * 1.0 * fs_dt;
                }
                VY(i + 1, j + 1) += r * edt / ((C(etatau, i, j) + C(etatau, i, j + 1)) * 0.5);
            }
        }
}

/* src/stokes/VelocityKernels.jl:246-269 */
void orc_compute_Res2d(const orc_fields2d *f, const orc_params2d *p) { orc_compute_Res2d_fs(f, p, 0.0); }

/* free-surface form (VelocityKernels.jl:271-307) */
void orc_compute_Res2d_fs(const orc_fields2d *f, const orc_params2d *p, double fs_dt)
{
    const int64_t nx = p->nx, ny = p->ny;
    const double *const *sp = p->inv_spacing;
    const double *P = f->P;
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < ny; j++)
        for (int64_t i = 0; i < nx; i++) {
            if (i < nx - 1) {
                const double _dx = SPC(sp[2], i, p->_dx), _dy = SPC(sp[1], j, p->_dy);
                f->Rx[IDX2(nx - 1, i, j)] = (-C(f->txx, i, j) + C(f->txx, i + 1, j)) * _dx +
                                            (-XY(f->txy, i + 1, j) + XY(f->txy, i + 1, j + 1)) * _dy -
                                            (-C(P, i, j) + C(P, i + 1, j)) * _dx - (C(f->fx, i, j) + C(f->fx, i + 1, j)) * 0.5;
            }
            if (j < ny - 1) {
                const double _dx = SPC(sp[0], i, p->_dx), _dy = SPC(sp[3], j, p->_dy);
                double r = (-C(f->tyy, i, j) + C(f->tyy, i, j + 1)) * _dy +
                           (-XY(f->txy, i, j + 1) + XY(f->txy, i + 1, j + 1)) * _dx -
                           (-C(P, i, j) + C(P, i, j + 1)) * _dy - (C(f->fy, i, j) + C(f->fy, i, j + 1)) * 0.5;
                if (fs_dt != 0.0) {
                    const int64_t jN = j + 1 < ny - 1 ? j + 1 : ny - 1;
                    const double drg = (C(f->fy, i, jN) - C(f->fy, i, j)) * _dy;
                    r += (f->Vy[IDX2(nx + 2, i + 1, j + 1)] * drg) * 1.0 * fs_dt;
                }
                f->Ry[IDX2(nx, i, j)] = r;
            }
        }
}

/* src/types/displacement.jl:17-28 */
void orc_velocity2displacement2d(const orc_fields2d *f, const orc_params2d *p)
{
    const size_t n1 = (size_t)(p->nx + 1) * (p->ny + 2), n2 = (size_t)(p->nx + 2) * (p->ny + 1);
    for (size_t c = 0; c < n1; c++) f->Ux[c] = f->Vx[c] * p->dt;
    for (size_t c = 0; c < n2; c++) f->Uy[c] = f->Vy[c] * p->dt;
}

/* BoundaryConditions.jl:86-100 ; no_slip.jl:1-18 ; free_slip.jl:1-13 ; periodic.jl:15-36.
 * 2D face naming: bot <-> j=1, top <-> j=end. */
void orc_flow_bcs2d(double *Vx, double *Vy, int64_t nx, int64_t ny,
                    uint32_t free_slip, uint32_t no_slip, uint32_t periodic)
{
    const int64_t x1 = nx + 1, x2 = nx + 2, y1 = ny + 1, y2 = ny + 2;
    if (no_slip) {
        if (no_slip & F_LEFT) {
            for (int64_t j = 0; j < y2; j++) VX(0, j) = 0.0;
            for (int64_t j = 0; j < y1; j++) VY(0, j) = -VY(1, j);
        }
        if (no_slip & F_RIGHT) {
            for (int64_t j = 0; j < y2; j++) VX(x1 - 1, j) = 0.0;
            for (int64_t j = 0; j < y1; j++) VY(x2 - 1, j) = -VY(x2 - 2, j);
        }
        if (no_slip & F_BOT) {
            for (int64_t i = 0; i < x1; i++) VX(i, 0) = -VX(i, 1);
            for (int64_t i = 0; i < x2; i++) VY(i, 0) = 0.0;
        }
        if (no_slip & F_TOP) {
            for (int64_t i = 0; i < x1; i++) VX(i, y2 - 1) = -VX(i, y2 - 2);
            for (int64_t i = 0; i < x2; i++) VY(i, y1 - 1) = 0.0;
        }
    }
    if (free_slip) {
        for (int64_t i = 0; i < x1; i++) {
            if (free_slip & F_BOT) VX(i, 0) = VX(i, 1);
            if (free_slip & F_TOP) VX(i, y2 - 1) = VX(i, y2 - 2);
        }
        for (int64_t j = 0; j < y1; j++) {
            if (free_slip & F_LEFT) VY(0, j) = VY(1, j);
            if (free_slip & F_RIGHT) VY(x2 - 1, j) = VY(x2 - 2, j);
        }
    }
    if (periodic) {
        for (int64_t j = 0; j < y2; j++)
            if (periodic & F_LEFT) VX(0, j) = VX(x1 - 1, j);
        for (int64_t j = 0; j < y1; j++) {
            if (periodic & F_LEFT) VY(0, j) = VY(x2 - 2, j);
            if (periodic & F_RIGHT) VY(x2 - 1, j) = VY(1, j);
        }
        for (int64_t i = 0; i < x1; i++) {
            if (periodic & F_BOT) VX(i, 0) = VX(i, y2 - 2);
            if (periodic & F_TOP) VX(i, y2 - 1) = VX(i, 1);
        }
        for (int64_t i = 0; i < x2; i++)
            if (periodic & F_BOT) VY(i, 0) = VY(i, y1 - 1);
    }
}

static double sumsq_inner2(const double *A, int64_t n1, int64_t n2)
{
    double s = 0.0;
    for (int64_t j = 1; j < n2 - 1; j++)
        for (int64_t i = 1; i < n1 - 1; i++) s += A[IDX2(n1, i, j)] * A[IDX2(n1, i, j)];
    return s;
}

/* src/stokes/Stokes2D.jl:278-284 (local Σx² parts) */
void orc_residual_sumsq2d(const orc_fields2d *f, const orc_params2d *p, double out[3])
{
    out[0] = sumsq_inner2(f->Rx, p->nx - 1, p->ny);
    out[1] = sumsq_inner2(f->Ry, p->nx, p->ny - 1);
    double s = 0.0;
    for (int64_t c = 0; c < p->nx * p->ny; c++) s += f->RP[c] * f->RP[c];
    out[2] = s;
}

/* src/stokes/Stokes2D.jl:229-269 ; note compute_P! receives ητ here (App. C #3) */
void orc_stokes2d_iteration(const orc_fields2d *f, const double *etatau, const orc_params2d *p)
{
    orc_compute_divV2d_sp(f->divV, f->Vx, f->Vy, p->nx, p->ny, p->_dx, p->_dy, p->inv_spacing);
    orc_compute_P3d(f->P, f->P0, f->RP, f->divV, f->Q, etatau, f->K, f->G, p->nx * p->ny, p->dt, p->r, p->theta_dtau);
    orc_compute_strain_rate2d(f, p);
    orc_compute_tau2d(f, p);
    orc_compute_V2d(f, etatau, p);
    orc_velocity2displacement2d(f, p);
    if (p->displacement_bcs) orc_flow_bcs2d(f->Ux, f->Uy, p->nx, p->ny, p->free_slip, p->no_slip, p->periodic);
    else orc_flow_bcs2d(f->Vx, f->Vy, p->nx, p->ny, p->free_slip, p->no_slip, p->periodic);
}

/* src/stokes/Stokes2D.jl:181-325 */
int32_t orc_stokes2d_solve(const orc_fields2d *f, const orc_params2d *p, orc_result *res)
{
    const int64_t nx = p->nx, ny = p->ny;
    const size_t n = (size_t)nx * ny;
    double *etatau = (double *)malloc(n * sizeof(double));
    orc_compute_maxloc2d(etatau, f->eta, nx, ny);   /* :208-209 */
    if (p->displacement_bcs) {                      /* displacement2velocity!(stokes, dt, flow_bcs) :223 */
        const double _dt = 1.0 / p->dt;
        for (size_t c = 0; c < (size_t)(nx + 1) * (ny + 2); c++) f->Vx[c] = f->Ux[c] * _dt;
        for (size_t c = 0; c < (size_t)(nx + 2) * (ny + 1); c++) f->Vy[c] = f->Uy[c] * _dt;
    }
    double err_it1 = 1.0, err = 1.0;
    int64_t iter = 0, cont = 0;
    res->status = 0;
    double t0 = omp_get_wtime();
    while (iter < 2 || (((err / err_it1) > p->eps_rel && err > p->eps_abs) && iter <= p->iterMax)) {
        orc_stokes2d_iteration(f, etatau, p);
        iter += 1;
        if (iter % p->nout == 0 && iter > 1) {
            orc_compute_Res2d(f, p);
            double s[3];
            orc_residual_sumsq2d(f, p, s);
            double nRx = sqrt(s[0]) / sqrt((double)((p->nxg - 2) * (p->nyg - 1)));
            double nRy = sqrt(s[1]) / sqrt((double)((p->nxg - 1) * (p->nyg - 2)));
            double nDV = sqrt(s[2]) / sqrt((double)(p->nxg * p->nyg));
            err = fmax(nRx, fmax(nRy, nDV));
            if (isnan(nRx) || isnan(nRy) || isnan(nDV)) err = NAN;
            if (cont < res->cap) {
                res->norm_Rx[cont] = nRx; res->norm_Ry[cont] = nRy; res->norm_divV[cont] = nDV;
                res->err_evo1[cont] = err; res->err_evo2[cont] = iter;
            }
            if (cont == 0) err_it1 = err;
            cont += 1;
        }
    }
    res->time_s = omp_get_wtime() - t0;
    res->iter = iter;
    res->nchecks = cont < res->cap ? cont : res->cap;
    memcpy(f->toxx, f->txx, n * sizeof(double));
    memcpy(f->toyy, f->tyy, n * sizeof(double));
    memcpy(f->toxy, f->txy, (size_t)(nx + 1) * (ny + 1) * sizeof(double));
    if (f->txy_c && f->toxy_c) memcpy(f->toxy_c, f->txy_c, n * sizeof(double));
    free(etatau);
    return 0;
}
