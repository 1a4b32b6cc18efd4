/* oracle/stokes3d.c -- TEST INFRASTRUCTURE ONLY (see jrx_oracle.h).
 * CPU restatement of the 3D isoviscous visco-elastic pseudo-transient Stokes path of
 * JustRelax.jl (src/stokes/Stokes3D.jl:25-186) -- one loop nest per reference kernel. */
#include "jrx_oracle.h"
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include <omp.h>

int orc_num_threads(void) { return omp_get_max_threads(); }
void orc_set_num_threads(int n) { if (n > 0) omp_set_num_threads(n); }

/* NUMA placement for the timed CPU baseline (bench.py cpu_baseline): copies `nslab` contiguous slabs of `slab` doubles with the static schedule over the
 * slowest index that every kernel of this file uses, so that a freshly allocated `dst` is first touched -- and therefore placed -- by the thread that
 * will work on it.  Values are unchanged. */
void orc_first_touch_copy(double *dst, const double *src, int64_t slab, int64_t nslab)
{
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < nslab; c++) memcpy(dst + (size_t)c * (size_t)slab, src + (size_t)c * (size_t)slab, (size_t)slab * sizeof(double));
}

/* array extents (src/types/constructors/stokes.jl:27-34,196-212,241-247) */
#define NVX1 (nx + 1)
#define NVX2 (ny + 2)
#define NVY1 (nx + 2)
#define NVY2 (ny + 1)
#define NVZ1 (nx + 2)
#define NVZ2 (ny + 2)

/* src/stokes/VelocityKernels.jl:3-6 + MiniKernels.jl:53-55,104-105 ; launch box ni */
void orc_compute_divV3d(double *divV, const double *Vx, const double *Vy, const double *Vz,
                        int64_t nx, int64_t ny, int64_t nz, double _dx, double _dy, double _dz)
{
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < nz; k++)
        for (int64_t j = 0; j < ny; j++)
            for (int64_t i = 0; i < nx; i++) {
                double dxi = (-Vx[IDX3(NVX1, NVX2, i, j + 1, k + 1)] + Vx[IDX3(NVX1, NVX2, i + 1, j + 1, k + 1)]) * _dx;
                double dyi = (-Vy[IDX3(NVY1, NVY2, i + 1, j, k + 1)] + Vy[IDX3(NVY1, NVY2, i + 1, j + 1, k + 1)]) * _dy;
                double dzi = (-Vz[IDX3(NVZ1, NVZ2, i + 1, j + 1, k)] + Vz[IDX3(NVZ1, NVZ2, i + 1, j + 1, k + 1)]) * _dz;
                divV[IDX3(nx, ny, i, j, k)] = dxi + dyi + dzi;
            }
}

/* src/stokes/PressureKernels.jl:10-15 (compressible array form) + :186-195 ; muladd -> fma */
void orc_compute_P3d(double *P, const double *P0, double *RP, const double *divV, const double *Q,
                     const double *eta, const double *K, const double *G, int64_t n,
                     double dt, double r, double theta_dtau)
{
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < n; c++) {
        double _Kdt = inv(K[c] * dt);
        double _Gdt = inv(G[c] * dt);
        double _dt = inv(dt);
        double rhs = -divV[c] + (Q[c] * _dt);
        double Pc = P[c];
        RP[c] = fma(-(Pc - P0[c]), _Kdt, rhs);
        double psi = inv(inv(eta[c]) + _Gdt) * r / theta_dtau;
        P[c] = (fma(P0[c], _Kdt, rhs) * psi + Pc) / (1.0 + _Kdt * psi);
    }
}

/* src/stokes/VelocityKernels.jl:59-104 ; launch box ni.+1, each block guarded by size(ε··) */
void orc_compute_strain_rate3d(const orc_fields3d *f, const orc_params3d *p)
{
    const int64_t nx = p->nx, ny = p->ny, nz = p->nz;
    const double _dx = p->_dx, _dy = p->_dy, _dz = p->_dz;
    const double *Vx = f->Vx, *Vy = f->Vy, *Vz = f->Vz;
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < nz + 1; k++)
        for (int64_t j = 0; j < ny + 1; j++)
            for (int64_t i = 0; i < nx + 1; i++) {
                if (i < nx && j < ny && k < nz) {
                    double d3 = f->divV[IDX3(nx, ny, i, j, k)] * inv(3.0);
                    double dxi = (-Vx[IDX3(NVX1, NVX2, i, j + 1, k + 1)] + Vx[IDX3(NVX1, NVX2, i + 1, j + 1, k + 1)]) * _dx;
                    double dyi = (-Vy[IDX3(NVY1, NVY2, i + 1, j, k + 1)] + Vy[IDX3(NVY1, NVY2, i + 1, j + 1, k + 1)]) * _dy;
                    double dzi = (-Vz[IDX3(NVZ1, NVZ2, i + 1, j + 1, k)] + Vz[IDX3(NVZ1, NVZ2, i + 1, j + 1, k + 1)]) * _dz;
                    f->exx[IDX3(nx, ny, i, j, k)] = dxi - d3;
                    f->eyy[IDX3(nx, ny, i, j, k)] = dyi - d3;
                    f->ezz[IDX3(nx, ny, i, j, k)] = dzi - d3;
                }
                /* εyz : (nx, ny+1, nz+1) */
                if (i < nx) {
                    f->eyz[IDX3(nx, ny + 1, i, j, k)] =
                        0.5 * (_dz * (Vy[IDX3(NVY1, NVY2, i + 1, j, k + 1)] - Vy[IDX3(NVY1, NVY2, i + 1, j, k)]) +
                               _dy * (Vz[IDX3(NVZ1, NVZ2, i + 1, j + 1, k)] - Vz[IDX3(NVZ1, NVZ2, i + 1, j, k)]));
                }
                /* εxz : (nx+1, ny, nz+1) */
                if (j < ny) {
                    f->exz[IDX3(nx + 1, ny, i, j, k)] =
                        0.5 * (_dz * (Vx[IDX3(NVX1, NVX2, i, j + 1, k + 1)] - Vx[IDX3(NVX1, NVX2, i, j + 1, k)]) +
                               _dx * (Vz[IDX3(NVZ1, NVZ2, i + 1, j + 1, k)] - Vz[IDX3(NVZ1, NVZ2, i, j + 1, k)]));
                }
                /* εxy : (nx+1, ny+1, nz) */
                if (k < nz) {
                    f->exy[IDX3(nx + 1, ny + 1, i, j, k)] =
                        0.5 * (_dy * (Vx[IDX3(NVX1, NVX2, i, j + 1, k + 1)] - Vx[IDX3(NVX1, NVX2, i, j, k + 1)]) +
                               _dx * (Vy[IDX3(NVY1, NVY2, i + 1, j, k + 1)] - Vy[IDX3(NVY1, NVY2, i, j, k + 1)]));
                }
            }
}

/* 4-cell clamped arithmetic means of a centre array A(nx,ny,nz) at a shear node
 * (src/MiniKernels.jl:133-147).  0-based: cells {i-1,i} x {j-1,j} clamped into range. */
static inline double av_xy_c(const double *A, int64_t nx, int64_t ny, int64_t i, int64_t j, int64_t k)
{
    int64_t i0 = clampi(i - 1, 0, nx - 1), i1 = clampi(i, 0, nx - 1);
    int64_t j0 = clampi(j - 1, 0, ny - 1), j1 = clampi(j, 0, ny - 1);
    return 0.25 * (A[IDX3(nx, ny, i0, j0, k)] + A[IDX3(nx, ny, i1, j0, k)] + A[IDX3(nx, ny, i0, j1, k)] + A[IDX3(nx, ny, i1, j1, k)]);
}
static inline double av_xz_c(const double *A, int64_t nx, int64_t ny, int64_t nz, int64_t i, int64_t j, int64_t k)
{
    int64_t i0 = clampi(i - 1, 0, nx - 1), i1 = clampi(i, 0, nx - 1);
    int64_t k0 = clampi(k - 1, 0, nz - 1), k1 = clampi(k, 0, nz - 1);
    return 0.25 * (A[IDX3(nx, ny, i0, j, k0)] + A[IDX3(nx, ny, i1, j, k0)] + A[IDX3(nx, ny, i0, j, k1)] + A[IDX3(nx, ny, i1, j, k1)]);
}
static inline double av_yz_c(const double *A, int64_t nx, int64_t ny, int64_t nz, int64_t i, int64_t j, int64_t k)
{
    int64_t j0 = clampi(j - 1, 0, ny - 1), j1 = clampi(j, 0, ny - 1);
    int64_t k0 = clampi(k - 1, 0, nz - 1), k1 = clampi(k, 0, nz - 1);
    return 0.25 * (A[IDX3(nx, ny, i, j0, k0)] + A[IDX3(nx, ny, i, j1, k0)] + A[IDX3(nx, ny, i, j0, k1)] + A[IDX3(nx, ny, i, j1, k1)]);
}

/* src/stokes/StressKernels.jl:149-230 ; launch box ni.+1 */
void orc_compute_tau3d(const orc_fields3d *f, const orc_params3d *p)
{
    const int64_t nx = p->nx, ny = p->ny, nz = p->nz;
    const double dt = p->dt, th = p->theta_dtau;
    const double *eta = f->eta, *G = f->G;
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < nz + 1; k++)
        for (int64_t j = 0; j < ny + 1; j++)
            for (int64_t i = 0; i < nx + 1; i++) {
                if (i < nx && j < ny && k < nz) {
                    size_t c = IDX3(nx, ny, i, j, k);
                    double _Gdt = inv(G[c] * dt);
                    double e = eta[c];
                    double dtr = compute_dtau_r(th, e, _Gdt);
                    f->txx[c] += stress_increment(f->txx[c], f->toxx[c], e, f->exx[c], _Gdt, dtr);
                    f->tyy[c] += stress_increment(f->tyy[c], f->toyy[c], e, f->eyy[c], _Gdt, dtr);
                    f->tzz[c] += stress_increment(f->tzz[c], f->tozz[c], e, f->ezz[c], _Gdt, dtr);
                }
                if (k < nz) { /* τxy (nx+1, ny+1, nz) */
                    size_t c = IDX3(nx + 1, ny + 1, i, j, k);
                    double e = av_xy_c(eta, nx, ny, i, j, k);
                    double _Gdt = inv(av_xy_c(G, nx, ny, i, j, k) * dt);
                    double dtr = compute_dtau_r(th, e, _Gdt);
                    f->txy[c] += stress_increment(f->txy[c], f->toxy[c], e, f->exy[c], _Gdt, dtr);
                }
                if (j < ny) { /* τxz (nx+1, ny, nz+1) */
                    size_t c = IDX3(nx + 1, ny, i, j, k);
                    double e = av_xz_c(eta, nx, ny, nz, i, j, k);
                    double _Gdt = inv(av_xz_c(G, nx, ny, nz, i, j, k) * dt);
                    double dtr = compute_dtau_r(th, e, _Gdt);
                    f->txz[c] += stress_increment(f->txz[c], f->toxz[c], e, f->exz[c], _Gdt, dtr);
                }
                if (i < nx) { /* τyz (nx, ny+1, nz+1) */
                    size_t c = IDX3(nx, ny + 1, i, j, k);
                    double e = av_yz_c(eta, nx, ny, nz, i, j, k);
                    double _Gdt = inv(av_yz_c(G, nx, ny, nz, i, j, k) * dt);
                    double dtr = compute_dtau_r(th, e, _Gdt);
                    f->tyz[c] += stress_increment(f->tyz[c], f->toyz[c], e, f->eyz[c], _Gdt, dtr);
                }
            }
}

/* src/stokes/VelocityKernels.jl:182-242 ; launch box inferred (nx+2,ny+2,nz+2), guards size(R·) */
void orc_compute_V3d(const orc_fields3d *f, const double *etatau, const orc_params3d *p)
{
    const int64_t nx = p->nx, ny = p->ny, nz = p->nz;
    const double _dx = p->_dx, _dy = p->_dy, _dz = p->_dz, edt = p->eta_dtau;
    const double *P = f->P;
#define C(A, i, j, k) (A)[IDX3(nx, ny, i, j, k)]
#define TXY(i, j, k) f->txy[IDX3(nx + 1, ny + 1, i, j, k)]
#define TXZ(i, j, k) f->txz[IDX3(nx + 1, ny, i, j, k)]
#define TYZ(i, j, k) f->tyz[IDX3(nx, ny + 1, i, j, k)]
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < nz; k++)
        for (int64_t j = 0; j < ny; j++)
            for (int64_t i = 0; i < nx; i++) {
                if (i < nx - 1) {
                    double R = (-C(f->txx, i, j, k) + C(f->txx, i + 1, j, k)) * _dx +
                               _dy * (TXY(i + 1, j + 1, k) - TXY(i + 1, j, k)) +
                               _dz * (TXZ(i + 1, j, k + 1) - TXZ(i + 1, j, k)) -
                               (-C(P, i, j, k) + C(P, i + 1, j, k)) * _dx -
                               0.5 * (C(f->fx, i, j, k) + C(f->fx, i + 1, j, k));
                    f->Rx[IDX3(nx - 1, ny, i, j, k)] = R;
                    f->Vx[IDX3(NVX1, NVX2, i + 1, j + 1, k + 1)] += R * edt / (0.5 * (C(etatau, i, j, k) + C(etatau, i + 1, j, k)));
                }
                if (j < ny - 1) {
                    double R = _dx * (TXY(i + 1, j + 1, k) - TXY(i, j + 1, k)) +
                               _dy * (C(f->tyy, i, j + 1, k) - C(f->tyy, i, j, k)) +
                               _dz * (TYZ(i, j + 1, k + 1) - TYZ(i, j + 1, k)) -
                               (-C(P, i, j, k) + C(P, i, j + 1, k)) * _dy -
                               0.5 * (C(f->fy, i, j, k) + C(f->fy, i, j + 1, k));
                    f->Ry[IDX3(nx, ny - 1, i, j, k)] = R;
                    f->Vy[IDX3(NVY1, NVY2, i + 1, j + 1, k + 1)] += R * edt / (0.5 * (C(etatau, i, j, k) + C(etatau, i, j + 1, k)));
                }
                if (k < nz - 1) {
                    double R = _dx * (TXZ(i + 1, j, k + 1) - TXZ(i, j, k + 1)) +
                               _dy * (TYZ(i, j + 1, k + 1) - TYZ(i, j, k + 1)) +
                               (-C(f->tzz, i, j, k) + C(f->tzz, i, j, k + 1)) * _dz -
                               (-C(P, i, j, k) + C(P, i, j, k + 1)) * _dz -
                               0.5 * (C(f->fz, i, j, k) + C(f->fz, i, j, k + 1));
                    f->Rz[IDX3(nx, ny, i, j, k)] = R;
                    f->Vz[IDX3(NVZ1, NVZ2, i + 1, j + 1, k + 1)] += R * edt / (0.5 * (C(etatau, i, j, k) + C(etatau, i, j, k + 1)));
                }
            }
#undef C
}

/* src/types/displacement.jl:8-28 ; U = V*dt over each array's own extent */
void orc_velocity2displacement3d(const orc_fields3d *f, const orc_params3d *p)
{
    const int64_t nx = p->nx, ny = p->ny, nz = p->nz;
    const size_t n1 = (size_t)(nx + 1) * (ny + 2) * (nz + 2), n2 = (size_t)(nx + 2) * (ny + 1) * (nz + 2),
                 n3 = (size_t)(nx + 2) * (ny + 2) * (nz + 1);
#pragma omp parallel for schedule(static)
    for (size_t c = 0; c < n1; c++) f->Ux[c] = f->Vx[c] * p->dt;
#pragma omp parallel for schedule(static)
    for (size_t c = 0; c < n2; c++) f->Uy[c] = f->Vy[c] * p->dt;
#pragma omp parallel for schedule(static)
    for (size_t c = 0; c < n3; c++) f->Uz[c] = f->Vz[c] * p->dt;
}

/* Boundary conditions.  Reference: src/boundaryconditions/BoundaryConditions.jl:86-100 (order:
 * no_slip, free_slip, periodic), no_slip.jl:20-54, free_slip.jl:15-70, periodic.jl:56-98.
 * The reference's free_slip!/periodic_boundary! kernels are one racy launch (edge ghosts are
 * written by several branches); here each face group runs to completion in source order, which
 * pins the edge/corner ghosts (never read by any stencil) to a deterministic value.
 * NOTE the reference's face naming in 3D: free_slip `top` <-> k=1, `bot` <-> k=end;
 * no_slip `bot` <-> k=1, `top` <-> k=end. */
#define VX(i, j, k) Vx[IDX3(nx + 1, ny + 2, i, j, k)]
#define VY(i, j, k) Vy[IDX3(nx + 2, ny + 1, i, j, k)]
#define VZ(i, j, k) Vz[IDX3(nx + 2, ny + 2, i, j, k)]
void orc_flow_bcs3d(double *Vx, double *Vy, double *Vz, int64_t nx, int64_t ny, int64_t nz,
                    uint32_t free_slip, uint32_t no_slip, uint32_t periodic)
{
    const int64_t x1 = nx + 1, x2 = nx + 2, y1 = ny + 1, y2 = ny + 2, z1 = nz + 1, z2 = nz + 2;
    /* sizes: Vx (x1,y2,z2)  Vy (x2,y1,z2)  Vz (x2,y2,z1) */
    if (no_slip) {
        if (no_slip & F_LEFT) {
            for (int64_t k = 0; k < z2; k++) for (int64_t j = 0; j < y2; j++) VX(0, j, k) = 0.0;
            for (int64_t k = 0; k < z2; k++) for (int64_t j = 0; j < y1; j++) VY(0, j, k) = -VY(1, j, k);
            for (int64_t k = 0; k < z1; k++) for (int64_t j = 0; j < y2; j++) VZ(0, j, k) = -VZ(1, j, k);
        }
        if (no_slip & F_RIGHT) {
            for (int64_t k = 0; k < z2; k++) for (int64_t j = 0; j < y2; j++) VX(x1 - 1, j, k) = 0.0;
            for (int64_t k = 0; k < z2; k++) for (int64_t j = 0; j < y1; j++) VY(x2 - 1, j, k) = -VY(x2 - 2, j, k);
            for (int64_t k = 0; k < z1; k++) for (int64_t j = 0; j < y2; j++) VZ(x2 - 1, j, k) = -VZ(x2 - 2, j, k);
        }
        if (no_slip & F_FRONT) {
            for (int64_t k = 0; k < z2; k++) for (int64_t i = 0; i < x1; i++) VX(i, 0, k) = -VX(i, 1, k);
            for (int64_t k = 0; k < z2; k++) for (int64_t i = 0; i < x2; i++) VY(i, 0, k) = 0.0;
            for (int64_t k = 0; k < z1; k++) for (int64_t i = 0; i < x2; i++) VZ(i, 0, k) = -VZ(i, 1, k);
        }
        if (no_slip & F_BACK) {
            for (int64_t k = 0; k < z2; k++) for (int64_t i = 0; i < x1; i++) VX(i, y2 - 1, k) = -VX(i, y2 - 2, k);
            for (int64_t k = 0; k < z2; k++) for (int64_t i = 0; i < x2; i++) VY(i, y1 - 1, k) = 0.0;
            for (int64_t k = 0; k < z1; k++) for (int64_t i = 0; i < x2; i++) VZ(i, y2 - 1, k) = -VZ(i, y2 - 2, k);
        }
        if (no_slip & F_BOT) { /* k = 1 in no_slip! */
            for (int64_t j = 0; j < y2; j++) for (int64_t i = 0; i < x1; i++) VX(i, j, 0) = -VX(i, j, 1);
            for (int64_t j = 0; j < y1; j++) for (int64_t i = 0; i < x2; i++) VY(i, j, 0) = -VY(i, j, 1);
            for (int64_t j = 0; j < y2; j++) for (int64_t i = 0; i < x2; i++) VZ(i, j, 0) = 0.0;
        }
        if (no_slip & F_TOP) { /* k = end in no_slip! */
            for (int64_t j = 0; j < y2; j++) for (int64_t i = 0; i < x1; i++) VX(i, j, z2 - 1) = -VX(i, j, z2 - 2);
            for (int64_t j = 0; j < y1; j++) for (int64_t i = 0; i < x2; i++) VY(i, j, z2 - 1) = -VY(i, j, z2 - 2);
            for (int64_t j = 0; j < y2; j++) for (int64_t i = 0; i < x2; i++) VZ(i, j, z1 - 1) = 0.0;
        }
    }
    if (free_slip) {
        if (free_slip & F_FRONT) {
            for (int64_t k = 0; k < z2; k++) for (int64_t i = 0; i < x1; i++) VX(i, 0, k) = VX(i, 1, k);
            for (int64_t k = 0; k < z1; k++) for (int64_t i = 0; i < x2; i++) VZ(i, 0, k) = VZ(i, 1, k);
        }
        if (free_slip & F_BACK) {
            for (int64_t k = 0; k < z2; k++) for (int64_t i = 0; i < x1; i++) VX(i, y2 - 1, k) = VX(i, y2 - 2, k);
            for (int64_t k = 0; k < z1; k++) for (int64_t i = 0; i < x2; i++) VZ(i, y2 - 1, k) = VZ(i, y2 - 2, k);
        }
        if (free_slip & F_TOP) { /* k = 1 in free_slip! */
            for (int64_t j = 0; j < y2; j++) for (int64_t i = 0; i < x1; i++) VX(i, j, 0) = VX(i, j, 1);
            for (int64_t j = 0; j < y1; j++) for (int64_t i = 0; i < x2; i++) VY(i, j, 0) = VY(i, j, 1);
        }
        if (free_slip & F_BOT) { /* k = end in free_slip! */
            for (int64_t j = 0; j < y2; j++) for (int64_t i = 0; i < x1; i++) VX(i, j, z2 - 1) = VX(i, j, z2 - 2);
            for (int64_t j = 0; j < y1; j++) for (int64_t i = 0; i < x2; i++) VY(i, j, z2 - 1) = VY(i, j, z2 - 2);
        }
        if (free_slip & F_LEFT) {
            for (int64_t k = 0; k < z2; k++) for (int64_t j = 0; j < y1; j++) VY(0, j, k) = VY(1, j, k);
            for (int64_t k = 0; k < z1; k++) for (int64_t j = 0; j < y2; j++) VZ(0, j, k) = VZ(1, j, k);
        }
        if (free_slip & F_RIGHT) {
            for (int64_t k = 0; k < z2; k++) for (int64_t j = 0; j < y1; j++) VY(x2 - 1, j, k) = VY(x2 - 2, j, k);
            for (int64_t k = 0; k < z1; k++) for (int64_t j = 0; j < y2; j++) VZ(x2 - 1, j, k) = VZ(x2 - 2, j, k);
        }
    }
    if (periodic) { /* periodic.jl:56-98 ; left/right, front/back, bot(k=1)/top(k=end) */
        if (periodic & F_LEFT) {
            for (int64_t k = 0; k < z2; k++) for (int64_t j = 0; j < y2; j++) VX(0, j, k) = VX(x1 - 1, j, k);
            for (int64_t k = 0; k < z2; k++) for (int64_t j = 0; j < y1; j++) VY(0, j, k) = VY(x2 - 2, j, k);
            for (int64_t k = 0; k < z1; k++) for (int64_t j = 0; j < y2; j++) VZ(0, j, k) = VZ(x2 - 2, j, k);
        }
        if (periodic & F_RIGHT) {
            for (int64_t k = 0; k < z2; k++) for (int64_t j = 0; j < y1; j++) VY(x2 - 1, j, k) = VY(1, j, k);
            for (int64_t k = 0; k < z1; k++) for (int64_t j = 0; j < y2; j++) VZ(x2 - 1, j, k) = VZ(1, j, k);
        }
        if (periodic & F_FRONT) {
            for (int64_t k = 0; k < z2; k++) for (int64_t i = 0; i < x1; i++) VX(i, 0, k) = VX(i, y2 - 2, k);
            for (int64_t k = 0; k < z2; k++) for (int64_t i = 0; i < x2; i++) VY(i, 0, k) = VY(i, y1 - 1, k);
            for (int64_t k = 0; k < z1; k++) for (int64_t i = 0; i < x2; i++) VZ(i, 0, k) = VZ(i, y2 - 2, k);
        }
        if (periodic & F_BACK) {
            for (int64_t k = 0; k < z2; k++) for (int64_t i = 0; i < x1; i++) VX(i, y2 - 1, k) = VX(i, 1, k);
            for (int64_t k = 0; k < z1; k++) for (int64_t i = 0; i < x2; i++) VZ(i, y2 - 1, k) = VZ(i, 1, k);
        }
        if (periodic & F_BOT) {
            for (int64_t j = 0; j < y2; j++) for (int64_t i = 0; i < x1; i++) VX(i, j, 0) = VX(i, j, z2 - 2);
            for (int64_t j = 0; j < y1; j++) for (int64_t i = 0; i < x2; i++) VY(i, j, 0) = VY(i, j, z2 - 2);
            for (int64_t j = 0; j < y2; j++) for (int64_t i = 0; i < x2; i++) VZ(i, j, 0) = VZ(i, j, z1 - 1);
        }
        if (periodic & F_TOP) {
            for (int64_t j = 0; j < y2; j++) for (int64_t i = 0; i < x1; i++) VX(i, j, z2 - 1) = VX(i, j, 1);
            for (int64_t j = 0; j < y1; j++) for (int64_t i = 0; i < x2; i++) VY(i, j, z2 - 1) = VY(i, j, 1);
        }
    }
}
#undef VX
#undef VY
#undef VZ

/* src/Utils.jl:409-461 ; window (1,1,1), indices clamped into range */
void orc_compute_maxloc3d(double *B, const double *A, int64_t nx, int64_t ny, int64_t nz)
{
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < nz; k++)
        for (int64_t j = 0; j < ny; j++)
            for (int64_t i = 0; i < nx; i++) {
                double x = -INFINITY;
                for (int64_t kk = k - 1; kk <= k + 1; kk++)
                    for (int64_t jj = j - 1; jj <= j + 1; jj++)
                        for (int64_t ii = i - 1; ii <= i + 1; ii++) {
                            double a = A[IDX3(nx, ny, clampi(ii, 0, nx - 1), clampi(jj, 0, ny - 1), clampi(kk, 0, nz - 1))];
                            if (a > x) x = a;
                        }
                B[IDX3(nx, ny, i, j, k)] = x;
            }
}

void orc_compute_maxloc2d(double *B, const double *A, int64_t nx, int64_t ny)
{
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < ny; j++)
        for (int64_t i = 0; i < nx; i++) {
            double x = -INFINITY;
            for (int64_t jj = j - 1; jj <= j + 1; jj++)
                for (int64_t ii = i - 1; ii <= i + 1; ii++) {
                    double a = A[IDX2(nx, clampi(ii, 0, nx - 1), clampi(jj, 0, ny - 1))];
                    if (a > x) x = a;
                }
            B[IDX2(nx, i, j)] = x;
        }
}

/* Σx² of A[2:end-1, 2:end-1, 2:end-1] (1-based) for an array of extents (n1,n2,n3) */
static double sumsq_inner3(const double *A, int64_t n1, int64_t n2, int64_t n3)
{
    double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
    for (int64_t k = 1; k < n3 - 1; k++)
        for (int64_t j = 1; j < n2 - 1; j++)
            for (int64_t i = 1; i < n1 - 1; i++) {
                double v = A[IDX3(n1, n2, i, j, k)];
                s += v * v;
            }
    return s;
}

/* src/stokes/Stokes3D.jl:127-142 + src/Utils.jl:698-701 (local part of norm_mpi) */
void orc_residual_sumsq3d(const orc_fields3d *f, const orc_params3d *p, double out[4])
{
    const int64_t nx = p->nx, ny = p->ny, nz = p->nz;
    out[0] = sumsq_inner3(f->Rx, nx - 1, ny, nz);
    out[1] = sumsq_inner3(f->Ry, nx, ny - 1, nz);
    out[2] = sumsq_inner3(f->Rz, nx, ny, nz - 1);
    double s = 0.0;
    const int64_t n = nx * ny * nz;
#pragma omp parallel for schedule(static) reduction(+ : s)
    for (int64_t c = 0; c < n; c++) s += f->RP[c] * f->RP[c];
    out[3] = s;
}

/* src/stokes/Stokes3D.jl:78-121 (single rank: update_halo! is a no-op) */
void orc_stokes3d_iteration(const orc_fields3d *f, const double *etatau, const orc_params3d *p)
{
    const int64_t n = p->nx * p->ny * p->nz;
    orc_compute_divV3d(f->divV, f->Vx, f->Vy, f->Vz, p->nx, p->ny, p->nz, p->_dx, p->_dy, p->_dz);
    orc_compute_P3d(f->P, f->P0, f->RP, f->divV, f->Q, f->eta, f->K, f->G, n, p->dt, p->r, p->theta_dtau);
    orc_compute_strain_rate3d(f, p);
    orc_compute_tau3d(f, p);
    orc_compute_V3d(f, etatau, p);
    orc_velocity2displacement3d(f, p);
    if (p->displacement_bcs) orc_flow_bcs3d(f->Ux, f->Uy, f->Uz, p->nx, p->ny, p->nz, p->free_slip, p->no_slip, p->periodic);
    else orc_flow_bcs3d(f->Vx, f->Vy, f->Vz, p->nx, p->ny, p->nz, p->free_slip, p->no_slip, p->periodic);
}

/* src/stokes/Stokes3D.jl:25-186 */
int32_t orc_stokes3d_solve(const orc_fields3d *f, const orc_params3d *p, orc_result *res)
{
    const int64_t nx = p->nx, ny = p->ny, nz = p->nz;
    const size_t n = (size_t)nx * ny * nz;
    double *etatau = (double *)malloc(n * sizeof(double));
    orc_compute_maxloc3d(etatau, f->eta, nx, ny, nz);     /* :55-57 */
    if (p->displacement_bcs) {                            /* displacement2velocity!(stokes, dt, flow_bcs) :72 */
        const double _dt = 1.0 / p->dt;
        for (size_t c = 0; c < (size_t)(nx + 1) * (ny + 2) * (nz + 2); c++) f->Vx[c] = f->Ux[c] * _dt;
        for (size_t c = 0; c < (size_t)(nx + 2) * (ny + 1) * (nz + 2); c++) f->Vy[c] = f->Uy[c] * _dt;
        for (size_t c = 0; c < (size_t)(nx + 2) * (ny + 2) * (nz + 1); c++) f->Vz[c] = f->Uz[c] * _dt;
    }

    double err_it1 = 1.0, err = 1.0;
    int64_t iter = 0, cont = 0;
    res->status = 0;
    double t0 = omp_get_wtime();
    while (iter < 2 || (((err / err_it1) > p->eps_rel && err > p->eps_abs) && iter <= p->iterMax)) {
        orc_stokes3d_iteration(f, etatau, p);
        iter += 1;
        if (iter % p->nout == 0 && iter > 1) {
            double s[4];
            orc_residual_sumsq3d(f, p, s);
            double nRx = sqrt(s[0]) / (double)((p->nxg - 2) * (p->nyg - 1) * (p->nzg - 1));
            double nRy = sqrt(s[1]) / (double)((p->nxg - 1) * (p->nyg - 2) * (p->nzg - 1));
            double nRz = sqrt(s[2]) / (double)((p->nxg - 1) * (p->nyg - 1) * (p->nzg - 2));
            double nDV = sqrt(s[3]) / (double)(p->nxg * p->nyg * p->nzg);
            if (cont < res->cap) {
                res->norm_Rx[cont] = nRx; res->norm_Ry[cont] = nRy; res->norm_Rz[cont] = nRz; res->norm_divV[cont] = nDV;
            }
            /* Julia's max() propagates NaN */
            err = fmax(fmax(nRx, nRy), fmax(nRz, nDV));
            if (isnan(nRx) || isnan(nRy) || isnan(nRz) || isnan(nDV)) err = NAN;
            if (cont < res->cap) { res->err_evo1[cont] = err; res->err_evo2[cont] = iter; }
            if (cont == 0) err_it1 = err;  /* max of the first sampled norms */
            cont += 1;
            if (isnan(err)) {                            /* error("NaN(s)") :162 -- the exception skips :172-173 */
                res->status = 1; res->iter = iter; res->nchecks = cont < res->cap ? cont : res->cap;
                res->time_s = omp_get_wtime() - t0;
                free(etatau);
                return 1;
            }
        }
    }
    res->time_s = omp_get_wtime() - t0;
    res->iter = iter;
    res->nchecks = cont < res->cap ? cont : res->cap;

    /* :172-173 multi_copy! τ -> τ_o (staggered set, then centre set) */
    memcpy(f->toxx, f->txx, n * sizeof(double));
    memcpy(f->toyy, f->tyy, n * sizeof(double));
    memcpy(f->tozz, f->tzz, n * sizeof(double));
    memcpy(f->toyz, f->tyz, (size_t)nx * (ny + 1) * (nz + 1) * sizeof(double));
    memcpy(f->toxz, f->txz, (size_t)(nx + 1) * ny * (nz + 1) * sizeof(double));
    memcpy(f->toxy, f->txy, (size_t)(nx + 1) * (ny + 1) * nz * sizeof(double));
    if (f->tyz_c && f->toyz_c) memcpy(f->toyz_c, f->tyz_c, n * sizeof(double));
    if (f->txz_c && f->toxz_c) memcpy(f->toxz_c, f->txz_c, n * sizeof(double));
    if (f->txy_c && f->toxy_c) memcpy(f->toxy_c, f->txy_c, n * sizeof(double));
    free(etatau);
    return res->status;
}
