/* oracle/gridops.c -- TEST INFRASTRUCTURE ONLY (see jrx_oracle.h).  CPU restatement of the grid operators a time step runs either side of
 * solve! / heatdiffusion_PT!: src/Interpolations.jl (vertex2center!, center2vertex_harm!, center2vertex! 3D, velocity2vertex!, velocity2center!),
 * src/rheology/BuoyancyForces.jl:6-60 (compute_ρg!), src/thermal_diffusion/ShearHeating.jl:14-71 (compute_shear_heating!).
 * One loop nest per reference kernel, sums in the reference's order.  Pinned by test/test_Interpolations.jl:43-65,150-209 (formulas on random
 * inputs) through tests/test_oracle_gridops.py; compute_shearheating is GeoParams' (absent): form assumed, parity unpinned. */
#include "common.h"
#include "material.h"

/* _velocity2vertex! 2D (Interpolations.jl:244-249); outputs (mx, my) */
void orc_velocity2vertex2d(double *Vxv, double *Vyv, const double *Vx, const double *Vy, int64_t nx, int64_t ny, int64_t mx, int64_t my)
{
    (void)ny;
    for (int64_t j = 0; j < my; j++)
        for (int64_t i = 0; i < mx; i++) {
            Vxv[IDX2(mx, i, j)] = (Vx[IDX2(nx + 1, i, j)] + Vx[IDX2(nx + 1, i, j + 1)]) / 2;
            Vyv[IDX2(mx, i, j)] = (Vy[IDX2(nx + 2, i, j)] + Vy[IDX2(nx + 2, i + 1, j)]) / 2;
        }
}

/* _velocity2vertex! 3D (Interpolations.jl:219-230) */
void orc_velocity2vertex3d(double *Vxv, double *Vyv, double *Vzv, const double *Vx, const double *Vy, const double *Vz, int64_t nx, int64_t ny, int64_t nz,
                           int64_t mx, int64_t my, int64_t mz)
{
    (void)nz;
#define VX_(i, j, k) Vx[IDX3(nx + 1, ny + 2, i, j, k)]
#define VY_(i, j, k) Vy[IDX3(nx + 2, ny + 1, i, j, k)]
#define VZ_(i, j, k) Vz[IDX3(nx + 2, ny + 2, i, j, k)]
#pragma omp parallel for
    for (int64_t k = 0; k < mz; k++)
        for (int64_t j = 0; j < my; j++)
            for (int64_t i = 0; i < mx; i++) {
                const size_t o = IDX3(mx, my, i, j, k);
                Vxv[o] = 0.25 * (VX_(i, j, k) + VX_(i, j + 1, k) + VX_(i, j, k + 1) + VX_(i, j + 1, k + 1));
                Vyv[o] = 0.25 * (VY_(i, j, k) + VY_(i + 1, j, k) + VY_(i, j, k + 1) + VY_(i + 1, j, k + 1));
                Vzv[o] = 0.25 * (VZ_(i, j, k) + VZ_(i, j + 1, k) + VZ_(i + 1, j, k) + VZ_(i + 1, j + 1, k));
            }
}

/* _velocity2center! 2D (Interpolations.jl:285-289) */
void orc_velocity2center2d(double *Vxc, double *Vyc, const double *Vx, const double *Vy, int64_t nx, int64_t ny)
{
    for (int64_t j = 0; j < ny; j++)
        for (int64_t i = 0; i < nx; i++) {
            Vxc[IDX2(nx, i, j)] = (Vx[IDX2(nx + 1, i, j + 1)] + Vx[IDX2(nx + 1, i + 1, j + 1)]) / 2;
            Vyc[IDX2(nx, i, j)] = (Vy[IDX2(nx + 2, i + 1, j)] + Vy[IDX2(nx + 2, i + 1, j + 1)]) / 2;
        }
}

/* _velocity2center! 3D (Interpolations.jl:264-270) */
void orc_velocity2center3d(double *Vxc, double *Vyc, double *Vzc, const double *Vx, const double *Vy, const double *Vz, int64_t nx, int64_t ny, int64_t nz)
{
#pragma omp parallel for
    for (int64_t k = 0; k < nz; k++)
        for (int64_t j = 0; j < ny; j++)
            for (int64_t i = 0; i < nx; i++) {
                const size_t o = IDX3(nx, ny, i, j, k);
                Vxc[o] = (VX_(i, j + 1, k + 1) + VX_(i + 1, j + 1, k + 1)) / 2;
                Vyc[o] = (VY_(i + 1, j, k + 1) + VY_(i + 1, j + 1, k + 1)) / 2;
                Vzc[o] = (VZ_(i + 1, j + 1, k) + VZ_(i + 1, j + 1, k + 1)) / 2;
            }
#undef VX_
#undef VY_
#undef VZ_
}

/* vertex2center! (Interpolations.jl:72-96): ni = size(vertex) .- 1; centre[I .+ ghost] = mean of the corners */
void orc_vertex2center(double *cen, const double *ver, const int64_t vdim[3], const int64_t cdim[3], int32_t ndim, int32_t gx, int32_t gy, int32_t gz)
{
    const int64_t v1 = vdim[0], v2 = vdim[1], c1 = cdim[0], c2 = cdim[1];
    if (ndim == 2) {
        for (int64_t j = 0; j < v2 - 1; j++)
            for (int64_t i = 0; i < v1 - 1; i++)
                cen[IDX2(c1, i + gx, j + gy)] = 0.25 * (ver[IDX2(v1, i, j)] + ver[IDX2(v1, i + 1, j)] + ver[IDX2(v1, i, j + 1)] + ver[IDX2(v1, i + 1, j + 1)]);
        return;
    }
#define V_(i, j, k) ver[IDX3(v1, v2, i, j, k)]
    for (int64_t k = 0; k < vdim[2] - 1; k++)
        for (int64_t j = 0; j < v2 - 1; j++)
            for (int64_t i = 0; i < v1 - 1; i++)
                cen[IDX3(c1, c2, i + gx, j + gy, k + gz)] = 0.125 * (V_(i, j, k) + V_(i + 1, j, k) + V_(i, j + 1, k) + V_(i + 1, j + 1, k) + V_(i, j, k + 1) +
                                                                     V_(i + 1, j, k + 1) + V_(i, j + 1, k + 1) + V_(i + 1, j + 1, k + 1));
#undef V_
}

/* center2vertex_kernel_harm! (Interpolations.jl:123-137) */
void orc_center2vertex_harm2d(double *ver, const double *cen, int64_t nx, int64_t ny)
{
    for (int64_t j = 0; j < ny + 1; j++)
        for (int64_t i = 0; i < nx + 1; i++) {
            const int64_t il = i - 1 > 0 ? i - 1 : 0, ir = i < nx - 1 ? i : nx - 1, jb = j - 1 > 0 ? j - 1 : 0, jt = j < ny - 1 ? j : ny - 1;
            ver[IDX2(nx + 1, i, j)] = 4 / (1 / cen[IDX2(nx, il, jb)] + 1 / cen[IDX2(nx, ir, jb)] + 1 / cen[IDX2(nx, il, jt)] + 1 / cen[IDX2(nx, ir, jt)]);
        }
}

/* center2vertex_kernel! 3D (Interpolations.jl:146-178) */
void orc_center2vertex3d(double *vyz, double *vxz, double *vxy, const double *cyz, const double *cxz, const double *cxy, int64_t nx, int64_t ny, int64_t nz)
{
#define C_(A, i, j, k) A[IDX3(nx, ny, i, j, k)]
    for (int64_t k = 0; k < nz; k++)
        for (int64_t j = 0; j < ny; j++)
            for (int64_t i = 0; i < nx; i++) {
                const int64_t i1 = i + 1, j1 = j + 1, k1 = k + 1;
                if (j1 < ny && k1 < nz) vyz[IDX3(nx, ny + 1, i, j1, k1)] = 0.25 * (C_(cyz, i, j, k) + C_(cyz, i, j1, k) + C_(cyz, i, j, k1) + C_(cyz, i, j1, k1));
                if (i1 < nx && k1 < nz) vxz[IDX3(nx + 1, ny, i1, j, k1)] = 0.25 * (C_(cxz, i, j, k) + C_(cxz, i1, j, k) + C_(cxz, i, j, k1) + C_(cxz, i1, j, k1));
                if (i1 < nx && j1 < ny) vxy[IDX3(nx + 1, ny + 1, i1, j1, k)] = 0.25 * (C_(cxy, i, j, k) + C_(cxy, i1, j, k) + C_(cxy, i, j1, k) + C_(cxy, i1, j1, k));
            }
#undef C_
}

/* compute_ρg_kernel! (BuoyancyForces.jl:17-21,50-54): the scalar-gravity form, one array (the caller passes the last component of ρg).  args.T is read at the
 * cell's own [i, j, k] of an array of extents tdim (getindex_NamedTuple(args, I...): no shift for a ghosted thermal.T) */
void orc_compute_rhog(double *rhog, const orc_rheology *rh, const double *phase_c, const double *T, const double *P, const int64_t n[3], const int64_t tdim[3],
                      int32_t ndim)
{
    const int64_t nx = n[0], ny = n[1], nz = ndim == 3 ? n[2] : 1, t1 = tdim ? tdim[0] : nx, t2 = tdim ? tdim[1] : ny;
    for (int64_t k = 0; k < nz; k++)
        for (int64_t j = 0; j < ny; j++)
            for (int64_t i = 0; i < nx; i++) {
                const size_t c = IDX3(nx, ny, i, j, k);
                const double t = T ? T[IDX3(t1, t2, i, j, k)] : 0.0, p = P ? P[c] : 0.0;
                rhog[c] = (phase_c ? mat_density_ratio(rh, phase_c + (size_t)rh->nphase * c, t, p) : mat_density(rh, 0, t, p)) * rh->gravity;
            }
}

/* compute_viscosity_kernel! for one MaterialParams (rheology/Viscosity.jl:136-167), creep laws of the table (no strain-rate dependence):
 * η <- clamp(continuation_linear(η_creep(T[I .+ 1] | T[I], P[I]), η, ν), cutoff) */
void orc_compute_viscosity_single(double *eta, const orc_rheology *rh, const double *T, const double *P, const int64_t n[3], const int64_t tdim[3], int32_t ndim,
                                  double nu, double lo, double hi, const double *AII, int32_t tau)
{
    const int64_t nx = n[0], ny = n[1], nz = ndim == 3 ? n[2] : 1, t1 = tdim ? tdim[0] : nx, t2 = tdim ? tdim[1] : ny;
    const int64_t sh = (tdim && tdim[0] == nx + 2) ? 1 : 0, shk = ndim == 3 ? sh : 0;
    for (int64_t k = 0; k < nz; k++)
        for (int64_t j = 0; j < ny; j++)
            for (int64_t i = 0; i < nx; i++) {
                const size_t c = IDX3(nx, ny, i, j, k);
                const double t = T ? T[IDX3(t1, t2, i + sh, j + sh, k + shk)] : 0.0, p = P ? P[c] : 0.0;
                /* array form compute_viscosity_εII! / compute_viscosity_τII!(η, ν, AII, args, rheology, cutoff) (Viscosity.jl:169-196) when AII is given */
                const double e = (1 - nu) * eta[c] + nu * mat_viscosity(rh, 0, AII ? AII[c] : 0.0, t, p, tau);
                eta[c] = fmin(fmax(e, lo), hi);
            }
}

/* compute_lithostatic_pressure!(P, ρg, dz) (src/Utils.jl:541-573): reverse(cumsum(reverse(w))) - w / 2 with w = ρg .* dz, accumulated from the top cell */
void orc_compute_lithostatic_pressure(double *P, const double *rhog, double dz, const double *dz_cells, const int64_t n[3], int32_t ndim)
{
    const int64_t ncol = ndim == 3 ? n[0] * n[1] : n[0], nlast = n[ndim - 1];
    for (int64_t c = 0; c < ncol; c++) {
        double acc = 0.0;
        for (int64_t k = nlast - 1; k >= 0; k--) {
            const double w = rhog[c + ncol * k] * (dz_cells ? dz_cells[k] : dz);
            acc += w;
            P[c + ncol * k] = acc - w / 2;
        }
    }
}

/* fn_ratio(fn, rheology, ratio) (src/phases/phases.jl:6-15) */
static double ratio_sum(const double *val, const double *r, int n)
{
    double x = 0.0;
    for (int q = 0; q < n; q++) x += (r[q] == 0.0) ? 0.0 : val[q] * r[q];
    return x;
}

/* compute_shear_heating_kernel! (ShearHeating.jl:31-41,58-71) with cache_tensors (StressUpdate.jl:190-205,252-276).  tau, tau_o: centre arrays in Voigt
 * order (2D xx, yy, xy_c; 3D xx, yy, zz, yz_c, xz_c, xy_c); eps: the staggered strain-rate tensor (shear on its edges).
 * compute_shearheating(ConstantShearheating(Χ), τ, ε, ε_el) = Χ Σ τ_ij (ε_ij - ε_el_ij), shear terms of the Voigt tuple twice [GeoParams; ASSUMED] */
void orc_compute_shear_heating(double *sh, const double *const *tau, const double *const *tau_o, const double *const *eps, const double *phase_c,
                               const orc_rheology *rh, const double *chi, double dt, const int64_t n[3], int32_t ndim)
{
    const int64_t nx = n[0], ny = n[1], nz = ndim == 3 ? n[2] : 1;
    const int N = ndim == 3 ? 6 : 3, NN = ndim == 3 ? 3 : 2;
    for (int64_t k = 0; k < nz; k++)
        for (int64_t j = 0; j < ny; j++)
            for (int64_t i = 0; i < nx; i++) {
                const size_t c = IDX3(nx, ny, i, j, k);
                double G = rh->G[0], X = chi[0];
                if (phase_c) {
                    const double *r = phase_c + (size_t)rh->nphase * c;
                    G = ratio_sum(rh->G, r, rh->nphase);
                    X = ratio_sum(chi, r, rh->nphase);
                }
                const double _Gdt = inv(G * dt);
                double e[6];
                for (int q = 0; q < NN; q++) e[q] = eps[q][c];
                if (ndim == 3) {
                    const double *yz = eps[3], *xz = eps[4], *xy = eps[5];
                    e[3] = 0.25 * (yz[IDX3(nx, ny + 1, i, j, k)] + yz[IDX3(nx, ny + 1, i, j + 1, k)] + yz[IDX3(nx, ny + 1, i, j, k + 1)] + yz[IDX3(nx, ny + 1, i, j + 1, k + 1)]);
                    e[4] = 0.25 * (xz[IDX3(nx + 1, ny, i, j, k)] + xz[IDX3(nx + 1, ny, i + 1, j, k)] + xz[IDX3(nx + 1, ny, i, j, k + 1)] + xz[IDX3(nx + 1, ny, i + 1, j, k + 1)]);
                    e[5] = 0.25 * (xy[IDX3(nx + 1, ny + 1, i, j, k)] + xy[IDX3(nx + 1, ny + 1, i + 1, j, k)] + xy[IDX3(nx + 1, ny + 1, i, j + 1, k)] + xy[IDX3(nx + 1, ny + 1, i + 1, j + 1, k)]);
                } else {
                    const double *xy = eps[2];
                    e[2] = (xy[IDX2(nx + 1, i, j)] + xy[IDX2(nx + 1, i + 1, j)] + xy[IDX2(nx + 1, i, j + 1)] + xy[IDX2(nx + 1, i + 1, j + 1)]) / 4;
                }
                double H = 0.0;
                for (int q = 0; q < N; q++) {
                    const double t = tau[q][c], eel = 0.5 * ((t - tau_o[q][c]) * _Gdt);
                    const double w = t * (e[q] - eel);
                    H += q < NN ? w : 2.0 * w;
                }
                sh[c] = fmax(0.0, X * H);
            }
}
