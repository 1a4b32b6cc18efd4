// gridops.hip -- the grid operators a time step runs either side of solve! / heatdiffusion_PT!, for gfx950.
//
// Reference being replaced (PTsolvers/JustRelax.jl): src/Interpolations.jl:72-96 (vertex2center!), :116-137 (center2vertex_harm!),
// :139-178 (center2vertex! 3D), :212-249 (velocity2vertex!), :257-289 (velocity2center!); src/rheology/BuoyancyForces.jl:6-60
// (compute_ρg!); src/thermal_diffusion/ShearHeating.jl:14-71 (compute_shear_heating!) with cache_tensors
// (rheology/StressUpdate.jl:190-205,252-276).  The methods the reference's AMDGPU extension forwards for these are
// src/ext/AMDGPU/2D.jl:301-352, 3D.jl:311-362.
// Every operator is one streaming pass (a few reads and one to three writes per output value): HBM-bound, one thread per output,
// x fastest across the lanes of a wave; nothing to tile.
#include "jrx_internal.hpp"
#include "jrx_kernels.hpp"
#include "jrx_material.hpp"

namespace {

// output (i, j, k) of an (n1, n2, n3) box: xy flattened over blockIdx.x, k = blockIdx.y
#define OUT_IJK(n1_, n2_)                                               \
    const int t_ = blockIdx.x * blockDim.x + threadIdx.x;               \
    const int j = t_ / (n1_), i = t_ - j * (n1_), k = blockIdx.y;       \
    if (j >= (n2_)) return;
#define OUT_GRID(n1_, n2_, n3_) dim3((unsigned)(((i64)(n1_) * (n2_) + 255) / 256), (unsigned)(n3_))

// _velocity2vertex! 2D (Interpolations.jl:244-249): Vx (nx+1, ny+2), Vy (nx+2, ny+1) -> (mx, my) outputs
__global__ __launch_bounds__(256) void k_vel2vertex2d(double *__restrict__ Vxv, double *__restrict__ Vyv, const double *__restrict__ Vx,
                                                      const double *__restrict__ Vy, int nx, int mx, int my)
{
    OUT_IJK(mx, my)
    const i64 o = i + (i64)mx * j;
    Vxv[o] = (Vx[i + (i64)(nx + 1) * j] + Vx[i + (i64)(nx + 1) * (j + 1)]) / 2;
    Vyv[o] = (Vy[i + (i64)(nx + 2) * j] + Vy[i + 1 + (i64)(nx + 2) * j]) / 2;
}

// _velocity2vertex! 3D (Interpolations.jl:219-230)
__global__ __launch_bounds__(256) void k_vel2vertex3d(double *__restrict__ Vxv, double *__restrict__ Vyv, double *__restrict__ Vzv,
                                                      const double *__restrict__ Vx, const double *__restrict__ Vy, const double *__restrict__ Vz,
                                                      int nx, int ny, int mx, int my)
{
    OUT_IJK(mx, my)
    const i64 o = i + (i64)mx * (j + (i64)my * k);
    const i64 x1 = nx + 1, xp = x1 * (ny + 2), y1 = nx + 2, yp = y1 * (ny + 1), z1 = nx + 2, zp = z1 * (ny + 2);
    const double *a = Vx + i + x1 * j + xp * k, *b = Vy + i + y1 * j + yp * k, *c = Vz + i + z1 * j + zp * k;
    Vxv[o] = 0.25 * (a[0] + a[x1] + a[xp] + a[x1 + xp]);
    Vyv[o] = 0.25 * (b[0] + b[1] + b[yp] + b[1 + yp]);
    Vzv[o] = 0.25 * (c[0] + c[z1] + c[1] + c[1 + z1]);
}

// _velocity2center! 2D (Interpolations.jl:285-289)
__global__ __launch_bounds__(256) void k_vel2center2d(double *__restrict__ Vxc, double *__restrict__ Vyc, const double *__restrict__ Vx,
                                                      const double *__restrict__ Vy, int nx, int ny)
{
    OUT_IJK(nx, ny)
    const i64 o = i + (i64)nx * j;
    Vxc[o] = (Vx[i + (i64)(nx + 1) * (j + 1)] + Vx[i + 1 + (i64)(nx + 1) * (j + 1)]) / 2;
    Vyc[o] = (Vy[i + 1 + (i64)(nx + 2) * j] + Vy[i + 1 + (i64)(nx + 2) * (j + 1)]) / 2;
}

// _velocity2center! 3D (Interpolations.jl:264-270)
__global__ __launch_bounds__(256) void k_vel2center3d(double *__restrict__ Vxc, double *__restrict__ Vyc, double *__restrict__ Vzc,
                                                      const double *__restrict__ Vx, const double *__restrict__ Vy, const double *__restrict__ Vz,
                                                      int nx, int ny)
{
    OUT_IJK(nx, ny)
    const i64 o = i + (i64)nx * (j + (i64)ny * k);
    const i64 x1 = nx + 1, xp = x1 * (ny + 2), y1 = nx + 2, yp = y1 * (ny + 1), z1 = nx + 2, zp = z1 * (ny + 2);
    const double *a = Vx + i + x1 * (j + 1) + xp * (k + 1), *b = Vy + i + 1 + y1 * j + yp * (k + 1), *c = Vz + i + 1 + z1 * (j + 1) + zp * k;
    Vxc[o] = (a[0] + a[1]) / 2;
    Vyc[o] = (b[0] + b[y1]) / 2;
    Vzc[o] = (c[0] + c[zp]) / 2;
}

// vertex2center_kernel! (Interpolations.jl:78-96): centre[I + ghost] = mean of the 4 (2D) / 8 (3D) corners; vertex (v1, v2[, v3]), centre (c1, c2[, c3])
template <bool D3>
__global__ __launch_bounds__(256) void k_vertex2center(double *__restrict__ cen, const double *__restrict__ ver, int v1, int v2, int c1, int c2, int g1,
                                                       int g2, int g3)
{
    OUT_IJK(v1 - 1, v2 - 1)
    const i64 vp = (i64)v1 * v2;
    const double *a = ver + i + (i64)v1 * j + vp * k;
    const i64 o = (i + g1) + (i64)c1 * ((j + g2) + (i64)c2 * (k + g3));
    if (D3) cen[o] = 0.125 * (a[0] + a[1] + a[v1] + a[v1 + 1] + a[vp] + a[vp + 1] + a[vp + v1] + a[vp + v1 + 1]);
    else cen[o] = 0.25 * (a[0] + a[1] + a[v1] + a[v1 + 1]);
}

// center2vertex_kernel_harm! (Interpolations.jl:123-137): harmonic mean of the clamped 2 x 2 cells around vertex (i, j)
__global__ __launch_bounds__(256) void k_center2vertex_harm2d(double *__restrict__ ver, const double *__restrict__ cen, int nx, int ny)
{
    OUT_IJK(nx + 1, ny + 1)
    const int il = max(i - 1, 0), ir = min(i, nx - 1), jb = max(j - 1, 0), jt = min(j, ny - 1);
    ver[i + (i64)(nx + 1) * j] = 4 / (1 / cen[il + (i64)nx * jb] + 1 / cen[ir + (i64)nx * jb] + 1 / cen[il + (i64)nx * jt] + 1 / cen[ir + (i64)nx * jt]);
}

// center2vertex_kernel! 3D (Interpolations.jl:146-178): interior edges of the three shear families from three centre arrays; the boundary edges keep
// their values (the reference's kernel never writes them)
__global__ __launch_bounds__(256) void k_center2vertex3d(double *__restrict__ vyz, double *__restrict__ vxz, double *__restrict__ vxy,
                                                         const double *__restrict__ cyz, const double *__restrict__ cxz, const double *__restrict__ cxy,
                                                         int nx, int ny, int nz)
{
    OUT_IJK(nx, ny)
    const i64 cp = (i64)nx * ny, c = i + (i64)nx * j + cp * k;
    const bool i1 = i + 1 < nx, j1 = j + 1 < ny, k1 = k + 1 < nz;
    if (j1 && k1) vyz[i + (i64)nx * ((j + 1) + (i64)(ny + 1) * (k + 1))] = 0.25 * (cyz[c] + cyz[c + nx] + cyz[c + cp] + cyz[c + nx + cp]);
    if (i1 && k1) vxz[(i + 1) + (i64)(nx + 1) * (j + (i64)ny * (k + 1))] = 0.25 * (cxz[c] + cxz[c + 1] + cxz[c + cp] + cxz[c + 1 + cp]);
    if (i1 && j1) vxy[(i + 1) + (i64)(nx + 1) * ((j + 1) + (i64)(ny + 1) * k)] = 0.25 * (cxy[c] + cxy[c + 1] + cxy[c + nx] + cxy[c + 1 + nx]);
}

// compute_ρg_kernel! (BuoyancyForces.jl:17-21,50-54): ρg = density(T, P[, ratios]) * gravity of the first phase.  args.T is indexed with the cell's own
// [i, j, k] in T's extents (t1, t2): a ghosted thermal.T passed as args.T is read WITHOUT the shift to the cell centre, exactly as getindex_NamedTuple does
// in the reference (test/test_WENO5.jl:208-214 calls it that way)
template <bool PH>
__global__ __launch_bounds__(256) void k_compute_rhog(double *__restrict__ rhog, const jrx_rheology rh, const double *__restrict__ phase_c,
                                                      const double *__restrict__ T, const double *__restrict__ P, int nx, int ny, int t1, int t2)
{
    OUT_IJK(nx, ny)
    const i64 c = i + (i64)nx * (j + (i64)ny * k);
    const double t = T ? T[i + (i64)t1 * (j + (i64)t2 * k)] : 0.0, p = P ? P[c] : 0.0;
    rhog[c] = (PH ? mat_density_ratio(rh, phase_c + (i64)rh.nphase * c, t, p) : mat_density(rh, 0, t, p)) * rh.gravity;
}

// compute_viscosity_kernel! for ONE MaterialParams (rheology/Viscosity.jl:136-167) with the table's creep laws (LinearViscous, Arrhenius: no strain-rate
// dependence, so the strain-rate operands drop out): η <- clamp(ν η_creep(T, P) + (1 - ν) η, cutoff).  args.T is read at I .+ 1 (local_viscosity_args,
// Viscosity.jl:513-523: the ghosted thermal.T) when sh = 1, at the cell's own index otherwise; args.P at I
__global__ __launch_bounds__(256) void k_viscosity_single(double *__restrict__ eta, const jrx_rheology rh, const double *__restrict__ T, const double *__restrict__ P,
                                                          int nx, int ny, int t1, int t2, int sh, int shk, double nu, double lo, double hi,
                                                          const double *__restrict__ AII, bool tau)
{
    OUT_IJK(nx, ny)
    const i64 c = i + (i64)nx * (j + (i64)ny * k);
    const double t = T ? T[(i + sh) + (i64)t1 * ((j + sh) + (i64)t2 * (k + shk))] : 0.0, p = P ? P[c] : 0.0;
    const double e = (1 - nu) * eta[c] + nu * mat_viscosity(rh, 0, AII ? AII[c] : 0.0, t, p, tau);
    eta[c] = fmin(fmax(e, lo), hi);
}

// compute_lithostatic_pressure!(P, ρg, dz) (src/Utils.jl:541-573): P[j] = Σ_{k>j} ρg[k] dz[k] + ρg[j] dz[j] / 2 down every column of the last dimension -- the
// reference's reverse(cumsum(reverse(w))) - w / 2 accumulated from the top cell downwards.  One thread per column, x fastest across the lanes.
__global__ __launch_bounds__(256) void k_lithostatic(double *__restrict__ P, const double *__restrict__ rhog, const double *__restrict__ dzv, const double dz,
                                                     const i64 ncol, const int nlast)
{
    const i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    double acc = 0.0;
    for (int k = nlast - 1; k >= 0; k--) {
        const i64 q = c + ncol * k;
        const double w = rhog[q] * (dzv ? dzv[k] : dz);
        acc += w;
        P[q] = acc - w / 2;
    }
}

struct ShArgs {
    double *sh;
    const double *t[6], *to[6], *e[6];      // Voigt order: 2D xx, yy, xy; 3D xx, yy, zz, yz, xz, xy.  τ, τ_o at the centres; ε shear on its edges
    const double *phase_c;
    double G[JRX_MAXPHASE], chi[JRX_MAXPHASE];
    double dt;
    int nphase, nx, ny, nz;
};

// fn_ratio(fn, rheology, ratio) without args (src/phases/phases.jl:6-15)
__device__ __forceinline__ double sh_ratio(const double *val, const double *r, int n)
{
    double x = 0.0;
    for (int q = 0; q < n; q++) x += (r[q] == 0.0) ? 0.0 : val[q] * r[q];
    return x;
}

// compute_shear_heating_kernel! (ShearHeating.jl:31-41,58-71).  ε_el = 0.5 (τ - τ_o) / (G dt); H = max(0, Χ τ : (ε - ε_el)) with the shear terms
// of the Voigt tuples counted twice [GeoParams' compute_shearheating(ConstantShearheating) -- form assumed]
template <bool D3>
__global__ __launch_bounds__(256) void k_shear_heating(const ShArgs a)
{
    const int nx = a.nx, ny = a.ny;
    OUT_IJK(nx, ny)
    const i64 c = i + (i64)nx * (j + (i64)ny * k);
    double G = a.G[0], chi = a.chi[0];
    if (a.phase_c) {
        const double *r = a.phase_c + (i64)a.nphase * c;
        G = sh_ratio(a.G, r, a.nphase);
        chi = sh_ratio(a.chi, r, a.nphase);
    }
    const double _Gdt = 1.0 / (G * a.dt);
    constexpr int N = D3 ? 6 : 3, NN = D3 ? 3 : 2;
    double e[6];
    for (int q = 0; q < NN; q++) e[q] = a.e[q][c];
    if (D3) {       // _av_yz, _av_xz, _av_xy of the edge arrays (cache_tensors, StressUpdate.jl:252-276)
        const i64 y1 = nx, yp = y1 * (ny + 1), x1 = nx + 1, xp = x1 * ny, w1 = nx + 1, wp = w1 * (ny + 1);
        const double *yz = a.e[3] + i + y1 * j + yp * k, *xz = a.e[4] + i + x1 * j + xp * k, *xy = a.e[5] + i + w1 * j + wp * k;
        e[3] = 0.25 * (yz[0] + yz[y1] + yz[yp] + yz[y1 + yp]);
        e[4] = 0.25 * (xz[0] + xz[1] + xz[xp] + xz[1 + xp]);
        e[5] = 0.25 * (xy[0] + xy[1] + xy[w1] + xy[1 + w1]);
    } else {        // av_shear: sum(_gather(A, I...)) / 4 (StressUpdate.jl:193)
        const double *xy = a.e[2] + i + (i64)(nx + 1) * j;
        e[2] = (xy[0] + xy[1] + xy[nx + 1] + xy[nx + 2]) / 4;
    }
    double H = 0.0;
#pragma unroll
    for (int q = 0; q < N; q++) {
        const double t = a.t[q][c], eel = 0.5 * ((t - a.to[q][c]) * _Gdt);
        const double w = t * (e[q] - eel);
        H += q < NN ? w : 2.0 * w;
    }
    a.sh[c] = fmax(0.0, chi * H);
}

jrx_status done(jrx_handle *h)
{
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

}  // namespace

extern "C" {

jrx_status jrx_velocity2vertex2d(jrx_handle *h, double *Vx_v, double *Vy_v, const double *Vx, const double *Vy, int64_t nx, int64_t ny, int64_t mx, int64_t my)
{
    if (!h) return JRX_ERR_ARG;
    JRX_TRY(jrx_check_device(h));
    if (!Vx_v || !Vy_v || !Vx || !Vy || nx < 1 || ny < 1 || mx < 1 || my < 1 || mx > nx + 1 || my > ny + 1)
        return jrx_fail(h, JRX_ERR_ARG, "velocity2vertex!: bad argument (outputs at most ni .+ 1)");
    hipLaunchKernelGGL(k_vel2vertex2d, OUT_GRID(mx, my, 1), dim3(256), 0, h->stream, Vx_v, Vy_v, Vx, Vy, (int)nx, (int)mx, (int)my);
    return done(h);
}

jrx_status jrx_velocity2vertex3d(jrx_handle *h, double *Vx_v, double *Vy_v, double *Vz_v, const double *Vx, const double *Vy, const double *Vz, int64_t nx,
                                 int64_t ny, int64_t nz, int64_t mx, int64_t my, int64_t mz)
{
    if (!h) return JRX_ERR_ARG;
    JRX_TRY(jrx_check_device(h));
    if (!Vx_v || !Vy_v || !Vz_v || !Vx || !Vy || !Vz || nx < 1 || ny < 1 || nz < 1 || mx < 1 || my < 1 || mz < 1 || mx > nx + 1 || my > ny + 1 || mz > nz + 1)
        return jrx_fail(h, JRX_ERR_ARG, "velocity2vertex!: bad argument (outputs at most ni .+ 1)");
    hipLaunchKernelGGL(k_vel2vertex3d, OUT_GRID(mx, my, mz), dim3(256), 0, h->stream, Vx_v, Vy_v, Vz_v, Vx, Vy, Vz, (int)nx, (int)ny, (int)mx, (int)my);
    return done(h);
}

jrx_status jrx_velocity2center2d(jrx_handle *h, double *Vx_c, double *Vy_c, const double *Vx, const double *Vy, int64_t nx, int64_t ny)
{
    if (!h) return JRX_ERR_ARG;
    JRX_TRY(jrx_check_device(h));
    if (!Vx_c || !Vy_c || !Vx || !Vy || nx < 1 || ny < 1) return jrx_fail(h, JRX_ERR_ARG, "velocity2center!: bad argument");
    hipLaunchKernelGGL(k_vel2center2d, OUT_GRID(nx, ny, 1), dim3(256), 0, h->stream, Vx_c, Vy_c, Vx, Vy, (int)nx, (int)ny);
    return done(h);
}

jrx_status jrx_velocity2center3d(jrx_handle *h, double *Vx_c, double *Vy_c, double *Vz_c, const double *Vx, const double *Vy, const double *Vz, int64_t nx,
                                 int64_t ny, int64_t nz)
{
    if (!h) return JRX_ERR_ARG;
    JRX_TRY(jrx_check_device(h));
    if (!Vx_c || !Vy_c || !Vz_c || !Vx || !Vy || !Vz || nx < 1 || ny < 1 || nz < 1) return jrx_fail(h, JRX_ERR_ARG, "velocity2center!: bad argument");
    hipLaunchKernelGGL(k_vel2center3d, OUT_GRID(nx, ny, nz), dim3(256), 0, h->stream, Vx_c, Vy_c, Vz_c, Vx, Vy, Vz, (int)nx, (int)ny);
    return done(h);
}

jrx_status jrx_vertex2center(jrx_handle *h, double *center, const double *vertex, const int64_t vdim[3], const int64_t cdim[3], int32_t ndim, int32_t ghost_x,
                             int32_t ghost_y, int32_t ghost_z)
{
    if (!h) return JRX_ERR_ARG;
    JRX_TRY(jrx_check_device(h));
    if (!center || !vertex || !vdim || !cdim || (ndim != 2 && ndim != 3)) return jrx_fail(h, JRX_ERR_ARG, "vertex2center!: bad argument");
    const int g[3] = {ghost_x != 0, ghost_y != 0, ndim == 3 && ghost_z != 0};
    for (int d = 0; d < ndim; d++)
        if (vdim[d] < 2 || vdim[d] - 1 + g[d] > cdim[d])
            return jrx_fail(h, JRX_ERR_ARG, "vertex2center!: the centre array is too small along dimension %d (size(vertex) - 1 + ghost = %lld > %lld)", d + 1,
                            (long long)(vdim[d] - 1 + g[d]), (long long)cdim[d]);
    if (ndim == 3)
        hipLaunchKernelGGL(k_vertex2center<true>, OUT_GRID(vdim[0] - 1, vdim[1] - 1, vdim[2] - 1), dim3(256), 0, h->stream, center, vertex, (int)vdim[0],
                           (int)vdim[1], (int)cdim[0], (int)cdim[1], g[0], g[1], g[2]);
    else
        hipLaunchKernelGGL(k_vertex2center<false>, OUT_GRID(vdim[0] - 1, vdim[1] - 1, 1), dim3(256), 0, h->stream, center, vertex, (int)vdim[0], (int)vdim[1],
                           (int)cdim[0], (int)cdim[1], g[0], g[1], 0);
    return done(h);
}

jrx_status jrx_center2vertex_harm2d(jrx_handle *h, double *vertex, const double *center, int64_t nx, int64_t ny)
{
    if (!h) return JRX_ERR_ARG;
    JRX_TRY(jrx_check_device(h));
    if (!vertex || !center || nx < 1 || ny < 1) return jrx_fail(h, JRX_ERR_ARG, "center2vertex_harm!: bad argument");
    hipLaunchKernelGGL(k_center2vertex_harm2d, OUT_GRID(nx + 1, ny + 1, 1), dim3(256), 0, h->stream, vertex, center, (int)nx, (int)ny);
    return done(h);
}

jrx_status jrx_center2vertex3d(jrx_handle *h, double *vertex_yz, double *vertex_xz, double *vertex_xy, const double *center_yz, const double *center_xz,
                               const double *center_xy, int64_t nx, int64_t ny, int64_t nz)
{
    if (!h) return JRX_ERR_ARG;
    JRX_TRY(jrx_check_device(h));
    if (!vertex_yz || !vertex_xz || !vertex_xy || !center_yz || !center_xz || !center_xy || nx < 1 || ny < 1 || nz < 1)
        return jrx_fail(h, JRX_ERR_ARG, "center2vertex!: bad argument");
    hipLaunchKernelGGL(k_center2vertex3d, OUT_GRID(nx, ny, nz), dim3(256), 0, h->stream, vertex_yz, vertex_xz, vertex_xy, center_yz, center_xz, center_xy,
                       (int)nx, (int)ny, (int)nz);
    return done(h);
}

jrx_status jrx_compute_rhog(jrx_handle *h, double *rhog, const jrx_rheology *rh, const double *phase_c, const double *T, const double *P, const int64_t n[3],
                            const int64_t tdim[3], int32_t ndim)
{
    if (!h) return JRX_ERR_ARG;
    JRX_TRY(jrx_check_device(h));
    if (!rhog || !rh || !n || (ndim != 2 && ndim != 3) || rh->nphase < 1 || rh->nphase > JRX_MAXPHASE) return jrx_fail(h, JRX_ERR_ARG, "compute_ρg!: bad argument");
    if (!rh->has_density) return jrx_fail(h, JRX_ERR_ARG, "compute_ρg!: the rheology table carries no density law (has_density = 0)");
    const int nx = (int)n[0], ny = (int)n[1], nz = ndim == 3 ? (int)n[2] : 1;
    if (nx < 1 || ny < 1 || nz < 1) return jrx_fail(h, JRX_ERR_ARG, "compute_ρg!: bad size");
    int t1 = nx, t2 = ny;
    if (tdim) {
        for (int d = 0; d < ndim; d++)
            if (tdim[d] < n[d]) return jrx_fail(h, JRX_ERR_ARG, "compute_ρg!: args.T is smaller than ρg along dimension %d", d + 1);
        t1 = (int)tdim[0]; t2 = (int)tdim[1];
    }
    if (phase_c) hipLaunchKernelGGL(k_compute_rhog<true>, OUT_GRID(nx, ny, nz), dim3(256), 0, h->stream, rhog, *rh, phase_c, T, P, nx, ny, t1, t2);
    else hipLaunchKernelGGL(k_compute_rhog<false>, OUT_GRID(nx, ny, nz), dim3(256), 0, h->stream, rhog, *rh, phase_c, T, P, nx, ny, t1, t2);
    return done(h);
}

jrx_status jrx_compute_lithostatic_pressure(jrx_handle *h, double *P, const double *rhog, double dz, const double *dz_cells, const int64_t n[3], int32_t ndim)
{
    if (!h) return JRX_ERR_ARG;
    JRX_TRY(jrx_check_device(h));
    if (!P || !rhog || !n || (ndim != 2 && ndim != 3)) return jrx_fail(h, JRX_ERR_ARG, "compute_lithostatic_pressure!: bad argument");
    for (int d = 0; d < ndim; d++)
        if (n[d] < 1) return jrx_fail(h, JRX_ERR_ARG, "compute_lithostatic_pressure!: bad size");
    if (jrx_comm_active(h) && (jrx_comm_has_neighbor(h, ndim - 1, 0) || jrx_comm_has_neighbor(h, ndim - 1, 1)))
        return jrx_fail(h, JRX_ERR_UNSUPPORTED, "compute_lithostatic_pressure!: the vertical direction is split across ranks (the IGG form that gathers the weight of the "
                                                "ranks above is not built)");
    const i64 ncol = ndim == 3 ? (i64)n[0] * n[1] : (i64)n[0];
    hipLaunchKernelGGL(k_lithostatic, dim3((unsigned)((ncol + 255) / 256)), dim3(256), 0, h->stream, P, rhog, dz_cells, dz, ncol, (int)n[ndim - 1]);
    return done(h);
}

jrx_status jrx_compute_viscosity_single(jrx_handle *h, double *eta, const jrx_rheology *rh, const double *T, const double *P, const int64_t n[3],
                                        const int64_t tdim[3], int32_t ndim, double nu, double cutoff_lo, double cutoff_hi, const double *AII, int32_t tau_form)
{
    if (!h) return JRX_ERR_ARG;
    JRX_TRY(jrx_check_device(h));
    if (!eta || !rh || !n || (ndim != 2 && ndim != 3) || rh->nphase < 1) return jrx_fail(h, JRX_ERR_ARG, "compute_viscosity!: bad argument");
    const int nx = (int)n[0], ny = (int)n[1], nz = ndim == 3 ? (int)n[2] : 1;
    if (nx < 1 || ny < 1 || nz < 1) return jrx_fail(h, JRX_ERR_ARG, "compute_viscosity!: bad size");
    int t1 = nx, t2 = ny, sh = 0;
    if (T && tdim) {
        bool same = true, ghosted = true;
        for (int d = 0; d < ndim; d++) { same = same && tdim[d] == n[d]; ghosted = ghosted && tdim[d] == n[d] + 2; }
        if (!same && !ghosted) return jrx_fail(h, JRX_ERR_ARG, "compute_viscosity!: args.T must be ni (cell centres) or ni .+ 2 (thermal.T, read at I .+ 1)");
        t1 = (int)tdim[0]; t2 = (int)tdim[1]; sh = ghosted ? 1 : 0;
    }
    if (rh->visc_kind[0] == 2 && !AII)
        return jrx_fail(h, JRX_ERR_ARG, "compute_viscosity!: a power-law creep needs the invariant array (compute_viscosity_εII!(η, ν, εII, args, rheology, cutoff))");
    hipLaunchKernelGGL(k_viscosity_single, OUT_GRID(nx, ny, nz), dim3(256), 0, h->stream, eta, *rh, T, P, nx, ny, t1, t2, sh, ndim == 3 ? sh : 0, nu, cutoff_lo,
                       cutoff_hi, AII, tau_form != 0);
    return done(h);
}

jrx_status jrx_compute_shear_heating(jrx_handle *h, double *shear_heating, const double *const *tau, const double *const *tau_o, const double *const *eps,
                                     const double *phase_c, const jrx_rheology *rh, const double *chi, double dt, const int64_t n[3], int32_t ndim)
{
    if (!h) return JRX_ERR_ARG;
    JRX_TRY(jrx_check_device(h));
    if (!shear_heating || !tau || !tau_o || !eps || !rh || !chi || !n || (ndim != 2 && ndim != 3) || rh->nphase < 1 || rh->nphase > JRX_MAXPHASE)
        return jrx_fail(h, JRX_ERR_ARG, "compute_shear_heating!: bad argument");
    const int N = ndim == 3 ? 6 : 3;
    ShArgs a{};
    a.sh = shear_heating; a.phase_c = phase_c; a.dt = dt; a.nphase = rh->nphase;
    a.nx = (int)n[0]; a.ny = (int)n[1]; a.nz = ndim == 3 ? (int)n[2] : 1;
    if (a.nx < 1 || a.ny < 1 || a.nz < 1) return jrx_fail(h, JRX_ERR_ARG, "compute_shear_heating!: bad size");
    for (int q = 0; q < N; q++) {
        if (!tau[q] || !tau_o[q] || !eps[q]) return jrx_fail(h, JRX_ERR_ARG, "compute_shear_heating!: component %d of a tensor is NULL", q);
        a.t[q] = tau[q]; a.to[q] = tau_o[q]; a.e[q] = eps[q];
    }
    for (int q = 0; q < rh->nphase; q++) { a.G[q] = rh->G[q]; a.chi[q] = chi[q]; }
    if (ndim == 3) hipLaunchKernelGGL(k_shear_heating<true>, OUT_GRID(a.nx, a.ny, a.nz), dim3(256), 0, h->stream, a);
    else hipLaunchKernelGGL(k_shear_heating<false>, OUT_GRID(a.nx, a.ny, 1), dim3(256), 0, h->stream, a);
    return done(h);
}

}  // extern "C"
