// stokes3d_kernels.hpp -- device kernels of the 3D Stokes PT iteration (shared by libjrx_hip and
// the kernel micro-benchmark scripts/kbench.hip).  See stokes3d.hip for the reference citations.
#pragma once
#include "jrx_internal.hpp"

namespace {

struct Dims3 {
    int nx, ny, nz;
};

// element strides of the staggered arrays
struct Lay3 {
    int nx, ny, nz;
    // row lengths (n1) and plane sizes (n1*n2)
    int vx1, vy1, vz1;          // nx+1, nx+2, nx+2
    i64 vxp, vyp, vzp;          // plane sizes
    i64 cp;                     // nx*ny
    int xy1; i64 xyp;           // (nx+1), (nx+1)*(ny+1)
    int xz1; i64 xzp;           // (nx+1), (nx+1)*ny
    int yz1; i64 yzp;           // nx, nx*(ny+1)
};

__host__ __device__ inline Lay3 make_lay(int nx, int ny, int nz)
{
    Lay3 L;
    L.nx = nx; L.ny = ny; L.nz = nz;
    L.vx1 = nx + 1; L.vxp = (i64)(nx + 1) * (ny + 2);
    L.vy1 = nx + 2; L.vyp = (i64)(nx + 2) * (ny + 1);
    L.vz1 = nx + 2; L.vzp = (i64)(nx + 2) * (ny + 2);
    L.cp = (i64)nx * ny;
    L.xy1 = nx + 1; L.xyp = (i64)(nx + 1) * (ny + 1);
    L.xz1 = nx + 1; L.xzp = (i64)(nx + 1) * ny;
    L.yz1 = nx;     L.yzp = (i64)nx * (ny + 1);
    return L;
}

struct SweepArgs {
    jrx_stokes3d_fields f;
    const double *etatau;
    double _dx, _dy, _dz, dt, r, theta_dtau, eta_dtau;
    Lay3 L;
    // sub-box of the launch (0-based, half-open) -- lets the driver split boundary slabs / interior
    int i0, i1, j0, j1, k0, k1;
};

// ------------------------------------------------------------------------------------------------
// Stress sweep, version 1: one thread per node of the ni.+1 box, xy-plane flattened over threadIdx
// so that rows of any length (nx, nx+1, nx+2) stay fully coalesced; blockIdx.y walks z.
// ------------------------------------------------------------------------------------------------
template <bool DIAG>
__global__ __launch_bounds__(256) void k_stress3d(const SweepArgs a)
{
    const Lay3 &L = a.L;
    const int nx = L.nx, ny = L.ny, nz = L.nz;
    const int wi = a.i1 - a.i0;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int jj = t / wi;
    const int i = a.i0 + (t - jj * wi);
    const int j = a.j0 + jj;
    const int k = a.k0 + blockIdx.y;
    if (j >= a.j1) return;

    const double *__restrict__ Vx = a.f.Vx, *__restrict__ Vy = a.f.Vy, *__restrict__ Vz = a.f.Vz;
    const double *__restrict__ eta = a.f.eta, *__restrict__ G = a.f.G;
    const double _dx = a._dx, _dy = a._dy, _dz = a._dz, dt = a.dt, th = a.theta_dtau;

#define VX(i_, j_, k_) Vx[(i_) + (i64)L.vx1 * (j_) + L.vxp * (k_)]
#define VY(i_, j_, k_) Vy[(i_) + (i64)L.vy1 * (j_) + L.vyp * (k_)]
#define VZ(i_, j_, k_) Vz[(i_) + (i64)L.vz1 * (j_) + L.vzp * (k_)]
#define CC(i_, j_, k_) ((i_) + (i64)nx * (j_) + L.cp * (k_))

    const bool ci = i < nx, cj = j < ny, ck = k < nz;

    if (ci && cj && ck) {
        const i64 c = CC(i, j, k);
        // compute_∇V! (VelocityKernels.jl:3-6)
        const double dxi = (-VX(i, j + 1, k + 1) + VX(i + 1, j + 1, k + 1)) * _dx;
        const double dyi = (-VY(i + 1, j, k + 1) + VY(i + 1, j + 1, k + 1)) * _dy;
        const double dzi = (-VZ(i + 1, j + 1, k) + VZ(i + 1, j + 1, k + 1)) * _dz;
        const double divV = dxi + dyi + dzi;
        // compute_P! (PressureKernels.jl:186-195), η (not ητ) in the 3D driver (Stokes3D.jl:85)
        const double e = eta[c];
        const double _Gdt = 1.0 / (G[c] * dt);
        {
            const double _Kdt = 1.0 / (a.f.K[c] * dt);
            const double _dt = 1.0 / dt;
            const double P = a.f.P[c], P0 = a.f.P0[c];
            const double rhs = -divV + (a.f.Q[c] * _dt);
            const double psi = 1.0 / (1.0 / e + _Gdt) * a.r / th;
            a.f.P[c] = (fma(P0, _Kdt, rhs) * psi + P) / (1.0 + _Kdt * psi);
            if (DIAG) {
                a.f.RP[c] = fma(-(P - P0), _Kdt, rhs);
                a.f.divV[c] = divV;
            }
        }
        // compute_strain_rate! normal components (VelocityKernels.jl:69-78)
        const double d3 = divV * (1.0 / 3.0);
        const double exx = dxi - d3, eyy = dyi - d3, ezz = dzi - d3;
        if (DIAG) { a.f.exx[c] = exx; a.f.eyy[c] = eyy; a.f.ezz[c] = ezz; }
        // compute_τ! normal components (StressKernels.jl:185-198)
        const double dtr = dev_dtau_r(th, e, _Gdt);
        double tv;
        tv = a.f.txx[c]; a.f.txx[c] = tv + dev_stress_inc(tv, a.f.toxx[c], e, exx, _Gdt, dtr);
        tv = a.f.tyy[c]; a.f.tyy[c] = tv + dev_stress_inc(tv, a.f.toyy[c], e, eyy, _Gdt, dtr);
        tv = a.f.tzz[c]; a.f.tzz[c] = tv + dev_stress_inc(tv, a.f.tozz[c], e, ezz, _Gdt, dtr);
    }

    // clamped neighbour cell indices (MiniKernels.jl:133-147)
    const int im = max(i - 1, 0), ip = min(i, nx - 1);
    const int jm = max(j - 1, 0), jp = min(j, ny - 1);
    const int km = max(k - 1, 0), kp = min(k, nz - 1);

    if (ck) {   // τxy at (i,j,k) of (nx+1, ny+1, nz)   (VelocityKernels.jl:95-101, StressKernels.jl:199-208)
        const double exy = 0.5 * (_dy * (VX(i, j + 1, k + 1) - VX(i, j, k + 1)) + _dx * (VY(i + 1, j, k + 1) - VY(i, j, k + 1)));
        const double e = 0.25 * (eta[CC(im, jm, k)] + eta[CC(ip, jm, k)] + eta[CC(im, jp, k)] + eta[CC(ip, jp, k)]);
        const double g = 0.25 * (G[CC(im, jm, k)] + G[CC(ip, jm, k)] + G[CC(im, jp, k)] + G[CC(ip, jp, k)]);
        const double _Gdt = 1.0 / (g * dt);
        const double dtr = dev_dtau_r(th, e, _Gdt);
        const i64 c = i + (i64)L.xy1 * j + L.xyp * k;
        const double tv = a.f.txy[c];
        a.f.txy[c] = tv + dev_stress_inc(tv, a.f.toxy[c], e, exy, _Gdt, dtr);
        if (DIAG) a.f.exy[c] = exy;
    }
    if (cj) {   // τxz at (i,j,k) of (nx+1, ny, nz+1)
        const double exz = 0.5 * (_dz * (VX(i, j + 1, k + 1) - VX(i, j + 1, k)) + _dx * (VZ(i + 1, j + 1, k) - VZ(i, j + 1, k)));
        const double e = 0.25 * (eta[CC(im, j, km)] + eta[CC(ip, j, km)] + eta[CC(im, j, kp)] + eta[CC(ip, j, kp)]);
        const double g = 0.25 * (G[CC(im, j, km)] + G[CC(ip, j, km)] + G[CC(im, j, kp)] + G[CC(ip, j, kp)]);
        const double _Gdt = 1.0 / (g * dt);
        const double dtr = dev_dtau_r(th, e, _Gdt);
        const i64 c = i + (i64)L.xz1 * j + L.xzp * k;
        const double tv = a.f.txz[c];
        a.f.txz[c] = tv + dev_stress_inc(tv, a.f.toxz[c], e, exz, _Gdt, dtr);
        if (DIAG) a.f.exz[c] = exz;
    }
    if (ci) {   // τyz at (i,j,k) of (nx, ny+1, nz+1)
        const double eyz = 0.5 * (_dz * (VY(i + 1, j, k + 1) - VY(i + 1, j, k)) + _dy * (VZ(i + 1, j + 1, k) - VZ(i + 1, j, k)));
        const double e = 0.25 * (eta[CC(i, jm, km)] + eta[CC(i, jp, km)] + eta[CC(i, jm, kp)] + eta[CC(i, jp, kp)]);
        const double g = 0.25 * (G[CC(i, jm, km)] + G[CC(i, jp, km)] + G[CC(i, jm, kp)] + G[CC(i, jp, kp)]);
        const double _Gdt = 1.0 / (g * dt);
        const double dtr = dev_dtau_r(th, e, _Gdt);
        const i64 c = i + (i64)L.yz1 * j + L.yzp * k;
        const double tv = a.f.tyz[c];
        a.f.tyz[c] = tv + dev_stress_inc(tv, a.f.toyz[c], e, eyz, _Gdt, dtr);
        if (DIAG) a.f.eyz[c] = eyz;
    }
#undef VX
#undef VY
#undef VZ
}

// ------------------------------------------------------------------------------------------------
// Velocity sweep, version 1: one thread per cell; compute_V! (VelocityKernels.jl:182-242).
// ------------------------------------------------------------------------------------------------
template <bool DIAG>
__global__ __launch_bounds__(256) void k_velocity3d(const SweepArgs a)
{
    const Lay3 &L = a.L;
    const int nx = L.nx, ny = L.ny, nz = L.nz;
    const int wi = a.i1 - a.i0;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int jj = t / wi;
    const int i = a.i0 + (t - jj * wi);
    const int j = a.j0 + jj;
    const int k = a.k0 + blockIdx.y;
    if (j >= a.j1) return;

    const double _dx = a._dx, _dy = a._dy, _dz = a._dz, edt = a.eta_dtau;
    const double *__restrict__ P = a.f.P, *__restrict__ et = a.etatau;
    const double *__restrict__ txy = a.f.txy, *__restrict__ txz = a.f.txz, *__restrict__ tyz = a.f.tyz;
#define TXY(i_, j_, k_) txy[(i_) + (i64)L.xy1 * (j_) + L.xyp * (k_)]
#define TXZ(i_, j_, k_) txz[(i_) + (i64)L.xz1 * (j_) + L.xzp * (k_)]
#define TYZ(i_, j_, k_) tyz[(i_) + (i64)L.yz1 * (j_) + L.yzp * (k_)]
    const i64 c = CC(i, j, k);
    const double Pc = P[c], ec = et[c];

    if (i < nx - 1) {
        const i64 cx = c + 1;
        const double R = (-a.f.txx[c] + a.f.txx[cx]) * _dx + _dy * (TXY(i + 1, j + 1, k) - TXY(i + 1, j, k)) +
                         _dz * (TXZ(i + 1, j, k + 1) - TXZ(i + 1, j, k)) - (-Pc + P[cx]) * _dx -
                         0.5 * (a.f.fx[c] + a.f.fx[cx]);
        const i64 v = (i + 1) + (i64)L.vx1 * (j + 1) + L.vxp * (k + 1);
        a.f.Vx[v] += R * edt / (0.5 * (ec + et[cx]));
        if (DIAG) a.f.Rx[i + (i64)(nx - 1) * j + (i64)(nx - 1) * ny * k] = R;
    }
    if (j < ny - 1) {
        const i64 cy = c + nx;
        const double R = _dx * (TXY(i + 1, j + 1, k) - TXY(i, j + 1, k)) + _dy * (a.f.tyy[cy] - a.f.tyy[c]) +
                         _dz * (TYZ(i, j + 1, k + 1) - TYZ(i, j + 1, k)) - (-Pc + P[cy]) * _dy -
                         0.5 * (a.f.fy[c] + a.f.fy[cy]);
        const i64 v = (i + 1) + (i64)L.vy1 * (j + 1) + L.vyp * (k + 1);
        a.f.Vy[v] += R * edt / (0.5 * (ec + et[cy]));
        if (DIAG) a.f.Ry[i + (i64)nx * j + (i64)nx * (ny - 1) * k] = R;
    }
    if (k < nz - 1) {
        const i64 cz = c + L.cp;
        const double R = _dx * (TXZ(i + 1, j, k + 1) - TXZ(i, j, k + 1)) + _dy * (TYZ(i, j + 1, k + 1) - TYZ(i, j, k + 1)) +
                         (-a.f.tzz[c] + a.f.tzz[cz]) * _dz - (-Pc + P[cz]) * _dz - 0.5 * (a.f.fz[c] + a.f.fz[cz]);
        const i64 v = (i + 1) + (i64)L.vz1 * (j + 1) + L.vzp * (k + 1);
        a.f.Vz[v] += R * edt / (0.5 * (ec + et[cz]));
        if (DIAG) a.f.Rz[c] = R;
    }
#undef TXY
#undef TXZ
#undef TYZ
#undef CC
}


// ================================================================================================
// Version 2 ("zm"): 2.5D z-marching sweeps.  A block owns a TX x TY tile of cell columns and walks
// KZ planes in z; the values that the next plane needs again (the k+1 velocity plane, the k plane
// of η and G, the upper τ/P/f/ητ plane of the velocity sweep) stay in registers instead of being
// re-read, x/y neighbours of the same plane are served by L1 (same or adjacent wave), so HBM sees
// each array plane once per sweep apart from tile halos.
// ================================================================================================
#define CC3(i_, j_, k_) ((i_) + (i64)nx * (j_) + L.cp * (k_))

// generic (all loads from memory) shear-node updates, used for the few nodes on the upper
// boundary planes i = nx, j = ny, k = nz that no cell-column thread owns
template <bool DIAG>
__device__ __forceinline__ void node_xy(const SweepArgs &a, int I, int J, int k)
{
    const Lay3 &L = a.L;
    const int nx = L.nx, ny = L.ny;
    const double *Vx = a.f.Vx, *Vy = a.f.Vy, *eta = a.f.eta, *G = a.f.G;
    const int im = max(I - 1, 0), ip = min(I, nx - 1), jm = max(J - 1, 0), jp = min(J, ny - 1);
    const double exy = 0.5 * (a._dy * (Vx[I + (i64)L.vx1 * (J + 1) + L.vxp * (k + 1)] - Vx[I + (i64)L.vx1 * J + L.vxp * (k + 1)]) +
                              a._dx * (Vy[(I + 1) + (i64)L.vy1 * J + L.vyp * (k + 1)] - Vy[I + (i64)L.vy1 * J + L.vyp * (k + 1)]));
    const double e = 0.25 * (eta[CC3(im, jm, k)] + eta[CC3(ip, jm, k)] + eta[CC3(im, jp, k)] + eta[CC3(ip, jp, k)]);
    const double g = 0.25 * (G[CC3(im, jm, k)] + G[CC3(ip, jm, k)] + G[CC3(im, jp, k)] + G[CC3(ip, jp, k)]);
    const double _Gdt = 1.0 / (g * a.dt);
    const double dtr = dev_dtau_r(a.theta_dtau, e, _Gdt);
    const i64 c = I + (i64)L.xy1 * J + L.xyp * k;
    const double tv = a.f.txy[c];
    a.f.txy[c] = tv + dev_stress_inc(tv, a.f.toxy[c], e, exy, _Gdt, dtr);
    if (DIAG) a.f.exy[c] = exy;
}
template <bool DIAG>
__device__ __forceinline__ void node_xz(const SweepArgs &a, int I, int j, int Kk)
{
    const Lay3 &L = a.L;
    const int nx = L.nx, nz = L.nz;
    const double *Vx = a.f.Vx, *Vz = a.f.Vz, *eta = a.f.eta, *G = a.f.G;
    const int im = max(I - 1, 0), ip = min(I, nx - 1), km = max(Kk - 1, 0), kp = min(Kk, nz - 1);
    const double exz = 0.5 * (a._dz * (Vx[I + (i64)L.vx1 * (j + 1) + L.vxp * (Kk + 1)] - Vx[I + (i64)L.vx1 * (j + 1) + L.vxp * Kk]) +
                              a._dx * (Vz[(I + 1) + (i64)L.vz1 * (j + 1) + L.vzp * Kk] - Vz[I + (i64)L.vz1 * (j + 1) + L.vzp * Kk]));
    const double e = 0.25 * (eta[CC3(im, j, km)] + eta[CC3(ip, j, km)] + eta[CC3(im, j, kp)] + eta[CC3(ip, j, kp)]);
    const double g = 0.25 * (G[CC3(im, j, km)] + G[CC3(ip, j, km)] + G[CC3(im, j, kp)] + G[CC3(ip, j, kp)]);
    const double _Gdt = 1.0 / (g * a.dt);
    const double dtr = dev_dtau_r(a.theta_dtau, e, _Gdt);
    const i64 c = I + (i64)L.xz1 * j + L.xzp * Kk;
    const double tv = a.f.txz[c];
    a.f.txz[c] = tv + dev_stress_inc(tv, a.f.toxz[c], e, exz, _Gdt, dtr);
    if (DIAG) a.f.exz[c] = exz;
}
template <bool DIAG>
__device__ __forceinline__ void node_yz(const SweepArgs &a, int i, int J, int Kk)
{
    const Lay3 &L = a.L;
    const int nx = L.nx, ny = L.ny, nz = L.nz;
    const double *Vy = a.f.Vy, *Vz = a.f.Vz, *eta = a.f.eta, *G = a.f.G;
    const int jm = max(J - 1, 0), jp = min(J, ny - 1), km = max(Kk - 1, 0), kp = min(Kk, nz - 1);
    const double eyz = 0.5 * (a._dz * (Vy[(i + 1) + (i64)L.vy1 * J + L.vyp * (Kk + 1)] - Vy[(i + 1) + (i64)L.vy1 * J + L.vyp * Kk]) +
                              a._dy * (Vz[(i + 1) + (i64)L.vz1 * (J + 1) + L.vzp * Kk] - Vz[(i + 1) + (i64)L.vz1 * J + L.vzp * Kk]));
    const double e = 0.25 * (eta[CC3(i, jm, km)] + eta[CC3(i, jp, km)] + eta[CC3(i, jm, kp)] + eta[CC3(i, jp, kp)]);
    const double g = 0.25 * (G[CC3(i, jm, km)] + G[CC3(i, jp, km)] + G[CC3(i, jm, kp)] + G[CC3(i, jp, kp)]);
    const double _Gdt = 1.0 / (g * a.dt);
    const double dtr = dev_dtau_r(a.theta_dtau, e, _Gdt);
    const i64 c = i + (i64)L.yz1 * J + L.yzp * Kk;
    const double tv = a.f.tyz[c];
    a.f.tyz[c] = tv + dev_stress_inc(tv, a.f.toyz[c], e, eyz, _Gdt, dtr);
    if (DIAG) a.f.eyz[c] = eyz;
}

template <bool DIAG, int TX, int TY, int KZ>
__global__ __launch_bounds__(TX *TY) void k_stress3d_zm(const SweepArgs a)
{
    const Lay3 &L = a.L;
    const int nx = L.nx, ny = L.ny, nz = L.nz;
    const int i = blockIdx.x * TX + threadIdx.x;
    const int j = blockIdx.y * TY + threadIdx.y;
    const int kb = blockIdx.z * KZ;
    if (i >= nx || j >= ny) return;
    const int kend = min(kb + KZ, nz);
    const int im = max(i - 1, 0), jm = max(j - 1, 0);
    const bool xhi = (i == nx - 1), yhi = (j == ny - 1);
    const double _dx = a._dx, _dy = a._dy, _dz = a._dz, dt = a.dt, th = a.theta_dtau, _dt = 1.0 / dt, rr = a.r;

    const double *__restrict__ Vx = a.f.Vx, *__restrict__ Vy = a.f.Vy, *__restrict__ Vz = a.f.Vz;
    const double *__restrict__ eta = a.f.eta, *__restrict__ G = a.f.G;

    // running offsets (advance by one plane per step)
    i64 oc = CC3(i, j, kb);                                             // centre arrays, level k
    i64 ocx = CC3(im, j, kb), ocy = CC3(i, jm, kb), ocxy = CC3(im, jm, kb);
    i64 ovx = i + (i64)L.vx1 * (j + 1) + L.vxp * (kb + 1);            // Vx[i, j+1, k+1]
    i64 ovy = (i + 1) + (i64)L.vy1 * j + L.vyp * (kb + 1);            // Vy[i+1, j, k+1]
    i64 ovz = (i + 1) + (i64)L.vz1 * (j + 1) + L.vzp * (kb + 1);      // Vz[i+1, j+1, k+1]
    i64 oxy = i + (i64)L.xy1 * j + L.xyp * kb;
    i64 oxz = i + (i64)L.xz1 * j + L.xzp * kb;
    i64 oyz = i + (i64)L.yz1 * j + L.yzp * kb;

    // carried values: level k of V, level k-1 (clamped) of η, G
    double a_p = Vx[ovx - L.vxp], b_p = Vy[ovy - L.vyp];
    double c_p = Vz[ovz - L.vzp], cx_p = Vz[ovz - L.vzp - 1], cy_p = Vz[ovz - L.vzp - L.vz1];
    const i64 back = kb > 0 ? L.cp : 0;
    double e_p = eta[oc - back], ex_p = eta[ocx - back], ey_p = eta[ocy - back];
    double g_p = G[oc - back], gx_p = G[ocx - back], gy_p = G[ocy - back];

    for (int k = kb; k < kend; ++k) {
        // level k+1 velocities
        const double va = Vx[ovx], vax = Vx[ovx + 1], vay = Vx[ovx - L.vx1];
        const double vb = Vy[ovy], vby = Vy[ovy + L.vy1], vbx = Vy[ovy - 1];
        const double vc = Vz[ovz], vcx = Vz[ovz - 1], vcy = Vz[ovz - L.vz1];
        // level k material
        const double e = eta[oc], ex = eta[ocx], ey = eta[ocy], exy_ = eta[ocxy];
        const double g = G[oc], gx = G[ocx], gy = G[ocy], gxy = G[ocxy];

        {   // centre (i,j,k)
            const double dxi = (-va + vax) * _dx;
            const double dyi = (-vb + vby) * _dy;
            const double dzi = (-c_p + vc) * _dz;
            const double divV = dxi + dyi + dzi;
            const double _Gdt = 1.0 / (g * dt);
            const double _Kdt = 1.0 / (a.f.K[oc] * dt);
            const double P = a.f.P[oc], P0 = a.f.P0[oc];
            const double rhs = -divV + (a.f.Q[oc] * _dt);
            const double psi = 1.0 / (1.0 / e + _Gdt) * rr / th;
            a.f.P[oc] = (fma(P0, _Kdt, rhs) * psi + P) / (1.0 + _Kdt * psi);
            const double d3 = divV * (1.0 / 3.0);
            const double exx = dxi - d3, eyy = dyi - d3, ezz = dzi - d3;
            if (DIAG) {
                a.f.RP[oc] = fma(-(P - P0), _Kdt, rhs);
                a.f.divV[oc] = divV;
                a.f.exx[oc] = exx; a.f.eyy[oc] = eyy; a.f.ezz[oc] = ezz;
            }
            const double dtr = dev_dtau_r(th, e, _Gdt);
            double tv;
            tv = a.f.txx[oc]; a.f.txx[oc] = tv + dev_stress_inc(tv, a.f.toxx[oc], e, exx, _Gdt, dtr);
            tv = a.f.tyy[oc]; a.f.tyy[oc] = tv + dev_stress_inc(tv, a.f.toyy[oc], e, eyy, _Gdt, dtr);
            tv = a.f.tzz[oc]; a.f.tzz[oc] = tv + dev_stress_inc(tv, a.f.tozz[oc], e, ezz, _Gdt, dtr);
        }
        {   // τxy (i,j,k)
            const double sxy = 0.5 * (_dy * (va - vay) + _dx * (vb - vbx));
            const double ee = 0.25 * (exy_ + ey + ex + e);
            const double gg = 0.25 * (gxy + gy + gx + g);
            const double _Gdt = 1.0 / (gg * dt);
            const double dtr = dev_dtau_r(th, ee, _Gdt);
            const double tv = a.f.txy[oxy];
            a.f.txy[oxy] = tv + dev_stress_inc(tv, a.f.toxy[oxy], ee, sxy, _Gdt, dtr);
            if (DIAG) a.f.exy[oxy] = sxy;
        }
        {   // τxz (i,j,k)
            const double sxz = 0.5 * (_dz * (va - a_p) + _dx * (c_p - cx_p));
            const double ee = 0.25 * (ex_p + e_p + ex + e);
            const double gg = 0.25 * (gx_p + g_p + gx + g);
            const double _Gdt = 1.0 / (gg * dt);
            const double dtr = dev_dtau_r(th, ee, _Gdt);
            const double tv = a.f.txz[oxz];
            a.f.txz[oxz] = tv + dev_stress_inc(tv, a.f.toxz[oxz], ee, sxz, _Gdt, dtr);
            if (DIAG) a.f.exz[oxz] = sxz;
        }
        {   // τyz (i,j,k)
            const double syz = 0.5 * (_dz * (vb - b_p) + _dy * (c_p - cy_p));
            const double ee = 0.25 * (ey_p + e_p + ey + e);
            const double gg = 0.25 * (gy_p + g_p + gy + g);
            const double _Gdt = 1.0 / (gg * dt);
            const double dtr = dev_dtau_r(th, ee, _Gdt);
            const double tv = a.f.tyz[oyz];
            a.f.tyz[oyz] = tv + dev_stress_inc(tv, a.f.toyz[oyz], ee, syz, _Gdt, dtr);
            if (DIAG) a.f.eyz[oyz] = syz;
        }
        // upper boundary planes i = nx, j = ny (one extra node column per boundary thread)
        if (xhi) { node_xy<DIAG>(a, nx, j, k); node_xz<DIAG>(a, nx, j, k); }
        if (yhi) { node_xy<DIAG>(a, i, ny, k); node_yz<DIAG>(a, i, ny, k); }
        if (xhi && yhi) node_xy<DIAG>(a, nx, ny, k);

        a_p = va; b_p = vb; c_p = vc; cx_p = vcx; cy_p = vcy;
        e_p = e; ex_p = ex; ey_p = ey; g_p = g; gx_p = gx; gy_p = gy;
        oc += L.cp; ocx += L.cp; ocy += L.cp; ocxy += L.cp;
        ovx += L.vxp; ovy += L.vyp; ovz += L.vzp;
        oxy += L.xyp; oxz += L.xzp; oyz += L.yzp;
    }
    if (kend == nz) {   // top plane k = nz of the xz / yz nodes
        node_xz<DIAG>(a, i, j, nz);
        node_yz<DIAG>(a, i, j, nz);
        if (xhi) node_xz<DIAG>(a, nx, j, nz);
        if (yhi) node_yz<DIAG>(a, i, ny, nz);
    }
}

// Velocity sweep, z-marching.  Sub-box [i0,i1) x [j0,j1) x [k0,k1) of the cell box.
template <bool DIAG, int TX, int TY, int KZ>
__global__ __launch_bounds__(TX *TY) void k_velocity3d_zm(const SweepArgs a)
{
    const Lay3 &L = a.L;
    const int nx = L.nx, ny = L.ny, nz = L.nz;
    const int i = a.i0 + blockIdx.x * TX + threadIdx.x;
    const int j = a.j0 + blockIdx.y * TY + threadIdx.y;
    const int kb = a.k0 + blockIdx.z * KZ;
    if (i >= a.i1 || j >= a.j1) return;
    const int kend = min(kb + KZ, a.k1);
    const double _dx = a._dx, _dy = a._dy, _dz = a._dz, edt = a.eta_dtau;
    const bool hx = i < nx - 1, hy = j < ny - 1;
    const double *__restrict__ P = a.f.P, *__restrict__ et = a.etatau;
    const double *__restrict__ txy = a.f.txy, *__restrict__ txz = a.f.txz, *__restrict__ tyz = a.f.tyz;

    i64 oc = CC3(i, j, kb);
    i64 oxy = i + (i64)L.xy1 * j + L.xyp * kb;          // τxy(i, j, k)
    i64 oxz = i + (i64)L.xz1 * j + L.xzp * (kb + 1);    // τxz(i, j, k+1)
    i64 oyz = i + (i64)L.yz1 * j + L.yzp * (kb + 1);    // τyz(i, j, k+1)
    i64 ovx = (i + 1) + (i64)L.vx1 * (j + 1) + L.vxp * (kb + 1);
    i64 ovy = (i + 1) + (i64)L.vy1 * (j + 1) + L.vyp * (kb + 1);
    i64 ovz = (i + 1) + (i64)L.vz1 * (j + 1) + L.vzp * (kb + 1);
    i64 orx = i + (i64)(nx - 1) * j + (i64)(nx - 1) * ny * kb;
    i64 ory = i + (i64)nx * j + (i64)nx * (ny - 1) * kb;

    // carried: level-k values that were the "upper" loads of the previous step
    double Pc = P[oc], ec = et[oc], tzz_c = a.f.tzz[oc], fz_c = a.f.fz[oc];
    double s10 = txz[oxz + 1 - L.xzp];      // τxz(i+1, j, k)
    double r10 = tyz[oyz + L.yz1 - L.yzp];  // τyz(i, j+1, k)

    for (int k = kb; k < kend; ++k) {
        const bool hz = k < nz - 1;
        const double q11 = txy[oxy + 1 + L.xy1], q10 = txy[oxy + 1], q01 = txy[oxy + L.xy1];
        const double s11 = txz[oxz + 1], s01 = txz[oxz];
        const double r11 = tyz[oyz + L.yz1], r01 = tyz[oyz];
        double Pz = 0.0, ez = 0.0, tzz_z = 0.0, fz_z = 0.0;
        if (hz) { Pz = P[oc + L.cp]; ez = et[oc + L.cp]; tzz_z = a.f.tzz[oc + L.cp]; fz_z = a.f.fz[oc + L.cp]; }
        if (hx) {
            const double R = (-a.f.txx[oc] + a.f.txx[oc + 1]) * _dx + _dy * (q11 - q10) + _dz * (s11 - s10) - (-Pc + P[oc + 1]) * _dx -
                             0.5 * (a.f.fx[oc] + a.f.fx[oc + 1]);
            a.f.Vx[ovx] += R * edt / (0.5 * (ec + et[oc + 1]));
            if (DIAG) a.f.Rx[orx] = R;
        }
        if (hy) {
            const double R = _dx * (q11 - q01) + _dy * (a.f.tyy[oc + nx] - a.f.tyy[oc]) + _dz * (r11 - r10) - (-Pc + P[oc + nx]) * _dy -
                             0.5 * (a.f.fy[oc] + a.f.fy[oc + nx]);
            a.f.Vy[ovy] += R * edt / (0.5 * (ec + et[oc + nx]));
            if (DIAG) a.f.Ry[ory] = R;
        }
        if (hz) {
            const double R = _dx * (s11 - s01) + _dy * (r11 - r01) + (-tzz_c + tzz_z) * _dz - (-Pc + Pz) * _dz - 0.5 * (fz_c + fz_z);
            a.f.Vz[ovz] += R * edt / (0.5 * (ec + ez));
            if (DIAG) a.f.Rz[oc] = R;
        }
        Pc = Pz; ec = ez; tzz_c = tzz_z; fz_c = fz_z; s10 = s11; r10 = r11;
        oc += L.cp; oxy += L.xyp; oxz += L.xzp; oyz += L.yzp;
        ovx += L.vxp; ovy += L.vyp; ovz += L.vzp;
        orx += (i64)(nx - 1) * ny; ory += (i64)nx * (ny - 1);
    }
}
#undef CC3

// ================================================================================================
// Version 3 ("zb"): the z-marching sweeps with (a) 32-bit byte offsets against SGPR base pointers
// (global_load saddr form: one VGPR of address state per array class instead of a 64-bit pointer
// per array -- v2 needed 256 VGPRs and ran at one wave per SIMD), and (b) an XCD-aware tile order:
// workgroups are dealt round-robin over the 8 XCDs, so workgroup b is mapped to logical tile
// (b % 8) * (tiles/8) + b / 8 -- each XCD walks one contiguous run of tiles, x/y-adjacent tiles
// share that XCD's L2 and are resident together.  Arrays must be < 4 GiB each (n <= ~800).
// ================================================================================================
typedef unsigned int u32;
#define LDB(p, off) (*(const double *)((const char *)(p) + (off)))
#define STB(p, off, v) (*(double *)((char *)(p) + (off)) = (v))

struct TileMap {
    int ntx, nty, ntz, ntiles, per;     // per = ceil(ntiles / 8)
};
__host__ __device__ inline TileMap make_tilemap(int ex, int ey, int ez, int TX, int TY, int KZ)
{
    TileMap m;
    m.ntx = (ex + TX - 1) / TX; m.nty = (ey + TY - 1) / TY; m.ntz = (ez + KZ - 1) / KZ;
    m.ntiles = m.ntx * m.nty * m.ntz;
    m.per = (m.ntiles + 7) / 8;
    return m;
}
template <bool XCD>
__device__ __forceinline__ bool tile_of_block(const TileMap &m, int &tx, int &ty, int &tz)
{
    const int b = blockIdx.x;
    const int lb = XCD ? (b & 7) * m.per + (b >> 3) : b;
    if (lb >= m.ntiles) return false;
    tx = lb % m.ntx;
    const int r = lb / m.ntx;
    ty = r % m.nty;
    tz = r / m.nty;
    return true;
}

template <bool DIAG, int TX, int TY, int KZ, int MINW, bool EDGES, bool XCD = true>
__global__ __launch_bounds__(TX *TY, MINW) void k_stress3d_zb(const SweepArgs a, const TileMap tm)
{
    const Lay3 &L = a.L;
    const int nx = L.nx, ny = L.ny, nz = L.nz;
    int tx, ty, tz;
    if (!tile_of_block<XCD>(tm, tx, ty, tz)) return;
    const int i = tx * TX + (int)(threadIdx.x % TX);
    const int j = ty * TY + (int)(threadIdx.x / TX);
    const int kb = tz * KZ;
    if (i >= nx || j >= ny) return;
    const int kend = min(kb + KZ, nz);
    const int im = max(i - 1, 0), jm = max(j - 1, 0);
    const bool xhi = (i == nx - 1), yhi = (j == ny - 1);
    const double _dx = a._dx, _dy = a._dy, _dz = a._dz, dt = a.dt, th = a.theta_dtau, _dt = 1.0 / dt, rr = a.r;
    const jrx_stokes3d_fields &f = a.f;

    // byte strides (uniform)
    const u32 sc = (u32)L.cp * 8u, svx = (u32)L.vxp * 8u, svy = (u32)L.vyp * 8u, svz = (u32)L.vzp * 8u;
    const u32 sxy = (u32)L.xyp * 8u, sxz = (u32)L.xzp * 8u, syz = (u32)L.yzp * 8u;
    const u32 rvx = (u32)L.vx1 * 8u, rvy = (u32)L.vy1 * 8u, rvz = (u32)L.vz1 * 8u;
    // running byte offsets
    u32 oc = 8u * (u32)(i + nx * j) + sc * (u32)kb;
    const u32 dcx = 8u * (u32)(i - im), dcy = 8u * (u32)(nx * (j - jm));     // distance to the clamped x / y neighbour cell
    u32 ovx = 8u * (u32)(i + L.vx1 * (j + 1)) + svx * (u32)(kb + 1);
    u32 ovy = 8u * (u32)((i + 1) + L.vy1 * j) + svy * (u32)(kb + 1);
    u32 ovz = 8u * (u32)((i + 1) + L.vz1 * (j + 1)) + svz * (u32)(kb + 1);
    u32 oxy = 8u * (u32)(i + L.xy1 * j) + sxy * (u32)kb;
    u32 oxz = 8u * (u32)(i + L.xz1 * j) + sxz * (u32)kb;
    u32 oyz = 8u * (u32)(i + L.yz1 * j) + syz * (u32)kb;

    double a_p = LDB(f.Vx, ovx - svx), b_p = LDB(f.Vy, ovy - svy);
    double c_p = LDB(f.Vz, ovz - svz), cx_p = LDB(f.Vz, ovz - svz - 8u), cy_p = LDB(f.Vz, ovz - svz - rvz);
    const u32 back = kb > 0 ? sc : 0u;
    double e_p = LDB(f.eta, oc - back), ex_p = LDB(f.eta, oc - back - dcx), ey_p = LDB(f.eta, oc - back - dcy);
    double g_p = LDB(f.G, oc - back), gx_p = LDB(f.G, oc - back - dcx), gy_p = LDB(f.G, oc - back - dcy);

    for (int k = kb; k < kend; ++k) {
        const double va = LDB(f.Vx, ovx), vax = LDB(f.Vx, ovx + 8u), vay = LDB(f.Vx, ovx - rvx);
        const double vb = LDB(f.Vy, ovy), vby = LDB(f.Vy, ovy + rvy), vbx = LDB(f.Vy, ovy - 8u);
        const double vc = LDB(f.Vz, ovz), vcx = LDB(f.Vz, ovz - 8u), vcy = LDB(f.Vz, ovz - rvz);
        const double e = LDB(f.eta, oc), ex = LDB(f.eta, oc - dcx), ey = LDB(f.eta, oc - dcy), exy_ = LDB(f.eta, oc - dcx - dcy);
        const double g = LDB(f.G, oc), gx = LDB(f.G, oc - dcx), gy = LDB(f.G, oc - dcy), gxy = LDB(f.G, oc - dcx - dcy);
        // issue the remaining independent loads of this plane early
        const double P = LDB(f.P, oc), P0 = LDB(f.P0, oc), Kc = LDB(f.K, oc), Qc = LDB(f.Q, oc);
        const double txx = LDB(f.txx, oc), tyy = LDB(f.tyy, oc), tzz = LDB(f.tzz, oc);
        const double toxx = LDB(f.toxx, oc), toyy = LDB(f.toyy, oc), tozz = LDB(f.tozz, oc);
        const double txy = LDB(f.txy, oxy), toxy = LDB(f.toxy, oxy);
        const double txz = LDB(f.txz, oxz), toxz = LDB(f.toxz, oxz);
        const double tyz = LDB(f.tyz, oyz), toyz = LDB(f.toyz, oyz);

        {   // centre (i,j,k)
            const double dxi = (-va + vax) * _dx;
            const double dyi = (-vb + vby) * _dy;
            const double dzi = (-c_p + vc) * _dz;
            const double divV = dxi + dyi + dzi;
            const double _Gdt = 1.0 / (g * dt);
            const double _Kdt = 1.0 / (Kc * dt);
            const double rhs = -divV + (Qc * _dt);
            const double psi = 1.0 / (1.0 / e + _Gdt) * rr / th;
            STB(f.P, oc, (fma(P0, _Kdt, rhs) * psi + P) / (1.0 + _Kdt * psi));
            const double d3 = divV * (1.0 / 3.0);
            const double exx = dxi - d3, eyy = dyi - d3, ezz = dzi - d3;
            if (DIAG) {
                STB(f.RP, oc, fma(-(P - P0), _Kdt, rhs));
                STB(f.divV, oc, divV);
                STB(f.exx, oc, exx); STB(f.eyy, oc, eyy); STB(f.ezz, oc, ezz);
            }
            const double dtr = dev_dtau_r(th, e, _Gdt);
            STB(f.txx, oc, txx + dev_stress_inc(txx, toxx, e, exx, _Gdt, dtr));
            STB(f.tyy, oc, tyy + dev_stress_inc(tyy, toyy, e, eyy, _Gdt, dtr));
            STB(f.tzz, oc, tzz + dev_stress_inc(tzz, tozz, e, ezz, _Gdt, dtr));
        }
        {   // τxy (i,j,k)
            const double s_ = 0.5 * (_dy * (va - vay) + _dx * (vb - vbx));
            const double ee = 0.25 * (exy_ + ey + ex + e);
            const double gg = 0.25 * (gxy + gy + gx + g);
            const double _Gdt = 1.0 / (gg * dt);
            const double dtr = dev_dtau_r(th, ee, _Gdt);
            STB(f.txy, oxy, txy + dev_stress_inc(txy, toxy, ee, s_, _Gdt, dtr));
            if (DIAG) STB(f.exy, oxy, s_);
        }
        {   // τxz (i,j,k)
            const double s_ = 0.5 * (_dz * (va - a_p) + _dx * (c_p - cx_p));
            const double ee = 0.25 * (ex_p + e_p + ex + e);
            const double gg = 0.25 * (gx_p + g_p + gx + g);
            const double _Gdt = 1.0 / (gg * dt);
            const double dtr = dev_dtau_r(th, ee, _Gdt);
            STB(f.txz, oxz, txz + dev_stress_inc(txz, toxz, ee, s_, _Gdt, dtr));
            if (DIAG) STB(f.exz, oxz, s_);
        }
        {   // τyz (i,j,k)
            const double s_ = 0.5 * (_dz * (vb - b_p) + _dy * (c_p - cy_p));
            const double ee = 0.25 * (ey_p + e_p + ey + e);
            const double gg = 0.25 * (gy_p + g_p + gy + g);
            const double _Gdt = 1.0 / (gg * dt);
            const double dtr = dev_dtau_r(th, ee, _Gdt);
            STB(f.tyz, oyz, tyz + dev_stress_inc(tyz, toyz, ee, s_, _Gdt, dtr));
            if (DIAG) STB(f.eyz, oyz, s_);
        }
        if (EDGES) {
            if (xhi) { node_xy<DIAG>(a, nx, j, k); node_xz<DIAG>(a, nx, j, k); }
            if (yhi) { node_xy<DIAG>(a, i, ny, k); node_yz<DIAG>(a, i, ny, k); }
            if (xhi && yhi) node_xy<DIAG>(a, nx, ny, k);
        }

        a_p = va; b_p = vb; c_p = vc; cx_p = vcx; cy_p = vcy;
        e_p = e; ex_p = ex; ey_p = ey; g_p = g; gx_p = gx; gy_p = gy;
        oc += sc; ovx += svx; ovy += svy; ovz += svz; oxy += sxy; oxz += sxz; oyz += syz;
    }
    if (EDGES && kend == nz) {
        node_xz<DIAG>(a, i, j, nz);
        node_yz<DIAG>(a, i, j, nz);
        if (xhi) node_xz<DIAG>(a, nx, j, nz);
        if (yhi) node_yz<DIAG>(a, i, ny, nz);
    }
}

template <bool DIAG, int TX, int TY, int KZ, int MINW, bool XCD = true>
__global__ __launch_bounds__(TX *TY, MINW) void k_velocity3d_zb(const SweepArgs a, const TileMap tm)
{
    const Lay3 &L = a.L;
    const int nx = L.nx, ny = L.ny, nz = L.nz;
    int tx, ty, tz;
    if (!tile_of_block<XCD>(tm, tx, ty, tz)) return;
    const int i = a.i0 + tx * TX + (int)(threadIdx.x % TX);
    const int j = a.j0 + ty * TY + (int)(threadIdx.x / TX);
    const int kb = a.k0 + tz * KZ;
    if (i >= a.i1 || j >= a.j1) return;
    const int kend = min(kb + KZ, a.k1);
    const double _dx = a._dx, _dy = a._dy, _dz = a._dz, edt = a.eta_dtau;
    const bool hx = i < nx - 1, hy = j < ny - 1;
    const jrx_stokes3d_fields &f = a.f;
    const double *et = a.etatau;

    const u32 sc = (u32)L.cp * 8u, svx = (u32)L.vxp * 8u, svy = (u32)L.vyp * 8u, svz = (u32)L.vzp * 8u;
    const u32 sxy = (u32)L.xyp * 8u, sxz = (u32)L.xzp * 8u, syz = (u32)L.yzp * 8u;
    const u32 rc = (u32)nx * 8u, rxy = (u32)L.xy1 * 8u, ryz = (u32)L.yz1 * 8u;
    const u32 srx = (u32)(nx - 1) * (u32)ny * 8u, sry = (u32)nx * (u32)(ny - 1) * 8u;
    u32 oc = 8u * (u32)(i + nx * j) + sc * (u32)kb;
    u32 oxy = 8u * (u32)(i + L.xy1 * j) + sxy * (u32)kb;
    u32 oxz = 8u * (u32)(i + L.xz1 * j) + sxz * (u32)(kb + 1);
    u32 oyz = 8u * (u32)(i + L.yz1 * j) + syz * (u32)(kb + 1);
    u32 ovx = 8u * (u32)((i + 1) + L.vx1 * (j + 1)) + svx * (u32)(kb + 1);
    u32 ovy = 8u * (u32)((i + 1) + L.vy1 * (j + 1)) + svy * (u32)(kb + 1);
    u32 ovz = 8u * (u32)((i + 1) + L.vz1 * (j + 1)) + svz * (u32)(kb + 1);
    u32 orx = 8u * (u32)(i + (nx - 1) * j) + srx * (u32)kb;
    u32 ory = 8u * (u32)(i + nx * j) + sry * (u32)kb;
    // x / y upper-neighbour distance, 0 on the last cell (value then unused but address stays in range)
    const u32 dx1 = hx ? 8u : 0u, dy1 = hy ? rc : 0u;

    double Pc = LDB(f.P, oc), ec = LDB(et, oc), tzz_c = LDB(f.tzz, oc), fz_c = LDB(f.fz, oc);
    double s10 = LDB(f.txz, oxz + 8u - sxz), r10 = LDB(f.tyz, oyz + ryz - syz);

    for (int k = kb; k < kend; ++k) {
        const bool hz = k < nz - 1;
        const u32 dz1 = hz ? sc : 0u;
        const double q11 = LDB(f.txy, oxy + 8u + rxy), q10 = LDB(f.txy, oxy + 8u), q01 = LDB(f.txy, oxy + rxy);
        const double s11 = LDB(f.txz, oxz + 8u), s01 = LDB(f.txz, oxz);
        const double r11 = LDB(f.tyz, oyz + ryz), r01 = LDB(f.tyz, oyz);
        const double Pz = LDB(f.P, oc + dz1), ez = LDB(et, oc + dz1), tzz_z = LDB(f.tzz, oc + dz1), fz_z = LDB(f.fz, oc + dz1);
        const double Px = LDB(f.P, oc + dx1), Py = LDB(f.P, oc + dy1), ex = LDB(et, oc + dx1), ey = LDB(et, oc + dy1);
        const double txx_c = LDB(f.txx, oc), txx_x = LDB(f.txx, oc + dx1), tyy_c = LDB(f.tyy, oc), tyy_y = LDB(f.tyy, oc + dy1);
        const double fx_c = LDB(f.fx, oc), fx_x = LDB(f.fx, oc + dx1), fy_c = LDB(f.fy, oc), fy_y = LDB(f.fy, oc + dy1);
        const double vx = LDB(f.Vx, ovx), vy = LDB(f.Vy, ovy), vz = LDB(f.Vz, ovz);
        if (hx) {
            const double R = (-txx_c + txx_x) * _dx + _dy * (q11 - q10) + _dz * (s11 - s10) - (-Pc + Px) * _dx - 0.5 * (fx_c + fx_x);
            STB(f.Vx, ovx, vx + R * edt / (0.5 * (ec + ex)));
            if (DIAG) STB(f.Rx, orx, R);
        }
        if (hy) {
            const double R = _dx * (q11 - q01) + _dy * (tyy_y - tyy_c) + _dz * (r11 - r10) - (-Pc + Py) * _dy - 0.5 * (fy_c + fy_y);
            STB(f.Vy, ovy, vy + R * edt / (0.5 * (ec + ey)));
            if (DIAG) STB(f.Ry, ory, R);
        }
        if (hz) {
            const double R = _dx * (s11 - s01) + _dy * (r11 - r01) + (-tzz_c + tzz_z) * _dz - (-Pc + Pz) * _dz - 0.5 * (fz_c + fz_z);
            STB(f.Vz, ovz, vz + R * edt / (0.5 * (ec + ez)));
            if (DIAG) STB(f.Rz, oc, R);
        }
        Pc = Pz; ec = ez; tzz_c = tzz_z; fz_c = fz_z; s10 = s11; r10 = r11;
        oc += sc; oxy += sxy; oxz += sxz; oyz += syz; ovx += svx; ovy += svy; ovz += svz; orx += srx; ory += sry;
    }
}

}   // namespace
