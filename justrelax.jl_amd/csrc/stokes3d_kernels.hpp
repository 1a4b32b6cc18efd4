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

// destination of the state arrays a sweep updates.  Equal to the source arrays for in-place
// sweeps; the fused iteration kernel ping-pongs between the caller's arrays and a scratch set.
struct Out10 {
    double *P, *txx, *tyy, *tzz, *tyz, *txz, *txy, *Vx, *Vy, *Vz;
};

struct SweepArgs {
    jrx_stokes3d_fields f;
    Out10 o;
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
// flow_bcs! by rule instead of by memory: what free_slip! / no_slip! leave in the boundary entries of V, derived from the interior on the
// fly.  t[face]: 0 none (memory holds the prescribed value), 1 free slip, 2 no slip; faces x-lo, x-hi, y-lo, y-hi, z-lo (k = 1), z-hi.
// Exact wherever a Stokes stencil reads (entries on at most one ghost plane); used by the boundary-layer launch behind the fused
// kernel so that it does not have to wait for -- or need at all -- a flow_bcs! launch on the new velocities.
struct GhostRule { int t[6]; };
__device__ __forceinline__ void ghost_tan(int &idx, double &sgn, const int n, const int tlo, const int thi)
{
    if (idx == 0 && tlo) { idx = 1; if (tlo == 2) sgn = -sgn; }
    else if (idx == n + 1 && thi) { idx = n; if (thi == 2) sgn = -sgn; }
}
__device__ __forceinline__ double vx_rule(const double *__restrict__ Vx, const Lay3 &L, const GhostRule &g, int i, int j, int k)
{
    if ((i == 0 && g.t[0] == 2) || (i == L.nx && g.t[1] == 2)) return 0.0;
    double sgn = 1.0;
    ghost_tan(j, sgn, L.ny, g.t[2], g.t[3]); ghost_tan(k, sgn, L.nz, g.t[4], g.t[5]);
    return sgn * Vx[i + (i64)L.vx1 * j + L.vxp * k];
}
__device__ __forceinline__ double vy_rule(const double *__restrict__ Vy, const Lay3 &L, const GhostRule &g, int i, int j, int k)
{
    if ((j == 0 && g.t[2] == 2) || (j == L.ny && g.t[3] == 2)) return 0.0;
    double sgn = 1.0;
    ghost_tan(i, sgn, L.nx, g.t[0], g.t[1]); ghost_tan(k, sgn, L.nz, g.t[4], g.t[5]);
    return sgn * Vy[i + (i64)L.vy1 * j + L.vyp * k];
}
__device__ __forceinline__ double vz_rule(const double *__restrict__ Vz, const Lay3 &L, const GhostRule &g, int i, int j, int k)
{
    if ((k == 0 && g.t[4] == 2) || (k == L.nz && g.t[5] == 2)) return 0.0;
    double sgn = 1.0;
    ghost_tan(i, sgn, L.nx, g.t[0], g.t[1]); ghost_tan(j, sgn, L.ny, g.t[2], g.t[3]);
    return sgn * Vz[i + (i64)L.vz1 * j + L.vzp * k];
}

// VISC: the viscous limit dt = Inf (see k_fused3d): 1/(G dt) = 1/(K dt) = 1/dt = 0, their operands τ_o, P0, K, G, Q are not loaded
template <bool DIAG, bool GH = false, bool VISC = false>
__device__ __forceinline__ void stress3d_node(const SweepArgs &a, const int i, const int j, const int k, const GhostRule *gr = nullptr)
{
    const Lay3 &L = a.L;
    const int nx = L.nx, ny = L.ny, nz = L.nz;

    const double *__restrict__ Vx = a.f.Vx, *__restrict__ Vy = a.f.Vy, *__restrict__ Vz = a.f.Vz;
    const double *__restrict__ eta = a.f.eta, *__restrict__ G = a.f.G;
    const double _dx = a._dx, _dy = a._dy, _dz = a._dz, dt = a.dt, th = a.theta_dtau;

#define VX(i_, j_, k_) (GH ? vx_rule(Vx, L, *gr, (i_), (j_), (k_)) : Vx[(i_) + (i64)L.vx1 * (j_) + L.vxp * (k_)])
#define VY(i_, j_, k_) (GH ? vy_rule(Vy, L, *gr, (i_), (j_), (k_)) : Vy[(i_) + (i64)L.vy1 * (j_) + L.vyp * (k_)])
#define VZ(i_, j_, k_) (GH ? vz_rule(Vz, L, *gr, (i_), (j_), (k_)) : Vz[(i_) + (i64)L.vz1 * (j_) + L.vzp * (k_)])
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
        const double _Gdt = VISC ? 0.0 : 1.0 / (G[c] * dt);
        {
            const double _Kdt = VISC ? 0.0 : 1.0 / (a.f.K[c] * dt);
            const double _dt = 1.0 / dt;
            const double P = a.f.P[c], P0 = VISC ? 0.0 : a.f.P0[c];
            const double rhs = -divV + ((VISC ? 0.0 : a.f.Q[c]) * _dt);
            const double psi = 1.0 / (1.0 / e + _Gdt) * a.r / th;
            a.o.P[c] = (fma(P0, _Kdt, rhs) * psi + P) / (1.0 + _Kdt * psi);
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
        tv = a.f.txx[c]; a.o.txx[c] = tv + dev_stress_inc(tv, (VISC ? 0.0 : a.f.toxx[c]), e, exx, _Gdt, dtr);
        tv = a.f.tyy[c]; a.o.tyy[c] = tv + dev_stress_inc(tv, (VISC ? 0.0 : a.f.toyy[c]), e, eyy, _Gdt, dtr);
        tv = a.f.tzz[c]; a.o.tzz[c] = tv + dev_stress_inc(tv, (VISC ? 0.0 : a.f.tozz[c]), e, ezz, _Gdt, dtr);
    }

    // clamped neighbour cell indices (MiniKernels.jl:133-147)
    const int im = max(i - 1, 0), ip = min(i, nx - 1);
    const int jm = max(j - 1, 0), jp = min(j, ny - 1);
    const int km = max(k - 1, 0), kp = min(k, nz - 1);

    if (ck) {   // τxy at (i,j,k) of (nx+1, ny+1, nz)   (VelocityKernels.jl:95-101, StressKernels.jl:199-208)
        const double exy = 0.5 * (_dy * (VX(i, j + 1, k + 1) - VX(i, j, k + 1)) + _dx * (VY(i + 1, j, k + 1) - VY(i, j, k + 1)));
        const double e = 0.25 * (eta[CC(im, jm, k)] + eta[CC(ip, jm, k)] + eta[CC(im, jp, k)] + eta[CC(ip, jp, k)]);
        const double _Gdt = VISC ? 0.0 : 1.0 / (0.25 * (G[CC(im, jm, k)] + G[CC(ip, jm, k)] + G[CC(im, jp, k)] + G[CC(ip, jp, k)]) * dt);
        const double dtr = dev_dtau_r(th, e, _Gdt);
        const i64 c = i + (i64)L.xy1 * j + L.xyp * k;
        const double tv = a.f.txy[c];
        a.o.txy[c] = tv + dev_stress_inc(tv, (VISC ? 0.0 : a.f.toxy[c]), e, exy, _Gdt, dtr);
        if (DIAG) a.f.exy[c] = exy;
    }
    if (cj) {   // τxz at (i,j,k) of (nx+1, ny, nz+1)
        const double exz = 0.5 * (_dz * (VX(i, j + 1, k + 1) - VX(i, j + 1, k)) + _dx * (VZ(i + 1, j + 1, k) - VZ(i, j + 1, k)));
        const double e = 0.25 * (eta[CC(im, j, km)] + eta[CC(ip, j, km)] + eta[CC(im, j, kp)] + eta[CC(ip, j, kp)]);
        const double _Gdt = VISC ? 0.0 : 1.0 / (0.25 * (G[CC(im, j, km)] + G[CC(ip, j, km)] + G[CC(im, j, kp)] + G[CC(ip, j, kp)]) * dt);
        const double dtr = dev_dtau_r(th, e, _Gdt);
        const i64 c = i + (i64)L.xz1 * j + L.xzp * k;
        const double tv = a.f.txz[c];
        a.o.txz[c] = tv + dev_stress_inc(tv, (VISC ? 0.0 : a.f.toxz[c]), e, exz, _Gdt, dtr);
        if (DIAG) a.f.exz[c] = exz;
    }
    if (ci) {   // τyz at (i,j,k) of (nx, ny+1, nz+1)
        const double eyz = 0.5 * (_dz * (VY(i + 1, j, k + 1) - VY(i + 1, j, k)) + _dy * (VZ(i + 1, j + 1, k) - VZ(i + 1, j, k)));
        const double e = 0.25 * (eta[CC(i, jm, km)] + eta[CC(i, jp, km)] + eta[CC(i, jm, kp)] + eta[CC(i, jp, kp)]);
        const double _Gdt = VISC ? 0.0 : 1.0 / (0.25 * (G[CC(i, jm, km)] + G[CC(i, jp, km)] + G[CC(i, jm, kp)] + G[CC(i, jp, kp)]) * dt);
        const double dtr = dev_dtau_r(th, e, _Gdt);
        const i64 c = i + (i64)L.yz1 * j + L.yzp * k;
        const double tv = a.f.tyz[c];
        a.o.tyz[c] = tv + dev_stress_inc(tv, (VISC ? 0.0 : a.f.toyz[c]), e, eyz, _Gdt, dtr);
        if (DIAG) a.f.eyz[c] = eyz;
    }
#undef VX
#undef VY
#undef VZ
}

template <bool DIAG>
__global__ __launch_bounds__(256) void k_stress3d(const SweepArgs a)
{
    const int wi = a.i1 - a.i0;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int jj = t / wi;
    const int j = a.j0 + jj;
    if (j >= a.j1) return;
    stress3d_node<DIAG>(a, a.i0 + (t - jj * wi), j, a.k0 + (int)blockIdx.y);
}

// the same over up to six disjoint node boxes in one launch (the thin boundary layers behind the fused kernel):
// blocks [start[b], start[b+1]) belong to box b, each covering 256 nodes of one xy-plane of the box
struct StressBoxes {
    int n;
    int box[6][6];        // i0, i1, j0, j1, k0, k1
    int start[7];         // first block of each box; start[n] = total
    int per_plane[6];     // blocks per xy-plane of the box
};
template <bool DIAG, bool GH = false, bool VISC = false>
__global__ __launch_bounds__(256) void k_stress3d_boxes(const SweepArgs a, const StressBoxes B, const GhostRule gr)
{
    int b = 0;
    while (b + 1 < B.n && (int)blockIdx.x >= B.start[b + 1]) b++;
    const int lb = (int)blockIdx.x - B.start[b];
    const int kz = lb / B.per_plane[b], bx = lb - kz * B.per_plane[b];
    const int wi = B.box[b][1] - B.box[b][0];
    const int t = bx * 256 + (int)threadIdx.x;
    const int jj = t / wi;
    const int j = B.box[b][2] + jj;
    if (j >= B.box[b][3]) return;
    stress3d_node<DIAG, GH, VISC>(a, B.box[b][0] + (t - jj * wi), j, B.box[b][4] + kz, &gr);
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
        a.o.Vx[v] = a.f.Vx[v] + R * edt / (0.5 * (ec + et[cx]));
        if (DIAG) a.f.Rx[i + (i64)(nx - 1) * j + (i64)(nx - 1) * ny * k] = R;
    }
    if (j < ny - 1) {
        const i64 cy = c + nx;
        const double R = _dx * (TXY(i + 1, j + 1, k) - TXY(i, j + 1, k)) + _dy * (a.f.tyy[cy] - a.f.tyy[c]) +
                         _dz * (TYZ(i, j + 1, k + 1) - TYZ(i, j + 1, k)) - (-Pc + P[cy]) * _dy -
                         0.5 * (a.f.fy[c] + a.f.fy[cy]);
        const i64 v = (i + 1) + (i64)L.vy1 * (j + 1) + L.vyp * (k + 1);
        a.o.Vy[v] = a.f.Vy[v] + R * edt / (0.5 * (ec + et[cy]));
        if (DIAG) a.f.Ry[i + (i64)nx * j + (i64)nx * (ny - 1) * k] = R;
    }
    if (k < nz - 1) {
        const i64 cz = c + L.cp;
        const double R = _dx * (TXZ(i + 1, j, k + 1) - TXZ(i, j, k + 1)) + _dy * (TYZ(i, j + 1, k + 1) - TYZ(i, j, k + 1)) +
                         (-a.f.tzz[c] + a.f.tzz[cz]) * _dz - (-Pc + P[cz]) * _dz - 0.5 * (a.f.fz[c] + a.f.fz[cz]);
        const i64 v = (i + 1) + (i64)L.vz1 * (j + 1) + L.vzp * (k + 1);
        a.o.Vz[v] = a.f.Vz[v] + R * edt / (0.5 * (ec + et[cz]));
        if (DIAG) a.f.Rz[c] = R;
    }
#undef TXY
#undef TXZ
#undef TYZ
#undef CC
}


// ================================================================================================
// 2.5D z-marching sweeps.  A block owns a TX x TY tile of cell columns and walks KZ planes in z;
// the values that the next plane needs again (the k+1 velocity plane, the k plane of η and G, the
// upper τ/P/f/ητ plane of the velocity sweep) stay in registers instead of being re-read, x/y
// neighbours of the same plane are served by L1 (same or adjacent wave), so HBM sees each array
// plane once per sweep apart from tile halos.  (A first version with 64-bit indices needed 256
// VGPRs -- one live pointer per array after loop strength reduction -- and ran at 1 wave/SIMD.)
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
    a.o.txy[c] = tv + dev_stress_inc(tv, a.f.toxy[c], e, exy, _Gdt, dtr);
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
    a.o.txz[c] = tv + dev_stress_inc(tv, a.f.toxz[c], e, exz, _Gdt, dtr);
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
    a.o.tyz[c] = tv + dev_stress_inc(tv, a.f.toyz[c], e, eyz, _Gdt, dtr);
    if (DIAG) a.f.eyz[c] = eyz;
}


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
// the same with an optional non-temporal (streaming) hint
template <bool NT> __device__ __forceinline__ double LDN(const double *p, u32 off)
{
    const double *q = (const double *)((const char *)p + off);
    return NT ? __builtin_nontemporal_load(q) : *q;
}
template <bool NT> __device__ __forceinline__ void STN(double *p, u32 off, double v)
{
    double *q = (double *)((char *)p + off);
    if (NT) __builtin_nontemporal_store(v, q);
    else *q = v;
}

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
// XM: 0 = tiles in dispatch order (x fastest, then y, then z);
//     1 = one contiguous run of tiles per XCD (measured slower: the XCDs stream through 8 distant DRAM regions);
//     G >= 2 = "banded": workgroup b runs on XCD b % 8 (round-robin dispatch), so within each band of 8*G tile rows XCD q
//         gets the G consecutive rows q*G .. q*G+G-1.  y-neighbour rows then live in the same XCD's L2 (no second
//         fetch through the fabric), while all XCDs still sweep the same 8*G-row band of every array together.
template <int XM>
__device__ __forceinline__ bool tile_of_block(const TileMap &m, int &tx, int &ty, int &tz)
{
    const int b = blockIdx.x;
    if (XM >= 2) {
        const int per_z = m.ntx * m.nty;
        if (m.nty % (8 * XM) == 0) {
            tz = b / per_z;
            const int bz = b - tz * per_z;
            const int q = bz & 7, r = bz >> 3;
            tx = r % m.ntx;
            const int r2 = r / m.ntx;
            ty = (r2 / XM) * (8 * XM) + q * XM + (r2 % XM);
            return b < m.ntiles;
        }
    }
    const int lb = XM == 1 ? (b & 7) * m.per + (b >> 3) : b;
    if (lb >= m.ntiles) return false;
    tx = lb % m.ntx;
    const int r = lb / m.ntx;
    ty = r % m.nty;
    tz = r / m.nty;
    return true;
}

// SHF: the x-neighbour operands (Vx at i+1; Vy, Vz, η, G at i-1) come from the adjacent lane; only the first / last lane of a
// wave (and the last cell column) load them
template <bool DIAG, int TX, int TY, int KZ, int MINW, bool EDGES, int XCD = 0, bool SHF = false, bool VISC = false>
__global__ __launch_bounds__(TX *TY, MINW) void k_stress3d_zb(const SweepArgs a, const TileMap tm)
{
    static_assert(!SHF || (TX % 64 == 0 && !EDGES), "SHF needs whole waves per row");
    static_assert(!VISC || (!DIAG && !EDGES), "the viscous-limit form is the state-only sweep over the interior tiles");
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
    double g_p = 1.0, gx_p = 1.0, gy_p = 1.0;     // VISC: G is only ever multiplied into 1/(G dt) = 0 (see k_fused3d)
    if (!VISC) { g_p = LDB(f.G, oc - back); gx_p = LDB(f.G, oc - back - dcx); gy_p = LDB(f.G, oc - back - dcy); }

    for (int k = kb; k < kend; ++k) {
        const double va = LDB(f.Vx, ovx), vay = LDB(f.Vx, ovx - rvx);
        const double vb = LDB(f.Vy, ovy), vby = LDB(f.Vy, ovy + rvy);
        const double vc = LDB(f.Vz, ovz), vcy = LDB(f.Vz, ovz - rvz);
        const double e = LDB(f.eta, oc), ey = LDB(f.eta, oc - dcy);
        const double g = VISC ? 1.0 : LDB(f.G, oc), gy = VISC ? 1.0 : LDB(f.G, oc - dcy);
        double vax, vbx, vcx, ex, exy_, gx = 1.0, gxy = 1.0;
        if (SHF) {
            const int lane = (int)(threadIdx.x & 63);
            vax = __shfl_down(va, 1, 64);
            vbx = __shfl_up(vb, 1, 64); vcx = __shfl_up(vc, 1, 64);
            ex = __shfl_up(e, 1, 64); exy_ = __shfl_up(ey, 1, 64);
            if (!VISC) { gx = __shfl_up(g, 1, 64); gxy = __shfl_up(gy, 1, 64); }
            if (lane == 63 || xhi) vax = LDB(f.Vx, ovx + 8u);
            if (lane == 0) {
                vbx = LDB(f.Vy, ovy - 8u); vcx = LDB(f.Vz, ovz - 8u);
                ex = LDB(f.eta, oc - dcx); exy_ = LDB(f.eta, oc - dcx - dcy);
                if (!VISC) { gx = LDB(f.G, oc - dcx); gxy = LDB(f.G, oc - dcx - dcy); }
            }
        } else {
            vax = LDB(f.Vx, ovx + 8u); vbx = LDB(f.Vy, ovy - 8u); vcx = LDB(f.Vz, ovz - 8u);
            ex = LDB(f.eta, oc - dcx); exy_ = LDB(f.eta, oc - dcx - dcy);
            if (!VISC) { gx = LDB(f.G, oc - dcx); gxy = LDB(f.G, oc - dcx - dcy); }
        }
        // issue the remaining independent loads of this plane early
        const double P = LDB(f.P, oc), P0 = VISC ? 0.0 : LDB(f.P0, oc), Kc = VISC ? 1.0 : LDB(f.K, oc), Qc = VISC ? 0.0 : LDB(f.Q, oc);
        const double txx = LDB(f.txx, oc), tyy = LDB(f.tyy, oc), tzz = LDB(f.tzz, oc);
        const double toxx = VISC ? 0.0 : LDB(f.toxx, oc), toyy = VISC ? 0.0 : LDB(f.toyy, oc), tozz = VISC ? 0.0 : LDB(f.tozz, oc);
        const double txy = LDB(f.txy, oxy), toxy = VISC ? 0.0 : LDB(f.toxy, oxy);
        const double txz = LDB(f.txz, oxz), toxz = VISC ? 0.0 : LDB(f.toxz, oxz);
        const double tyz = LDB(f.tyz, oyz), toyz = VISC ? 0.0 : LDB(f.toyz, oyz);

        {   // centre (i,j,k)
            const double dxi = (-va + vax) * _dx;
            const double dyi = (-vb + vby) * _dy;
            const double dzi = (-c_p + vc) * _dz;
            const double divV = dxi + dyi + dzi;
            const double _Gdt = VISC ? 0.0 : 1.0 / (g * dt);
            const double _Kdt = VISC ? 0.0 : 1.0 / (Kc * dt);
            const double rhs = -divV + (Qc * _dt);
            const double psi = 1.0 / (1.0 / e + _Gdt) * rr / th;
            STB(a.o.P, oc, (fma(P0, _Kdt, rhs) * psi + P) / (1.0 + _Kdt * psi));
            const double d3 = divV * (1.0 / 3.0);
            const double exx = dxi - d3, eyy = dyi - d3, ezz = dzi - d3;
            if (DIAG) {
                STB(f.RP, oc, fma(-(P - P0), _Kdt, rhs));
                STB(f.divV, oc, divV);
                STB(f.exx, oc, exx); STB(f.eyy, oc, eyy); STB(f.ezz, oc, ezz);
            }
            const double dtr = dev_dtau_r(th, e, _Gdt);
            STB(a.o.txx, oc, txx + dev_stress_inc(txx, toxx, e, exx, _Gdt, dtr));
            STB(a.o.tyy, oc, tyy + dev_stress_inc(tyy, toyy, e, eyy, _Gdt, dtr));
            STB(a.o.tzz, oc, tzz + dev_stress_inc(tzz, tozz, e, ezz, _Gdt, dtr));
        }
        {   // τxy (i,j,k)
            const double s_ = 0.5 * (_dy * (va - vay) + _dx * (vb - vbx));
            const double ee = 0.25 * (exy_ + ey + ex + e);
            const double gg = 0.25 * (gxy + gy + gx + g);
            const double _Gdt = VISC ? 0.0 : 1.0 / (gg * dt);
            const double dtr = dev_dtau_r(th, ee, _Gdt);
            STB(a.o.txy, oxy, txy + dev_stress_inc(txy, toxy, ee, s_, _Gdt, dtr));
            if (DIAG) STB(f.exy, oxy, s_);
        }
        {   // τxz (i,j,k)
            const double s_ = 0.5 * (_dz * (va - a_p) + _dx * (c_p - cx_p));
            const double ee = 0.25 * (ex_p + e_p + ex + e);
            const double gg = 0.25 * (gx_p + g_p + gx + g);
            const double _Gdt = VISC ? 0.0 : 1.0 / (gg * dt);
            const double dtr = dev_dtau_r(th, ee, _Gdt);
            STB(a.o.txz, oxz, txz + dev_stress_inc(txz, toxz, ee, s_, _Gdt, dtr));
            if (DIAG) STB(f.exz, oxz, s_);
        }
        {   // τyz (i,j,k)
            const double s_ = 0.5 * (_dz * (vb - b_p) + _dy * (c_p - cy_p));
            const double ee = 0.25 * (ey_p + e_p + ey + e);
            const double gg = 0.25 * (gy_p + g_p + gy + g);
            const double _Gdt = VISC ? 0.0 : 1.0 / (gg * dt);
            const double dtr = dev_dtau_r(th, ee, _Gdt);
            STB(a.o.tyz, oyz, tyz + dev_stress_inc(tyz, toyz, ee, s_, _Gdt, dtr));
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

// SHF: the operands at i+1 that the right-hand lane holds as its own (τxy(·,j+1), τxz, P, ητ, τxx, fx) come by lane shuffle; the last
// lane of a wave and the last column of the (sub-)box load them
// NOF: body-force arrays known to hold only +0.0 are not loaded (1: fx, fy; 2: all three), see k_fused3d; the caller states it per launch
template <bool DIAG, int TX, int TY, int KZ, int MINW, int XCD = 0, bool SHF = false, int NOF = 0>
__global__ __launch_bounds__(TX *TY, MINW) void k_velocity3d_zb(const SweepArgs a, const TileMap tm)
{
    static_assert(!SHF || TX % 64 == 0, "SHF needs whole waves per row");
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

    double Pc = LDB(f.P, oc), ec = LDB(et, oc), tzz_c = LDB(f.tzz, oc), fz_c = NOF >= 2 ? 0.0 : LDB(f.fz, oc);
    double s10 = LDB(f.txz, oxz + 8u - sxz), r10 = LDB(f.tyz, oyz + ryz - syz);

    for (int k = kb; k < kend; ++k) {
        const bool hz = k < nz - 1;
        const u32 dz1 = hz ? sc : 0u;
        const double q10 = LDB(f.txy, oxy + 8u), q01 = LDB(f.txy, oxy + rxy);
        const double s01 = LDB(f.txz, oxz);
        const double r11 = LDB(f.tyz, oyz + ryz), r01 = LDB(f.tyz, oyz);
        const double Pz = LDB(f.P, oc + dz1), ez = LDB(et, oc + dz1), tzz_z = LDB(f.tzz, oc + dz1), fz_z = NOF >= 2 ? 0.0 : LDB(f.fz, oc + dz1);
        const double Py = LDB(f.P, oc + dy1), ey = LDB(et, oc + dy1);
        const double txx_c = LDB(f.txx, oc), tyy_c = LDB(f.tyy, oc), tyy_y = LDB(f.tyy, oc + dy1);
        const double fx_c = NOF >= 1 ? 0.0 : LDB(f.fx, oc), fy_c = NOF >= 1 ? 0.0 : LDB(f.fy, oc), fy_y = NOF >= 1 ? 0.0 : LDB(f.fy, oc + dy1);
        const double vx = LDB(f.Vx, ovx), vy = LDB(f.Vy, ovy), vz = LDB(f.Vz, ovz);
        double q11, s11, Px, ex, txx_x, fx_x;
        if (SHF) {
            q11 = __shfl_down(q01, 1, 64); s11 = __shfl_down(s01, 1, 64);
            Px = __shfl_down(Pc, 1, 64); ex = __shfl_down(ec, 1, 64); txx_x = __shfl_down(txx_c, 1, 64); fx_x = NOF >= 1 ? 0.0 : __shfl_down(fx_c, 1, 64);
            if ((threadIdx.x & 63) == 63 || i == a.i1 - 1) {
                q11 = LDB(f.txy, oxy + 8u + rxy); s11 = LDB(f.txz, oxz + 8u);
                Px = LDB(f.P, oc + dx1); ex = LDB(et, oc + dx1); txx_x = LDB(f.txx, oc + dx1); fx_x = NOF >= 1 ? 0.0 : LDB(f.fx, oc + dx1);
            }
        } else {
            q11 = LDB(f.txy, oxy + 8u + rxy); s11 = LDB(f.txz, oxz + 8u);
            Px = LDB(f.P, oc + dx1); ex = LDB(et, oc + dx1); txx_x = LDB(f.txx, oc + dx1); fx_x = NOF >= 1 ? 0.0 : LDB(f.fx, oc + dx1);
        }
        if (hx) {
            const double R0 = (-txx_c + txx_x) * _dx + _dy * (q11 - q10) + _dz * (s11 - s10) - (-Pc + Px) * _dx;
            const double R = NOF >= 1 ? R0 : R0 - 0.5 * (fx_c + fx_x);       // x - (+0.0) = x for every x
            STB(a.o.Vx, ovx, vx + R * edt / (0.5 * (ec + ex)));
            if (DIAG) STB(f.Rx, orx, R);
        }
        if (hy) {
            const double R0 = _dx * (q11 - q01) + _dy * (tyy_y - tyy_c) + _dz * (r11 - r10) - (-Pc + Py) * _dy;
            const double R = NOF >= 1 ? R0 : R0 - 0.5 * (fy_c + fy_y);
            STB(a.o.Vy, ovy, vy + R * edt / (0.5 * (ec + ey)));
            if (DIAG) STB(f.Ry, ory, R);
        }
        if (hz) {
            const double R0 = _dx * (s11 - s01) + _dy * (r11 - r01) + (-tzz_c + tzz_z) * _dz - (-Pc + Pz) * _dz;
            const double R = NOF >= 2 ? R0 : R0 - 0.5 * (fz_c + fz_z);
            STB(a.o.Vz, ovz, vz + R * edt / (0.5 * (ec + ez)));
            if (DIAG) STB(f.Rz, oc, R);
        }
        Pc = Pz; ec = ez; tzz_c = tzz_z; fz_c = fz_z; s10 = s11; r10 = r11;
        oc += sc; oxy += sxy; oxz += sxz; oyz += syz; ovx += svx; ovy += svy; ovz += svz; orx += srx; ory += sry;
    }
}

// ================================================================================================
// Fused iteration kernel: velocity sweep of iteration m + flow BCs (low faces) + stress sweep of
// iteration m+1 in one pass over memory.
//
//   reads  (src set)  P τ(6) V(3) f(3) ητ | P0 Q η K G τ_o(6)      = 25 array passes
//   writes (dst set)  V(3) P τ(6)                                    = 10 array passes
//
// i.e. 35 instead of the 45 passes of the two separate sweeps: P and τ(6) are not re-read and the new
// V(3) goes from the velocity update to the strain rates through LDS.  src and dst are different
// buffers (ping-pong between the caller's arrays and a library-owned scratch set): the velocity update
// of a neighbouring tile still needs the old P, τ while this tile already writes the new ones.
//
// Tiles overlap by one cell row/column on the low side: a block of TX x TY threads updates the
// velocities of TX x TY cell columns ("B cells") and the stresses of the (TX-1) x (TY-1) columns
// whose low-side neighbours it has ("A cells"): stress at cell (i,j,k) needs the new velocities of
// the cells (i,j,k) (i-1,j,k) (i,j-1,k) (i-1,j-1,k) (i-1,j,k-1) (i,j-1,k-1) (i,j,k-1) only.
// The block marches KZ planes in z (plus one prologue plane below the chunk); per plane: B phase
// -> new V into LDS (+ global for the cells the block owns) -> __syncthreads -> A phase.
// Ghost / boundary-plane velocities on the LOW faces are produced on the fly from the boundary
// condition of that face (free slip: copy, no slip: negate / zero, none: the prescribed value in
// memory); the high-face planes i = nx, j = ny, k = nz are finished by the per-node kernel after the
// regular BC kernels have run on the new V.  Periodic faces and multi-rank halos are not fused.
// ================================================================================================
struct FusedBC {
    // low faces: free-slip / no-slip flags; high faces: "normal velocity = 0" for the cells, the free-slip flags for the folded high-face node layers (HIF)
    int fsL, nsL, fsF, nsF, fsK0, nsK0, nsR, nsBk, nsK1, fsR, fsBk, fsK1;
    // NBR: faces with a neighbour (their fs / ns flags are cleared): the planes of the DESTINATION set there hold received velocities
    int nbL, nbR, nbF, nbBk, nbK0, nbK1;
    int feed;      // NBR: the idle feeder lane of the tiles on a low x face with a neighbour holds the received plane (tuning switch "nbr_feeder", default 1)
};
// NBR: the tiles of a block as a list of disjoint boxes of tiles, launched in two classes.  Box 0 = tiles that touch no face with a neighbour (a first share of them: enough work
// to cover update_halo!(V), which runs beside it); boxes 1.. = everything else -- the rest of the block in its natural XCD-banded order, shell tiles included, plus the shell
// tiles beside box 0 -- launched (blk0 = start[1]) once the exchange has delivered the planes they read
struct FusedShell {
    int nbox, box[7][6], start[8], banded[7];
    int blk0, cls;           // first block of the launch; cls (host side only): 0 = box 0, 1 = the boxes behind it
};
// coherent load (device scope): a value another kernel of this device has written while this one runs
__device__ __forceinline__ double LDC(const double *p, u32 off)
{
    const unsigned long long v = __hip_atomic_load((const unsigned long long *)((const char *)p + off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __longlong_as_double((long long)v);
}

// OVX: x-overlap of neighbouring tiles in cells (1, or 16 = one 128-B line so that row segments stay line-aligned; only
// the last overlap column is computed).  LOWREG: the previous velocity plane is re-read from a third LDS slot and the
// previous η/G plane is carried as two partial sums instead of 11 carried doubles.
// SHFL: x-neighbour operands come from the adjacent lane instead of a second load of the same array (7 velocity-phase loads at i+1, 4
// stress-phase loads at i-1): fewer vector-memory instructions through the L1/TA path (measured -4.7 %).  Lane TX-1 then only feeds
// its left neighbour (tile stride TX - OVX - 1), and the lanes left of the stress tile also load η, G for theirs.
// TX = 32 (with SHFL): a wave holds two rows of the tile (lane shuffles of width 32), so that 256 threads form a 32 x 8 tile with 30 x 7
// stress columns (82 % of the threads) instead of 64 x 4 with 62 x 3 (73 %), and rows quantise in steps of 30 columns (nx = 256: 9 half-wave
// tiles = 288 lanes instead of 5 x 64 = 320).
// YLDS (with SHFL): y-neighbour operands come from the adjacent row of the tile through LDS: every lane publishes P, ητ, τyy, fy, τxy, τyz
// (the row below reads them as its j+1 operands) and η, G (the row above reads them as its j-1 operands); only the top row of the tile
// and the row on the domain's back face still load the j+1 operands from memory.  One more barrier per plane, 8 fewer loads per lane.
// VISC: the viscous limit dt = Inf (SolVi3D, Burstedde, TaylorGreen run there).  1/(G dt), 1/(K dt) and 1/dt are exactly 0 then, so the
// operands they multiply -- the six old stresses, P0, K, G, Q -- cannot change the result for finite inputs and are not loaded: 15 read
// and 10 written arrays per launch instead of 25 and 10.  The arithmetic is the general one with those factors set to 0.
// TAG: only gives the launches over the high-face tiles (halo stream, see iter_step) a kernel name of their own, so that a profile
// separates them from the launch over the interior tiles.
// HIF (viscous-limit form, no neighbours): the stress nodes on the high faces i = nx, j = ny, k = nz -- which no cell column owns and which the boundary-layer launch
// (k_stress3d_boxes with the flow_bcs! rules) otherwise updates behind this kernel -- are updated here by the threads of the last cell column / row / plane, from the
// new velocities they hold anyway and the same rules (GhostRule), operation for operation as stress3d_node<false, true, true>: one launch per iteration.
template <int TX, int TY, int KZ, int MINW, int OVX = 1, bool LOWREG = false, int XG = 0, bool LATEA = false, bool SHFL = false, int YLDS = 0, int NT = 0, int TAG = 0, bool VISC = false, bool HIF = false, bool VFOLD = false, bool NBR = false, int NOF = 0, int YM = 1>
__global__ __launch_bounds__(TX *TY, MINW) void k_fused3d(const SweepArgs a, const FusedBC bc, int ntx, int nty, int tx0 = 0, int ty0 = 0, int tz0 = 0, const FusedShell sh = FusedShell{})
{
    // the launch covers the box of tiles [tx0, tx0+ntx) x [ty0, ty0+nty) x [tz0, tz0 + gridDim.x/(ntx*nty))
    static_assert(!(SHFL && (LATEA || (TX != 64 && TX != 32))), "SHFL is implemented for rows of one wave or half a wave");
    static_assert(!YLDS || SHFL, "YLDS builds on the SHFL operand layout");
    static_assert(!VISC || YLDS, "the viscous-limit form is built on the YLDS operand layout");
    static_assert(!HIF || (!LOWREG && SHFL && OVX == 1 && YLDS == 3), "the folded high-face layers are built on the forms with carried planes");
    static_assert(!VFOLD || VISC, "VFOLD simplifies the viscous-limit arithmetic");
    static_assert(!NBR || HIF, "the in-kernel neighbour faces are built on the one-launch forms");
    // YM > 1 (round 6): the block marches YM consecutive tiles in y.  Row 0 of a tile only feeds the velocity phase -- it recomputes what the top row of the tile below it
    // computes as its own (the y halo: 1 / (TY - 1) of every velocity-phase operand is read twice) -- so the top row leaves its new velocities and its η, plane by plane, in
    // LDS (sT) and row 0 of the next tile of the march takes them from there instead of loading anything: the halo is paid once per YM tiles.  Same values, same bits.
    static_assert(YM == 1 || (VISC && HIF && VFOLD && !NBR && !LOWREG && YLDS == 3), "the y march is built on the one-launch viscous-limit form");
    constexpr int NS = LOWREG ? 3 : 2;
    __shared__ double sT[YM > 1 ? KZ + 1 : 1][4][YM > 1 ? TX : 1];
    __shared__ double sV[NS][3][TY][TX];
    __shared__ double sY[YLDS ? (VISC ? 7 : 8) : 1][YLDS ? TY : 1][YLDS ? TX : 1];
    const Lay3 &L = a.L;
    const int nx = L.nx, ny = L.ny, nz = L.nz;
    const jrx_stokes3d_fields &f = a.f;
    const double *et = a.etatau;
    // a boundary entry of V read from memory: on a face with a neighbour it is in the destination set (received in this iteration), else in the source set (prescribed, static)
#define NBV(flag_, arr_, off_) ((NBR && bc.flag_) ? LDC(a.o.arr_, (off_)) : LDB(f.arr_, (off_)))
    // ... where such an entry also lies on the normal plane of a physical no-slip face it is zero by rule (vx_rule / vy_rule / vz_rule check that first): flow_bcs! is not applied in
    // memory in this pipeline, so the received plane does not carry the zero.  ii_ / jj_ / kk_: the entry's index along the array's own direction
#define NBVX(flag_, off_, ii_) ((NBR && bc.flag_) ? ((((ii_) == 0 && bc.nsL) || ((ii_) == nx && bc.nsR)) ? 0.0 : LDC(a.o.Vx, (off_))) : LDB(f.Vx, (off_)))
#define NBVY(flag_, off_, jj_) ((NBR && bc.flag_) ? ((((jj_) == 0 && bc.nsF) || ((jj_) == ny && bc.nsBk)) ? 0.0 : LDC(a.o.Vy, (off_))) : LDB(f.Vy, (off_)))
#define NBVZ(flag_, off_, kk_) ((NBR && bc.flag_) ? ((((kk_) == 0 && bc.nsK0) || ((kk_) == nz && bc.nsK1)) ? 0.0 : LDC(a.o.Vz, (off_))) : LDB(f.Vz, (off_)))
    // NOF: body-force arrays that are +0.0 in every entry (the driver's operand pass has seen all their bits zero) are not loaded: 1 = fx, fy (gravity along z), 2 = all three
    static_assert(NOF == 0 || (VISC && HIF && VFOLD) || (!VISC && (LOWREG || HIF) && YLDS == 3), "the forms without body-force loads exist for the one-launch viscous-limit form and for the general forms");
#define LFX(off_) (NOF >= 1 ? 0.0 : LDN<(NT & 2) != 0>(f.fx, (off_)))
#define LFY(off_) (NOF >= 1 ? 0.0 : LDN<(NT & 2) != 0>(f.fy, (off_)))
#define LFZ(off_) (NOF >= 2 ? 0.0 : LDN<(NT & 2) != 0>(f.fz, (off_)))
    const int tx = (int)(threadIdx.x % TX), ty = (int)(threadIdx.x / TX);
    int tile = blockIdx.x;
    int tix, tiy, tiz;
    if (NBR) {
        tile += sh.blk0;
        int b = 0;
        while (b + 1 < sh.nbox && tile >= sh.start[b + 1]) b++;
        int l = tile - sh.start[b];
        const int bw = sh.box[b][1] - sh.box[b][0], bh = sh.box[b][3] - sh.box[b][2], bd = sh.box[b][5] - sh.box[b][4];
        if (XG > 0 && sh.banded[b]) {       // XCD-banded order inside the box (see below)
            const int full = ((bh * bd) / (8 * XG)) * (8 * XG) * bw;
            if (l < full) {
                const int q = l & 7, r = l >> 3, r2 = r / bw;
                l = ((r2 / XG) * (8 * XG) + q * XG + r2 % XG) * bw + r % bw;
            }
        }
        tix = sh.box[b][0] + l % bw; tiy = sh.box[b][2] + (l / bw) % bh; tiz = sh.box[b][4] + l / (bw * bh);
    } else {
        const int ntyS = (nty + YM - 1) / YM;      // rows of blocks: a block marches YM tile rows
        if (XG > 0) {
            // XCD-banded order (blocks are dealt round-robin to the 8 XCDs): XCD q takes XG consecutive tile rows
            // q*XG .. q*XG+XG-1 of every group of 8*XG rows (rows run over y, then z), so the y-halo rows of
            // neighbouring tiles are served by the same L2; the tail that does not fill a group keeps plain order
            const int full = ((ntyS * (int)(gridDim.x / (unsigned)(ntx * ntyS))) / (8 * XG)) * (8 * XG) * ntx;
            if (tile < full) {
                const int q = tile & 7, r = tile >> 3, r2 = r / ntx;
                tile = ((r2 / XG) * (8 * XG) + q * XG + r2 % XG) * ntx + r % ntx;
            }
        }
        const int tr = tile / ntx;
        tix = tx0 + tile % ntx; tiy = ty0 + (tr % ntyS) * YM; tiz = tz0 + tr / ntyS;
    }
    const int tiy_end = NBR ? tiy + 1 : min(tiy + YM, ty0 + nty);       // the tiles [tiy, tiy_end) of the march
    for (const int tiy_first = tiy; tiy < tiy_end; ++tiy) {
    const bool fed = YM > 1 && tiy > tiy_first && threadIdx.x < TX;        // row 0 of a tile behind the first of the march: fed from sT (a whole wave: uniform)
    const bool feeds = YM > 1 && tiy + 1 < tiy_end && (int)(threadIdx.x / TX) == TY - 1;     // the top row of a tile with a successor in the march
    const int i = tix * (TX - OVX - (SHFL ? 1 : 0)) - OVX + tx;  // cell column of this thread
    const int j = tiy * (TY - 1) - 1 + ty;
    const int kb = tiz * KZ;
    const int kend = min(kb + KZ, nz);
    const bool bvalid = tx >= OVX - 1 && i >= 0 && j >= 0 && i < nx && j < ny;
    const bool avalid = bvalid && tx >= OVX && ty >= 1 && (!SHFL || tx < TX - 1);
    // NBR, a tile on the low x face of a block with a neighbour there: lane 0 sits on column i = -1 and is idle otherwise -- it holds the RECEIVED plane (Vx[0], the ghost columns
    // Vy[0], Vz[0] of the new set) as "the velocities of column -1": loaded with the other operands of the plane, published to sV like any lane's, read by the lane of column 0
    // through the generic i - 1 path.  (Round 6: before, column 0 fetched those entries itself behind the second barrier -- four dependent single-lane loads in the stress phase
    // of every plane: the rank with a low x neighbour ran its kernel 6 % slower than the rank with a high one, profiles/r06_low_face_feeder.txt.)
    const bool feedL = NBR && SHFL && OVX == 1 && YLDS && !LOWREG && bc.feed && bc.nbL && tx == 0 && i == -1 && j >= 0 && j < ny;
    const bool hx = i < nx - 1, hy = j < ny - 1;
    const double _dx = a._dx, _dy = a._dy, _dz = a._dz, dt = a.dt, th = a.theta_dtau, _dt = 1.0 / dt, rr = a.r, edt = a.eta_dtau;
    // VFOLD: what the viscous-limit arithmetic reduces to for FINITE η (the driver's operand check guarantees it), bit for bit:
    //   dev_dtau_r(θ, η, 0) = 1 / (θ + fma(η, 0, 1)) = 1 / (θ + 1): one division per thread instead of four per cell and plane;
    //   INC(τ, 0, η, ε, 0, dτ_r) = dτ_r fma(2η, ε, fma(-(τ - 0) η, 0, -τ)) = dτ_r fma(2η, ε, -τ)   (x · 0 = ±0, and ±0 + (-τ) = -τ, the zeros included);
    //   compute_P!: fma(0, 0, rhs) = rhs (rhs = -∇V + 0 is never -0) and the division by 1 + 0 ψ = 1 is the identity.
    // The fused kernel spends more than half of its time issuing fp64 VALU instructions, a good part of them division sequences (profiles/r04_temporal_blocking.txt §4).
    const double dtr0 = VFOLD ? 1.0 / (th + 1.0) : 0.0;
    auto DTR = [&](const double eta_, const double Gdt_) { return VFOLD ? dtr0 : dev_dtau_r(th, eta_, Gdt_); };
    auto INC = [&](const double t_, const double to_, const double eta_, const double e_, const double Gdt_, const double dtr_) {
        return VFOLD ? dtr_ * fma(2.0 * eta_, e_, -t_) : dev_stress_inc(t_, to_, eta_, e_, Gdt_, dtr_);
    };

    const u32 sc = (u32)L.cp * 8u, svx = (u32)L.vxp * 8u, svy = (u32)L.vyp * 8u, svz = (u32)L.vzp * 8u;
    const u32 sxy = (u32)L.xyp * 8u, sxz = (u32)L.xzp * 8u, syz = (u32)L.yzp * 8u;
    const u32 rc = (u32)nx * 8u, rxy = (u32)L.xy1 * 8u, ryz = (u32)L.yz1 * 8u;
    const u32 rvx = (u32)L.vx1 * 8u, rvy = (u32)L.vy1 * 8u, rvz = (u32)L.vz1 * 8u;
    const int ic = bvalid ? i : (feedL ? -1 : 0), jc = (bvalid || feedL) ? j : 0;          // keep addresses in range for idle threads (a feeder lane only uses ovx / ovy / ovz: column ic + 1 = 0)
    const int kfirst = kb > 0 ? kb - 1 : 0;
    u32 oc = 8u * (u32)(ic + nx * jc) + sc * (u32)kfirst;
    u32 oxy = 8u * (u32)(ic + L.xy1 * jc) + sxy * (u32)kfirst;
    u32 oxz = 8u * (u32)(ic + L.xz1 * jc) + sxz * (u32)(kfirst + 1);
    u32 oyz = 8u * (u32)(ic + L.yz1 * jc) + syz * (u32)(kfirst + 1);
    u32 ovx = 8u * (u32)((ic + 1) + L.vx1 * (jc + 1)) + svx * (u32)(kfirst + 1);   // Vx[i+1, j+1, k+1]
    u32 ovy = 8u * (u32)((ic + 1) + L.vy1 * (jc + 1)) + svy * (u32)(kfirst + 1);
    u32 ovz = 8u * (u32)((ic + 1) + L.vz1 * (jc + 1)) + svz * (u32)(kfirst + 1);
    const u32 dx1 = hx ? 8u : 0u, dy1 = hy ? rc : 0u;
    const int im = max(ic - 1, 0), jm = max(jc - 1, 0);
    const u32 dcx = 8u * (u32)(ic - im), dcy = rc * (u32)(jc - jm);

    // velocity-sweep carries (plane k values that were the upper loads of the previous plane)
    double Pc = 0, ec = 0, tzz_c = 0, fz_c = 0, s10 = 0, r10 = 0, s01p = 0, r01p = 0;
    if (bvalid && !fed) {
        Pc = LDB(f.P, oc); ec = LDB(et, oc); tzz_c = LDB(f.tzz, oc); fz_c = LFZ(oc);
        s10 = LDB(f.txz, oxz + 8u - sxz); r10 = LDB(f.tyz, oyz + ryz - syz);
        s01p = LDB(f.txz, oxz - sxz); r01p = LDB(f.tyz, oyz - syz);
    }
    // stress-sweep carries
    double a_p = 0, b_p = 0, c_p = 0, cx_p = 0, cy_p = 0, e_p = 0, ex_p = 0, ey_p = 0, g_p = 0, gx_p = 0, gy_p = 0;
    double exe_p = 0, eye_p = 0, gxg_p = 0, gyg_p = 0;      // LOWREG: (ex_p + e_p), (ey_p + e_p), same for G
    double vax_p = 0, vby_p = 0;                            // HIF: the previous plane of Vx[nx, j+1, ·] / Vy[i+1, ny, ·] (last cell column / row only)

    for (int k = kfirst; k < kend; ++k) {
        const bool hz = k < nz - 1;
        const bool live = k >= kb;                 // false only on the prologue plane below the chunk
        const int slot = LOWREG ? k % 3 : (k & 1);
        double vxn = 0, vyn = 0, vzn = 0, txx_c = 0, tyy_c = 0, P_k = Pc, tzz_k = tzz_c, s01k = s01p, r01k = r01p;
        const double s10k = s10, r10k = r10;       // HIF: the old τxz[i+1, j, k] / τyz[i, j+1, k] (s10 / r10 move on to plane k + 1 in the velocity phase)
        // stress-sweep operands that do not depend on the new velocities: issued before the barrier so that
        // one memory round trip per plane serves both phases
        double e = 0, ex = 0, ey = 0, exy_ = 0, g = 0, gx = 0, gy = 0, gxy = 0, P0 = 0, Kc = 0, Qc = 0;
        double toxx = 0, toyy = 0, tozz = 0, txy = 0, toxy = 0, toxz = 0, toyz = 0;
        auto preload_stress = [&]() {
            if (SHFL) {
                // η, G of the own column for every lane that has one; the i-1 column arrives by lane shuffle (clamped at i = 0)
                if (YLDS) {
                    if (fed) e = sT[k - kfirst][3][tx];
                    else if (bvalid) { e = LDB(f.eta, oc); if (!VISC) g = LDB(f.G, oc); }
                } else if (bvalid) {
                    e = LDB(f.eta, oc); ey = LDB(f.eta, oc - dcy); g = LDB(f.G, oc); gy = LDB(f.G, oc - dcy);
                    const double e_l = __shfl_up(e, 1, TX), g_l = __shfl_up(g, 1, TX), ey_l = __shfl_up(ey, 1, TX), gy_l = __shfl_up(gy, 1, TX);
                    ex = i > 0 ? e_l : e; gx = i > 0 ? g_l : g; exy_ = i > 0 ? ey_l : ey; gxy = i > 0 ? gy_l : gy;
                }
                if (YLDS < 2 && !VISC && avalid && live) {
                    P0 = LDN<(NT & 2) != 0>(f.P0, oc); Kc = LDN<(NT & 2) != 0>(f.K, oc); Qc = LDN<(NT & 2) != 0>(f.Q, oc);
                    toxx = LDN<(NT & 2) != 0>(f.toxx, oc); toyy = LDN<(NT & 2) != 0>(f.toyy, oc); tozz = LDN<(NT & 2) != 0>(f.tozz, oc);
                    toxy = LDN<(NT & 2) != 0>(f.toxy, oxy);
                    toxz = LDN<(NT & 2) != 0>(f.toxz, oxz - sxz); toyz = LDN<(NT & 2) != 0>(f.toyz, oyz - syz);
                }
                return;
            }
            if (avalid) {
                e = LDB(f.eta, oc); ex = LDB(f.eta, oc - dcx); ey = LDB(f.eta, oc - dcy);
                g = LDB(f.G, oc); gx = LDB(f.G, oc - dcx); gy = LDB(f.G, oc - dcy);
                if (live) {
                    exy_ = LDB(f.eta, oc - dcx - dcy); gxy = LDB(f.G, oc - dcx - dcy);
                    P0 = LDN<(NT & 2) != 0>(f.P0, oc); Kc = LDN<(NT & 2) != 0>(f.K, oc); Qc = LDN<(NT & 2) != 0>(f.Q, oc);
                    toxx = LDN<(NT & 2) != 0>(f.toxx, oc); toyy = LDN<(NT & 2) != 0>(f.toyy, oc); tozz = LDN<(NT & 2) != 0>(f.tozz, oc);
                    txy = LDB(f.txy, oxy); toxy = LDN<(NT & 2) != 0>(f.toxy, oxy);
                    toxz = LDN<(NT & 2) != 0>(f.toxz, oxz - sxz); toyz = LDN<(NT & 2) != 0>(f.toyz, oyz - syz);
                }
            }
        };
        if (!LATEA) preload_stress();
        // velocity-phase operands
        double q01 = 0, s01 = 0, r11 = 0, r01 = 0, Pz = 0, ez = 0, tzz_z = 0, fz_z = 0, Py = 0, eyb = 0, tyy_y = 0;
        double fx_c = 0, fy_c = 0, fy_y = 0, vx = 0, vy = 0, vz = 0, txy_own = 0;
        const bool yrow = YLDS && ty < TY - 1 && hy;     // the j+1 operands are the next row's own operands (row = wave: uniform)
        if (fed) {
            // the new velocities of this row and plane, as the top row of the previous tile of the march computed them (read before the barrier: that row writes this plane's
            // entry of sT again behind it)
            vxn = sT[k - kfirst][0][tx]; vyn = sT[k - kfirst][1][tx]; vzn = sT[k - kfirst][2][tx];
            sY[6][ty][tx] = e;
        } else if (NBR && feedL) {
            vx = LDC(a.o.Vx, ovx); vy = LDC(a.o.Vy, ovy); vz = LDC(a.o.Vz, ovz);
        } else if (bvalid) {
            const u32 dz1 = hz ? sc : 0u;
            if (YLDS) {
                // published operands first (loads return in order), then the rest
                tyy_c = LDB(f.tyy, oc); fy_c = LFY(oc); txy_own = LDB(f.txy, oxy); r01 = LDB(f.tyz, oyz);
                if (!yrow) {
                    q01 = LDB(f.txy, oxy + rxy); r11 = LDB(f.tyz, oyz + ryz);
                    if (hy) { Py = LDB(f.P, oc + rc); eyb = LDB(et, oc + rc); tyy_y = LDB(f.tyy, oc + rc); fy_y = LFY(oc + rc); }
                }
                s01 = LDB(f.txz, oxz);
                Pz = LDB(f.P, oc + dz1); ez = LDB(et, oc + dz1); tzz_z = LDB(f.tzz, oc + dz1); fz_z = LFZ(oc + dz1);
                txx_c = LDB(f.txx, oc); fx_c = LFX(oc);
                vx = LDB(f.Vx, ovx); vy = LDB(f.Vy, ovy); vz = LDB(f.Vz, ovz);
                if (NBR) {      // the boundary planes i = nx, j = ny, k = nz of a face with a neighbour: received values
                    if (!hx && bc.nbR) vx = LDC(a.o.Vx, ovx);
                    if (!hy && bc.nbBk) vy = LDC(a.o.Vy, ovy);
                    if (!hz && bc.nbK1) vz = LDC(a.o.Vz, ovz);
                }
                if (YLDS == 2 && !VISC && avalid && live) {
                    // the stress phase's remaining operands queue behind the published ones
                    P0 = LDN<(NT & 2) != 0>(f.P0, oc); Kc = LDN<(NT & 2) != 0>(f.K, oc); Qc = LDN<(NT & 2) != 0>(f.Q, oc);
                    toxx = LDN<(NT & 2) != 0>(f.toxx, oc); toyy = LDN<(NT & 2) != 0>(f.toyy, oc); tozz = LDN<(NT & 2) != 0>(f.tozz, oc);
                    toxy = LDN<(NT & 2) != 0>(f.toxy, oxy);
                    toxz = LDN<(NT & 2) != 0>(f.toxz, oxz - sxz); toyz = LDN<(NT & 2) != 0>(f.toyz, oyz - syz);
                }
                sY[0][ty][tx] = Pc; sY[1][ty][tx] = ec; sY[2][ty][tx] = tyy_c; if (NOF < 1) sY[3][ty][tx] = fy_c; sY[4][ty][tx] = txy_own; sY[5][ty][tx] = r01;
                sY[6][ty][tx] = e;
                if (!VISC) sY[7][ty][tx] = g;
            } else {
                q01 = LDB(f.txy, oxy + rxy); s01 = LDB(f.txz, oxz);
                r11 = LDB(f.tyz, oyz + ryz); r01 = LDB(f.tyz, oyz);
                Pz = LDB(f.P, oc + dz1); ez = LDB(et, oc + dz1); tzz_z = LDB(f.tzz, oc + dz1); fz_z = LFZ(oc + dz1);
                Py = LDB(f.P, oc + dy1); eyb = LDB(et, oc + dy1);
                txx_c = LDB(f.txx, oc); tyy_c = LDB(f.tyy, oc);
                tyy_y = LDB(f.tyy, oc + dy1);
                fx_c = LFX(oc); fy_c = LFY(oc); fy_y = LFY(oc + dy1);
                vx = LDB(f.Vx, ovx); vy = LDB(f.Vy, ovy); vz = LDB(f.Vz, ovz);
            }
        }
        if (YLDS) __syncthreads();
        if (fed) {
            sV[slot][0][ty][tx] = vxn; sV[slot][1][ty][tx] = vyn; sV[slot][2][ty][tx] = vzn;
        } else if (NBR && feedL) {
            // (Vz on the normal plane k + 1 = nz of a physical no-slip face is zero by rule -- flow_bcs! is not applied in memory in this pipeline, NBVZ)
            sV[slot][0][ty][tx] = vx; sV[slot][1][ty][tx] = vy; sV[slot][2][ty][tx] = (!hz && bc.nsK1) ? 0.0 : vz;
        } else if (bvalid) {
            if (YLDS) {
                if (yrow) {
                    Py = sY[0][ty + 1][tx]; eyb = sY[1][ty + 1][tx]; tyy_y = sY[2][ty + 1][tx]; if (NOF < 1) fy_y = sY[3][ty + 1][tx];
                    q01 = sY[4][ty + 1][tx]; r11 = sY[5][ty + 1][tx];
                }
                // stress phase: η, G at j-1 from the row below (clamped at j = 0), then the i-1 column by lane shuffle (clamped at i = 0)
                if (ty > 0 && j > 0) { ey = sY[6][ty - 1][tx]; if (!VISC) gy = sY[7][ty - 1][tx]; }
                else { ey = e; gy = g; }
                const double e_l = __shfl_up(e, 1, TX), ey_l = __shfl_up(ey, 1, TX);
                ex = i > 0 ? e_l : e; exy_ = i > 0 ? ey_l : ey;
                if (!VISC) {
                    const double g_l = __shfl_up(g, 1, TX), gy_l = __shfl_up(gy, 1, TX);
                    gx = i > 0 ? g_l : g; gxy = i > 0 ? gy_l : gy;
                }
            }
            double q11, q10, s11, Px, ex, txx_x, fx_x;
            if (SHFL) {
                // the operands at i+1 are the right-hand lane's operands at i (every lane of the row holds them, incl. the feeder lane TX-1);
                // they are only used where hx, i.e. where that lane exists
                if (!YLDS) txy_own = LDB(f.txy, oxy);
                txy = txy_own;                                  // also the stress phase's own τxy
                q11 = __shfl_down(q01, 1, TX); q10 = __shfl_down(txy_own, 1, TX); s11 = __shfl_down(s01, 1, TX);
                Px = __shfl_down(Pc, 1, TX); ex = __shfl_down(ec, 1, TX); txx_x = __shfl_down(txx_c, 1, TX); fx_x = NOF >= 1 ? 0.0 : __shfl_down(fx_c, 1, TX);
                // the last cell column has no lane to its right, but its y- and z-momentum still need the shear stresses on the
                // domain's right face (τxy, τxz have nx+1 columns)
                if (!hx) { q11 = LDB(f.txy, oxy + 8u + rxy); s11 = LDB(f.txz, oxz + 8u); }
            } else {
                q11 = LDB(f.txy, oxy + 8u + rxy); q10 = LDB(f.txy, oxy + 8u); s11 = LDB(f.txz, oxz + 8u);
                Px = LDB(f.P, oc + dx1); ex = LDB(et, oc + dx1); txx_x = LDB(f.txx, oc + dx1); fx_x = LFX(oc + dx1);
            }
            const bool own = avalid && live;
            if (hx) {
                // NOF: x - (+0.0) = x for every x, -0.0 and NaN included, so the term is left out when the array is known to hold nothing but +0.0
                const double R0 = (-txx_c + txx_x) * _dx + _dy * (q11 - q10) + _dz * (s11 - s10) - (-Pc + Px) * _dx;
                const double R = NOF >= 1 ? R0 : R0 - 0.5 * (fx_c + fx_x);
                vxn = vx + R * edt / (0.5 * (ec + ex));
                if (own) STN<(NT & 1) != 0>(a.o.Vx, ovx, vxn);
            } else vxn = bc.nsR ? 0.0 : vx;
            if (hy) {
                const double R0 = _dx * (q11 - q01) + _dy * (tyy_y - tyy_c) + _dz * (r11 - r10) - (-Pc + Py) * _dy;
                const double R = NOF >= 1 ? R0 : R0 - 0.5 * (fy_c + fy_y);
                vyn = vy + R * edt / (0.5 * (ec + eyb));
                if (own) STN<(NT & 1) != 0>(a.o.Vy, ovy, vyn);
            } else vyn = bc.nsBk ? 0.0 : vy;
            if (hz) {
                const double R0 = _dx * (s11 - s01) + _dy * (r11 - r01) + (-tzz_c + tzz_z) * _dz - (-Pc + Pz) * _dz;
                const double R = NOF >= 2 ? R0 : R0 - 0.5 * (fz_c + fz_z);
                vzn = vz + R * edt / (0.5 * (ec + ez));
                if (own) STN<(NT & 1) != 0>(a.o.Vz, ovz, vzn);
            } else vzn = bc.nsK1 ? 0.0 : vz;
            Pc = Pz; ec = ez; tzz_c = tzz_z; fz_c = fz_z; s10 = s11; r10 = r11; s01p = s01; r01p = r01;
            sV[slot][0][ty][tx] = vxn; sV[slot][1][ty][tx] = vyn; sV[slot][2][ty][tx] = vzn;
            if (feeds) { sT[k - kfirst][0][tx] = vxn; sT[k - kfirst][1][tx] = vyn; sT[k - kfirst][2][tx] = vzn; sT[k - kfirst][3][tx] = e; }
        }
        if (YLDS == 3 && !VISC && avalid && live) {
            // lower register peak: the stress-only operands are requested once the velocity operands are consumed
            P0 = LDB(f.P0, oc); Kc = LDB(f.K, oc); Qc = LDB(f.Q, oc);
            toxx = LDB(f.toxx, oc); toyy = LDB(f.toyy, oc); tozz = LDB(f.tozz, oc);
            toxy = LDB(f.toxy, oxy);
            toxz = LDB(f.toxz, oxz - sxz); toyz = LDB(f.toyz, oyz - syz);
        }
        if (LATEA) {
            // lower register peak (4 waves/SIMD): the stress operands are requested only once the velocity
            // operands are consumed; their round trip overlaps the barrier
            __builtin_amdgcn_sched_barrier(0);
            preload_stress();
        }
        __syncthreads();
        if (avalid) {
            // ---- new velocities around cell (i,j,k): LDS, own registers, or the low-face boundary rule
            double va, vay, vb, vbx, vcx, vcy;
            const double vax = vxn, vby = vyn, vc = vzn;
            const u32 gvx = ovx - 8u, gvy = ovy - rvy, gvz = ovz;       // Vx[i,j+1,k+1], Vy[i+1,j,k+1], Vz[i+1,j+1,k+1]
            const bool lx = NBR && SHFL && OVX == 1 && YLDS && !LOWREG && bc.feed && bc.nbL && i == 0;     // column 0 next to a received plane: the lane to its left is the feeder lane (feedL)
            va = (i > 0 || lx) ? sV[slot][0][ty][tx - 1] : (bc.nsL ? 0.0 : NBV(nbL, Vx, gvx));
            if (j > 0) vay = (i > 0 || lx) ? sV[slot][0][ty - 1][tx - 1] : (bc.nsL ? 0.0 : NBV(nbL, Vx, gvx - rvx));
            else vay = bc.fsF ? va : (bc.nsF ? -va : NBVX(nbF, gvx - rvx, i));
            vb = j > 0 ? sV[slot][1][ty - 1][tx] : (bc.nsF ? 0.0 : NBV(nbF, Vy, gvy));
            if (i > 0) vbx = j > 0 ? sV[slot][1][ty - 1][tx - 1] : (bc.nsF ? 0.0 : NBV(nbF, Vy, gvy - 8u));
            else vbx = (lx && j > 0) ? sV[slot][1][ty - 1][tx - 1] : (bc.fsL ? vb : (bc.nsL ? -vb : NBVY(nbL, gvy - 8u, j)));
            vcx = (i > 0 || lx) ? sV[slot][2][ty][tx - 1] : (bc.fsL ? vc : (bc.nsL ? -vc : NBVZ(nbL, gvz - 8u, k + 1)));
            vcy = j > 0 ? sV[slot][2][ty - 1][tx] : (bc.fsF ? vc : (bc.nsF ? -vc : NBVZ(nbF, gvz - rvz, k + 1)));
            if (LOWREG && k > 0 && live) {
                const int ps = (k + 2) % 3;       // slot of plane k-1
                a_p = i > 0 ? sV[ps][0][ty][tx - 1] : (bc.nsL ? 0.0 : LDB(f.Vx, gvx - svx));
                b_p = j > 0 ? sV[ps][1][ty - 1][tx] : (bc.nsF ? 0.0 : LDB(f.Vy, gvy - svy));
                c_p = sV[ps][2][ty][tx];
                cx_p = i > 0 ? sV[ps][2][ty][tx - 1] : (bc.fsL ? c_p : (bc.nsL ? -c_p : LDB(f.Vz, gvz - svz - 8u)));
                cy_p = j > 0 ? sV[ps][2][ty - 1][tx] : (bc.fsF ? c_p : (bc.nsF ? -c_p : LDB(f.Vz, gvz - svz - rvz)));
            }
            if (k == 0) {
                // plane K = 0 of V: ghost of Vx, Vy (tangential), boundary plane of Vz (normal)
                a_p = bc.fsK0 ? va : (bc.nsK0 ? -va : NBVX(nbK0, gvx - svx, i));
                b_p = bc.fsK0 ? vb : (bc.nsK0 ? -vb : NBVY(nbK0, gvy - svy, j));
                c_p = bc.nsK0 ? 0.0 : NBV(nbK0, Vz, gvz - svz);
                cx_p = i > 0 ? (bc.nsK0 ? 0.0 : NBV(nbK0, Vz, gvz - svz - 8u)) : (bc.fsL ? c_p : (bc.nsL ? -c_p : NBVZ(nbL, gvz - svz - 8u, 0)));
                cy_p = j > 0 ? (bc.nsK0 ? 0.0 : NBV(nbK0, Vz, gvz - svz - rvz)) : (bc.fsF ? c_p : (bc.nsF ? -c_p : NBVZ(nbF, gvz - svz - rvz, 0)));
                e_p = e; ex_p = ex; ey_p = ey; g_p = g; gx_p = gx; gy_p = gy;
                exe_p = ex + e; eye_p = ey + e; gxg_p = gx + g; gyg_p = gy + g;
            }
            if (live) {
                {   // centre
                    const double dxi = (-va + vax) * _dx;
                    const double dyi = (-vb + vby) * _dy;
                    const double dzi = (-c_p + vc) * _dz;
                    const double divV = dxi + dyi + dzi;
                    const double _Gdt = VISC ? 0.0 : 1.0 / (g * dt);
                    const double _Kdt = VISC ? 0.0 : 1.0 / (Kc * dt);
                    const double rhs = -divV + (Qc * _dt);
                    const double psi = 1.0 / (1.0 / e + _Gdt) * rr / th;
                    if (VFOLD) STN<(NT & 1) != 0>(a.o.P, oc, rhs * psi + P_k);
                    else STN<(NT & 1) != 0>(a.o.P, oc, (fma(P0, _Kdt, rhs) * psi + P_k) / (1.0 + _Kdt * psi));
                    const double d3 = divV * (1.0 / 3.0);
                    const double exx = dxi - d3, eyy = dyi - d3, ezz = dzi - d3;
                    const double dtr = DTR(e, _Gdt);
                    STN<(NT & 1) != 0>(a.o.txx, oc, txx_c + INC(txx_c, toxx, e, exx, _Gdt, dtr));
                    STN<(NT & 1) != 0>(a.o.tyy, oc, tyy_c + INC(tyy_c, toyy, e, eyy, _Gdt, dtr));
                    STN<(NT & 1) != 0>(a.o.tzz, oc, tzz_k + INC(tzz_k, tozz, e, ezz, _Gdt, dtr));
                }
                {   // τxy (i,j,k)
                    const double s_ = 0.5 * (_dy * (va - vay) + _dx * (vb - vbx));
                    const double ee = 0.25 * (exy_ + ey + ex + e);
                    const double gg = 0.25 * (gxy + gy + gx + g);
                    const double _Gdt = VISC ? 0.0 : 1.0 / (gg * dt);
                    const double dtr = DTR(ee, _Gdt);
                    STN<(NT & 1) != 0>(a.o.txy, oxy, txy + INC(txy, toxy, ee, s_, _Gdt, dtr));
                }
                {   // τxz (i,j,k)
                    const double s_ = 0.5 * (_dz * (va - a_p) + _dx * (c_p - cx_p));
                    const double ee = 0.25 * ((LOWREG ? exe_p : ex_p + e_p) + ex + e);
                    const double gg = 0.25 * ((LOWREG ? gxg_p : gx_p + g_p) + gx + g);
                    const double _Gdt = VISC ? 0.0 : 1.0 / (gg * dt);
                    const double dtr = DTR(ee, _Gdt);
                    STN<(NT & 1) != 0>(a.o.txz, oxz - sxz, s01k + INC(s01k, toxz, ee, s_, _Gdt, dtr));
                }
                {   // τyz (i,j,k)
                    const double s_ = 0.5 * (_dz * (vb - b_p) + _dy * (c_p - cy_p));
                    const double ee = 0.25 * ((LOWREG ? eye_p : ey_p + e_p) + ey + e);
                    const double gg = 0.25 * ((LOWREG ? gyg_p : gy_p + g_p) + gy + g);
                    const double _Gdt = VISC ? 0.0 : 1.0 / (gg * dt);
                    const double dtr = DTR(ee, _Gdt);
                    STN<(NT & 1) != 0>(a.o.tyz, oyz - syz, r01k + INC(r01k, toyz, ee, s_, _Gdt, dtr));
                }
            }
            if (HIF) {
                // ---- high-face node layers i = nx (last cell column) and j = ny (last cell row) of plane k; values on ghost planes by the flow_bcs! rules
                const bool xl = i == nx - 1, yl = j == ny - 1;
                if (k == 0) {
                    // plane K = 0 of Vx[nx, j+1, ·] / Vy[i+1, ny, ·]: tangential ghosts of the low z face (normal planes of no-slip faces are zero before the rule's sign)
                    if (xl) vax_p = bc.nsR ? 0.0 : (bc.fsK0 ? vax : (bc.nsK0 ? -vax : NBV(nbK0, Vx, ovx - svx)));
                    if (yl) vby_p = bc.nsBk ? 0.0 : (bc.fsK0 ? vby : (bc.nsK0 ? -vby : NBV(nbK0, Vy, ovy - svy)));
                }
                if (live) {
                    if (xl) {
                        {   // τxy (nx, j, k): Vx[nx, j, k+1] is the row below's boundary value, Vy[nx+1, j, k+1] the ghost column
                            const double vxl = j > 0 ? sV[slot][0][ty - 1][tx] : (bc.nsR ? 0.0 : (bc.fsF ? vax : (bc.nsF ? -vax : NBV(nbF, Vx, ovx - rvx))));
                            const double vyg = (j == 0 && bc.nsF) ? 0.0 : (bc.fsR ? vb : (bc.nsR ? -vb : NBV(nbR, Vy, ovy - rvy + 8u)));
                            const double s_ = 0.5 * (_dy * (vax - vxl) + _dx * (vyg - vb));
                            const double ee = 0.25 * (ey + ey + e + e);
                            const double _Gdt = VISC ? 0.0 : 1.0 / (0.25 * (gy + gy + g + g) * dt);
                            const double to_ = VISC ? 0.0 : LDB(f.toxy, oxy + 8u);
                            const double dtr = DTR(ee, _Gdt);
                            const double t0 = LDB(f.txy, oxy + 8u);
                            STN<(NT & 1) != 0>(a.o.txy, oxy + 8u, t0 + INC(t0, to_, ee, s_, _Gdt, dtr));
                        }
                        {   // τxz (nx, j, k): Vz[nx+1, j+1, k] is the ghost column of the previous plane's own Vz
                            const double vzg = (k == 0 && bc.nsK0) ? 0.0 : (bc.fsR ? c_p : (bc.nsR ? -c_p : NBV(nbR, Vz, ovz - svz + 8u)));
                            const double s_ = 0.5 * (_dz * (vax - vax_p) + _dx * (vzg - c_p));
                            const double ee = 0.25 * (e_p + e_p + e + e);
                            const double _Gdt = VISC ? 0.0 : 1.0 / (0.25 * (g_p + g_p + g + g) * dt);
                            const double to_ = VISC ? 0.0 : LDB(f.toxz, oxz - sxz + 8u);
                            const double dtr = DTR(ee, _Gdt);
                            STN<(NT & 1) != 0>(a.o.txz, oxz - sxz + 8u, s10k + INC(s10k, to_, ee, s_, _Gdt, dtr));
                        }
                    }
                    if (yl) {
                        {   // τxy (i, ny, k): Vx[i, ny+1, k+1] is the ghost row, Vy[i, ny, k+1] the left neighbour's boundary value
                            const double vxg = (i == 0 && bc.nsL) ? 0.0 : (bc.fsBk ? va : (bc.nsBk ? -va : NBV(nbBk, Vx, ovx - 8u + rvx)));
                            const double vyl = i > 0 ? sV[slot][1][ty][tx - 1] : (bc.nsBk ? 0.0 : (bc.fsL ? vby : (bc.nsL ? -vby : NBV(nbL, Vy, ovy - 8u))));
                            const double s_ = 0.5 * (_dy * (vxg - va) + _dx * (vby - vyl));
                            const double ee = 0.25 * (ex + e + ex + e);
                            const double _Gdt = VISC ? 0.0 : 1.0 / (0.25 * (gx + g + gx + g) * dt);
                            const double to_ = VISC ? 0.0 : LDB(f.toxy, oxy + rxy);
                            const double dtr = DTR(ee, _Gdt);
                            const double t0 = LDB(f.txy, oxy + rxy);
                            STN<(NT & 1) != 0>(a.o.txy, oxy + rxy, t0 + INC(t0, to_, ee, s_, _Gdt, dtr));
                        }
                        {   // τyz (i, ny, k)
                            const double vzg = (k == 0 && bc.nsK0) ? 0.0 : (bc.fsBk ? c_p : (bc.nsBk ? -c_p : NBV(nbBk, Vz, ovz - svz + rvz)));
                            const double s_ = 0.5 * (_dz * (vby - vby_p) + _dy * (vzg - c_p));
                            const double ee = 0.25 * (e_p + e_p + e + e);
                            const double _Gdt = VISC ? 0.0 : 1.0 / (0.25 * (g_p + g_p + g + g) * dt);
                            const double to_ = VISC ? 0.0 : LDB(f.toyz, oyz - syz + ryz);
                            const double dtr = DTR(ee, _Gdt);
                            STN<(NT & 1) != 0>(a.o.tyz, oyz - syz + ryz, r10k + INC(r10k, to_, ee, s_, _Gdt, dtr));
                        }
                    }
                    if (xl && yl) {   // τxy (nx, ny, k): both velocities on ghost lines of their own boundary values
                        const double vxg = bc.nsR ? 0.0 : (bc.fsBk ? vax : (bc.nsBk ? -vax : NBV(nbBk, Vx, ovx + rvx)));
                        const double vyg = bc.nsBk ? 0.0 : (bc.fsR ? vby : (bc.nsR ? -vby : NBV(nbR, Vy, ovy + 8u)));
                        const double s_ = 0.5 * (_dy * (vxg - vax) + _dx * (vyg - vby));
                        const double ee = 0.25 * (e + e + e + e);
                        const double _Gdt = VISC ? 0.0 : 1.0 / (0.25 * (g + g + g + g) * dt);
                        const double to_ = VISC ? 0.0 : LDB(f.toxy, oxy + 8u + rxy);
                        const double dtr = DTR(ee, _Gdt);
                        const double t0 = LDB(f.txy, oxy + 8u + rxy);
                        STN<(NT & 1) != 0>(a.o.txy, oxy + 8u + rxy, t0 + INC(t0, to_, ee, s_, _Gdt, dtr));
                    }
                }
                vax_p = vax; vby_p = vby;
            }
            if (LOWREG) { exe_p = ex + e; eye_p = ey + e; gxg_p = gx + g; gyg_p = gy + g; }
            else {
                a_p = va; b_p = vb; c_p = vc; cx_p = vcx; cy_p = vcy;
                e_p = e; ex_p = ex; ey_p = ey; g_p = g; gx_p = gx; gy_p = gy;
            }
        }
        oc += sc; oxy += sxy; oxz += sxz; oyz += syz; ovx += svx; ovy += svy; ovz += svz;
    }
    if (HIF && avalid && kend == nz) {
        // ---- node plane k = nz above the last cell plane: the carried registers hold plane nz of the new V (a_p = Vx[i, j+1, nz], b_p = Vy[i+1, j, nz], c_p = Vz[i+1, j+1, nz] --
        // the boundary plane --, cx_p / cy_p its left / front neighbours, vax_p / vby_p the boundary column / row), e_p, ex_p, ey_p plane nz - 1 of η; the byte offsets have moved
        // on to plane nz + 1 of V and of τxz / τyz (so `- sxz` is plane nz), s10 / r10 / s01p / r01p hold the old stresses of plane nz
        const bool xl = i == nx - 1, yl = j == ny - 1;
        {   // τxz (i, j, nz): Vx[i, j+1, nz+1] is the ghost plane
            const double vxg = (i == 0 && bc.nsL) ? 0.0 : (bc.fsK1 ? a_p : (bc.nsK1 ? -a_p : NBV(nbK1, Vx, ovx - 8u)));
            const double s_ = 0.5 * (_dz * (vxg - a_p) + _dx * (c_p - cx_p));
            const double ee = 0.25 * (ex_p + e_p + ex_p + e_p);
            const double _Gdt = VISC ? 0.0 : 1.0 / (0.25 * (gx_p + g_p + gx_p + g_p) * dt);
            const double to_ = VISC ? 0.0 : LDB(f.toxz, oxz - sxz);
            const double dtr = DTR(ee, _Gdt);
            STN<(NT & 1) != 0>(a.o.txz, oxz - sxz, s01p + INC(s01p, to_, ee, s_, _Gdt, dtr));
        }
        {   // τyz (i, j, nz)
            const double vyg = (j == 0 && bc.nsF) ? 0.0 : (bc.fsK1 ? b_p : (bc.nsK1 ? -b_p : NBV(nbK1, Vy, ovy - rvy)));
            const double s_ = 0.5 * (_dz * (vyg - b_p) + _dy * (c_p - cy_p));
            const double ee = 0.25 * (ey_p + e_p + ey_p + e_p);
            const double _Gdt = VISC ? 0.0 : 1.0 / (0.25 * (gy_p + g_p + gy_p + g_p) * dt);
            const double to_ = VISC ? 0.0 : LDB(f.toyz, oyz - syz);
            const double dtr = DTR(ee, _Gdt);
            STN<(NT & 1) != 0>(a.o.tyz, oyz - syz, r01p + INC(r01p, to_, ee, s_, _Gdt, dtr));
        }
        if (xl) {   // τxz (nx, j, nz)
            const double vxg = bc.nsR ? 0.0 : (bc.fsK1 ? vax_p : (bc.nsK1 ? -vax_p : NBV(nbK1, Vx, ovx)));
            const double vzg = bc.nsK1 ? 0.0 : (bc.fsR ? c_p : (bc.nsR ? -c_p : NBV(nbR, Vz, ovz - svz + 8u)));
            const double s_ = 0.5 * (_dz * (vxg - vax_p) + _dx * (vzg - c_p));
            const double ee = 0.25 * (e_p + e_p + e_p + e_p);
            const double _Gdt = VISC ? 0.0 : 1.0 / (0.25 * (g_p + g_p + g_p + g_p) * dt);
            const double to_ = VISC ? 0.0 : LDB(f.toxz, oxz - sxz + 8u);
            const double dtr = DTR(ee, _Gdt);
            STN<(NT & 1) != 0>(a.o.txz, oxz - sxz + 8u, s10 + INC(s10, to_, ee, s_, _Gdt, dtr));
        }
        if (yl) {   // τyz (i, ny, nz)
            const double vyg = bc.nsBk ? 0.0 : (bc.fsK1 ? vby_p : (bc.nsK1 ? -vby_p : NBV(nbK1, Vy, ovy)));
            const double vzg = bc.nsK1 ? 0.0 : (bc.fsBk ? c_p : (bc.nsBk ? -c_p : NBV(nbBk, Vz, ovz - svz + rvz)));
            const double s_ = 0.5 * (_dz * (vyg - vby_p) + _dy * (vzg - c_p));
            const double ee = 0.25 * (e_p + e_p + e_p + e_p);
            const double _Gdt = VISC ? 0.0 : 1.0 / (0.25 * (g_p + g_p + g_p + g_p) * dt);
            const double to_ = VISC ? 0.0 : LDB(f.toyz, oyz - syz + ryz);
            const double dtr = DTR(ee, _Gdt);
            STN<(NT & 1) != 0>(a.o.tyz, oyz - syz + ryz, r10 + INC(r10, to_, ee, s_, _Gdt, dtr));
        }
    }
    }   // the march over tiy
#undef NBV
#undef NBVX
#undef NBVY
#undef NBVZ
#undef LFX
#undef LFY
#undef LFZ
}

}   // namespace
