// stokes2d.hip -- 2D visco-elastic pseudo-transient Stokes path for gfx950.
//
// Reference being replaced: src/stokes/Stokes2D.jl:181-325 and its kernels
// (VelocityKernels.jl:3-6,10-44,108-131,246-269; PressureKernels.jl:10-15,186-195;
// StressKernels.jl:63-91; MiniKernels.jl:76-80; boundaryconditions/*.jl).  Same two-sweep fusion as
// the 3D path; at the reference's 2D sizes (<= 1024^2) the working set is Infinity-Cache resident,
// so these kernels are launch/latency bound rather than HBM bound.
#include "jrx_internal.hpp"
#include "jrx_kernels.hpp"
#include "jrx_material.hpp"

namespace {

// inverse spacings of a non-uniform Geometry (src/grid/Cartesian.jl:77-100): device arrays, all NULL on a uniform grid (then the scalars _dx, _dy apply).
// vx, vy = _di.vertex (nx | ny entries: cell sizes), cx, cy = _di.center (nx-1 | ny-1: distances of the cell centres), vxy = _di.velocity[1][2] (y spacing of the
// Vx grid with its ghost rows, ny+1), vyx = _di.velocity[2][1] (x spacing of the Vy grid, nx+1).  Which one a stencil takes is the reference's choice, kernel by
// kernel (VelocityKernels.jl:3-44,108-180,246-307, stress_rotation_particles.jl:17-29).
struct Sp2 { const double *vx, *vy, *cx, *cy, *vxy, *vyx; };
__device__ __forceinline__ double spc(const double *a, const int i, const double u) { return a ? a[i] : u; }

struct Args2 {
    jrx_stokes2d_fields f;
    const double *etatau;
    Sp2 sp;
    double _dx, _dy, dt, r, theta_dtau, eta_dtau;
    int nx, ny;
    unsigned fs, ns;      // free_slip / no_slip face masks for the velocity kernel's fused ghost update (BCF)
    double fs_dt = 0.0;   // dt * free_surface of the free-surface forms of compute_V! / compute_Res! (VelocityKernels.jl:134-180,271-307)
};

#define VX(i_, j_) Vx[(i_) + (i64)(nx + 1) * (j_)]
#define VY(i_, j_) Vy[(i_) + (i64)(nx + 2) * (j_)]
#define CC(i_, j_) ((i_) + (i64)nx * (j_))
// Blocks are dealt round-robin to the 8 XCDs, each with its own L2: block L of a 1D launch takes position (L % 8) * (T / 8) + L / 8 of the flattened
// (x fastest) node sequence, so that every XCD works on a contiguous band of rows and finds the rows j +- 1 of its stencils in its own L2
// (shear band 1024^2: k_vep_stress2d fetched 37.8 array passes from HBM for 19 needed, profiles/r02_bench2d_xcd_slabs.txt)
__device__ __forceinline__ unsigned xcd_slab_block()
{
    const unsigned L = blockIdx.x, per = gridDim.x / 8u;
    return L < per * 8u ? (L & 7u) * per + (L >> 3) : L;
}

template <bool DIAG>
__global__ __launch_bounds__(256) void k_stress2d(const Args2 a)
{
    const int nx = a.nx, ny = a.ny;
    const int t = xcd_slab_block() * blockDim.x + threadIdx.x;
    const int j = t / (nx + 1), i = t - j * (nx + 1);
    if (j > ny) return;
    const double *__restrict__ Vx = a.f.Vx, *__restrict__ Vy = a.f.Vy, *__restrict__ eta = a.f.eta, *__restrict__ G = a.f.G;
    const double dt = a.dt, th = a.theta_dtau;
    if (i < nx && j < ny) {
        const i64 c = CC(i, j);
        const double dxi = (-VX(i, j + 1) + VX(i + 1, j + 1)) * spc(a.sp.vx, i, a._dx);
        const double dyi = (-VY(i + 1, j) + VY(i + 1, j + 1)) * spc(a.sp.vy, j, a._dy);
        const double divV = dxi + dyi;
        const double _Gdt = 1.0 / (G[c] * dt);
        {   // compute_P! with ητ (Stokes2D.jl:231-233)
            const double _Kdt = 1.0 / (a.f.K[c] * dt);
            const double _dt = 1.0 / dt;
            const double P = a.f.P[c], P0 = a.f.P0[c];
            const double rhs = -divV + (a.f.Q[c] * _dt);
            const double psi = 1.0 / (1.0 / a.etatau[c] + _Gdt) * a.r / th;
            a.f.P[c] = (fma(P0, _Kdt, rhs) * psi + P) / (1.0 + _Kdt * psi);
            if (DIAG) { a.f.RP[c] = fma(-(P - P0), _Kdt, rhs); a.f.divV[c] = divV; }
        }
        const double d3 = divV * (1.0 / 3.0);
        const double exx = dxi - d3, eyy = dyi - d3;
        if (DIAG) { a.f.exx[c] = exx; a.f.eyy[c] = eyy; }
        const double e = eta[c];
        const double dtr = dev_dtau_r(th, e, _Gdt);
        double tv;
        tv = a.f.txx[c]; a.f.txx[c] = tv + dev_stress_inc(tv, a.f.toxx[c], e, exx, _Gdt, dtr);
        tv = a.f.tyy[c]; a.f.tyy[c] = tv + dev_stress_inc(tv, a.f.toyy[c], e, eyy, _Gdt, dtr);
    }
    {   // vertex (i,j) of (nx+1, ny+1)
        const int im = max(i - 1, 0), ip = min(i, nx - 1), jm = max(j - 1, 0), jp = min(j, ny - 1);
        const double exy = 0.5 * (spc(a.sp.vxy, j, a._dy) * (VX(i, j + 1) - VX(i, j)) + spc(a.sp.vyx, i, a._dx) * (VY(i + 1, j) - VY(i, j)));
        const double e = 0.25 * (eta[CC(im, jm)] + eta[CC(ip, jm)] + eta[CC(im, jp)] + eta[CC(ip, jp)]);
        const double g = 0.25 * (G[CC(im, jm)] + G[CC(ip, jm)] + G[CC(im, jp)] + G[CC(ip, jp)]);
        const double _Gdt = 1.0 / (g * dt);
        const double dtr = dev_dtau_r(th, e, _Gdt);
        const i64 v = i + (i64)(nx + 1) * j;
        const double tv = a.f.txy[v];
        a.f.txy[v] = tv + dev_stress_inc(tv, a.f.toxy[v], e, exy, _Gdt, dtr);
        if (DIAG) a.f.exy[v] = exy;
    }
}

// compute_V! (VelocityKernels.jl:108-131); RES_ONLY stores compute_Res! (:246-269) values instead.
// BCF: the thread that updates a velocity node next to a free-slip / no-slip face also refreshes that node's ghost copy
// (free_slip.jl:1-13, no_slip.jl:1-18), which is all flow_bcs! changes once it has been applied in full one time: the other ghost
// and boundary values it writes are copies of nodes compute_V! never updates.  Saves the flow_bcs! launches of the launch-bound loop.
template <bool RES_ONLY, bool BCF>
__device__ __forceinline__ void velocity2d_cell(const Args2 &a, const int i, const int j)
{
    const int nx = a.nx, ny = a.ny;
    const double edt = a.eta_dtau;
    const double *__restrict__ P = a.f.P, *__restrict__ txy = a.f.txy, *__restrict__ et = a.etatau;
#define TXY(i_, j_) txy[(i_) + (i64)(nx + 1) * (j_)]
    const i64 c = CC(i, j);
    if (i < nx - 1) {
        const double _dx = spc(a.sp.cx, i, a._dx), _dy = spc(a.sp.vy, j, a._dy);        // _dx_c, _dy_v
        const double dP = (-P[c] + P[c + 1]) * _dx, dT = (-a.f.txx[c] + a.f.txx[c + 1]) * _dx;
        const double dS = (-TXY(i + 1, j) + TXY(i + 1, j + 1)) * _dy, av = (a.f.fx[c] + a.f.fx[c + 1]) * 0.5;
        if (RES_ONLY) a.f.Rx[i + (i64)(nx - 1) * j] = dT + dS - dP - av;
        else {
            const i64 q = (i + 1) + (i64)(nx + 1) * (j + 1);
            const double v = a.f.Vx[q] + (-dP + dT + dS - av) * edt / ((et[c] + et[c + 1]) * 0.5);
            a.f.Vx[q] = v;
            if (BCF) {      // Vx ghost rows j = 0 (bot) and j = ny+1 (top)
                if (j == 0) { if (a.fs & JRX_FACE_BOT) a.f.Vx[q - (nx + 1)] = v; else if (a.ns & JRX_FACE_BOT) a.f.Vx[q - (nx + 1)] = -v; }
                if (j == ny - 1) { if (a.fs & JRX_FACE_TOP) a.f.Vx[q + (nx + 1)] = v; else if (a.ns & JRX_FACE_TOP) a.f.Vx[q + (nx + 1)] = -v; }
            }
        }
    }
    if (j < ny - 1) {
        const double _dx = spc(a.sp.vx, i, a._dx), _dy = spc(a.sp.cy, j, a._dy);        // _dx_v, _dy_c
        const double dP = (-P[c] + P[c + nx]) * _dy, dT = (-a.f.tyy[c] + a.f.tyy[c + nx]) * _dy;
        const double dS = (-TXY(i, j + 1) + TXY(i + 1, j + 1)) * _dx, av = (a.f.fy[c] + a.f.fy[c + nx]) * 0.5;
        double corr = 0.0;
        const bool fsurf = a.fs_dt != 0.0;
        if (fsurf) {      // ρg_correction = Vy ∂(ρg_y)/∂y θ dt with θ = 1, j_N = min(j + 1, ny)
            const int jN = min(j + 1, ny - 1);
            const double drg = (a.f.fy[i + (i64)nx * jN] - a.f.fy[c]) * _dy;
            const double vy0 = a.f.Vy[(i + 1) + (i64)(nx + 2) * (j + 1)];
            corr = RES_ONLY ? (vy0 * drg) * 1.0 * a.fs_dt : vy0 * drg * 1.0 * a.fs_dt;
        }
        if (RES_ONLY) a.f.Ry[c] = fsurf ? dT + dS - dP - av + corr : dT + dS - dP - av;
        else {
            const i64 q = (i + 1) + (i64)(nx + 2) * (j + 1);
            const double rhs = fsurf ? -dP + dT + dS - av + corr : -dP + dT + dS - av;
            const double v = a.f.Vy[q] + rhs * edt / ((et[c] + et[c + nx]) * 0.5);
            a.f.Vy[q] = v;
            if (BCF) {      // Vy ghost columns i = 0 (left) and i = nx+1 (right)
                if (i == 0) { if (a.fs & JRX_FACE_LEFT) a.f.Vy[q - 1] = v; else if (a.ns & JRX_FACE_LEFT) a.f.Vy[q - 1] = -v; }
                if (i == nx - 1) { if (a.fs & JRX_FACE_RIGHT) a.f.Vy[q + 1] = v; else if (a.ns & JRX_FACE_RIGHT) a.f.Vy[q + 1] = -v; }
            }
        }
    }
#undef TXY
}

template <bool RES_ONLY, bool BCF = false>
__global__ __launch_bounds__(256) void k_velocity2d(const Args2 a)
{
    const int t = xcd_slab_block() * blockDim.x + threadIdx.x;
    const int j = t / a.nx, i = t - j * a.nx;
    if (j >= a.ny) return;
    velocity2d_cell<RES_ONLY, BCF>(a, i, j);
}

// ------------------------------------------------------------------------------------------------
// One PT iteration in one launch for iterations nobody observes (the 2D loop is launch-bound): compute_V! of iteration m, flow_bcs! by
// rule, and compute_∇V! / compute_P! / compute_strain_rate! / compute_τ! of iteration m+1, reading the state (P, τ, V) from one set and
// writing it to the other.  A thread owns node (i, j) of the (nx+1) x (ny+1) node grid: it computes the new velocities of its cell and
// of the cell below (the row below recomputes them for itself), takes those of the column to its left from the neighbouring lane
// (waves overlap by one lane: lane 0 only feeds lane 1), derives boundary and ghost entries from the flow_bcs! rules, and then does
// the stress update of its cell centre and its vertex exactly as k_stress2d does.  Same arithmetic, in the same order.
// ------------------------------------------------------------------------------------------------
struct Out6_2d { double *P, *txx, *tyy, *txy, *Vx, *Vy; };
struct BC2 { int tL, tR, tB, tT; };      // 0 none (memory holds the prescribed value), 1 free slip, 2 no slip; B: j = 1, T: j = end
__global__ __launch_bounds__(256) void k_fused2d(const Args2 a, const Out6_2d o, const BC2 bc, const int nwx)
{
    const int nx = a.nx, ny = a.ny;
    const int wg = (int)xcd_slab_block() * 4 + (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const int j = wg / nwx, i = (wg - j * nwx) * 63 + lane - 1;
    if (j > ny) return;
    const double _dx = a._dx, _dy = a._dy, edt = a.eta_dtau, dt = a.dt, th = a.theta_dtau;
    const double *__restrict__ Vx = a.f.Vx, *__restrict__ Vy = a.f.Vy, *__restrict__ P = a.f.P, *__restrict__ et = a.etatau;
    const double *__restrict__ txy = a.f.txy, *__restrict__ eta = a.f.eta, *__restrict__ G = a.f.G;
#define TXY(i_, j_) txy[(i_) + (i64)(nx + 1) * (j_)]
    // compute_V! of cell (ci, cj): the new Vx[ci+1, cj+1] and Vy[ci+1, cj+1]; nodes on the right / top boundary planes keep their value
    auto Bcell = [&](const int ci, const int cj, double &vx_, double &vy_) {
        const i64 c = CC(ci, cj);
        const double Pc = P[c], ec = et[c];
        if (ci < nx - 1) {
            const double dP = (-Pc + P[c + 1]) * _dx, dT = (-a.f.txx[c] + a.f.txx[c + 1]) * _dx;
            const double dS = (-TXY(ci + 1, cj) + TXY(ci + 1, cj + 1)) * _dy, av = (a.f.fx[c] + a.f.fx[c + 1]) * 0.5;
            vx_ = VX(ci + 1, cj + 1) + (-dP + dT + dS - av) * edt / ((ec + et[c + 1]) * 0.5);
        } else vx_ = bc.tR == 2 ? 0.0 : VX(nx, cj + 1);
        if (cj < ny - 1) {
            const double dP = (-Pc + P[c + nx]) * _dy, dT = (-a.f.tyy[c] + a.f.tyy[c + nx]) * _dy;
            const double dS = (-TXY(ci, cj + 1) + TXY(ci + 1, cj + 1)) * _dx, av = (a.f.fy[c] + a.f.fy[c + nx]) * 0.5;
            vy_ = VY(ci + 1, cj + 1) + (-dP + dT + dS - av) * edt / ((ec + et[c + nx]) * 0.5);
        } else vy_ = bc.tT == 2 ? 0.0 : VY(ci + 1, ny);
    };
    auto rule = [](const int t, const double v, const double mem) { return t == 1 ? v : (t == 2 ? -v : mem); };
    const bool col = i >= 0 && i < nx;                 // this lane has a cell column
    double vxn = 0.0, vyn = 0.0, vxb = 0.0, vyb = 0.0;   // new Vx[i+1, j+1], Vy[i+1, j+1] (own row) and Vx[i+1, j], Vy[i+1, j] (row below)
    if (col) {
        if (j < ny) Bcell(i, j, vxn, vyn);
        if (j >= 1) Bcell(i, j - 1, vxb, vyb);
        else {
            vxb = rule(bc.tB, vxn, VX(i + 1, 0));                 // ghost row below the bottom face
            vyb = bc.tB == 2 ? 0.0 : VY(i + 1, 0);                // Vy on the bottom face
        }
        if (j == ny) vxn = rule(bc.tT, vxb, VX(i + 1, ny + 1));   // ghost row above the top face
    }
    const double Lvxn = __shfl_up(vxn, 1, 64), Lvxb = __shfl_up(vxb, 1, 64), Lvyb = __shfl_up(vyb, 1, 64);
    if (lane == 0 || i < 0 || i > nx) return;          // feeder lane / beyond the row
    // new velocities on the left of the node: Vx[i, j+1], Vx[i, j], Vy[i, j]
    double X1, X0, Y0;
    if (i >= 1) { X1 = Lvxn; X0 = Lvxb; Y0 = Lvyb; }
    else {
        // left boundary plane of Vx (ghost rows by the bottom / top rule) and ghost column of Vy
        auto VxL = [&](const int jr) -> double {
            const double lo = bc.tL == 2 ? 0.0 : VX(0, 1), hi = bc.tL == 2 ? 0.0 : VX(0, ny);
            if (jr == 0) return rule(bc.tB, lo, VX(0, 0));
            if (jr == ny + 1) return rule(bc.tT, hi, VX(0, ny + 1));
            return bc.tL == 2 ? 0.0 : VX(0, jr);
        };
        X1 = VxL(j + 1); X0 = VxL(j);
        Y0 = rule(bc.tL, vyb, VY(0, j));
    }
    const double Yr = i < nx ? vyb : rule(bc.tR, Lvyb, VY(nx + 1, j));        // Vy[i+1, j]; beyond the right face: ghost column
    if (i < nx && j < ny) {
        const i64 c = CC(i, j);
        const double dxi = (-X1 + vxn) * _dx;
        const double dyi = (-vyb + vyn) * _dy;
        const double divV = dxi + dyi;
        const double _Gdt = 1.0 / (G[c] * dt);
        {   // compute_P! with ητ (Stokes2D.jl:231-233)
            const double _Kdt = 1.0 / (a.f.K[c] * dt);
            const double _dt = 1.0 / dt;
            const double Pc = P[c], P0 = a.f.P0[c];
            const double rhs = -divV + (a.f.Q[c] * _dt);
            const double psi = 1.0 / (1.0 / et[c] + _Gdt) * a.r / th;
            o.P[c] = (fma(P0, _Kdt, rhs) * psi + Pc) / (1.0 + _Kdt * psi);
        }
        const double d3 = divV * (1.0 / 3.0);
        const double exx = dxi - d3, eyy = dyi - d3;
        const double e = eta[c];
        const double dtr = dev_dtau_r(th, e, _Gdt);
        double tv;
        tv = a.f.txx[c]; o.txx[c] = tv + dev_stress_inc(tv, a.f.toxx[c], e, exx, _Gdt, dtr);
        tv = a.f.tyy[c]; o.tyy[c] = tv + dev_stress_inc(tv, a.f.toyy[c], e, eyy, _Gdt, dtr);
        if (i < nx - 1) o.Vx[(i + 1) + (i64)(nx + 1) * (j + 1)] = vxn;
        if (j < ny - 1) o.Vy[(i + 1) + (i64)(nx + 2) * (j + 1)] = vyn;
    }
    {   // vertex (i, j)
        const int im = max(i - 1, 0), ip = min(i, nx - 1), jm = max(j - 1, 0), jp = min(j, ny - 1);
        const double exy = 0.5 * (_dy * (X1 - X0) + _dx * (Yr - Y0));
        const double e = 0.25 * (eta[CC(im, jm)] + eta[CC(ip, jm)] + eta[CC(im, jp)] + eta[CC(ip, jp)]);
        const double g = 0.25 * (G[CC(im, jm)] + G[CC(ip, jm)] + G[CC(im, jp)] + G[CC(ip, jp)]);
        const double _Gdt = 1.0 / (g * dt);
        const double dtr = dev_dtau_r(th, e, _Gdt);
        const i64 v = i + (i64)(nx + 1) * j;
        const double tv = txy[v];
        o.txy[v] = tv + dev_stress_inc(tv, a.f.toxy[v], e, exy, _Gdt, dtr);
    }
#undef TXY
}

// ------------------------------------------------------------------------------------------------
// k_fused2d with every operand requested up front (option "fused2d_batch", default).  The kernel above follows the reference's control flow -- compute_V! of the own cell,
// of the cell below, the boundary rules, compute_P!, compute_τ! of the centre, of the vertex -- and every `if` on the way ends a basic block, so its ~50 loads reach the
// memory system as a chain of 12-14 dependent groups: on the grids this kernel runs on (everything sits in L2 / Infinity Cache) an iteration IS that chain of round trips.
// Here the loads are unconditional (clamped indices; what a boundary lane does not use it does not use), issued as one batch and pinned ahead of the arithmetic; the
// arithmetic is the kernel above, expression for expression, with the boundary cases as selects.  VISC (dt = Inf, SolCx and every purely viscous 2D run): the seven arrays
// that only ever meet 1/(G dt), 1/(K dt), 1/dt -- τ_o (3), P0, K, G, Q -- are not loaded (finite operands: the driver checks them once per solve), as in 3D.
// ------------------------------------------------------------------------------------------------
template <bool VISC>
__global__ __launch_bounds__(256) void k_fused2d_b(const Args2 a, const Out6_2d o, const BC2 bc, const int nwx)
{
    const int nx = a.nx, ny = a.ny;
    const int wg = (int)xcd_slab_block() * 4 + (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const int j = wg / nwx, i = (wg - j * nwx) * 63 + lane - 1;
    if (j > ny) return;
    const double _dx = a._dx, _dy = a._dy, edt = a.eta_dtau, dt = a.dt, th = a.theta_dtau;
    const double *__restrict__ Vx = a.f.Vx, *__restrict__ Vy = a.f.Vy, *__restrict__ P = a.f.P, *__restrict__ et = a.etatau;
    const double *__restrict__ txy = a.f.txy, *__restrict__ eta = a.f.eta, *__restrict__ G = a.f.G;
    typedef unsigned int u32;
#define LB(p_, off_) (*(const double *)((const char *)(p_) + (off_)))
    // clamped cell column / rows: r0 = j - 1, r1 = j, r2 = j + 1 (cells), ip = ic + 1 (cells)
    const int ic = min(max(i, 0), nx - 1), ip = min(ic + 1, nx - 1);
    const int r0 = min(max(j - 1, 0), ny - 1), r1 = min(j, ny - 1), r2 = min(j + 1, ny - 1);
    const u32 c0 = 8u * (u32)(ic + nx * r0), c1 = 8u * (u32)(ic + nx * r1), c2 = 8u * (u32)(ic + nx * r2), dxp = 8u * (u32)(ip - ic);
    // node rows of τxy (nx+1 columns, rows 0..ny): j - 1, j, j + 1 clamped; node column ic + 1 and ic
    const int n0 = max(j - 1, 0), n2 = min(j + 1, ny);
    const u32 t0 = 8u * (u32)(ic + (nx + 1) * n0), t1 = 8u * (u32)(ic + (nx + 1) * j), t2 = 8u * (u32)(ic + (nx + 1) * n2);
    // Vx (nx+1 columns, rows 0..ny+1): (i+1, j), (i+1, j+1);  Vy (nx+2 columns, rows 0..ny): (i+1, j), (i+1, min(j+1, ny)).  The feeder lane of the first segment (i = -1) thereby
    // loads the boundary column Vx[0, ·], Vy[0, ·] for the lane on the left face, and the lane on the right face (i = nx, no cell of its own) the ghost column Vy[nx+1, j]
    const int cvx = min(max(i + 1, 0), nx), cvy = min(max(i + 1, 0), nx + 1);
    const u32 vx0 = 8u * (u32)(cvx + (nx + 1) * j), vx1 = vx0 + 8u * (u32)(nx + 1);
    const u32 vy0 = 8u * (u32)(cvy + (nx + 2) * j), vy1 = 8u * (u32)(cvy + (nx + 2) * n2);
    // vertex (i, j): the four cells around it, clamped
    const int iq = min(max(i, 0), nx - 1), jm = max(j - 1, 0), jq = min(j, ny - 1);
    const u32 v10 = 8u * (u32)(iq + nx * jm), v11 = 8u * (u32)(iq + nx * jq);
    const u32 vv = 8u * (u32)(min(max(i, 0), nx) + (nx + 1) * j);
    // ---- every operand
    const double P00 = LB(P, c0), P10 = LB(P, c0 + dxp), P01 = LB(P, c1), P11 = LB(P, c1 + dxp), P02 = LB(P, c2);
    const double E00 = LB(et, c0), E10 = LB(et, c0 + dxp), E01 = LB(et, c1), E11 = LB(et, c1 + dxp), E02 = LB(et, c2);
    const double X00 = LB(a.f.txx, c0), X10 = LB(a.f.txx, c0 + dxp), X01 = LB(a.f.txx, c1), X11 = LB(a.f.txx, c1 + dxp);
    const double Y00 = LB(a.f.tyy, c0), Y01 = LB(a.f.tyy, c1), Y02 = LB(a.f.tyy, c2);
    const double Fx00 = LB(a.f.fx, c0), Fx10 = LB(a.f.fx, c0 + dxp), Fx01 = LB(a.f.fx, c1), Fx11 = LB(a.f.fx, c1 + dxp);
    const double Fy00 = LB(a.f.fy, c0), Fy01 = LB(a.f.fy, c1), Fy02 = LB(a.f.fy, c2);
    const double S10 = LB(txy, t0 + 8u), S01 = LB(txy, t1), S11 = LB(txy, t1 + 8u), S02 = LB(txy, t2), S12 = LB(txy, t2 + 8u);     // τxy[ic+1, j-1], [ic, j], [ic+1, j], [ic, j+1], [ic+1, j+1]
    const double Ux0 = LB(Vx, vx0), Ux1 = LB(Vx, vx1), Uy0 = LB(Vy, vy0), Uy1 = LB(Vy, vy1);
    const double e10 = LB(eta, v10), e11 = LB(eta, v11);
    double g10 = 0, g11 = 0, Gc = 0, Kc = 0, P0c = 0, Qc = 0, toxx = 0, toyy = 0, toxy = 0;
    if (!VISC) {
        g10 = LB(G, v10); g11 = LB(G, v11);
        Gc = g11;          // (where the centre is updated, i < nx and j < ny, the vertex's cell (iq, jq) is the own cell)
    }
    const double ec1 = e11, tvv = i < nx ? S01 : S11;       // η of the own cell = the vertex's (iq, jq); τxy[i, j]: the own node column, or the last one (ic + 1 = nx)
    __builtin_amdgcn_sched_barrier(0);
    // ---- compute_V! of cell (ci, cj) from its operands: Pc, P[c+1], P[c+nx], ... (k_fused2d::Bcell)
    auto Bcell = [&](const int ci, const int cj, const double Pc, const double Px, const double Py, const double ec, const double ex, const double ey, const double txc,
                     const double txr, const double tyc, const double tyu, const double s_r0, const double s_r1, const double s_l1, const double fxc, const double fxr,
                     const double fyc, const double fyu, const double vxo, const double vyo, double &vx_, double &vy_) {
        if (ci < nx - 1) {
            const double dP = (-Pc + Px) * _dx, dT = (-txc + txr) * _dx;
            const double dS = (-s_r0 + s_r1) * _dy, av = (fxc + fxr) * 0.5;
            vx_ = vxo + (-dP + dT + dS - av) * edt / ((ec + ex) * 0.5);
        } else vx_ = bc.tR == 2 ? 0.0 : vxo;
        if (cj < ny - 1) {
            const double dP = (-Pc + Py) * _dy, dT = (-tyc + tyu) * _dy;
            const double dS = (-s_l1 + s_r1) * _dx, av = (fyc + fyu) * 0.5;
            vy_ = vyo + (-dP + dT + dS - av) * edt / ((ec + ey) * 0.5);
        } else vy_ = bc.tT == 2 ? 0.0 : vyo;
    };
    auto rule = [](const int t, const double v, const double mem) { return t == 1 ? v : (t == 2 ? -v : mem); };
    const bool col = i >= 0 && i < nx;
    double vxn = 0.0, vyn = 0.0, vxb = 0.0, vyb = 0.0;
    if (col) {
        //                         Pc   P[c+1] P[c+nx] ec  e[c+1] e[c+nx] txx  txx+1 tyy  tyy+nx τxy[ci+1,cj] [ci+1,cj+1] [ci,cj+1]
        if (j < ny) Bcell(i, j, P01, P11, P02, E01, E11, E02, X01, X11, Y01, Y02, S11, S12, S02, Fx01, Fx11, Fy01, Fy02, Ux1, Uy1, vxn, vyn);
    }
    if (!VISC) {
        // the six operands only the stress update reads, requested once the first compute_V! has consumed its share of the batch (the register peak that decides between four
        // and five waves per SIMD -- 512^2 is 4,617 waves for 4,096 or 5,120 slots) and pinned here: in flight under the second compute_V! and the lane exchanges
        __builtin_amdgcn_sched_barrier(0);
        Kc = LB(a.f.K, c1); P0c = LB(a.f.P0, c1); Qc = LB(a.f.Q, c1); toxx = LB(a.f.toxx, c1); toyy = LB(a.f.toyy, c1); toxy = LB(a.f.toxy, vv);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (col) {
        if (j >= 1) Bcell(i, j - 1, P00, P10, P01, E00, E10, E01, X00, X10, Y00, Y01, S10, S11, S01, Fx00, Fx10, Fy00, Fy01, Ux0, Uy0, vxb, vyb);
        else {
            vxb = rule(bc.tB, vxn, Ux0);                 // ghost row below the bottom face: Vx[i+1, 0]
            vyb = bc.tB == 2 ? 0.0 : Uy0;                // Vy on the bottom face: Vy[i+1, 0]
        }
        if (j == ny) vxn = rule(bc.tT, vxb, Ux1);        // ghost row above the top face: Vx[i+1, ny+1]
    }
    const double Lvxn = __shfl_up(vxn, 1, 64), Lvxb = __shfl_up(vxb, 1, 64), Lvyb = __shfl_up(vyb, 1, 64);
    // from the lane to the left: its velocity loads (for the lane on the left face: the boundary column) and the vertex's two left-hand cells (im, jm), (im, jq) = that lane's (iq, jm), (iq, jq)
    const double LUx0 = __shfl_up(Ux0, 1, 64), LUx1 = __shfl_up(Ux1, 1, 64), LUy0 = __shfl_up(Uy0, 1, 64);
    const double Le10 = __shfl_up(e10, 1, 64), Le11 = __shfl_up(e11, 1, 64);
    const double e00 = i > 0 ? Le10 : e10, e01 = i > 0 ? Le11 : e11;
    double g00 = 0, g01 = 0;
    if (!VISC) {
        const double Lg10 = __shfl_up(g10, 1, 64), Lg11 = __shfl_up(g11, 1, 64);
        g00 = i > 0 ? Lg10 : g10; g01 = i > 0 ? Lg11 : g11;
    }
    if (lane == 0 || i < 0 || i > nx) return;          // feeder lane / beyond the row
    const double B0 = LUx0, B1 = LUx1, BY = i == 0 ? LUy0 : Uy0;      // i = 0: Vx[0, j], Vx[0, j+1], Vy[0, j];  i = nx: Vy[nx+1, j]
    double X1, X0, Y0;
    if (i >= 1) { X1 = Lvxn; X0 = Lvxb; Y0 = Lvyb; }
    else {
        // left boundary plane of Vx (ghost rows by the bottom / top rule) and ghost column of Vy: B0 = Vx[0, j], B1 = Vx[0, j+1], BY = Vy[0, j]
        const double b0 = bc.tL == 2 ? 0.0 : B0, b1 = bc.tL == 2 ? 0.0 : B1;
        X1 = (j + 1 == ny + 1) ? rule(bc.tT, b0, B1) : b1;
        X0 = (j == 0) ? rule(bc.tB, b1, B0) : b0;
        Y0 = rule(bc.tL, vyb, BY);
    }
    const double Yr = i < nx ? vyb : rule(bc.tR, Lvyb, BY);        // Vy[i+1, j]; beyond the right face: ghost column Vy[nx+1, j]
    if (i < nx && j < ny) {
        const double dxi = (-X1 + vxn) * _dx;
        const double dyi = (-vyb + vyn) * _dy;
        const double divV = dxi + dyi;
        const double _Gdt = VISC ? 0.0 : 1.0 / (Gc * dt);
        {   // compute_P! with ητ (Stokes2D.jl:231-233)
            const double _Kdt = VISC ? 0.0 : 1.0 / (Kc * dt);
            const double _dt = 1.0 / dt;
            const double rhs = -divV + (Qc * _dt);
            const double psi = 1.0 / (1.0 / E01 + _Gdt) * a.r / th;
            *(double *)((char *)o.P + c1) = (fma(P0c, _Kdt, rhs) * psi + P01) / (1.0 + _Kdt * psi);
        }
        const double d3 = divV * (1.0 / 3.0);
        const double exx = dxi - d3, eyy = dyi - d3;
        const double dtr = dev_dtau_r(th, ec1, _Gdt);
        *(double *)((char *)o.txx + c1) = X01 + dev_stress_inc(X01, toxx, ec1, exx, _Gdt, dtr);
        *(double *)((char *)o.tyy + c1) = Y01 + dev_stress_inc(Y01, toyy, ec1, eyy, _Gdt, dtr);
        if (i < nx - 1) o.Vx[(i + 1) + (i64)(nx + 1) * (j + 1)] = vxn;
        if (j < ny - 1) o.Vy[(i + 1) + (i64)(nx + 2) * (j + 1)] = vyn;
    }
    {   // vertex (i, j)
        const double exy = 0.5 * (_dy * (X1 - X0) + _dx * (Yr - Y0));
        const double e = 0.25 * (e00 + e10 + e01 + e11);
        const double g = 0.25 * (g00 + g10 + g01 + g11);
        const double _Gdt = VISC ? 0.0 : 1.0 / (g * dt);
        const double dtr = dev_dtau_r(th, e, _Gdt);
        *(double *)((char *)o.txy + vv) = tvv + dev_stress_inc(tvv, toxy, e, exy, _Gdt, dtr);
    }
#undef LB
}

// dt = Inf (see k_fused2d_b<VISC>): one streaming pass checks that every entry of the arrays the viscous-limit form does not load is harmless -- τ_o, P0, Q finite, K and G neither
// NaN nor 0 (0 * Inf) -- as visc_operands_check does in 3D; if not, the general form runs and produces the reference's NaNs
__global__ __launch_bounds__(256) void k_visc_operands_ok2d(const double *__restrict__ P0, const double *__restrict__ Q, const double *__restrict__ toxx, const double *__restrict__ toyy,
                                                            const double *__restrict__ K, const double *__restrict__ G, i64 nc, const double *__restrict__ toxy, i64 nv, int *bad)
{
    const i64 stride = (i64)gridDim.x * blockDim.x;
    bool b = false;
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < nv; t += stride) {
        if (t < nc) {
            b |= !(isfinite(P0[t]) && isfinite(Q[t]) && isfinite(toxx[t]) && isfinite(toyy[t]));
            const double k = K[t], g = G[t];
            b |= (k != k) || (g != g) || k == 0.0 || g == 0.0;
        }
        b |= !isfinite(toxy[t]);
    }
    if (__any(b) && (threadIdx.x & 63) == 0) atomicOr(bad, 1);
}
#undef VX
#undef VY
#undef CC

Args2 make_args2(const jrx_stokes2d_fields *f, const double *etatau, const jrx_stokes2d_params *p)
{
    Args2 a;
    a.f = *f; a.etatau = etatau;
    a._dx = p->_dx; a._dy = p->_dy; a.dt = p->dt; a.r = p->r; a.theta_dtau = p->theta_dtau; a.eta_dtau = p->eta_dtau;
    a.nx = (int)p->nx; a.ny = (int)p->ny;
    a.fs = p->free_slip; a.ns = p->no_slip;
    a.sp = Sp2{p->inv_spacing[0], p->inv_spacing[1], p->inv_spacing[2], p->inv_spacing[3], p->inv_spacing[4], p->inv_spacing[5]};
    return a;
}

// the six inverse-spacing arrays of a non-uniform grid come together or not at all
bool spacing_ok(const double *const sp[6])
{
    int n = 0;
    for (int q = 0; q < 6; q++) n += sp[q] != nullptr;
    return n == 0 || n == 6;
}

jrx_status check2(jrx_handle *h, const jrx_stokes2d_fields *f, const jrx_stokes2d_params *p)
{
    if (!h) return JRX_ERR_ARG;
    if (!f || !p) return jrx_fail(h, JRX_ERR_ARG, "null fields/params");
    JRX_TRY(jrx_check_device(h));
    if (p->nx < 3 || p->ny < 3) return jrx_fail(h, JRX_ERR_ARG, "2D Stokes needs at least 3 cells per dimension");
    if ((double)(p->nx + 2) * (double)(p->ny + 2) >= 2147483647.0) return jrx_fail(h, JRX_ERR_UNSUPPORTED, "grid too large");
    const void *req[] = {f->P, f->P0, f->divV, f->Q, f->Vx, f->Vy, f->Ux, f->Uy, f->txx, f->tyy, f->txy, f->toxx, f->toyy, f->toxy,
                         f->exx, f->eyy, f->exy, f->eta, f->K, f->G, f->fx, f->fy, f->RP, f->Rx, f->Ry};
    for (const void *q : req)
        if (!q) return jrx_fail(h, JRX_ERR_ARG, "a required 2D field pointer is NULL");
    if (!spacing_ok(p->inv_spacing)) return jrx_fail(h, JRX_ERR_ARG, "non-uniform grid: all six inverse-spacing arrays are required");
    return JRX_OK;
}

jrx_status launch_bcs2(jrx_handle *h, hipStream_t s, double *Vx, double *Vy, int nx, int ny, uint32_t fs, uint32_t ns, uint32_t pe)
{
    BcArr A[3] = {{Vx, {nx + 1, ny + 2, 1}}, {Vy, {nx + 2, ny + 1, 1}}, {nullptr, {0, 0, 0}}};
    auto run = [&](int type, int dim, bool lo, bool hi) -> jrx_status {
        if (!lo && !hi) return JRX_OK;
        const int d1 = dim == 0 ? 1 : 0;
        const int na = A[0].n[d1] > A[1].n[d1] ? A[0].n[d1] : A[1].n[d1];
        hipLaunchKernelGGL(k_bc3d, dim3((na + 255) / 256, 1), dim3(256), 0, s, A[0], A[1], A[2], type, dim, (int)lo, (int)hi);
        JRX_LAUNCH_CHECK(h);
        return JRX_OK;
    };
    // 2D naming: bot <-> j = 1, top <-> j = end for every condition (no_slip.jl:1-18, free_slip.jl:1-13, periodic.jl:15-36)
    if (ns) {
        JRX_TRY(run(1, 0, ns & JRX_FACE_LEFT, ns & JRX_FACE_RIGHT));
        JRX_TRY(run(1, 1, ns & JRX_FACE_BOT, ns & JRX_FACE_TOP));
    }
    if (fs) {
        JRX_TRY(run(0, 1, fs & JRX_FACE_BOT, fs & JRX_FACE_TOP));
        JRX_TRY(run(0, 0, fs & JRX_FACE_LEFT, fs & JRX_FACE_RIGHT));
    }
    if (pe) {
        JRX_TRY(run(2, 0, pe & JRX_FACE_LEFT, pe & JRX_FACE_RIGHT));
        JRX_TRY(run(2, 1, pe & JRX_FACE_BOT, pe & JRX_FACE_TOP));
    }
    return JRX_OK;
}

jrx_status launch_sumsq2(jrx_handle *h, hipStream_t s, const jrx_stokes2d_fields *f, const jrx_stokes2d_params *p)
{
    const int nx = (int)p->nx, ny = (int)p->ny;
    RedArr A0 = {f->Rx, {nx - 1, ny, 1}, 1}, A1 = {f->Ry, {nx, ny - 1, 1}, 1}, A2 = {nullptr, {0, 0, 0}, 0}, A3 = {f->RP, {nx, ny, 1}, 0};
    int nb = (int)(((i64)nx * ny + 2047) / 2048);
    nb = nb < 1 ? 1 : (nb > kMaxRedBlocks ? kMaxRedBlocks : nb);
    hipLaunchKernelGGL(k_sumsq_partial, dim3(nb), dim3(256), 0, s, A0, A1, A2, A3, h->d_partials);
    JRX_LAUNCH_CHECK(h);
    hipLaunchKernelGGL(k_sumsq_final, dim3(1), dim3(256), 0, s, h->d_partials, nb, h->d_sums);
    JRX_LAUNCH_CHECK(h);
    return JRX_OK;
}

// fuse_bc: flow_bcs! has already been applied in full once in this solve, nothing observes U in this iteration and no face is
// periodic, so the velocity kernel refreshes the ghosts itself
jrx_status enqueue_iteration2(jrx_handle *h, const jrx_stokes2d_fields *f, const double *etatau, const jrx_stokes2d_params *p, bool diag,
                              bool fuse_bc, bool skip_stress, bool *bcs_full_done);
jrx_status enqueue_iteration2(jrx_handle *h, const jrx_stokes2d_fields *f, const double *etatau, const jrx_stokes2d_params *p, bool diag,
                              bool fuse_bc = false)
{
    return enqueue_iteration2(h, f, etatau, p, diag, fuse_bc, false, nullptr);
}
// skip_stress: the stress sweep of this iteration has already been applied (by a fused launch); bcs_full_done (optional): set when
// flow_bcs! has been launched in full
jrx_status enqueue_iteration2(jrx_handle *h, const jrx_stokes2d_fields *f, const double *etatau, const jrx_stokes2d_params *p, bool diag,
                              bool fuse_bc, bool skip_stress, bool *bcs_full_done)
{
    const int nx = (int)p->nx, ny = (int)p->ny;
    Args2 a = make_args2(f, etatau, p);
    hipStream_t s = h->stream;
    const unsigned gA = (unsigned)(((i64)(nx + 1) * (ny + 1) + 255) / 256), gB = (unsigned)(((i64)nx * ny + 255) / 256);
    if (!skip_stress) {
        if (diag) hipLaunchKernelGGL(k_stress2d<true>, dim3(gA), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(k_stress2d<false>, dim3(gA), dim3(256), 0, s, a);
        JRX_LAUNCH_CHECK(h);
    }
    if (fuse_bc && !diag && p->periodic == 0 && !jrx_comm_active(h) && !p->displacement_bcs) {
        hipLaunchKernelGGL((k_velocity2d<false, true>), dim3(gB), dim3(256), 0, s, a);
        JRX_LAUNCH_CHECK(h);
        return JRX_OK;
    }
    hipLaunchKernelGGL(k_velocity2d<false>, dim3(gB), dim3(256), 0, s, a);
    JRX_LAUNCH_CHECK(h);
    if (diag) {
        hipLaunchKernelGGL(k_scale3, dim3(256), dim3(256), 0, s, f->Ux, f->Vx, (i64)(nx + 1) * (ny + 2), f->Uy, f->Vy, (i64)(nx + 2) * (ny + 1),
                           (double *)nullptr, (const double *)nullptr, (i64)0, p->dt);
        JRX_LAUNCH_CHECK(h);
    }
    if (p->displacement_bcs) {    // flow_bcs! on U = V dt (overwritten by the next iteration: only observable ones matter); V's ghosts stay as they are
        if (diag) JRX_TRY(launch_bcs2(h, s, f->Ux, f->Uy, nx, ny, p->free_slip, p->no_slip, p->periodic));
    } else {
        JRX_TRY(launch_bcs2(h, s, f->Vx, f->Vy, nx, ny, p->free_slip, p->no_slip, p->periodic));
        if (bcs_full_done) *bcs_full_done = true;
    }
    if (jrx_comm_active(h)) {
        double *arrs[2] = {f->Vx, f->Vy};
        const int64_t ext[2][3] = {{nx + 1, ny + 2, 1}, {nx + 2, ny + 1, 1}};
        const int64_t n[3] = {nx, ny, 1};
        JRX_TRY(jrx_halo_exchange(h, s, 2, arrs, ext, n));
    }
    return JRX_OK;
}

}   // namespace

extern "C" {

jrx_status jrx_stokes2d_sweep_stress(jrx_handle *h, const jrx_stokes2d_fields *f, const double *etatau, const jrx_stokes2d_params *p, int32_t flags)
{
    JRX_TRY(check2(h, f, p));
    if (!etatau) return jrx_fail(h, JRX_ERR_ARG, "etatau is NULL");
    Args2 a = make_args2(f, etatau, p);
    const unsigned gA = (unsigned)(((i64)(p->nx + 1) * (p->ny + 1) + 255) / 256);
    if (flags & JRX_OUT_DIAG) hipLaunchKernelGGL(k_stress2d<true>, dim3(gA), dim3(256), 0, h->stream, a);
    else hipLaunchKernelGGL(k_stress2d<false>, dim3(gA), dim3(256), 0, h->stream, a);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_stokes2d_sweep_velocity(jrx_handle *h, const jrx_stokes2d_fields *f, const double *etatau, const jrx_stokes2d_params *p, int32_t flags)
{
    JRX_TRY(check2(h, f, p));
    if (!etatau) return jrx_fail(h, JRX_ERR_ARG, "etatau is NULL");
    Args2 a = make_args2(f, etatau, p);
    const unsigned gB = (unsigned)(((i64)p->nx * p->ny + 255) / 256);
    hipLaunchKernelGGL(k_velocity2d<false>, dim3(gB), dim3(256), 0, h->stream, a);
    JRX_LAUNCH_CHECK(h);
    if (flags & JRX_OUT_DIAG) {
        hipLaunchKernelGGL(k_scale3, dim3(256), dim3(256), 0, h->stream, f->Ux, f->Vx, (i64)(p->nx + 1) * (p->ny + 2), f->Uy, f->Vy,
                           (i64)(p->nx + 2) * (p->ny + 1), (double *)nullptr, (const double *)nullptr, (i64)0, p->dt);
        JRX_LAUNCH_CHECK(h);
    }
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_stokes2d_compute_res(jrx_handle *h, const jrx_stokes2d_fields *f, const jrx_stokes2d_params *p)
{
    JRX_TRY(check2(h, f, p));
    Args2 a = make_args2(f, nullptr, p);
    const unsigned gB = (unsigned)(((i64)p->nx * p->ny + 255) / 256);
    hipLaunchKernelGGL(k_velocity2d<true>, dim3(gB), dim3(256), 0, h->stream, a);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_flow_bcs2d(jrx_handle *h, double *Vx, double *Vy, int64_t nx, int64_t ny, uint32_t free_slip, uint32_t no_slip, uint32_t periodic)
{
    if (!h) return JRX_ERR_ARG;
    if (!Vx || !Vy) return jrx_fail(h, JRX_ERR_ARG, "null velocity pointer");
    JRX_TRY(launch_bcs2(h, h->stream, Vx, Vy, (int)nx, (int)ny, free_slip, no_slip, periodic));
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_stokes2d_residual_sumsq(jrx_handle *h, const jrx_stokes2d_fields *f, const jrx_stokes2d_params *p, double out[3])
{
    JRX_TRY(check2(h, f, p));
    JRX_TRY(launch_sumsq2(h, h->stream, f, p));
    JRX_HIP(h, hipMemcpyAsync(h->h_sums, h->d_sums, 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    out[0] = h->h_sums[0]; out[1] = h->h_sums[1]; out[2] = h->h_sums[3];
    return JRX_OK;
}

jrx_status jrx_stokes2d_solve(jrx_handle *h, const jrx_stokes2d_fields *f, const jrx_stokes2d_params *p, jrx_solve_result *res)
{
    JRX_TRY(check2(h, f, p));
    if (!res) return jrx_fail(h, JRX_ERR_ARG, "null result");
    if (p->nout < 1) return jrx_fail(h, JRX_ERR_ARG, "nout must be >= 1");
    const int nx = (int)p->nx, ny = (int)p->ny;
    const size_t n = (size_t)nx * ny;
    hipStream_t s = h->stream;
    JRX_TRY(jrx_ensure_etatau(h, n));
    // compute_maxloc!(ητ, η; window=(1,1)); update_halo!(ητ)   (Stokes2D.jl:206-210)
    hipLaunchKernelGGL(k_maxloc, dim3((unsigned)((n + 255) / 256), 1), dim3(256), 0, s, h->etatau, f->eta, nx, ny, 1);
    JRX_LAUNCH_CHECK(h);
    if (jrx_comm_active(h)) {
        double *arrs[1] = {h->etatau};
        const int64_t ext[1][3] = {{nx, ny, 1}};
        const int64_t nn[3] = {nx, ny, 1};
        JRX_TRY(jrx_halo_exchange(h, s, 1, arrs, ext, nn));
    }
    if (p->displacement_bcs) {    // displacement2velocity!(stokes, dt, flow_bcs) (Stokes2D.jl:223)
        hipLaunchKernelGGL(k_scale3, dim3(256), dim3(256), 0, s, f->Vx, (const double *)f->Ux, (i64)(nx + 1) * (ny + 2), f->Vy,
                           (const double *)f->Uy, (i64)(nx + 2) * (ny + 1), (double *)nullptr, (const double *)nullptr, (i64)0, 1.0 / p->dt);
        JRX_LAUNCH_CHECK(h);
    }
    double err_it1 = 1.0, err = 1.0;
    int64_t iter = 0, cont = 0;
    const int rank = jrx_comm_rank(h);
    JRX_HIP(h, hipEventRecord(h->ev[6], s));
    auto keep_going = [&](int64_t it) { return it < 2 || (((err / err_it1) > p->eps_rel && err > p->eps_abs) && it <= p->iterMax); };
    auto is_check = [&](int64_t i1) { return (i1 % p->nout == 0) && i1 > 1; };
    // Fused pipeline (as in 3D): when nothing observes iteration it1 and iteration it1+1 certainly runs unobserved, compute_V! of it1,
    // flow_bcs! (by rule) and the stress sweep of it1+1 run as one launch that ping-pongs (P, τ, V) between the caller's arrays and a
    // library-owned set; flow_bcs! itself is applied lazily before anything reads the boundary entries of V from memory.
    // measured with the XCD slab block order (SolCx, profiles/r02_bench2d_xcd_slabs.txt; it/s two kernels vs fused): 128^2 175.9 k / 176.2 k, 256^2 135.2 k /
    // 144.2 k, 384^2 101.5 k / 106.7 k, 512^2 79.1 k / 73.6 k, 768^2 40.1 k / 38.8 k, 1024^2 equal -- fused up to 200,000 nodes (~ 440^2)
    const bool fusable = !p->displacement_bcs && h->fused2d && h->scratch_sets && (h->kernel_variant == 3 || (h->kernel_variant == 0 && (i64)(nx + 1) * (ny + 1) <= (i64)(h->fused2d_batch ? h->fused2d_max_nodes : 200000))) &&
                         !jrx_comm_active(h) && p->periodic == 0 && nx >= 2 && ny >= 2 && !p->inv_spacing[0];
    const size_t nvx = (size_t)(nx + 1) * (ny + 2), nvy = (size_t)(nx + 2) * (ny + 1), nvt = (size_t)(nx + 1) * (ny + 1);
    Out6_2d setU = {f->P, f->txx, f->tyy, f->txy, f->Vx, f->Vy}, setS = setU;
    if (fusable) {
        if (!(h->scratch2d[0] && h->scratch2d_dims[0] == nx && h->scratch2d_dims[1] == ny)) {
            for (int q = 0; q < 6; q++) { if (h->scratch2d[q]) JRX_HIP(h, hipFree(h->scratch2d[q])); h->scratch2d[q] = nullptr; }
            h->scratch2d_dims[0] = h->scratch2d_dims[1] = 0;
            const size_t sz[6] = {n, n, n, nvt, nvx, nvy};
            for (int q = 0; q < 6; q++) JRX_HIP(h, hipMalloc(&h->scratch2d[q], sz[q] * sizeof(double)));
            h->scratch2d_dims[0] = nx; h->scratch2d_dims[1] = ny;
        }
        double **S = h->scratch2d;
        setS = Out6_2d{S[0], S[1], S[2], S[3], S[4], S[5]};
        // boundary and ghost entries of V that no fused launch writes
        JRX_HIP(h, hipMemcpyAsync(setS.Vx, f->Vx, nvx * sizeof(double), hipMemcpyDeviceToDevice, s));
        JRX_HIP(h, hipMemcpyAsync(setS.Vy, f->Vy, nvy * sizeof(double), hipMemcpyDeviceToDevice, s));
    }
    BC2 bc2;
    {
        auto ty = [&](uint32_t bit) { return (p->no_slip & bit) ? 2 : ((p->free_slip & bit) ? 1 : 0); };
        bc2.tL = ty(JRX_FACE_LEFT); bc2.tR = ty(JRX_FACE_RIGHT); bc2.tB = ty(JRX_FACE_BOT); bc2.tT = ty(JRX_FACE_TOP);
    }
    jrx_stokes2d_fields cur = *f;
    bool cur_is_user = true, stress_done = false, ghosts_stale = false;
    bool bcs_full_done[2] = {false, false};      // flow_bcs! launched in full on the V of the caller's set / the second set
    const int nwx = (nx + 1 + 62) / 63;
    // the one-launch iteration: "fused2d_batch" (default) = the form with every operand requested up front; its viscous-limit instantiation when dt = Inf and the operand check passed
    bool visc2 = false;
    if (fusable && h->fused2d_batch && h->viscous_limit && p->dt == INFINITY) {
        int *d_bad = reinterpret_cast<int *>(h->d_sums + 6), *h_bad = reinterpret_cast<int *>(h->h_sums + 6);
        JRX_HIP(h, hipMemsetAsync(d_bad, 0, sizeof(double), s));
        hipLaunchKernelGGL(k_visc_operands_ok2d, dim3(1024), dim3(256), 0, s, (const double *)f->P0, (const double *)f->Q, (const double *)f->toxx, (const double *)f->toyy,
                           (const double *)f->K, (const double *)f->G, (i64)n, (const double *)f->toxy, (i64)nvt, d_bad);
        JRX_LAUNCH_CHECK(h);
        JRX_HIP(h, hipMemcpyAsync(h_bad, d_bad, sizeof(double), hipMemcpyDeviceToHost, s));
        JRX_HIP(h, hipStreamSynchronize(s));
        visc2 = (*h_bad == 0);
        h->stat_visc_checks++;
        if (!visc2) h->stat_visc_fallbacks++;
    }
    const bool batch2 = h->fused2d_batch && (double)(nx + 2) * (double)(ny + 2) < 536870912.0;      // (32-bit byte offsets)
    auto launch_fused2d = [&](hipStream_t st, const Args2 &aa, const Out6_2d &dst) {
        const dim3 g((unsigned)((nwx * (ny + 1) + 3) / 4));
        if (batch2 && visc2) hipLaunchKernelGGL(k_fused2d_b<true>, g, dim3(256), 0, st, aa, dst, bc2, nwx);
        else if (batch2) hipLaunchKernelGGL(k_fused2d_b<false>, g, dim3(256), 0, st, aa, dst, bc2, nwx);
        else hipLaunchKernelGGL(k_fused2d, g, dim3(256), 0, st, aa, dst, bc2, nwx);
    };
    Args2 a = make_args2(&cur, h->etatau, p);
    // Runs of unobserved iterations in the steady state of the loop replay as a captured graph of GIT iterations (the gap between dependent launches is shorter
    // inside a graph: scripts/graph_probe.hip, 4.6 vs 5.7 - 6.1 us per pair of short kernels): GIT x k_fused2d (an even count, so that the ping-pong sets end where
    // they started; one graph per parity) on the grids that run the one-launch iteration (SolCx 128^2 173 k -> 184 k it/s, 256^2 143 k -> 150 k).  The two-kernel form
    // of the larger grids gains nothing from it (512^2: 77.9 k plain, 76.7 k replayed) and keeps plain launches.  Option "loop_graphs" = 0: plain launches everywhere.
    constexpr int GIT = 32;
    GraphExecs gexec;        // released on every exit path
    bool graphs = h->loop_graphs && fusable;
    auto capture = [&](hipGraphExec_t *out, auto &&body) -> bool {
        hipGraph_t g = nullptr;
        bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) { body(); ok = hipStreamEndCapture(s, &g) == hipSuccess && g != nullptr; }
        if (ok) ok = hipGraphInstantiate(out, g, nullptr, nullptr, 0) == hipSuccess;
        if (g) (void)hipGraphDestroy(g);
        if (!ok) { (void)hipGetLastError(); *out = nullptr; }
        return ok;
    };
    while (keep_going(iter)) {
        if (graphs && iter >= 2) {
            // observed iterations: the multiples of nout and iteration iterMax + 1 (Stokes2D.jl:265,312); between them err does not change, so keep_going holds
            int64_t nxt = ((iter / p->nout) + 1) * p->nout;
            if (nxt > p->iterMax + 1) nxt = p->iterMax + 1;
            int64_t run = nxt - 1 - iter;                     // unobserved iterations from here
            if (fusable) {
                int64_t frun = run - 1;                       // the last unobserved iteration before an observed one is not fused with it
                if (stress_done && frun >= GIT) {
                    const int par = cur_is_user ? 0 : 1;
                    if (!gexec[par]) {
                        const bool ok = capture(&gexec[par], [&]() {
                            jrx_stokes2d_fields c = cur;
                            bool cu = cur_is_user;
                            for (int q = 0; q < GIT; q++) {
                                const Args2 aa = make_args2(&c, h->etatau, p);
                                const Out6_2d dst = cu ? setS : setU;
                                launch_fused2d(s, aa, dst);
                                c.P = dst.P; c.txx = dst.txx; c.tyy = dst.tyy; c.txy = dst.txy; c.Vx = dst.Vx; c.Vy = dst.Vy;
                                cu = !cu;
                            }
                        });
                        if (!ok) graphs = false;
                    }
                    if (gexec[par]) {
                        while (frun >= GIT) {
                            JRX_HIP(h, hipGraphLaunch(gexec[par], s));
                            iter += GIT; frun -= GIT;
                            h->stat_fused2d += GIT;
                        }
                        continue;
                    }
                }
            }
        }
        const int64_t it1 = iter + 1;
        const bool check = is_check(it1);
        const bool diag = check || !keep_going(it1);
        const bool fuse_next = fusable && !diag && keep_going(it1) && !(is_check(it1 + 1) || !keep_going(it1 + 1));
        a = make_args2(&cur, h->etatau, p);
        if (fuse_next) {
            if (!stress_done) {
                hipLaunchKernelGGL(k_stress2d<false>, dim3((unsigned)((nvt + 255) / 256)), dim3(256), 0, s, a);
                JRX_LAUNCH_CHECK(h);
            }
            const Out6_2d dst = cur_is_user ? setS : setU;
            launch_fused2d(s, a, dst);
            h->stat_fused2d++;
            JRX_LAUNCH_CHECK(h);
            cur.P = dst.P; cur.txx = dst.txx; cur.tyy = dst.tyy; cur.txy = dst.txy; cur.Vx = dst.Vx; cur.Vy = dst.Vy;
            cur_is_user = !cur_is_user;
            stress_done = true; ghosts_stale = true;
        } else {
            if (ghosts_stale && diag) {
                // U = V dt copies the boundary entries of V as flow_bcs! of the previous iteration left them: apply the pending flow_bcs! now
                JRX_TRY(launch_bcs2(h, s, cur.Vx, cur.Vy, nx, ny, p->free_slip, p->no_slip, p->periodic));
                bcs_full_done[cur_is_user ? 0 : 1] = true;
            }
            ghosts_stale = false;
            bool &full = bcs_full_done[cur_is_user ? 0 : 1];
            JRX_TRY(enqueue_iteration2(h, &cur, h->etatau, p, diag, full && iter >= 1, stress_done, &full));
            stress_done = false;
        }
        iter = it1;
        if (check) {
            a = make_args2(&cur, h->etatau, p);
            hipLaunchKernelGGL(k_velocity2d<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);   // compute_Res! (Stokes2D.jl:274-276)
            JRX_LAUNCH_CHECK(h);
            JRX_TRY(launch_sumsq2(h, s, f, p));
            JRX_HIP(h, hipMemcpyAsync(h->h_sums, h->d_sums, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
            JRX_HIP(h, hipStreamSynchronize(s));
            double ss[3] = {h->h_sums[0], h->h_sums[1], h->h_sums[3]};
            JRX_TRY(jrx_allreduce_sum_host(h, ss, 3));
            const double nRx = sqrt(ss[0]) / sqrt((double)((p->nxg - 2) * (p->nyg - 1)));
            const double nRy = sqrt(ss[1]) / sqrt((double)((p->nxg - 1) * (p->nyg - 2)));
            const double nDV = sqrt(ss[2]) / sqrt((double)(p->nxg * p->nyg));
            err = fmax(nRx, fmax(nRy, nDV));
            if (std::isnan(nRx) || std::isnan(nRy) || std::isnan(nDV)) err = NAN;
            if (cont < res->cap) {
                if (res->norm_Rx) res->norm_Rx[cont] = nRx;
                if (res->norm_Ry) res->norm_Ry[cont] = nRy;
                if (res->norm_divV) res->norm_divV[cont] = nDV;
                if (res->err_evo1) res->err_evo1[cont] = err;
                if (res->err_evo2) res->err_evo2[cont] = iter;
            }
            if (cont == 0) err_it1 = err;
            cont++;
            if (rank == 0 && ((p->verbose && (err / err_it1) > p->eps_rel && err > p->eps_abs) || iter == p->iterMax))
                printf("Total steps = %lld, abs_err = %1.3e , rel_err = %1.3e [norm_Rx=%1.3e, norm_Ry=%1.3e, norm_∇V=%1.3e] \n",
                       (long long)iter, err, err / err_it1, nRx, nRy, nDV);
        }
    }
    gexec.reset();
    JRX_HIP(h, hipEventRecord(h->ev[7], s));
    if (!cur_is_user) {       // leave the state in the caller's arrays
        JRX_HIP(h, hipMemcpyAsync(setU.P, setS.P, n * sizeof(double), hipMemcpyDeviceToDevice, s));
        JRX_HIP(h, hipMemcpyAsync(setU.txx, setS.txx, n * sizeof(double), hipMemcpyDeviceToDevice, s));
        JRX_HIP(h, hipMemcpyAsync(setU.tyy, setS.tyy, n * sizeof(double), hipMemcpyDeviceToDevice, s));
        JRX_HIP(h, hipMemcpyAsync(setU.txy, setS.txy, nvt * sizeof(double), hipMemcpyDeviceToDevice, s));
        JRX_HIP(h, hipMemcpyAsync(setU.Vx, setS.Vx, nvx * sizeof(double), hipMemcpyDeviceToDevice, s));
        JRX_HIP(h, hipMemcpyAsync(setU.Vy, setS.Vy, nvy * sizeof(double), hipMemcpyDeviceToDevice, s));
    }
    // multi_copy! (Stokes2D.jl:308-309)
    hipLaunchKernelGGL(k_copy6, dim3(256), dim3(256), 0, s, f->toxx, f->txx, (i64)n, f->toyy, f->tyy, (i64)n, f->toxy, f->txy, (i64)(nx + 1) * (ny + 1),
                       (f->txy_c && f->toxy_c) ? f->toxy_c : nullptr, (const double *)f->txy_c, (i64)n, (double *)nullptr, (const double *)nullptr,
                       (i64)0, (double *)nullptr, (const double *)nullptr, (i64)0);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(s));
    float ms = 0.f;
    JRX_HIP(h, hipEventElapsedTime(&ms, h->ev[6], h->ev[7]));
    res->iter = iter;
    res->nchecks = cont < res->cap ? cont : res->cap;
    res->time_s = ms * 1e-3;
    res->av_time_s = iter > 1 ? res->time_s / (double)(iter - 1) : res->time_s;
    return JRX_OK;
}

}   // extern "C"


// ================================================================================================
// 2D multiphase visco-elasto-plastic driver (config 5: shear band) -- src/stokes/Stokes2D.jl:577-866
// ================================================================================================
namespace {

struct VepArgs {
    jrx_vep2d_fields f;
    jrx_rheology rh;
    const double *theta, *etatau, *Kc, *Gc;
    const double *eta_lin_c, *eta_lin_v;                // linear laws: phase-averaged η at centres / vertices, computed once per solve (nullptr: from the ratios per call)
    double *lam, *lamv;
    double *txx_out = nullptr, *tyy_out = nullptr;      // where the centre half writes τxx, τyy (nullptr: in place)
    Sp2 sp;                                             // non-uniform grid: inverse spacing arrays (all NULL: _dx, _dy)
    double _dx, _dy, dt, r, theta_dtau, rel, nu, cut_lo, cut_hi;
    int nx, ny;
    bool soft;            // some phase has a softening law (EII_pl is then read by the yield function)
    bool si;              // strain_increment variant
    bool tg;              // args.T is the ghosted thermal.T (nx+2, ny+2): densities read it at the cell's own [i, j], unshifted (BuoyancyForces.jl:52)
    bool vfields;         // some phase's creep law reads T, P or the invariant (visc_kind != 0)
    bool vtau;            // the viscosity is taken from the stress (update_viscosity_τII!, the in-loop form) rather than from the strain rate (compute_viscosity!)
    bool obs = true;      // the outputs nothing inside the PT loop reads -- ∇V, RP, ε_pl (3), ε_vol_pl, τII, η_vep -- are stored; the solve loop clears it on iterations whose
                          // results cannot be observed (not a norm check, not the last one): the next iteration overwrites them anyway
};

__device__ __forceinline__ double sinv2(double xx, double yy, double xy) { return sqrt(0.5 * (xx * xx + yy * yy) + xy * xy); }
// GeoParams second_invariant_staggered: the shear slot enters as the mean of the squared vertex values
// (pinned by the extrema of test/test_shearband2D.jl:198-199: mean-then-square misses them by 2.8e-3)
__device__ __forceinline__ double sinv_stag(double xx, double yy, double a, double b, double c, double d)
{
    return sqrt(0.5 * (xx * xx + yy * yy) + 0.25 * (a * a + b * b + c * c + d * d));
}
__device__ __forceinline__ double ratio_avg(const double *val, const double *r, int n)
{   // fn_ratio, src/phases/phases.jl:6-15
    double x = 0.0;
    for (int q = 0; q < n; q++) x += (r[q] == 0.0) ? 0.0 : val[q] * r[q];
    return x;
}
// NP > 0: the number of phases as a compile-time constant (the phase loops unroll and the caller hands the ratios in registers, loaded in one batch), else rh.nphase
template <int NP = 0>
__device__ __forceinline__ void plastic_params(const jrx_rheology &rh, const double *r, bool &is_pl, double &eta_reg)
{   // plastic_params_phase, rheology/StressUpdate.jl:152-176
    is_pl = false; eta_reg = 0.0;
    const int np = NP > 0 ? NP : rh.nphase;
#pragma unroll
    for (int q = 0; q < np; q++)
        if (rh.is_pl[q]) { is_pl = true; eta_reg += rh.eta_vp[q] * r[q]; }
}
// SOFT: some phase has a softening law (compiled out otherwise)
template <bool SOFT, int NP = 0>
__device__ __forceinline__ double yield_F(const jrx_rheology &rh, const double *r, double P, double tII, double EII)
{   // compute_yieldfunction_phase, StressUpdate.jl:399-410 ; DP: F = τII - cosϕ(EII) C(EII) - sinϕ(EII) P (softening at the EII keyword)
    double F = 0.0;
    const int np = NP > 0 ? NP : rh.nphase;
#pragma unroll
    for (int q = 0; q < np; q++) {
        if (r[q] == 0.0) continue;
        double Fq = tII;
        if (rh.is_pl[q]) {
            if (SOFT) {
                double sp, cp;
                mat_friction(rh, q, EII, sp, cp);
                Fq = tII - cp * mat_cohesion(rh, q, EII) - sp * P;
            } else Fq = tII - rh.cosphi[q] * rh.C[q] - rh.sinphi[q] * P;
        }
        F += r[q] * Fq;
    }
    return F;
}
template <int NP = 0>
__device__ __forceinline__ void plastic_grad(const jrx_rheology &rh, const double *r, const double t[3], double dQdt[3], double &dQdP, double &dFdP)
{   // compute_plastic_gradients_phase, StressUpdate.jl:476-495 ; ∂Q/∂τ = τ/(2 τII), ∂Q/∂P = -sinψ, ∂F/∂P = -sinϕ
    dQdt[0] = dQdt[1] = dQdt[2] = 0.0; dQdP = 0.0; dFdP = 0.0;
    const double tII = sinv2(t[0], t[1], t[2]);
    const int np = NP > 0 ? NP : rh.nphase;
    // (the quotients do not depend on the phase: one division per component, not one per component and phase -- the same bits)
    bool any_pl = false;
#pragma unroll
    for (int q = 0; q < np; q++) any_pl |= rh.is_pl[q] != 0;
    double g0 = 0.0, g1 = 0.0, g2 = 0.0;
    if (any_pl) { g0 = 0.5 * t[0] / tII; g1 = 0.5 * t[1] / tII; g2 = 0.5 * (t[2] / tII); }
#pragma unroll
    for (int q = 0; q < np; q++) {
        if (r[q] == 0.0 || !rh.is_pl[q]) continue;
        dQdt[0] = fma(r[q], g0, dQdt[0]); dQdt[1] = fma(r[q], g1, dQdt[1]); dQdt[2] = fma(r[q], g2, dQdt[2]);
        dQdP = fma(r[q], -rh.sinpsi[q], dQdP);
        dFdP = fma(r[q], -rh.sinphi[q], dFdP);
    }
}

#define C2(A, i_, j_) (A)[(i_) + (i64)nx * (j_)]
#define V2(A, i_, j_) (A)[(i_) + (i64)(nx + 1) * (j_)]

// compute_∇V! + compute_P! (phase form: K, G phase-averaged once per solve; writes θ) + compute_strain_rate!
// ML: compute_maxloc!(ητ, η) of the own cell first (clamped 3 x 3 window, same comparison order as k_maxloc) and store it: saves the
// separate launch of the launch-bound 2D loop
// RHO: update_ρg! of the own cell (args.T, args.P = stokes.P; phase-ratio density times the scalar gravity, into the last component of ρg)
template <bool ML, bool RHO = false>
__global__ __launch_bounds__(256) void k_vep_pre(const VepArgs a, double *__restrict__ theta)
{
    const int nx = a.nx, ny = a.ny;
    const int t = xcd_slab_block() * blockDim.x + threadIdx.x;
    const int j = t / (nx + 1), i = t - j * (nx + 1);
    if (j > ny) return;
    const double *__restrict__ Vx = a.f.Vx, *__restrict__ Vy = a.f.Vy;
#define VX(i_, j_) Vx[(i_) + (i64)(nx + 1) * (j_)]
#define VY(i_, j_) Vy[(i_) + (i64)(nx + 2) * (j_)]
    if (i < nx && j < ny) {
        const i64 c = i + (i64)nx * j;
        const double dxi = (-VX(i, j + 1) + VX(i + 1, j + 1)) * spc(a.sp.vx, i, a._dx);
        const double dyi = (-VY(i + 1, j) + VY(i + 1, j + 1)) * spc(a.sp.vy, j, a._dy);
        const double divV = dxi + dyi;
        if (a.obs) a.f.divV[c] = divV;
        const double _Kdt = 1.0 / (a.Kc[c] * a.dt), _Gdt = 1.0 / (a.Gc[c] * a.dt), _dt = 1.0 / a.dt;
        const double P = theta[c], P0 = a.f.P0[c];
        const double rhs = -divV + (a.f.Q[c] * _dt);
        if (a.obs) a.f.RP[c] = fma(-(P - P0), _Kdt, rhs);
        double et;
        if (ML) {
            et = -INFINITY;
            for (int jj = j - 1; jj <= j + 1; jj++) {
                const int jc = clampi(jj, 0, ny - 1);
                for (int ii = i - 1; ii <= i + 1; ii++) {
                    const double v = a.f.eta[clampi(ii, 0, nx - 1) + (i64)nx * jc];
                    if (v > et) et = v;
                }
            }
            const_cast<double *>(a.etatau)[c] = et;
        } else et = a.etatau[c];
        const double psi = 1.0 / (1.0 / et + _Gdt) * a.r / a.theta_dtau;
        theta[c] = (fma(P0, _Kdt, rhs) * psi + P) / (1.0 + _Kdt * psi);
        const double d3 = divV * (1.0 / 3.0);
        a.f.exx[c] = dxi - d3;
        a.f.eyy[c] = dyi - d3;
        if (RHO) a.f.fy[c] = mat_density_ratio(a.rh, a.f.phase_c + (i64)a.rh.nphase * c, !a.f.T ? 0.0 : (a.tg ? a.f.T[i + (i64)(nx + 2) * j] : a.f.T[c]), a.f.P[c]) * a.rh.gravity;
    }
    a.f.exy[i + (i64)(nx + 1) * j] = 0.5 * (spc(a.sp.vxy, j, a._dy) * (VX(i, j + 1) - VX(i, j)) + spc(a.sp.vyx, i, a._dx) * (VY(i + 1, j) - VY(i, j)));
#undef VX
#undef VY
}

// strain_increment variant (Stokes2D.jl:659-661, 680-692): ∇U and Δε from the displacements (compute_∇V!, compute_strain_rate! on U), then
// ε = Δε * _dt (compute_strain_rate_from_increment!, VelocityKernels.jl:46-57) -- overwrites the ε that k_vep_pre derived from V
__global__ __launch_bounds__(256) void k_vep_strain_inc(const VepArgs a)
{
    const int nx = a.nx, ny = a.ny;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = t / (nx + 1), i = t - j * (nx + 1);
    if (j > ny) return;
    const double *__restrict__ Ux = a.f.Ux, *__restrict__ Uy = a.f.Uy;
    const double _dt = 1.0 / a.dt;
#define UX(i_, j_) Ux[(i_) + (i64)(nx + 1) * (j_)]
#define UY(i_, j_) Uy[(i_) + (i64)(nx + 2) * (j_)]
    if (i < nx && j < ny) {
        const i64 c = i + (i64)nx * j;
        const double dxi = (-UX(i, j + 1) + UX(i + 1, j + 1)) * a._dx;
        const double dyi = (-UY(i + 1, j) + UY(i + 1, j + 1)) * a._dy;
        const double divU = dxi + dyi;
        a.f.divU[c] = divU;
        const double d3 = divU * (1.0 / 3.0);
        const double dexx = dxi - d3, deyy = dyi - d3;
        a.f.dexx[c] = dexx; a.f.deyy[c] = deyy;
        a.f.exx[c] = dexx * _dt; a.f.eyy[c] = deyy * _dt;
    }
    const double dexy = 0.5 * (a._dy * (UX(i, j + 1) - UX(i, j)) + a._dx * (UY(i + 1, j) - UY(i, j)));
    a.f.dexy[i + (i64)(nx + 1) * j] = dexy;
    a.f.exy[i + (i64)(nx + 1) * j] = dexy * _dt;
#undef UX
#undef UY
}
// compute_stress_increment(τ, τ_o, η, Δε, _G, dτ_r, dt) -- StressKernels.jl:18-21
// k_vep_pre<ML = true> for uniform grids and constant densities with every operand requested up front (option "fused2d_batch"): the control-flow form issues its 24 loads in five
// dependent groups (the 3 x 3 window of η compares as it loads, every `if (a.obs)` ends a basic block); same arithmetic, expression for expression.  OBS: a.obs as a constant.
template <bool OBS>
__global__ __launch_bounds__(256) void k_vep_pre_b(const VepArgs a, double *__restrict__ theta)
{
    const int nx = a.nx, ny = a.ny;
    const int t = xcd_slab_block() * blockDim.x + threadIdx.x;
    const int j = t / (nx + 1), i = t - j * (nx + 1);
    if (j > ny) return;
    const double *__restrict__ Vx = a.f.Vx, *__restrict__ Vy = a.f.Vy;
    const bool cell = i < nx && j < ny;
    const int ic = min(i, nx - 1), jc = min(j, ny - 1);
    const i64 c = ic + (i64)nx * jc;
    // velocities: Vx[i, j], Vx[i, j+1], Vx[i+1, j+1] (cells only), Vy[i, j], Vy[i+1, j], Vy[i+1, j+1] (cells only)
    const i64 qx = i + (i64)(nx + 1) * j, qy = i + (i64)(nx + 2) * j;
    const double x00 = Vx[qx], x01 = Vx[qx + (nx + 1)], x11 = Vx[qx + (nx + 1) + (i < nx ? 1 : 0)];
    const double y00 = Vy[qy], y10 = Vy[qy + 1], y11 = Vy[qy + 1 + (j < ny ? nx + 2 : 0)];
    const double Kc = a.Kc[c], Gc = a.Gc[c], P = theta[c], P0 = a.f.P0[c], Q = a.f.Q[c];
    double w[9];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const i64 r = (i64)nx * clampi(jc + q - 1, 0, ny - 1);
#pragma unroll
        for (int m = 0; m < 3; m++) w[3 * q + m] = a.f.eta[clampi(ic + m - 1, 0, nx - 1) + r];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (cell) {
        const double dxi = (-x01 + x11) * a._dx;
        const double dyi = (-y10 + y11) * a._dy;
        const double divV = dxi + dyi;
        if (OBS) a.f.divV[c] = divV;
        const double _Kdt = 1.0 / (Kc * a.dt), _Gdt = 1.0 / (Gc * a.dt), _dt = 1.0 / a.dt;
        const double rhs = -divV + (Q * _dt);
        if (OBS) a.f.RP[c] = fma(-(P - P0), _Kdt, rhs);
        double et = -INFINITY;
#pragma unroll
        for (int q = 0; q < 9; q++)
            if (w[q] > et) et = w[q];
        const_cast<double *>(a.etatau)[c] = et;
        const double psi = 1.0 / (1.0 / et + _Gdt) * a.r / a.theta_dtau;
        theta[c] = (fma(P0, _Kdt, rhs) * psi + P) / (1.0 + _Kdt * psi);
        const double d3 = divV * (1.0 / 3.0);
        a.f.exx[c] = dxi - d3;
        a.f.eyy[c] = dyi - d3;
    }
    a.f.exy[qx] = 0.5 * (a._dy * (x01 - x00) + a._dx * (y10 - y00));
}

__device__ __forceinline__ double dev_stress_inc_dt(double t, double to, double eta, double de, double _G, double dtr, double dt)
{
    return dtr * fma(2.0 * eta, de, fma(-(t - to) * eta, _G, -t * dt));
}

// update_stresses_center_vertex_ps! -- vertex half.  Runs before the centre half so that the vertex averages
// see the old centre stresses (the reference's single launch races on them).
// SI: strain_increment form (StressKernels.jl:1147-1302): Δε instead of ε, _G and dτ_r = inv(θ_dτ dt + η _G + dt), plastic terms times dt
template <bool SOFT, bool SI = false, int NP = 0>
__device__ __forceinline__ void vep_vertex_at(const VepArgs &a, const int i, const int j)
{
    const int nx = a.nx, ny = a.ny, np = NP > 0 ? NP : a.rh.nphase;
    const int i0 = clampi(i - 1, 0, nx - 1), ic = clampi(i, 0, nx - 1), j0 = clampi(j - 1, 0, ny - 1), jc = clampi(j, 0, ny - 1);
#define AVC(A) (0.25 * (C2(A, i0, j0) + C2(A, ic, jc) + C2(A, i0, jc) + C2(A, ic, j0)))
    const double Pv = AVC(a.theta), exxv = SI ? AVC(a.f.dexx) : AVC(a.f.exx), eyyv = SI ? AVC(a.f.deyy) : AVC(a.f.eyy), txxv = AVC(a.f.txx), tyyv = AVC(a.f.tyy);
    const double toxxv = AVC(a.f.toxx), toyyv = AVC(a.f.toyy);
    const double EIIv = SOFT ? AVC(a.f.EII_pl) : 0.0;      // EIIv_ij = av_clamped(EII, Ic...) (StressKernels.jl:1030); only softening laws read it
#undef AVC
    const i64 v = i + (i64)(nx + 1) * j;
    double rvv[NP > 0 ? NP : 1];
    if (NP > 0) {
#pragma unroll
        for (int q = 0; q < NP; q++) rvv[q] = a.f.phase_v[(i64)NP * v + q];
    }
    const double *rv = NP > 0 ? rvv : a.f.phase_v + (i64)np * v;
    bool is_pl; double eta_reg;
    plastic_params<NP>(a.rh, rv, is_pl, eta_reg);
    const double _Gdt = SI ? 1.0 / ratio_avg(a.rh.G, rv, np) : 1.0 / (ratio_avg(a.rh.G, rv, np) * a.dt);      // SI: _Gv
    const double Kv = ratio_avg(a.rh.Kb, rv, np);
    const double etav = 4.0 / (1.0 / C2(a.f.eta, i0, j0) + 1.0 / C2(a.f.eta, ic, jc) + 1.0 / C2(a.f.eta, i0, jc) + 1.0 / C2(a.f.eta, ic, j0));
    const double dtr = SI ? 1.0 / (a.theta_dtau * a.dt + etav * _Gdt + a.dt) : 1.0 / (a.theta_dtau + etav * _Gdt + 1.0);
    const double txy = a.f.txy[v];
    const double dxx = SI ? dev_stress_inc_dt(txxv, toxxv, etav, exxv, _Gdt, dtr, a.dt) : dev_stress_inc(txxv, toxxv, etav, exxv, _Gdt, dtr);
    const double dyy = SI ? dev_stress_inc_dt(tyyv, toyyv, etav, eyyv, _Gdt, dtr, a.dt) : dev_stress_inc(tyyv, toyyv, etav, eyyv, _Gdt, dtr);
    const double dxy = SI ? dev_stress_inc_dt(txy, a.f.toxy[v], etav, a.f.dexy[v], _Gdt, dtr, a.dt) : dev_stress_inc(txy, a.f.toxy[v], etav, a.f.exy[v], _Gdt, dtr);
    const double tt[3] = {txxv + dxx, tyyv + dyy, txy + dxy};
    const double tIIv = sinv2(dxx + txxv, dyy + tyyv, dxy + txy);
    double dQdt[3], dQdP, dFdP;
    plastic_grad<NP>(a.rh, rv, tt, dQdt, dQdP, dFdP);
    const double vol = isinf(Kv) ? 0.0 : Kv * a.dt * dFdP * dQdP;
    const double F = yield_F<SOFT, NP>(a.rh, rv, Pv, tIIv, EIIv);
    if (is_pl && tIIv != 0.0 && F > 0) {
        const double l = fma(1.0 - a.rel, a.lamv[v], a.rel * (fmax(F, 0.0) / (SI ? etav * dtr * a.dt + eta_reg + vol : etav * dtr + eta_reg + vol)));
        a.lamv[v] = l;
        const double epl = l * dQdt[2];
        a.f.txy[v] = txy + (SI ? fma(-2.0 * etav * a.dt * epl, dtr, dxy) : fma(-2.0 * etav * epl, dtr, dxy));
        if (a.obs) a.f.eplxy[v] = epl;
    } else {
        a.f.txy[v] = txy + dxy;
        if (a.obs) a.f.eplxy[v] = 0.0;
    }
}

__global__ __launch_bounds__(256) void k_vep_vertex(const VepArgs a)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = t / (a.nx + 1), i = t - j * (a.nx + 1);
    if (j > a.ny) return;
    if (a.si) { if (a.soft) vep_vertex_at<true, true>(a, i, j); else vep_vertex_at<false, true>(a, i, j); }
    else if (a.soft) vep_vertex_at<true>(a, i, j);
    else vep_vertex_at<false>(a, i, j);
}

// operands of the centre half, loaded up front: in the merged launch they are requested BEFORE the vertex half stores anything (the compiler cannot move
// loads above stores through unrelated pointers), so that the two halves' memory round trips overlap
struct CentreOps { double e, exyc, exx, eyy, txx, tyy, txyc, toxx, toyy, toxyc, theta, lam, EII, dexx, deyy, dexyc; };
template <bool SOFT, bool SI>
__device__ __forceinline__ CentreOps vep_centre_load(const VepArgs &a, const int i, const int j)
{
    const int nx = a.nx;
    const i64 c = i + (i64)nx * j;
    CentreOps o;
    o.e = a.f.eta[c];
    o.exyc = (V2(a.f.exy, i, j) + V2(a.f.exy, i + 1, j) + V2(a.f.exy, i, j + 1) + V2(a.f.exy, i + 1, j + 1)) / 4;
    o.exx = a.f.exx[c]; o.eyy = a.f.eyy[c];
    o.txx = a.f.txx[c]; o.tyy = a.f.tyy[c]; o.txyc = a.f.txy_c[c];
    o.toxx = a.f.toxx[c]; o.toyy = a.f.toyy[c]; o.toxyc = a.f.toxy_c[c];
    o.theta = a.theta[c]; o.lam = a.lam[c];
    o.EII = SOFT ? a.f.EII_pl[c] : 0.0;
    if (SI) {      // Δεij = (Δε.xx, Δε.yy, av_shear(Δε.xy)) -- cache_tensors, StressUpdate.jl:226-246
        o.dexyc = (V2(a.f.dexy, i, j) + V2(a.f.dexy, i + 1, j) + V2(a.f.dexy, i, j + 1) + V2(a.f.dexy, i + 1, j + 1)) / 4;
        o.dexx = a.f.dexx[c]; o.deyy = a.f.deyy[c];
    } else o.dexyc = o.dexx = o.deyy = 0.0;
    return o;
}

// update_stresses_center_vertex_ps! -- centre half (+ Pr_c, τII, η_vep)
template <bool SOFT, bool SI = false, int NP = 0>
__device__ __forceinline__ void vep_centre_at(const VepArgs &a, const int i, const int j, const CentreOps &o)
{
    const int nx = a.nx, np = NP > 0 ? NP : a.rh.nphase;
    const i64 c = i + (i64)nx * j;
    double *__restrict__ txx_o = a.txx_out ? a.txx_out : a.f.txx, *__restrict__ tyy_o = a.tyy_out ? a.tyy_out : a.f.tyy;
    double rcv[NP > 0 ? NP : 1];
    if (NP > 0) {
#pragma unroll
        for (int q = 0; q < NP; q++) rcv[q] = a.f.phase_c[(i64)NP * c + q];
    }
    const double *rc = NP > 0 ? rcv : a.f.phase_c + (i64)np * c;
    const double _Gdt = SI ? 1.0 / ratio_avg(a.rh.G, rc, np) : 1.0 / (ratio_avg(a.rh.G, rc, np) * a.dt);
    bool is_pl; double eta_reg;
    plastic_params<NP>(a.rh, rc, is_pl, eta_reg);
    const double K = ratio_avg(a.rh.Kb, rc, np);
    const double e = o.e;
    const double dtr = SI ? 1.0 / (a.theta_dtau * a.dt + e * _Gdt + a.dt) : 1.0 / (a.theta_dtau + e * _Gdt + 1.0);
    const double eij[3] = {o.exx, o.eyy, o.exyc};
    double tij[3] = {o.txx, o.tyy, o.txyc};
    const double toij[3] = {o.toxx, o.toyy, o.toxyc};
    double d[3];
    if (SI) {
        const double deij[3] = {o.dexx, o.deyy, o.dexyc};
#pragma unroll
        for (int q = 0; q < 3; q++) d[q] = dev_stress_inc_dt(tij[q], toij[q], e, deij[q], _Gdt, dtr, a.dt);
    } else {
#pragma unroll
        for (int q = 0; q < 3; q++) d[q] = dev_stress_inc(tij[q], toij[q], e, eij[q], _Gdt, dtr);
    }
    double tII = sinv2(d[0] + tij[0], d[1] + tij[1], d[2] + tij[2]);
    const double tt[3] = {tij[0] + d[0], tij[1] + d[1], tij[2] + d[2]};
    double dQdt[3], dQdP, dFdP;
    plastic_grad<NP>(a.rh, rc, tt, dQdt, dQdP, dFdP);
    const double vol = isinf(K) ? 0.0 : K * a.dt * dFdP * dQdP;
    const double Pr = o.theta;
    const double F = yield_F<SOFT, NP>(a.rh, rc, Pr, tII, o.EII);
    double l = o.lam;
    if (is_pl && tII != 0.0 && F > 0) {
        l = fma(1.0 - a.rel, l, a.rel * (fmax(F, 0.0) / (SI ? e * dtr * a.dt + eta_reg + vol : e * dtr + eta_reg + vol)));
        a.lam[c] = l;
        double epl[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            epl[q] = l * dQdt[q];
            d[q] = SI ? fma(-2.0 * e * a.dt * epl[q], dtr, d[q]) : fma(-2.0 * e * epl[q], dtr, d[q]);
            tij[q] = d[q] + tij[q];
        }
        if (a.obs) a.f.evol_pl[c] = -l * dQdP;
        txx_o[c] = tij[0]; tyy_o[c] = tij[1]; a.f.txy_c[c] = tij[2];
        if (a.obs) { a.f.eplxx[c] = epl[0]; a.f.eplyy[c] = epl[1]; }
        tII = sinv2(tij[0], tij[1], tij[2]);
    } else {
        if (a.obs) a.f.evol_pl[c] = 0.0;
        txx_o[c] = d[0] + tij[0]; tyy_o[c] = d[1] + tij[1]; a.f.txy_c[c] = d[2] + tij[2];
        if (a.obs) { a.f.eplxx[c] = 0.0; a.f.eplyy[c] = 0.0; }
    }
    if (a.obs) {
        a.f.tII[c] = tII;
        a.f.eta_vep[c] = tII * 0.5 * (1.0 / sinv2(eij[0], eij[1], eij[2]));
    }
    a.f.P[c] = Pr - (isinf(K) ? 0.0 : K * a.dt * l * dQdP);
}
__global__ __launch_bounds__(256) void k_vep_centre(const VepArgs a)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = t / a.nx, i = t - j * a.nx;
    if (j >= a.ny) return;
    if (a.si) { if (a.soft) vep_centre_at<true, true>(a, i, j, vep_centre_load<true, true>(a, i, j)); else vep_centre_at<false, true>(a, i, j, vep_centre_load<false, true>(a, i, j)); }
    else if (a.soft) vep_centre_at<true>(a, i, j, vep_centre_load<true, false>(a, i, j));
    else vep_centre_at<false>(a, i, j, vep_centre_load<false, false>(a, i, j));
}
// both halves in one launch: the vertex half averages the OLD centre stresses, so the centre half must write τxx, τyy elsewhere
// (a.txx_out / a.tyy_out; the caller then swaps the pointers)
template <bool SOFT, bool SI = false, int NP = 0>
__global__ __launch_bounds__(256) void k_vep_stress2d(const VepArgs a)
{
    const int t = xcd_slab_block() * blockDim.x + threadIdx.x;
    const int j = t / (a.nx + 1), i = t - j * (a.nx + 1);
    if (j > a.ny) return;
    const bool cell = i < a.nx && j < a.ny;
    CentreOps o = {};
    if (cell) o = vep_centre_load<SOFT, SI>(a, i, j);          // before the vertex half's stores
    vep_vertex_at<SOFT, SI, NP>(a, i, j);
    if (cell) vep_centre_at<SOFT, SI, NP>(a, i, j, o);
}

// compute_τ_nonlinear! 2D: single phase (StressKernels.jl:266-307) / phases at the cell centres (:310-351) with
// _compute_τ_nonlinear! (rheology/StressUpdate.jl:2-57).  Centre-only; τ_old.xy and ε_pl.xy are the vertex arrays
// addressed with the centre's [i,j], as the reference's caller passes them (Stokes2D.jl:442-458).
template <bool MULTI>
__global__ __launch_bounds__(256) void k_tau_nonlinear2d(const VepArgs a, double *__restrict__ theta_out)
{
    const int nx = a.nx, ny = a.ny, np = a.rh.nphase;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = t / nx, i = t - j * nx;
    if (j >= ny) return;
    const i64 c = i + (i64)nx * j;
    const double one = 1.0;
    const double *r = MULTI ? a.f.phase_c + (i64)np * c : &one;
    const int n = MULTI ? np : 1;
    const double e = a.f.eta[c], dt = a.dt;
    const double _Gdt = 1.0 / ((MULTI ? ratio_avg(a.rh.G, r, n) : a.rh.G[0]) * dt);
    const double dtr = dev_dtau_r(a.theta_dtau, e, _Gdt);
    bool is_pl = false;
    double C = 0.0, sinphi = 0.0, cosphi = 0.0, sinpsi = 0.0, eta_reg = 0.0;
    for (int q = 0; q < n; q++) {
        if (r[q] == 0.0 || !a.rh.is_pl[q]) continue;
        is_pl = true;
        const double EII = a.soft ? a.f.EII_pl[c] : 0.0;         // soften_cohesion / soften_friction_angle at EII[I...] (StressUpdate.jl:305-381)
        double sp, cp;
        mat_friction(a.rh, q, EII, sp, cp);
        C += mat_cohesion(a.rh, q, EII) * r[q]; sinphi += sp * r[q]; cosphi += cp * r[q];
        sinpsi += a.rh.sinpsi[q] * r[q]; eta_reg += a.rh.eta_vp[q] * r[q];
    }
    const double K = MULTI ? ratio_avg(a.rh.Kb, r, n) : a.rh.Kb[0];
    const double volume = isinf(K) ? 0.0 : K * dt * sinphi * sinpsi;
    const double eij[3] = {a.f.exx[c], a.f.eyy[c], (V2(a.f.exy, i, j) + V2(a.f.exy, i + 1, j) + V2(a.f.exy, i, j + 1) + V2(a.f.exy, i + 1, j + 1)) / 4};
    const double tij[3] = {a.f.txx[c], a.f.tyy[c], a.f.txy_c[c]};
    const double toij[3] = {a.f.toxx[c], a.f.toyy[c], V2(a.f.toxy, i, j)};
    const double P = a.f.P[c];
    double d[3], ldq[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 3; q++) d[q] = dev_stress_inc(tij[q], toij[q], e, eij[q], _Gdt, dtr);
    const double tII_trial = sinv2(tij[0] + d[0], tij[1] + d[1], tij[2] + d[2]);
    const double ty = fmax(C * cosphi + P * sinphi, 0.0);
    double l = a.lam[c];
    if (is_pl && tII_trial > ty) {
        const double F = tII_trial - ty;
        l = 0.5 * l + (1 - 0.5) * (F > 0.0 ? 1.0 : 0.0) * F * (1.0 / (e * dtr + eta_reg + volume));
        const double l_tII = l * 0.5 * (1.0 / tII_trial);
#pragma unroll
        for (int q = 0; q < 3; q++) {
            ldq[q] = (tij[q] + d[q]) * l_tII;
            d[q] = fma(-dtr * 2.0, e * ldq[q], d[q]);
        }
        a.lam[c] = l;
    }
    a.f.eplxx[c] = isinf(ldq[0]) ? 0.0 : ldq[0];
    a.f.eplyy[c] = isinf(ldq[1]) ? 0.0 : ldq[1];
    V2(a.f.eplxy, i, j) = isinf(ldq[2]) ? 0.0 : ldq[2];
    a.f.txx[c] = tij[0] + d[0]; a.f.tyy[c] = tij[1] + d[1]; a.f.txy_c[c] = tij[2] + d[2];
    const double tII = sinv2(tij[0] + d[0], tij[1] + d[1], tij[2] + d[2]);
    a.f.tII[c] = tII;
    a.f.eta_vep[c] = tII * 0.5 * (1.0 / sinv2(eij[0], eij[1], eij[2]));
    theta_out[c] = P + (isinf(K) ? 0.0 : K * dt * l * sinpsi);
}

// center2vertex! 2D (Interpolations.jl:101-114): pass 0 inner vertices, pass 1 the x-edge rows, pass 2 the y-edge columns; pass 3 = the three at once: after
// them every edge / corner vertex is a copy of the inner vertex its indices clamp to, so each thread evaluates that one
__global__ __launch_bounds__(256) void k_center2vertex2d(double *__restrict__ v, const double *__restrict__ cc, int nx, int ny, int pass)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (pass == 3) {
        const int j = t / (nx + 1), i = t - j * (nx + 1);
        if (j > ny) return;
        const int ii = clampi(i, 1, nx - 1), jj = clampi(j, 1, ny - 1);
        v[i + (i64)(nx + 1) * j] = 0.25 * (cc[(ii - 1) + (i64)nx * (jj - 1)] + cc[ii + (i64)nx * (jj - 1)] + cc[(ii - 1) + (i64)nx * jj] + cc[ii + (i64)nx * jj]);
    } else if (pass == 0) {
        const int j = t / (nx + 1), i = t - j * (nx + 1);
        if (j > ny || i < 1 || i >= nx || j < 1 || j >= ny) return;
        v[i + (i64)(nx + 1) * j] = 0.25 * (cc[(i - 1) + (i64)nx * (j - 1)] + cc[i + (i64)nx * (j - 1)] + cc[(i - 1) + (i64)nx * j] + cc[i + (i64)nx * j]);
    } else if (pass == 1) {
        if (t > ny) return;
        v[0 + (i64)(nx + 1) * t] = v[1 + (i64)(nx + 1) * t];
        v[nx + (i64)(nx + 1) * t] = v[nx - 1 + (i64)(nx + 1) * t];
    } else {
        if (t > nx) return;
        v[t] = v[t + (i64)(nx + 1)];
        v[t + (i64)(nx + 1) * ny] = v[t + (i64)(nx + 1) * (ny - 1)];
    }
}

__device__ __forceinline__ double phase_viscosity(const jrx_rheology &rh, const double *r)
{   // compute_phase_viscosity, rheology/Viscosity.jl:605-625 (LinearViscous elements)
    for (int q = 0; q < rh.nphase; q++)
        if (r[q] > 0.999) return rh.eta[q];
    double s = 0.0;
    for (int q = 0; q < rh.nphase; q++)
        if (r[q] != 0.0) s += (1.0 / rh.eta[q]) * r[q];
    return 1.0 / s;
}
// compute_viscosity_kernel! at a centre / a vertex for creep laws that read fields (rheology/Viscosity.jl:382-418): the invariant of @stress_center /
// @strain_center, args at the cell (T at I .+ 1 of the ghosted thermal.T, local_viscosity_args :513-523); at a vertex (xx_v, yy_v, xy) -- the PT solvers
// never write xx_v, yy_v: zero -- and args averaged over the clamped surrounding centres, T over its 2 x 2 nodes (local_viscosity_args_vertex :528-552)
__device__ __forceinline__ double vep_visc_fields_centre(const VepArgs &a, const i64 t)
{
    const int nx = a.nx, j = (int)(t / nx), i = (int)(t - (i64)j * nx);
    const double AII = a.vtau ? mat_visc_invariant2(a.f.txx[t], a.f.tyy[t], a.f.txy_c[t]) : mat_visc_invariant2(a.f.exx[t], a.f.eyy[t], a.f.exy_c[t]);
    const double T = !a.f.T ? 0.0 : (a.tg ? a.f.T[(i + 1) + (i64)(nx + 2) * (j + 1)] : a.f.T[t]);
    return mat_phase_viscosity(a.rh, a.f.phase_c + (i64)a.rh.nphase * t, AII, T, a.f.P[t], a.vtau);
}
__device__ __forceinline__ double vep_visc_fields_vertex(const VepArgs &a, const i64 t)
{
    const int nx = a.nx, ny = a.ny, j = (int)(t / (nx + 1)), i = (int)(t - (i64)j * (nx + 1));
    const int il = max(i - 1, 0), ir = min(i, nx - 1), jb = max(j - 1, 0), jt = min(j, ny - 1);
    const double AII = mat_visc_invariant2(0.0, 0.0, a.vtau ? a.f.txy[t] : a.f.exy[t]);
    const double P = 0.25 * (a.f.P[il + (i64)nx * jb] + a.f.P[ir + (i64)nx * jb] + a.f.P[il + (i64)nx * jt] + a.f.P[ir + (i64)nx * jt]);
    double T = 0.0;
    if (a.f.T && a.tg) {
        const double *q = a.f.T + i + (i64)(nx + 2) * j;
        T = 0.25 * (q[0] + q[1] + q[nx + 2] + q[nx + 3]);
    } else if (a.f.T) T = 0.25 * (a.f.T[il + (i64)nx * jb] + a.f.T[ir + (i64)nx * jb] + a.f.T[il + (i64)nx * jt] + a.f.T[ir + (i64)nx * jt]);
    return mat_phase_viscosity(a.rh, a.f.phase_v + (i64)a.rh.nphase * t, AII, T, P, a.vtau);
}
__device__ __forceinline__ void vep_visc_at(const VepArgs &a, const i64 t)
{
    const int nx = a.nx, ny = a.ny, np = a.rh.nphase;
    if (a.vfields) {
        if (t < (i64)nx * ny) {
            const double e = vep_visc_fields_centre(a, t) * a.nu + a.f.eta[t] * (1.0 - a.nu);
            a.f.eta[t] = fmin(fmax(e, a.cut_lo), a.cut_hi);
        }
        if (a.f.eta_v && t < (i64)(nx + 1) * (ny + 1)) {
            const double e = vep_visc_fields_vertex(a, t) * a.nu + a.f.eta_v[t] * (1.0 - a.nu);
            a.f.eta_v[t] = fmin(fmax(e, a.cut_lo), a.cut_hi);
        }
        return;
    }
    if (t < (i64)nx * ny) {
        double e = a.eta_lin_c ? a.eta_lin_c[t] : phase_viscosity(a.rh, a.f.phase_c + np * t);
        e = e * a.nu + a.f.eta[t] * (1.0 - a.nu);
        a.f.eta[t] = fmin(fmax(e, a.cut_lo), a.cut_hi);
    }
    if (a.f.eta_v && t < (i64)(nx + 1) * (ny + 1)) {
        double e = a.eta_lin_v ? a.eta_lin_v[t] : phase_viscosity(a.rh, a.f.phase_v + np * t);
        e = e * a.nu + a.f.eta_v[t] * (1.0 - a.nu);
        a.f.eta_v[t] = fmin(fmax(e, a.cut_lo), a.cut_hi);
    }
}
__global__ __launch_bounds__(256) void k_vep_visc(const VepArgs a) { vep_visc_at(a, (i64)blockIdx.x * blockDim.x + threadIdx.x); }
// compute_viscosity! and compute_V! in one launch: the velocity update reads ητ (already taken from the previous η), P, τ, ρg, never η
template <bool BCF>
__global__ __launch_bounds__(256) void k_vep_visc_velocity(const VepArgs a, const Args2 b)
{
    const i64 t = (i64)xcd_slab_block() * blockDim.x + threadIdx.x;
    vep_visc_at(a, t);
    const int j = (int)(t / b.nx), i = (int)(t - (i64)j * b.nx);
    if (j < b.ny) velocity2d_cell<false, BCF>(b, i, j);
}

// k_vep_visc_velocity for laws whose η reads no field (the phase average is precomputed), uniform grids and no free surface, with every operand requested up front (option
// "fused2d_batch").  The general kernel carries the field-reading creep laws, the non-uniform spacings and the free-surface correction as run-time branches: 3,200 ISA lines, 68
// branches, its 48 loads in ~20 dependent groups.  Same arithmetic on the path it takes for these inputs, expression for expression.
template <bool BCF>
__global__ __launch_bounds__(256) void k_vep_visc_velocity_b(const VepArgs a, const Args2 b)
{
    const i64 t = (i64)xcd_slab_block() * blockDim.x + threadIdx.x;
    const int nx = b.nx, ny = b.ny;
    const i64 nc = (i64)nx * ny, nv = (i64)(nx + 1) * (ny + 1);
    const bool cell = t < nc, vert = a.f.eta_v != nullptr && t < nv;
    const i64 c = cell ? t : 0, tv = t < nv ? t : 0;
    const int j = (int)(c / nx), i = (int)(c - (i64)j * nx);
    const i64 cx = c + (i < nx - 1 ? 1 : 0), cy = c + (j < ny - 1 ? nx : 0);
    const double *__restrict__ P = b.f.P, *__restrict__ txy = b.f.txy, *__restrict__ et = b.etatau;
    const i64 qx = (i + 1) + (i64)(nx + 1) * (j + 1), qy = (i + 1) + (i64)(nx + 2) * (j + 1);
    // ---- every operand
    const double el = a.eta_lin_c[c], eo = a.f.eta[c];
    double elv = 0.0, eov = 0.0;
    if (a.f.eta_v) { elv = a.eta_lin_v[tv]; eov = a.f.eta_v[tv]; }
    const double P0 = P[c], Px = P[cx], Py = P[cy], X0 = b.f.txx[c], X1 = b.f.txx[cx], Y0 = b.f.tyy[c], Y1 = b.f.tyy[cy];
    const double S10 = txy[(i + 1) + (i64)(nx + 1) * j], S11 = txy[(i + 1) + (i64)(nx + 1) * (j + 1)], S01 = txy[i + (i64)(nx + 1) * (j + 1)];
    const double fx0 = b.f.fx[c], fx1 = b.f.fx[cx], fy0 = b.f.fy[c], fy1 = b.f.fy[cy];
    const double E0 = et[c], Ex = et[cx], Ey = et[cy];
    const double vx0 = b.f.Vx[qx], vy0 = b.f.Vy[qy];
    __builtin_amdgcn_sched_barrier(0);
    // ---- compute_viscosity! (vep_visc_at, laws without fields)
    if (cell) {
        double e = el;
        e = e * a.nu + eo * (1.0 - a.nu);
        a.f.eta[c] = fmin(fmax(e, a.cut_lo), a.cut_hi);
    }
    if (vert) {
        double e = elv;
        e = e * a.nu + eov * (1.0 - a.nu);
        a.f.eta_v[tv] = fmin(fmax(e, a.cut_lo), a.cut_hi);
    }
    if (!cell) return;
    // ---- compute_V! (velocity2d_cell<false, BCF>, uniform spacing, fs_dt = 0)
    const double edt = b.eta_dtau, _dx = b._dx, _dy = b._dy;
    if (i < nx - 1) {
        const double dP = (-P0 + Px) * _dx, dT = (-X0 + X1) * _dx;
        const double dS = (-S10 + S11) * _dy, av = (fx0 + fx1) * 0.5;
        const double v = vx0 + (-dP + dT + dS - av) * edt / ((E0 + Ex) * 0.5);
        b.f.Vx[qx] = v;
        if (BCF) {      // Vx ghost rows j = 0 (bot) and j = ny+1 (top)
            if (j == 0) { if (b.fs & JRX_FACE_BOT) b.f.Vx[qx - (nx + 1)] = v; else if (b.ns & JRX_FACE_BOT) b.f.Vx[qx - (nx + 1)] = -v; }
            if (j == ny - 1) { if (b.fs & JRX_FACE_TOP) b.f.Vx[qx + (nx + 1)] = v; else if (b.ns & JRX_FACE_TOP) b.f.Vx[qx + (nx + 1)] = -v; }
        }
    }
    if (j < ny - 1) {
        const double dP = (-P0 + Py) * _dy, dT = (-Y0 + Y1) * _dy;
        const double dS = (-S01 + S11) * _dx, av = (fy0 + fy1) * 0.5;
        const double rhs = -dP + dT + dS - av;
        const double v = vy0 + rhs * edt / ((E0 + Ey) * 0.5);
        b.f.Vy[qy] = v;
        if (BCF) {      // Vy ghost columns i = 0 (left) and i = nx+1 (right)
            if (i == 0) { if (b.fs & JRX_FACE_LEFT) b.f.Vy[qy - 1] = v; else if (b.ns & JRX_FACE_LEFT) b.f.Vy[qy - 1] = -v; }
            if (i == nx - 1) { if (b.fs & JRX_FACE_RIGHT) b.f.Vy[qy + 1] = v; else if (b.ns & JRX_FACE_RIGHT) b.f.Vy[qy + 1] = -v; }
        }
    }
}

// rho: also compute_ρg!(ρg, phase_ratios, rheology, args) (Stokes2D.jl:646)
__global__ __launch_bounds__(256) void k_phase_avg(double *__restrict__ Kc, double *__restrict__ Gc, const VepArgs a, const bool rho, double *__restrict__ elc = nullptr,
                                                   double *__restrict__ elv = nullptr)
{
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (elv && t < (i64)(a.nx + 1) * (a.ny + 1)) elv[t] = phase_viscosity(a.rh, a.f.phase_v + a.rh.nphase * t);
    if (t >= (i64)a.nx * a.ny) return;
    if (elc) elc[t] = phase_viscosity(a.rh, a.f.phase_c + a.rh.nphase * t);
    Kc[t] = ratio_avg(a.rh.Kb, a.f.phase_c + a.rh.nphase * t, a.rh.nphase);
    Gc[t] = ratio_avg(a.rh.G, a.f.phase_c + a.rh.nphase * t, a.rh.nphase);
    if (rho) a.f.fy[t] = mat_density_ratio(a.rh, a.f.phase_c + a.rh.nphase * t, !a.f.T ? 0.0 : (a.tg ? a.f.T[(t % a.nx) + (i64)(a.nx + 2) * (t / a.nx)] : a.f.T[t]), a.f.P[t]) * a.rh.gravity;
}

// Single-phase driver (Stokes2D.jl:345-557): compute_ρg!/update_ρg!(ρg[2], rheology, args) and compute_viscosity!/compute_viscosity_τII!
// (Viscosity.jl:142-167) for creep laws without strain-rate dependence: η <- clamp((1 - ν) η + ν η_creep(T, P), cutoff).
// args.T: cell-centred (nx, ny), or -- tg -- thermal.T (nx+2, ny+2) indexed as the reference does: density at [i, j]
// (getindex_NamedTuple(args, I...), BuoyancyForces.jl:17), viscosity at [i+1, j+1] (local_viscosity_args, Viscosity.jl:513-523).
// A power-law creep takes its invariant from @strain(stokes) = (ε.xx, ε.yy, ε.xy[i, j] -- the vertex array at the cell's index) in both forms, as
// _compute_viscosity!(stokes, ν, args, rheology, cutoff, fn_viscosity) does (Viscosity.jl:136-167); a.vtau: fn_viscosity is compute_viscosity_τII.
__global__ __launch_bounds__(256) void k_single_material(const VepArgs a, const double nu, const bool rho, const bool visc, const bool tg)
{
    const int nx = a.nx, ny = a.ny;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = t / nx, i = t - j * nx;
    if (j >= ny) return;
    const i64 c = i + (i64)nx * j;
    const double P = a.f.P[c];
    if (rho) {
        const double T = !a.f.T ? 0.0 : (tg ? a.f.T[i + (i64)(nx + 2) * j] : a.f.T[c]);
        a.f.fy[c] = mat_density(a.rh, 0, T, P) * a.rh.gravity;
    }
    if (visc) {
        const double T = !a.f.T ? 0.0 : (tg ? a.f.T[(i + 1) + (i64)(nx + 2) * (j + 1)] : a.f.T[c]);
        const double AII = a.rh.visc_kind[0] == 2 ? mat_visc_invariant2(a.f.exx[c], a.f.eyy[c], a.f.exy[i + (i64)(nx + 1) * j]) : 0.0;
        const double e = (1 - nu) * a.f.eta[c] + nu * mat_viscosity(a.rh, 0, AII, T, P, a.vtau);
        a.f.eta[c] = fmin(fmax(e, a.cut_lo), a.cut_hi);
    }
}
__global__ __launch_bounds__(256) void k_fill2(double *__restrict__ A, double va, double *__restrict__ B, double vb, i64 n)
{
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) { A[t] = va; B[t] = vb; }
}

__global__ __launch_bounds__(256) void k_tensor_invariant2d(double *__restrict__ II, const double *__restrict__ xx, const double *__restrict__ yy,
                                                           const double *__restrict__ xy, int nx, int ny)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = t / nx, i = t - j * nx;
    if (j >= ny) return;
    II[t] = sinv_stag(xx[t], yy[t], V2(xy, i, j), V2(xy, i + 1, j), V2(xy, i, j + 1), V2(xy, i + 1, j + 1));
}

__global__ __launch_bounds__(256) void k_axpy_dt(double *__restrict__ y, const double *__restrict__ x, double dt, i64 n)
{
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) y[t] += dt * x[t];
}

// the epilogue operators alone: mode 0 shear2center_kernel! (Interpolations.jl:306-311), 1 accumulate_tensor_kernel!
// (StressKernels.jl:379-392), 2 compute_vorticity! (stress_rotation_particles.jl:17-29; over the vertices)
__global__ __launch_bounds__(256) void k_epilogue_op2d(int mode, double *__restrict__ out, const double *A, const double *B, const double *Cv, double s1,
                                                       double s2, int nx, int ny)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (mode == 2) {
        const int j = t / (nx + 1), i = t - j * (nx + 1);
        if (j > ny) return;
        // A = Vx (nx+1, ny+2), B = Vy (nx+2, ny+1); s1 = _dx, s2 = _dy
        V2(out, i, j) = 0.5 * ((-B[i + (i64)(nx + 2) * j] + B[(i + 1) + (i64)(nx + 2) * j]) * s1 - (-A[i + (i64)(nx + 1) * j] + A[i + (i64)(nx + 1) * (j + 1)]) * s2);
        return;
    }
    const int j = t / nx, i = t - j * nx;
    if (j >= ny) return;
    const i64 c = i + (i64)nx * j;
    if (mode == 0) out[c] = 0.25 * (V2(Cv, i, j) + V2(Cv, i + 1, j) + V2(Cv, i, j + 1) + V2(Cv, i + 1, j + 1));
    else out[c] += sinv_stag(A[c], B[c], V2(Cv, i, j), V2(Cv, i + 1, j), V2(Cv, i, j + 1), V2(Cv, i + 1, j + 1)) * s1;
}

// post-loop epilogue: compute_vorticity!, shear2center! x3, accumulate_tensor!, accumulate_vol! (Stokes2D.jl:831-843)
__global__ __launch_bounds__(256) void k_vep_epilogue(const VepArgs a)
{
    const int nx = a.nx, ny = a.ny;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = t / (nx + 1), i = t - j * (nx + 1);
    if (j > ny) return;
    if (a.f.omega_xy)
        V2(a.f.omega_xy, i, j) = 0.5 * ((-a.f.Vy[i + (i64)(nx + 2) * j] + a.f.Vy[(i + 1) + (i64)(nx + 2) * j]) * spc(a.sp.vyx, i, a._dx) -
                                        (-a.f.Vx[i + (i64)(nx + 1) * j] + a.f.Vx[i + (i64)(nx + 1) * (j + 1)]) * spc(a.sp.vxy, j, a._dy));
    if (i < nx && j < ny) {
        const i64 c = i + (i64)nx * j;
#define S2C(V) (0.25 * (V2(V, i, j) + V2(V, i + 1, j) + V2(V, i, j + 1) + V2(V, i + 1, j + 1)))
        if (a.f.exy_c) a.f.exy_c[c] = S2C(a.f.exy);
        if (a.f.eplxy_c) a.f.eplxy_c[c] = S2C(a.f.eplxy);
        if (a.f.dexy_c && a.f.dexy) a.f.dexy_c[c] = S2C(a.f.dexy);
#undef S2C
        a.f.EII_pl[c] += sinv_stag(a.f.eplxx[c], a.f.eplyy[c], V2(a.f.eplxy, i, j), V2(a.f.eplxy, i + 1, j), V2(a.f.eplxy, i, j + 1),
                                   V2(a.f.eplxy, i + 1, j + 1)) * a.dt;
        a.f.EVol_pl[c] += a.dt * a.f.evol_pl[c];
    }
}
#undef C2
#undef V2

jrx_status check_vep(jrx_handle *h, const jrx_vep2d_fields *f, const jrx_rheology *rh, const jrx_vep2d_params *p)
{
    if (!h) return JRX_ERR_ARG;
    if (!f || !rh || !p) return jrx_fail(h, JRX_ERR_ARG, "null VEP argument");
    JRX_TRY(jrx_check_device(h));
    if (p->nx < 3 || p->ny < 3) return jrx_fail(h, JRX_ERR_ARG, "2D Stokes needs at least 3 cells per dimension");
    if (rh->nphase < 1 || rh->nphase > JRX_MAXPHASE) return jrx_fail(h, JRX_ERR_ARG, "nphase must be in 1..%d", JRX_MAXPHASE);
    const void *req[] = {f->P, f->P0, f->divV, f->Q, f->Vx, f->Vy, f->Ux, f->Uy, f->exx, f->eyy, f->exy, f->eplxx, f->eplyy, f->eplxy, f->eplxy_c,
                         f->txx, f->tyy, f->txy, f->txy_c, f->tII, f->toxx, f->toyy, f->toxy, f->toxy_c, f->eta, f->eta_vep, f->EII_pl, f->evol_pl,
                         f->EVol_pl, f->fx, f->fy, f->RP, f->Rx, f->Ry, f->phase_c, f->phase_v};
    for (const void *q : req)
        if (!q) return jrx_fail(h, JRX_ERR_ARG, "a required VEP field pointer is NULL");
    if (p->strain_increment && (!f->dexx || !f->deyy || !f->dexy || !f->divU))
        return jrx_fail(h, JRX_ERR_ARG, "strain_increment: the Δε (xx, yy, xy) and ∇U arrays are required");
    if (!spacing_ok(p->inv_spacing)) return jrx_fail(h, JRX_ERR_ARG, "non-uniform grid: all six inverse-spacing arrays are required");
    if (p->inv_spacing[0] && p->strain_increment)
        return jrx_fail(h, JRX_ERR_UNSUPPORTED, "strain_increment on a non-uniform grid is not built (the reference's own kernel indexes _di.center beyond its extent there)");
    return JRX_OK;
}

VepArgs make_vep(const jrx_vep2d_fields *f, const jrx_rheology *rh, const jrx_vep2d_params *p)
{
    VepArgs a;
    memset(&a, 0, sizeof(a));
    a.f = *f; a.rh = *rh;
    a._dx = p->_dx; a._dy = p->_dy; a.dt = p->dt; a.r = p->r; a.theta_dtau = p->theta_dtau; a.rel = p->lambda_relaxation;
    a.nu = p->viscosity_relaxation; a.cut_lo = p->cutoff_lo; a.cut_hi = p->cutoff_hi;
    a.nx = (int)p->nx; a.ny = (int)p->ny;
    a.soft = mat_has_softening(rh);
    a.si = p->strain_increment != 0;
    a.tg = p->T_ghosted != 0;
    a.vfields = mat_viscosity_reads_fields(rh); a.vtau = true;
    a.obs = true;
    a.sp = Sp2{p->inv_spacing[0], p->inv_spacing[1], p->inv_spacing[2], p->inv_spacing[3], p->inv_spacing[4], p->inv_spacing[5]};
    return a;
}

// the velocity / residual kernels of the visco-elastic path work on this view (P = stokes.P = Pr_c, τxy at vertices)
jrx_stokes2d_fields view2d(const jrx_vep2d_fields *f)
{
    jrx_stokes2d_fields g;
    memset(&g, 0, sizeof(g));
    g.P = f->P; g.P0 = f->P0; g.divV = f->divV; g.Q = f->Q; g.Vx = f->Vx; g.Vy = f->Vy; g.Ux = f->Ux; g.Uy = f->Uy;
    g.txx = f->txx; g.tyy = f->tyy; g.txy = f->txy; g.exx = f->exx; g.eyy = f->eyy; g.exy = f->exy; g.eta = f->eta;
    g.fx = f->fx; g.fy = f->fy; g.RP = f->RP; g.Rx = f->Rx; g.Ry = f->Ry;
    return g;
}

}   // namespace

extern "C" {

jrx_status jrx_tensor_invariant2d(jrx_handle *h, double *II, const double *xx, const double *yy, const double *xy, int64_t nx, int64_t ny)
{
    if (!h) return JRX_ERR_ARG;
    if (!II || !xx || !yy || !xy || nx < 1 || ny < 1) return jrx_fail(h, JRX_ERR_ARG, "tensor_invariant!: bad argument");
    hipLaunchKernelGGL(k_tensor_invariant2d, dim3((unsigned)((nx * ny + 255) / 256)), dim3(256), 0, h->stream, II, xx, yy, xy, (int)nx, (int)ny);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_vep2d_compute_viscosity(jrx_handle *h, const jrx_vep2d_fields *f, const jrx_rheology *rh, const jrx_vep2d_params *p, double nu)
{
    JRX_TRY(check_vep(h, f, rh, p));
    VepArgs a = make_vep(f, rh, p);
    a.nu = nu; a.vtau = false;
    hipLaunchKernelGGL(k_vep_visc, dim3((unsigned)(((p->nx + 1) * (p->ny + 1) + 255) / 256)), dim3(256), 0, h->stream, a);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}
jrx_status jrx_vep2d_compute_viscosity_tauII(jrx_handle *h, const jrx_vep2d_fields *f, const jrx_rheology *rh, const jrx_vep2d_params *p, double nu)
{
    JRX_TRY(check_vep(h, f, rh, p));
    VepArgs a = make_vep(f, rh, p);
    a.nu = nu; a.vtau = true;
    hipLaunchKernelGGL(k_vep_visc, dim3((unsigned)(((p->nx + 1) * (p->ny + 1) + 255) / 256)), dim3(256), 0, h->stream, a);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_vep2d_update_stresses(jrx_handle *h, const jrx_vep2d_fields *f, const double *theta, double *lambda, double *lambda_v,
                                     const jrx_rheology *rh, const jrx_vep2d_params *p)
{
    JRX_TRY(check_vep(h, f, rh, p));
    if (!theta || !lambda || !lambda_v) return jrx_fail(h, JRX_ERR_ARG, "θ / λ / λv is NULL");
    VepArgs a = make_vep(f, rh, p);
    a.theta = theta; a.lam = lambda; a.lamv = lambda_v;
    const unsigned gv = (unsigned)(((p->nx + 1) * (p->ny + 1) + 255) / 256), gc = (unsigned)((p->nx * p->ny + 255) / 256);
    hipLaunchKernelGGL(k_vep_vertex, dim3(gv), dim3(256), 0, h->stream, a);
    JRX_LAUNCH_CHECK(h);
    hipLaunchKernelGGL(k_vep_centre, dim3(gc), dim3(256), 0, h->stream, a);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_shear2center2d(jrx_handle *h, double *xy_c, const double *xy, int64_t nx, int64_t ny)
{
    if (!h) return JRX_ERR_ARG;
    if (!xy_c || !xy || nx < 1 || ny < 1) return jrx_fail(h, JRX_ERR_ARG, "shear2center!: bad argument");
    hipLaunchKernelGGL(k_epilogue_op2d, dim3((unsigned)((nx * ny + 255) / 256)), dim3(256), 0, h->stream, 0, xy_c, (const double *)nullptr,
                       (const double *)nullptr, xy, 0.0, 0.0, (int)nx, (int)ny);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_accumulate_tensor2d(jrx_handle *h, double *II, const double *xx, const double *yy, const double *xy, double dt, int64_t nx, int64_t ny)
{
    if (!h) return JRX_ERR_ARG;
    if (!II || !xx || !yy || !xy || nx < 1 || ny < 1) return jrx_fail(h, JRX_ERR_ARG, "accumulate_tensor!: bad argument");
    hipLaunchKernelGGL(k_epilogue_op2d, dim3((unsigned)((nx * ny + 255) / 256)), dim3(256), 0, h->stream, 1, II, xx, yy, xy, dt, 0.0, (int)nx, (int)ny);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_compute_vorticity2d(jrx_handle *h, double *wxy, const double *Vx, const double *Vy, int64_t nx, int64_t ny, double _dx, double _dy)
{
    if (!h) return JRX_ERR_ARG;
    if (!wxy || !Vx || !Vy || nx < 1 || ny < 1) return jrx_fail(h, JRX_ERR_ARG, "compute_vorticity!: bad argument");
    hipLaunchKernelGGL(k_epilogue_op2d, dim3((unsigned)(((nx + 1) * (ny + 1) + 255) / 256)), dim3(256), 0, h->stream, 2, wxy, Vx, Vy, (const double *)nullptr,
                       _dx, _dy, (int)nx, (int)ny);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

// accumulate_vol!(EVol_pl, ε_vol_pl, dt): EVol_pl += dt * ε_vol_pl (StressKernels.jl:410-431), any dimension
jrx_status jrx_accumulate_vol(jrx_handle *h, double *EVol, const double *evol, double dt, int64_t n)
{
    if (!h) return JRX_ERR_ARG;
    if (!EVol || !evol || n < 1) return jrx_fail(h, JRX_ERR_ARG, "accumulate_vol!: bad argument");
    hipLaunchKernelGGL(k_axpy_dt, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, EVol, evol, dt, (i64)n);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_compute_tau_nonlinear2d(jrx_handle *h, const jrx_vep2d_fields *f, double *theta, double *lambda, const jrx_rheology *rh,
                                       const jrx_vep2d_params *p, int32_t multiphase)
{
    if (!h) return JRX_ERR_ARG;
    if (!f || !rh || !p) return jrx_fail(h, JRX_ERR_ARG, "null VEP argument");
    if (p->nx < 1 || p->ny < 1) return jrx_fail(h, JRX_ERR_ARG, "compute_τ_nonlinear!: empty grid");
    if (rh->nphase < 1 || rh->nphase > JRX_MAXPHASE) return jrx_fail(h, JRX_ERR_ARG, "nphase must be in 1..%d", JRX_MAXPHASE);
    if (!theta || !lambda) return jrx_fail(h, JRX_ERR_ARG, "θ / λ is NULL");
    const void *req[] = {f->P, f->exx, f->eyy, f->exy, f->eplxx, f->eplyy, f->eplxy, f->txx, f->tyy, f->txy_c, f->tII, f->toxx, f->toyy,
                         f->toxy, f->eta, f->eta_vep, multiphase ? (const void *)f->phase_c : (const void *)f->P};
    for (const void *q : req)
        if (!q) return jrx_fail(h, JRX_ERR_ARG, "compute_τ_nonlinear!: a required field pointer is NULL");
    VepArgs a = make_vep(f, rh, p);
    a.lam = lambda;
    const unsigned gc = (unsigned)((p->nx * p->ny + 255) / 256);
    if (multiphase) hipLaunchKernelGGL(k_tau_nonlinear2d<true>, dim3(gc), dim3(256), 0, h->stream, a, theta);
    else hipLaunchKernelGGL(k_tau_nonlinear2d<false>, dim3(gc), dim3(256), 0, h->stream, a, theta);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_center2vertex2d(jrx_handle *h, double *vertex, const double *center, int64_t nx, int64_t ny)
{
    if (!h) return JRX_ERR_ARG;
    if (!vertex || !center || nx < 2 || ny < 2) return jrx_fail(h, JRX_ERR_ARG, "center2vertex!: bad argument");
    const i64 nv = (nx + 1) * (ny + 1);
    hipLaunchKernelGGL(k_center2vertex2d, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, h->stream, vertex, center, (int)nx, (int)ny, 3);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_stokes2d_vep_solve(jrx_handle *h, const jrx_vep2d_fields *f, const jrx_rheology *rh, const jrx_vep2d_params *p,
                                  jrx_solve_result *res)
{
    JRX_TRY(check_vep(h, f, rh, p));
    if (!res) return jrx_fail(h, JRX_ERR_ARG, "null result");
    if (p->nout < 1) return jrx_fail(h, JRX_ERR_ARG, "nout must be >= 1");
    const bool comm = jrx_comm_active(h);
    const int nx = (int)p->nx, ny = (int)p->ny;
    const int64_t nn[3] = {nx, ny, 1};
    const size_t n = (size_t)nx * ny, nv = (size_t)(nx + 1) * (ny + 1);
    hipStream_t s = h->stream;
    // library scratch: ητ, θ, λ, K, G (centre), λv (vertex) and the second set of τxx, τyy, carved out of one allocation
    JRX_TRY(jrx_ensure_etatau(h, 8 * n + 2 * nv));
    double *etatau = h->etatau, *theta = etatau + n, *lam = theta + n, *Kc = lam + n, *Gc = Kc + n, *lamv = Gc + n;
    VepArgs a = make_vep(f, rh, p);
    a.theta = theta; a.etatau = etatau; a.Kc = Kc; a.Gc = Gc; a.lam = lam; a.lamv = lamv;
    a.txx_out = lamv + nv; a.tyy_out = a.txx_out + n;
    double *eta_lin_c = a.tyy_out + n, *eta_lin_v = eta_lin_c + n;
    jrx_stokes2d_fields g = view2d(f);
    jrx_stokes2d_params q;
    memset(&q, 0, sizeof(q));
    q.nx = nx; q.ny = ny; q.nxg = p->nxg; q.nyg = p->nyg; q._dx = p->_dx; q._dy = p->_dy; q.dt = p->dt; q.r = p->r;
    q.theta_dtau = p->theta_dtau; q.eta_dtau = p->eta_dtau; q.free_slip = p->free_slip; q.no_slip = p->no_slip; q.periodic = p->periodic;
    for (int d = 0; d < 6; d++) q.inv_spacing[d] = p->inv_spacing[d];
    Args2 b = make_args2(&g, etatau, &q);
    const unsigned gv = (unsigned)((nv + 255) / 256), gc = (unsigned)((n + 255) / 256);

    JRX_HIP(h, hipMemcpyAsync(f->P0, f->P, n * sizeof(double), hipMemcpyDeviceToDevice, s));        // @copy stokes.P0 stokes.P
    JRX_HIP(h, hipMemcpyAsync(theta, f->P, n * sizeof(double), hipMemcpyDeviceToDevice, s));        // θ = deepcopy(stokes.P)
    JRX_HIP(h, hipMemsetAsync(lam, 0, n * sizeof(double), s));
    JRX_HIP(h, hipMemsetAsync(lamv, 0, nv * sizeof(double), s));
    JRX_HIP(h, hipMemsetAsync(f->eplxx, 0, n * sizeof(double), s));                                 // @tensor_center(ε_pl) .= 0
    JRX_HIP(h, hipMemsetAsync(f->eplyy, 0, n * sizeof(double), s));
    JRX_HIP(h, hipMemsetAsync(f->eplxy_c, 0, n * sizeof(double), s));
    // linear laws: η of a cell / vertex depends on its phase ratios only -- averaged once per solve, compute_viscosity! then reads one array instead of the ratios
    const bool lin = !a.vfields;
    hipLaunchKernelGGL(k_phase_avg, dim3(lin && f->eta_v ? gv : gc), dim3(256), 0, s, Kc, Gc, a, rh->has_density != 0, lin ? eta_lin_c : (double *)nullptr,
                       lin && f->eta_v ? eta_lin_v : (double *)nullptr);
    JRX_LAUNCH_CHECK(h);
    if (lin) { a.eta_lin_c = eta_lin_c; a.eta_lin_v = f->eta_v ? eta_lin_v : nullptr; }
    const bool upd_rho = rh->has_density && !mat_density_is_constant(rh);       // update_ρg!: a no-op for constant densities
    const bool ubc = p->displacement_bcs != 0;
    if (ubc) {    // displacement2velocity!(stokes, dt, flow_bcs) (Stokes2D.jl:647): V = U * inv(dt)
        hipLaunchKernelGGL(k_scale3, dim3(256), dim3(256), 0, s, f->Vx, (const double *)f->Ux, (i64)(nx + 1) * (ny + 2), f->Vy,
                           (const double *)f->Uy, (i64)(nx + 2) * (ny + 1), (double *)nullptr, (const double *)nullptr, (i64)0, 1.0 / p->dt);
        JRX_LAUNCH_CHECK(h);
    }
    b.fs_dt = p->free_surface ? p->dt : 0.0;      // dt * free_surface with a Bool: Inf * false == 0.0 in Julia (solve! with dt = Inf)
    // option "fused2d_batch" (default): the forms of the pre and viscosity + velocity kernels that request every operand up front -- uniform grids, constant densities, viscosity laws
    // without fields (phase average precomputed), no free surface; everything else keeps the general kernels
    const bool batch_pre = h->fused2d_batch && !p->inv_spacing[0] && !upd_rho;
    const bool batch_vv = h->fused2d_batch && !p->inv_spacing[0] && b.fs_dt == 0.0 && !a.vfields && a.eta_lin_c != nullptr && (a.f.eta_v == nullptr || a.eta_lin_v != nullptr);

    double err_it1 = 1.0, err = 1.0;
    int64_t iter = 0, cont = 0;
    JRX_HIP(h, hipEventRecord(h->ev[6], s));
    // Runs of unobserved iterations replay as a captured graph of GIT iterations (three launches each: at the sizes where the loop is launch-bound -- 17 - 18 us
    // per iteration up to 256^2 -- the gap between dependent launches is shorter inside a graph).  An even count, so that the (τxx, τyy) sets end where they
    // started.  Only in the plain steady state: one rank, no periodic face, velocity boundary conditions, strain-rate form.  Option "loop_graphs" = 0: plain launches.
    constexpr int GIT = 32;
    GraphExecs gexec;        // released on every exit path
    bool graphs = h->loop_graphs && !comm && !ubc && !a.si && p->periodic == 0 && (i64)(nx + 1) * (ny + 1) <= 200000;
    while (iter <= p->iterMax) {
        if (p->iterMin < iter && ((err / err_it1) < p->eps_rel || err < p->eps_abs)) break;          // Stokes2D.jl:650-651
        if (graphs && iter >= 1 && !((err / err_it1) < p->eps_rel || err < p->eps_abs)) {
            // observed iterations (checks: multiples of nout; the last one: iterMax + 1) end a run; err does not change inside one
            int64_t nxt = ((iter / p->nout) + 1) * p->nout;
            if (nxt > p->iterMax + 1) nxt = p->iterMax + 1;
            int64_t run = nxt - 1 - iter;
            if (run >= GIT) {
                const int par = a.f.txx == f->txx ? 0 : 1;
                if (!gexec[par]) {
                    hipGraph_t gr = nullptr;
                    bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
                    if (ok) {
                        VepArgs aa = a;
                        aa.obs = h->vep_store_all;          // a run of unobserved iterations
                        Args2 bb = b;
                        for (int q = 0; q < GIT; q++) {
                            if (upd_rho) hipLaunchKernelGGL((k_vep_pre<true, true>), dim3(gv), dim3(256), 0, s, aa, theta);
                            else if (batch_pre) { if (aa.obs) hipLaunchKernelGGL(k_vep_pre_b<true>, dim3(gv), dim3(256), 0, s, aa, theta); else hipLaunchKernelGGL(k_vep_pre_b<false>, dim3(gv), dim3(256), 0, s, aa, theta); }
                            else hipLaunchKernelGGL(k_vep_pre<true>, dim3(gv), dim3(256), 0, s, aa, theta);
                            if (aa.soft) hipLaunchKernelGGL(k_vep_stress2d<true>, dim3(gv), dim3(256), 0, s, aa);
                            else switch (h->vep3_np_const ? aa.rh.nphase : 0) {
                            case 1: hipLaunchKernelGGL((k_vep_stress2d<false, false, 1>), dim3(gv), dim3(256), 0, s, aa); break;
                            case 2: hipLaunchKernelGGL((k_vep_stress2d<false, false, 2>), dim3(gv), dim3(256), 0, s, aa); break;
                            case 3: hipLaunchKernelGGL((k_vep_stress2d<false, false, 3>), dim3(gv), dim3(256), 0, s, aa); break;
                            case 4: hipLaunchKernelGGL((k_vep_stress2d<false, false, 4>), dim3(gv), dim3(256), 0, s, aa); break;
                            default: hipLaunchKernelGGL(k_vep_stress2d<false>, dim3(gv), dim3(256), 0, s, aa);
                            }
                            { double *t_ = aa.f.txx; aa.f.txx = aa.txx_out; aa.txx_out = t_; }
                            { double *t_ = aa.f.tyy; aa.f.tyy = aa.tyy_out; aa.tyy_out = t_; }
                            bb.f.txx = aa.f.txx; bb.f.tyy = aa.f.tyy;
                            if (batch_vv) hipLaunchKernelGGL(k_vep_visc_velocity_b<true>, dim3(gv), dim3(256), 0, s, aa, bb);
                            else hipLaunchKernelGGL(k_vep_visc_velocity<true>, dim3(gv), dim3(256), 0, s, aa, bb);
                        }
                        ok = hipStreamEndCapture(s, &gr) == hipSuccess && gr != nullptr;
                    }
                    if (ok) ok = hipGraphInstantiate(&gexec[par], gr, nullptr, nullptr, 0) == hipSuccess;
                    if (gr) (void)hipGraphDestroy(gr);
                    if (!ok) { (void)hipGetLastError(); gexec[par] = nullptr; graphs = false; }
                }
                if (gexec[par]) {
                    while (run >= GIT) {
                        JRX_HIP(h, hipGraphLaunch(gexec[par], s));
                        iter += GIT; run -= GIT;
                    }
                    continue;
                }
            }
        }
        {   // can the loop stop after the iteration launched now (a check, the last allowed one, or already converged)?  Only then are its output-only arrays stored
            const int64_t it1 = iter + 1;
            a.obs = ((it1 % p->nout == 0) && it1 > 1) || it1 > p->iterMax || (p->iterMin < it1 && ((err / err_it1) < p->eps_rel || err < p->eps_abs)) || h->vep_store_all;
        }
        if (comm) {
            hipLaunchKernelGGL(k_maxloc, dim3(gc, 1), dim3(256), 0, s, etatau, (const double *)f->eta, nx, ny, 1);
            JRX_LAUNCH_CHECK(h);
            // update_halo!(ητ) (Stokes2D.jl:655)
            double *arrs[1] = {etatau};
            const int64_t ext[1][3] = {{nx, ny, 1}};
            JRX_TRY(jrx_halo_exchange(h, s, 1, arrs, ext, nn));
            if (upd_rho) hipLaunchKernelGGL((k_vep_pre<false, true>), dim3(gv), dim3(256), 0, s, a, theta);
            else hipLaunchKernelGGL(k_vep_pre<false>, dim3(gv), dim3(256), 0, s, a, theta);
        } else if (upd_rho) hipLaunchKernelGGL((k_vep_pre<true, true>), dim3(gv), dim3(256), 0, s, a, theta);
        else if (batch_pre) { if (a.obs) hipLaunchKernelGGL(k_vep_pre_b<true>, dim3(gv), dim3(256), 0, s, a, theta); else hipLaunchKernelGGL(k_vep_pre_b<false>, dim3(gv), dim3(256), 0, s, a, theta); }
        else hipLaunchKernelGGL(k_vep_pre<true>, dim3(gv), dim3(256), 0, s, a, theta);      // compute_maxloc! folded in
        JRX_LAUNCH_CHECK(h);
        if (a.si) {
            hipLaunchKernelGGL(k_vep_strain_inc, dim3(gv), dim3(256), 0, s, a);
            JRX_LAUNCH_CHECK(h);
        }
        // update_stresses_center_vertex_ps!: vertex and centre halves in one launch; the new τxx, τyy go to the other set, then swap
        if (a.si) {
            if (a.soft) hipLaunchKernelGGL((k_vep_stress2d<true, true>), dim3(gv), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((k_vep_stress2d<false, true>), dim3(gv), dim3(256), 0, s, a);
        } else if (a.soft) hipLaunchKernelGGL(k_vep_stress2d<true>, dim3(gv), dim3(256), 0, s, a);
        else switch (h->vep3_np_const ? a.rh.nphase : 0) {
        case 1: hipLaunchKernelGGL((k_vep_stress2d<false, false, 1>), dim3(gv), dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((k_vep_stress2d<false, false, 2>), dim3(gv), dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL((k_vep_stress2d<false, false, 3>), dim3(gv), dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL((k_vep_stress2d<false, false, 4>), dim3(gv), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL(k_vep_stress2d<false>, dim3(gv), dim3(256), 0, s, a);
        }
        JRX_LAUNCH_CHECK(h);
        { double *t_ = a.f.txx; a.f.txx = a.txx_out; a.txx_out = t_; }
        { double *t_ = a.f.tyy; a.f.tyy = a.tyy_out; a.tyy_out = t_; }
        b.f.txx = a.f.txx; b.f.tyy = a.f.tyy; g.txx = a.f.txx; g.tyy = a.f.tyy;
        if (comm) {   // update_halo!(stokes.τ.xy) (Stokes2D.jl:757)
            double *arrs[1] = {f->txy};
            const int64_t ext[1][3] = {{nx + 1, ny + 1, 1}};
            JRX_TRY(jrx_halo_exchange(h, s, 1, arrs, ext, nn));
        }
        // flow_bcs! applied in full by iteration 1: refresh ghosts in-kernel (never with DisplacementBoundaryConditions: flow_bcs! then acts on U)
        // (nor with strain_increment: U = V dt must copy the ghosts of V as the previous flow_bcs! left them)
        const bool bcf = iter >= 1 && p->periodic == 0 && !comm && !ubc && !a.si;
        bool used_bcf = false;
        {
            const bool next_check = ((iter + 1) % p->nout == 0) && iter + 1 > 1;
            const bool next_last = next_check || iter + 1 > p->iterMax || (p->iterMin < iter + 1 && ((err / err_it1) < p->eps_rel || err < p->eps_abs));
            used_bcf = bcf && !next_last;
            // compute_viscosity! + compute_V! (free-surface form with dt*free_surface = 0) in one launch
            if (batch_vv && used_bcf) hipLaunchKernelGGL(k_vep_visc_velocity_b<true>, dim3(gv), dim3(256), 0, s, a, b);
            else if (batch_vv) hipLaunchKernelGGL(k_vep_visc_velocity_b<false>, dim3(gv), dim3(256), 0, s, a, b);
            else if (used_bcf) hipLaunchKernelGGL(k_vep_visc_velocity<true>, dim3(gv), dim3(256), 0, s, a, b);
            else hipLaunchKernelGGL(k_vep_visc_velocity<false>, dim3(gv), dim3(256), 0, s, a, b);
        }
        JRX_LAUNCH_CHECK(h);
        iter += 1;
        const bool check = (iter % p->nout == 0) && iter > 1;
        // the loop can stop after this iteration if it is a check, the last allowed one, or already converged
        const bool last = check || iter > p->iterMax || (p->iterMin < iter && ((err / err_it1) < p->eps_rel || err < p->eps_abs));
        if (last || a.si) {   // U = V*dt is only observable after an iteration the loop can stop at -- or every iteration when the strains are taken from U
            hipLaunchKernelGGL(k_scale3, dim3(256), dim3(256), 0, s, f->Ux, (const double *)f->Vx, (i64)(nx + 1) * (ny + 2), f->Uy,
                               (const double *)f->Vy, (i64)(nx + 2) * (ny + 1), (double *)nullptr, (const double *)nullptr, (i64)0, p->dt);
            JRX_LAUNCH_CHECK(h);
        }
        if (ubc) {    // flow_bcs!(stokes, ::DisplacementBoundaryConditions) acts on U = V dt, which the next iteration overwrites: only the last one is observable
            if (last || a.si) JRX_TRY(launch_bcs2(h, s, f->Ux, f->Uy, nx, ny, p->free_slip, p->no_slip, p->periodic));
        } else if (!used_bcf) JRX_TRY(launch_bcs2(h, s, f->Vx, f->Vy, nx, ny, p->free_slip, p->no_slip, p->periodic));
        if (comm) {   // update_halo!(@velocity(stokes)...) (Stokes2D.jl:784)
            double *arrs[2] = {f->Vx, f->Vy};
            const int64_t ext[2][3] = {{nx + 1, ny + 2, 1}, {nx + 2, ny + 1, 1}};
            JRX_TRY(jrx_halo_exchange(h, s, 2, arrs, ext, nn));
        }
        if (check) {
            hipLaunchKernelGGL(k_velocity2d<true>, dim3(gc), dim3(256), 0, s, b);  // compute_Res!
            JRX_LAUNCH_CHECK(h);
            JRX_TRY(launch_sumsq2(h, s, &g, &q));
            JRX_HIP(h, hipMemcpyAsync(h->h_sums, h->d_sums, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
            JRX_HIP(h, hipStreamSynchronize(s));
            double ss[3] = {h->h_sums[0], h->h_sums[1], h->h_sums[3]};
            JRX_TRY(jrx_allreduce_sum_host(h, ss, 3));                                   // norm_mpi (Stokes2D.jl:803-808)
            const double nRx = sqrt(ss[0]) / sqrt((double)((p->nxg - 2) * (p->nyg - 1)));
            const double nRy = sqrt(ss[1]) / sqrt((double)((p->nxg - 1) * (p->nyg - 2)));
            const double nDV = sqrt(ss[2]) / sqrt((double)(p->nxg * p->nyg));
            err = fmax(nRx, fmax(nRy, nDV));
            if (std::isnan(nRx) || std::isnan(nRy) || std::isnan(nDV)) err = NAN;
            if (cont < res->cap) {
                if (res->norm_Rx) res->norm_Rx[cont] = nRx;
                if (res->norm_Ry) res->norm_Ry[cont] = nRy;
                if (res->norm_divV) res->norm_divV[cont] = nDV;
                if (res->err_evo1) res->err_evo1[cont] = err;
                if (res->err_evo2) res->err_evo2[cont] = iter;
            }
            if (cont == 0) err_it1 = err;
            cont++;
            if (p->verbose && jrx_comm_rank(h) == 0)      // igg.me == 0 (Stokes2D.jl:814)
                printf("Total steps = %lld, abs_err = %1.3e , rel_err = %1.3e [norm_Rx=%1.3e, norm_Ry=%1.3e, norm_∇V=%1.3e] \n",
                       (long long)iter, err, err / err_it1, nRx, nRy, nDV);
            if (std::isnan(err)) {
                // error("NaN(s)"): leave the caller's arrays consistent (the current τxx, τyy may live in the second set) and the stream drained
                gexec.reset();
                if (a.f.txx != f->txx) {
                    (void)hipMemcpyAsync(f->txx, a.f.txx, n * sizeof(double), hipMemcpyDeviceToDevice, s);
                    (void)hipMemcpyAsync(f->tyy, a.f.tyy, n * sizeof(double), hipMemcpyDeviceToDevice, s);
                }
                (void)hipEventRecord(h->ev[7], s);
                (void)hipStreamSynchronize(s);
                float msn = 0.f;
                (void)hipEventElapsedTime(&msn, h->ev[6], h->ev[7]);
                res->iter = iter; res->nchecks = cont < res->cap ? cont : res->cap;
                res->time_s = msn * 1e-3; res->av_time_s = iter > 1 ? res->time_s / (double)(iter - 1) : res->time_s;
                return jrx_fail(h, JRX_ERR_NAN, "NaN(s)");
            }
        }
    }
    gexec.reset();
    JRX_HIP(h, hipEventRecord(h->ev[7], s));
    if (a.f.txx != f->txx) {      // odd number of swaps: leave τxx, τyy in the caller's arrays
        JRX_HIP(h, hipMemcpyAsync(f->txx, a.f.txx, n * sizeof(double), hipMemcpyDeviceToDevice, s));
        JRX_HIP(h, hipMemcpyAsync(f->tyy, a.f.tyy, n * sizeof(double), hipMemcpyDeviceToDevice, s));
        a.f.txx = f->txx; a.f.tyy = f->tyy; b.f.txx = f->txx; b.f.tyy = f->tyy; g.txx = f->txx; g.tyy = f->tyy;
    }
    a.txx_out = a.tyy_out = nullptr;
    hipLaunchKernelGGL(k_vep_epilogue, dim3(gv), dim3(256), 0, s, a);
    JRX_LAUNCH_CHECK(h);
    hipLaunchKernelGGL(k_copy6, dim3(256), dim3(256), 0, s, f->toxx, (const double *)f->txx, (i64)n, f->toyy, (const double *)f->tyy, (i64)n,
                       f->toxy, (const double *)f->txy, (i64)nv, f->toxy_c, (const double *)f->txy_c, (i64)n, (double *)nullptr,
                       (const double *)nullptr, (i64)0, (double *)nullptr, (const double *)nullptr, (i64)0);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(s));
    float ms = 0.f;
    JRX_HIP(h, hipEventElapsedTime(&ms, h->ev[6], h->ev[7]));
    res->iter = iter;
    res->nchecks = cont < res->cap ? cont : res->cap;
    res->time_s = ms * 1e-3;
    res->av_time_s = iter > 1 ? res->time_s / (double)(iter - 1) : res->time_s;
    return JRX_OK;
}


// solve!(stokes, pt_stokes, grid, flow_bcs, ρg, rheology::MaterialParams, args, dt, igg; kwargs) -- src/stokes/Stokes2D.jl:345-557: the
// single-phase visco-elasto-plastic driver, the caller of compute_τ_nonlinear! and center2vertex! (test/test_WENO5.jl:226-291).
// rheology = phase 0 of the table.  compute_P! takes η (not ητ) and updates stokes.P in place (:418-420); θ = P + K dt λ sinψ only
// replaces P after the loop (:523).  The first compute_maxloc! of an iteration (:413) is dead (ητ is recomputed at :437 before its
// only reader, compute_V!) and is not launched.
jrx_status jrx_stokes2d_nonlinear_solve(jrx_handle *h, const jrx_vep2d_fields *f, const jrx_rheology *rh, const jrx_vep2d_params *p,
                                        jrx_solve_result *res)
{
    if (!h) return JRX_ERR_ARG;
    if (!f || !rh || !p || !res) return jrx_fail(h, JRX_ERR_ARG, "null argument");
    JRX_TRY(jrx_check_device(h));
    if (p->nx < 3 || p->ny < 3) return jrx_fail(h, JRX_ERR_ARG, "2D Stokes needs at least 3 cells per dimension");
    if (p->nout < 1) return jrx_fail(h, JRX_ERR_ARG, "nout must be >= 1");
    if (rh->nphase < 1 || rh->nphase > JRX_MAXPHASE) return jrx_fail(h, JRX_ERR_ARG, "nphase must be in 1..%d", JRX_MAXPHASE);
    const void *req[] = {f->P, f->P0, f->divV, f->Q, f->Vx, f->Vy, f->Ux, f->Uy, f->exx, f->eyy, f->exy, f->eplxx, f->eplyy, f->eplxy, f->eplxy_c,
                         f->txx, f->tyy, f->txy, f->txy_c, f->tII, f->toxx, f->toyy, f->toxy, f->toxy_c, f->eta, f->eta_vep, f->EII_pl, f->evol_pl,
                         f->EVol_pl, f->fx, f->fy, f->RP, f->Rx, f->Ry};
    for (const void *q : req)
        if (!q) return jrx_fail(h, JRX_ERR_ARG, "a required field pointer is NULL");
    const bool comm = jrx_comm_active(h);
    const int nx = (int)p->nx, ny = (int)p->ny;
    const int64_t nn[3] = {nx, ny, 1};
    const size_t n = (size_t)nx * ny, nv = (size_t)(nx + 1) * (ny + 1);
    hipStream_t s = h->stream;
    JRX_TRY(jrx_ensure_etatau(h, 5 * n));
    double *etatau = h->etatau, *theta = etatau + n, *lam = theta + n, *Kc = lam + n, *Gc = Kc + n;
    VepArgs a = make_vep(f, rh, p);
    a.theta = f->P; a.etatau = f->eta; a.Kc = Kc; a.Gc = Gc; a.lam = lam;          // the view compute_P! works on: P in place, η instead of ητ
    jrx_stokes2d_fields g = view2d(f);
    jrx_stokes2d_params q;
    memset(&q, 0, sizeof(q));
    q.nx = nx; q.ny = ny; q.nxg = p->nxg; q.nyg = p->nyg; q._dx = p->_dx; q._dy = p->_dy; q.dt = p->dt; q.r = p->r;
    q.theta_dtau = p->theta_dtau; q.eta_dtau = p->eta_dtau; q.free_slip = p->free_slip; q.no_slip = p->no_slip; q.periodic = p->periodic;
    for (int d = 0; d < 6; d++) q.inv_spacing[d] = p->inv_spacing[d];
    Args2 b = make_args2(&g, etatau, &q);
    b.fs_dt = p->free_surface ? p->dt : 0.0;      // dt * free_surface with a Bool: Inf * false == 0.0 in Julia (solve! with dt = Inf)
    const unsigned gv = (unsigned)((nv + 255) / 256), gc = (unsigned)((n + 255) / 256);
    const bool tg = p->T_ghosted != 0, ubc = p->displacement_bcs != 0;
    const bool upd_rho = rh->has_density && rh->rho_kind[0] != 0;

    JRX_HIP(h, hipMemsetAsync(theta, 0, n * sizeof(double), s));                                    // θ = @zeros(ni...) :398
    JRX_HIP(h, hipMemsetAsync(lam, 0, n * sizeof(double), s));
    JRX_HIP(h, hipMemsetAsync(f->eplxx, 0, n * sizeof(double), s));                                 // @tensor_center(ε_pl) .= 0 :391-393
    JRX_HIP(h, hipMemsetAsync(f->eplyy, 0, n * sizeof(double), s));
    JRX_HIP(h, hipMemsetAsync(f->eplxy_c, 0, n * sizeof(double), s));
    hipLaunchKernelGGL(k_fill2, dim3(gc), dim3(256), 0, s, Kc, rh->Kb[0], Gc, rh->G[0], (i64)n);      // Kb = get_Kb(rheology); G = get_G(rheology)
    // compute_ρg!(ρg[end], rheology, args); compute_viscosity!(stokes, args, rheology, cutoff) :406-407
    {
        VepArgs a0 = a;
        a0.vtau = false;                    // compute_viscosity! is the εII form; the in-loop compute_viscosity_τII! the τII one
        hipLaunchKernelGGL(k_single_material, dim3(gc), dim3(256), 0, s, a0, 1.0, rh->has_density != 0, true, tg);
    }
    JRX_LAUNCH_CHECK(h);
    if (ubc) {    // displacement2velocity!(stokes, dt, flow_bcs) :410
        hipLaunchKernelGGL(k_scale3, dim3(256), dim3(256), 0, s, f->Vx, (const double *)f->Ux, (i64)(nx + 1) * (ny + 2), f->Vy,
                           (const double *)f->Uy, (i64)(nx + 2) * (ny + 1), (double *)nullptr, (const double *)nullptr, (i64)0, 1.0 / p->dt);
        JRX_LAUNCH_CHECK(h);
    }
    double err_it1 = 1.0, err = 1.0;
    int64_t iter = 0, cont = 0;
    res->iter = 0; res->nchecks = 0;
    JRX_HIP(h, hipEventRecord(h->ev[6], s));
    auto keep_going = [&](int64_t it) { return it < 2 || (((err / err_it1) > p->eps_rel && err > p->eps_abs) && it <= p->iterMax); };
    // Runs of unobserved iterations (ten short launches each as the reference orders them, every argument constant: the loop is launch-bound at any 2D size,
    // 38 us per iteration) replay as a captured graph of GIT iterations of six launches (center2vertex! in one pass, flow_bcs! folded into compute_V!); one rank,
    // velocity boundary conditions.  Option "loop_graphs" = 0: plain launches.
    constexpr int GIT = 16;
    GraphExecs gexecs;       // released on every exit path
    hipGraphExec_t &gexec = gexecs[0];
    bool graphs = h->loop_graphs && !comm && !ubc;
    auto unobserved_iteration = [&]() {
        hipLaunchKernelGGL(k_vep_pre<false>, dim3(gv), dim3(256), 0, s, a, f->P);
        hipLaunchKernelGGL(k_single_material, dim3(gc), dim3(256), 0, s, a, p->viscosity_relaxation, upd_rho, true, tg);
        hipLaunchKernelGGL(k_maxloc, dim3(gc, 1), dim3(256), 0, s, etatau, (const double *)f->eta, nx, ny, 1);
        hipLaunchKernelGGL(k_tau_nonlinear2d<false>, dim3(gc), dim3(256), 0, s, a, theta);
        hipLaunchKernelGGL(k_center2vertex2d, dim3(gv), dim3(256), 0, s, f->txy, (const double *)f->txy_c, nx, ny, 3);        // the three passes of center2vertex! in one
        if (p->periodic == 0) {      // flow_bcs! has been applied in full by now (iter >= 2): the velocity kernel refreshes the ghosts next to what it updates
            hipLaunchKernelGGL((k_velocity2d<false, true>), dim3(gc), dim3(256), 0, s, b);
            return JRX_OK;
        }
        hipLaunchKernelGGL(k_velocity2d<false>, dim3(gc), dim3(256), 0, s, b);
        return launch_bcs2(h, s, f->Vx, f->Vy, nx, ny, p->free_slip, p->no_slip, p->periodic);
    };
    while (keep_going(iter)) {
        if (graphs && iter >= 2) {
            int64_t nxt = ((iter / p->nout) + 1) * p->nout;        // observed: the multiples of nout and iteration iterMax + 1
            if (nxt > p->iterMax + 1) nxt = p->iterMax + 1;
            int64_t run = nxt - 1 - iter;
            if (run >= GIT) {
                if (!gexec) {
                    hipGraph_t gr = nullptr;
                    bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
                    if (ok) {
                        for (int q = 0; q < GIT && ok; q++) ok = unobserved_iteration() == JRX_OK;
                        ok = (hipStreamEndCapture(s, &gr) == hipSuccess && gr != nullptr) && ok;
                    }
                    if (ok) ok = hipGraphInstantiate(&gexec, gr, nullptr, nullptr, 0) == hipSuccess;
                    if (gr) (void)hipGraphDestroy(gr);
                    if (!ok) { (void)hipGetLastError(); gexec = nullptr; graphs = false; }
                }
                if (gexec) {
                    while (run >= GIT) {
                        JRX_HIP(h, hipGraphLaunch(gexec, s));
                        iter += GIT; run -= GIT;
                    }
                    continue;
                }
            }
        }
        const int64_t it1 = iter + 1;
        const bool check = (it1 % p->nout == 0) && it1 > 1;
        const bool diag = check || !keep_going(it1);      // U is only observable after such an iteration
        hipLaunchKernelGGL(k_vep_pre<false>, dim3(gv), dim3(256), 0, s, a, f->P);                    // compute_∇V!, compute_P!, compute_strain_rate!
        hipLaunchKernelGGL(k_single_material, dim3(gc), dim3(256), 0, s, a, p->viscosity_relaxation, upd_rho, true, tg);   // update_ρg!, compute_viscosity_τII!
        hipLaunchKernelGGL(k_maxloc, dim3(gc, 1), dim3(256), 0, s, etatau, (const double *)f->eta, nx, ny, 1);            // compute_maxloc!(ητ, η) :437
        JRX_LAUNCH_CHECK(h);
        if (comm) {
            double *arrs[1] = {etatau};
            const int64_t ext[1][3] = {{nx, ny, 1}};
            JRX_TRY(jrx_halo_exchange(h, s, 1, arrs, ext, nn));
        }
        hipLaunchKernelGGL(k_tau_nonlinear2d<false>, dim3(gc), dim3(256), 0, s, a, theta);           // compute_τ_nonlinear! :440-458
        hipLaunchKernelGGL(k_center2vertex2d, dim3(gv), dim3(256), 0, s, f->txy, (const double *)f->txy_c, nx, ny, 3);   // center2vertex! :459, its three passes in one
        JRX_LAUNCH_CHECK(h);
        if (comm) {   // update_halo!(stokes.τ.xy) :460
            double *arrs[1] = {f->txy};
            const int64_t ext[1][3] = {{nx + 1, ny + 1, 1}};
            JRX_TRY(jrx_halo_exchange(h, s, 1, arrs, ext, nn));
        }
        hipLaunchKernelGGL(k_velocity2d<false>, dim3(gc), dim3(256), 0, s, b);                       // compute_V! (free-surface form) :463-474
        JRX_LAUNCH_CHECK(h);
        if (diag) {
            hipLaunchKernelGGL(k_scale3, dim3(256), dim3(256), 0, s, f->Ux, (const double *)f->Vx, (i64)(nx + 1) * (ny + 2), f->Uy,
                               (const double *)f->Vy, (i64)(nx + 2) * (ny + 1), (double *)nullptr, (const double *)nullptr, (i64)0, p->dt);
            JRX_LAUNCH_CHECK(h);
        }
        if (!ubc) JRX_TRY(launch_bcs2(h, s, f->Vx, f->Vy, nx, ny, p->free_slip, p->no_slip, p->periodic));
        else if (diag) JRX_TRY(launch_bcs2(h, s, f->Ux, f->Uy, nx, ny, p->free_slip, p->no_slip, p->periodic));
        if (comm) {
            double *arrs[2] = {f->Vx, f->Vy};
            const int64_t ext[2][3] = {{nx + 1, ny + 2, 1}, {nx + 2, ny + 1, 1}};
            JRX_TRY(jrx_halo_exchange(h, s, 2, arrs, ext, nn));
        }
        iter = it1;
        if (check) {
            hipLaunchKernelGGL(k_velocity2d<true>, dim3(gc), dim3(256), 0, s, b);                    // compute_Res! :479-490
            JRX_LAUNCH_CHECK(h);
            JRX_TRY(launch_sumsq2(h, s, &g, &q));
            JRX_HIP(h, hipMemcpyAsync(h->h_sums, h->d_sums, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
            JRX_HIP(h, hipStreamSynchronize(s));
            double ss[3] = {h->h_sums[0], h->h_sums[1], h->h_sums[3]};
            JRX_TRY(jrx_allreduce_sum_host(h, ss, 3));
            const double nRx = sqrt(ss[0]) / sqrt((double)((p->nxg - 2) * (p->nyg - 1)));
            const double nRy = sqrt(ss[1]) / sqrt((double)((p->nxg - 1) * (p->nyg - 2)));
            const double nDV = sqrt(ss[2]) / sqrt((double)(p->nxg * p->nyg));
            err = fmax(nRx, fmax(nRy, nDV));
            if (std::isnan(nRx) || std::isnan(nRy) || std::isnan(nDV)) err = NAN;
            if (cont < res->cap) {
                if (res->norm_Rx) res->norm_Rx[cont] = nRx;
                if (res->norm_Ry) res->norm_Ry[cont] = nRy;
                if (res->norm_divV) res->norm_divV[cont] = nDV;
                if (res->err_evo1) res->err_evo1[cont] = err;
                if (res->err_evo2) res->err_evo2[cont] = iter;
            }
            if (cont == 0) err_it1 = err;
            cont++;
            if (jrx_comm_rank(h) == 0 && ((p->verbose && (err / err_it1) > p->eps_rel && err > p->eps_abs) || iter == p->iterMax))
                printf("Total steps = %lld, abs_err = %1.3e , rel_err = %1.3e [norm_Rx=%1.3e, norm_Ry=%1.3e, norm_∇V=%1.3e] \n",
                       (long long)iter, err, err / err_it1, nRx, nRy, nDV);
            if (std::isnan(err)) {
                res->iter = iter; res->nchecks = cont < res->cap ? cont : res->cap;
                (void)hipStreamSynchronize(s);
                return jrx_fail(h, JRX_ERR_NAN, "NaN(s)");
            }
        }
    }
    gexecs.reset();
    JRX_HIP(h, hipEventRecord(h->ev[7], s));
    JRX_HIP(h, hipMemcpyAsync(f->P, theta, n * sizeof(double), hipMemcpyDeviceToDevice, s));        // stokes.P .= θ :523
    a.txx_out = a.tyy_out = nullptr;
    hipLaunchKernelGGL(k_vep_epilogue, dim3(gv), dim3(256), 0, s, a);
    JRX_LAUNCH_CHECK(h);
    hipLaunchKernelGGL(k_copy6, dim3(256), dim3(256), 0, s, f->toxx, (const double *)f->txx, (i64)n, f->toyy, (const double *)f->tyy, (i64)n,
                       f->toxy, (const double *)f->txy, (i64)nv, f->toxy_c, (const double *)f->txy_c, (i64)n, (double *)nullptr,
                       (const double *)nullptr, (i64)0, (double *)nullptr, (const double *)nullptr, (i64)0);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(s));
    float ms = 0.f;
    JRX_HIP(h, hipEventElapsedTime(&ms, h->ev[6], h->ev[7]));
    res->iter = iter;
    res->nchecks = cont < res->cap ? cont : res->cap;
    res->time_s = ms * 1e-3;
    res->av_time_s = iter > 1 ? res->time_s / (double)(iter - 1) : res->time_s;
    return JRX_OK;
}

}   // extern "C"
