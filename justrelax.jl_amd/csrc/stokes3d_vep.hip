// stokes3d_vep.hip -- 3D multiphase visco-elasto-plastic pseudo-transient Stokes driver for gfx950.
//
// Reference being replaced (PTsolvers/JustRelax.jl): src/stokes/Stokes3D.jl:447-668 (driver),
// src/stokes/StressKernels.jl:604-989 (update_stresses_center_vertex_ps! 3D + clamped stencils),
// PressureKernels.jl:47-106 (compute_P! with phase ratios), rheology/Viscosity.jl:67-106,282-300
// (update_viscosity_τII!), rheology/StressUpdate.jl:146-176,435-550 (plastic parameters, yield function, gradients),
// stress_rotation_particles.jl:31-50 (vorticity), Interpolations.jl:314-323 (shear2center!),
// StressKernels.jl:394-431 (accumulate_tensor!, accumulate_vol!), as test/test_shearband3D_MPI.jl drives them.
// Rheology table as in the 2D driver (stokes2d.hip): per-phase LinearViscous η, ConstantElasticity (G, Kb),
// DruckerPrager_regularised (C, ϕ, ψ, η_vp); constant densities (ρg given).
//
// Per PT iteration: k_vep3_pre (∇V, θ, RP, ε(6)) -> k_vep3_visc -> 3 edge kernels (yz, xz, xy; new edge stresses to
// temporaries: every update reads last iteration's stresses, where the reference's single launch races) -> commit ->
// k_vep3_centre -> velocity sweep of the visco-elastic path (compute_V! is the same kernel) -> BCs.
// HBM-bound fp64 stencils with branches; no MFMA.
#include "jrx_internal.hpp"
#include "jrx_kernels.hpp"
#include "jrx_material.hpp"

namespace {

struct Vep3Args {
    jrx_vep3d_fields f;
    jrx_rheology rh;
    const double *etatau, *Kc, *Gc;
    const double *eta_lin = nullptr;   // linear laws: the phase average of η (constant over a solve), precomputed by k_vep3_phase_avg; nullptr: computed per call
    double *theta, *lam;
    double *lamv[3], *tnew[3];
    double *cnew[3] = {nullptr, nullptr, nullptr};   // centre pass: where the new τxx, τyy, τzz go (nullptr: in place).  A second set lets the centre pass run BESIDE the edge pass, which averages the OLD normal stresses
    double _dx, _dy, _dz, dt, r, theta_dtau, rel, nu, cut_lo, cut_hi;
    int nx, ny, nz;
    bool soft;            // some phase has a softening law: the yield function then reads EII_pl
    bool nt;              // outputs with non-temporal stores
    bool tg;              // args.T is the ghosted thermal.T (ni .+ 2): densities read it at the cell's own [i, j, k], unshifted
    bool obs;             // the outputs nothing inside the PT loop reads -- ∇V, RP, ε_pl (6), ε_vol_pl, η_vep, τII (update_viscosity_τII! takes its invariant from the stress arrays) -- are stored; the solve loop clears it on iterations whose
                          // results cannot be observed (not a norm check, not the last one): the next iteration overwrites them anyway (11 of 29 written passes)
};

// node (i, j, k) of an (n1, n2, n3) box: xy flattened over blockIdx.x (no nearly empty blocks when n1 = nx + 1), k = blockIdx.y
#define NODE_IJK(n1_, n2_)                                              \
    const int t_ = blockIdx.x * blockDim.x + threadIdx.x;               \
    const int j = t_ / (n1_), i = t_ - j * (n1_), k = blockIdx.y;       \
    if (j >= (n2_)) return;
#define GRID_IJK(n1_, n2_, n3_) dim3((unsigned)(((i64)(n1_) * (n2_) + 255) / 256), (unsigned)(n3_))
// the same box, XCD-aware: blocks are dealt round-robin to the 8 XCDs (each with its own L2), so block L of the launch is mapped to position
// (L % 8) * (T / 8) + L / 8 of the (flattened xy, z) sequence -- every XCD then works on one contiguous slab of planes and finds the rows j +- 1 and
// planes k +- 1 of its stencils in its own L2 (measured on k_vep3_pre at 256^3: 26 array passes fetched from HBM without, ~11 with)
#define NODE_IJK_XS(n1_, n2_)                                                         \
    unsigned bx_ = blockIdx.x, by_ = blockIdx.y;                                      \
    {                                                                                 \
        const unsigned L_ = by_ * gridDim.x + bx_, per_ = (gridDim.x * gridDim.y) / 8u; \
        if (L_ < per_ * 8u) { const unsigned Ln_ = (L_ & 7u) * per_ + (L_ >> 3); bx_ = Ln_ % gridDim.x; by_ = Ln_ / gridDim.x; } \
    }                                                                                 \
    const int t_ = bx_ * blockDim.x + threadIdx.x;                                    \
    const int j = t_ / (n1_), i = t_ - j * (n1_), k = by_;                            \
    if (j >= (n2_)) return;
// block = 64 consecutive nodes of the flattened xy plane x 4 consecutive planes (one plane per wave): the k-1 / k+1 operands of the
// gathering kernels are the neighbouring waves' k operands and meet in the CU's L1 (JRX_VEP_MAP=0 keeps the one-plane blocks)
#define NODE_IJK4(n1_, n2_, n3_)                                                     \
    const int t_ = blockIdx.x * 64 + (threadIdx.x & 63);                             \
    const int j = t_ / (n1_), i = t_ - j * (n1_), k = blockIdx.y * 4 + (threadIdx.x >> 6); \
    if (j >= (n2_) || k >= (n3_)) return;
#define GRID_IJK4(n1_, n2_, n3_) dim3((unsigned)(((i64)(n1_) * (n2_) + 63) / 64), (unsigned)(((n3_) + 3) / 4))

__device__ __forceinline__ int clampi3(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ double sinv3(const double t[6])
{
    return sqrt(0.5 * (t[0] * t[0] + t[1] * t[1] + t[2] * t[2]) + t[3] * t[3] + t[4] * t[4] + t[5] * t[5]);
}
__device__ __forceinline__ double ratio_avg3(const double *val, const double *r, int n)
{   // fn_ratio, src/phases/phases.jl:6-15
    double x = 0.0;
#pragma unroll
    for (int q = 0; q < n; q++) x += (r[q] == 0.0) ? 0.0 : val[q] * r[q];
    return x;
}
// NP > 0: the number of phases as a compile-time constant (the phase loops unroll), else rh.nphase
template <int NP = 0>
__device__ __forceinline__ void plastic_params3(const jrx_rheology &rh, const double *r, bool &is_pl, double &eta_reg)
{   // plastic_params_phase, rheology/StressUpdate.jl:152-176
    is_pl = false; eta_reg = 0.0;
    const int np = NP > 0 ? NP : rh.nphase;
#pragma unroll
    for (int q = 0; q < np; q++)
        if (rh.is_pl[q]) { is_pl = true; eta_reg += rh.eta_vp[q] * r[q]; }
}
// SOFT: some phase has a softening law (compiled out otherwise: the erfc / sincos paths cost the edge kernel its second wave per SIMD)
template <bool SOFT, int NP = 0>
__device__ __forceinline__ double yield_F3(const jrx_rheology &rh, const double *r, double P, double tII, double EII)
{   // compute_yieldfunction_phase, StressUpdate.jl:435-452 ; DP: F = τII - cosϕ(EII) C(EII) - sinϕ(EII) P (softening at the EII keyword)
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
__device__ __forceinline__ void plastic_grad3(const jrx_rheology &rh, const double *r, const double t[6], double dQdt[6], double &dQdP, double &dFdP)
{   // compute_plastic_gradients_phase, StressUpdate.jl:463-550 (shear slots halved once, :466-472)
#pragma unroll
    for (int q = 0; q < 6; q++) dQdt[q] = 0.0;
    dQdP = 0.0; dFdP = 0.0;
    const double tII = sinv3(t);
    const int np = NP > 0 ? NP : rh.nphase;
    // ∂Q/∂τ of a Drucker-Prager phase does not depend on the phase: one division per component instead of one per component and phase (the same quotient, so the same bits)
    bool any_pl = false;
#pragma unroll
    for (int q = 0; q < np; q++) any_pl |= rh.is_pl[q] != 0;
    double g[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (any_pl) {
#pragma unroll
        for (int s = 0; s < 3; s++) g[s] = 0.5 * t[s] / tII;
#pragma unroll
        for (int s = 3; s < 6; s++) g[s] = 0.5 * (t[s] / tII);
    }
#pragma unroll
    for (int q = 0; q < np; q++) {
        if (r[q] == 0.0 || !rh.is_pl[q]) continue;
#pragma unroll
        for (int s = 0; s < 6; s++) dQdt[s] = fma(r[q], g[s], dQdt[s]);
        dQdP = fma(r[q], -rh.sinpsi[q], dQdP);
        dFdP = fma(r[q], -rh.sinphi[q], dFdP);
    }
}

// store of an output array that no thread of this launch reads back: non-temporal when a.nt (the lines do not displace the operands the neighbouring
// blocks are about to re-read from L2; tuning switch "vep3_nt")
#define VST(a_, ptr_, val_) do { double *q_ = &(ptr_); const double v_ = (val_); if ((a_).nt) __builtin_nontemporal_store(v_, q_); else *q_ = v_; } while (0)
#define C3(A, i, j, k) (A)[(i) + (i64)nx * ((j) + (i64)ny * (k))]
#define EYZ(A, i, j, k) (A)[(i) + (i64)nx * ((j) + (i64)(ny + 1) * (k))]
#define EXZ(A, i, j, k) (A)[(i) + (i64)(nx + 1) * ((j) + (i64)ny * (k))]
#define EXY(A, i, j, k) (A)[(i) + (i64)(nx + 1) * ((j) + (i64)(ny + 1) * (k))]

// compute_∇V!, compute_P! (phase form: K, G per cell, η = ητ, P = θ) and compute_strain_rate! 3D over the ni.+1 box
// (VelocityKernels.jl:3-6,59-104; PressureKernels.jl:47-106,186-195)
// ML: compute_maxloc!(ητ, η) of the own cell first (clamped 3 x 3 x 3 window, the comparison order of k_maxloc) and store it
// RHO: update_ρg! of the own cell (phase-ratio density at args.T, args.P = stokes.P, times the scalar gravity, into ρg_z)
// A thread walks KZ planes of its (i, j) column (blockIdx.y = z chunk): the 3 x 3 x 3 maximum is then the maximum of three 3 x 3 plane maxima, two of
// which are carried from the previous planes -- 9 loads of η per cell instead of 27 (max is order-independent: same result).
template <bool ML, bool RHO = false, int KZ = 8>
__global__ __launch_bounds__(256) void k_vep3_pre(const Vep3Args a)
{
    const int nx = a.nx, ny = a.ny, nz = a.nz;
    unsigned bx_ = blockIdx.x, by_ = blockIdx.y;
    {   // XCD slab order of the (flattened xy, z chunk) block sequence, see NODE_IJK_XS
        const unsigned L_ = by_ * gridDim.x + bx_, per_ = (gridDim.x * gridDim.y) / 8u;
        if (L_ < per_ * 8u) { const unsigned Ln_ = (L_ & 7u) * per_ + (L_ >> 3); bx_ = Ln_ % gridDim.x; by_ = Ln_ / gridDim.x; }
    }
    const int t_ = bx_ * blockDim.x + threadIdx.x;
    const int j = t_ / (nx + 1), i = t_ - j * (nx + 1);
    if (j >= ny + 1) return;
    const int k0 = (int)by_ * KZ, k1 = min(k0 + KZ, nz + 1);
    const double *__restrict__ Vx = a.f.Vx, *__restrict__ Vy = a.f.Vy, *__restrict__ Vz = a.f.Vz;
    const double _dx = a._dx, _dy = a._dy, _dz = a._dz;
#define VX(i_, j_, k_) Vx[(i_) + (i64)(nx + 1) * ((j_) + (i64)(ny + 2) * (k_))]
#define VY(i_, j_, k_) Vy[(i_) + (i64)(nx + 2) * ((j_) + (i64)(ny + 1) * (k_))]
#define VZ(i_, j_, k_) Vz[(i_) + (i64)(nx + 2) * ((j_) + (i64)(ny + 2) * (k_))]
    const bool cellcol = i < nx && j < ny;
    const int il = clampi3(i - 1, 0, nx - 1), ic = clampi3(i, 0, nx - 1), ir = clampi3(i + 1, 0, nx - 1);
    const int jl = clampi3(j - 1, 0, ny - 1), jc = clampi3(j, 0, ny - 1), jr = clampi3(j + 1, 0, ny - 1);
    auto plane_max = [&](int kk) {          // clamped 3 x 3 maximum of η in plane clamp(kk), the comparison order of k_maxloc within the plane
        const i64 pk = (i64)nx * ny * clampi3(kk, 0, nz - 1);
        double m = -INFINITY;
        const int js[3] = {jl, jc, jr}, is[3] = {il, ic, ir};
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const double v = a.f.eta[is[r] + (i64)nx * js[q] + pk];
                if (v > m) m = v;
            }
        return m;
    };
    double m_prev = -INFINITY, m_cur = -INFINITY, m_next = -INFINITY;
    if (ML && cellcol) { m_prev = plane_max(k0 - 1); m_cur = plane_max(k0); }
#pragma unroll 1
    for (int k = k0; k < k1; k++) {
        if (cellcol && k < nz) {
            const i64 c = i + (i64)nx * (j + (i64)ny * k);
            const double dxi = (-VX(i, j + 1, k + 1) + VX(i + 1, j + 1, k + 1)) * _dx;
            const double dyi = (-VY(i + 1, j, k + 1) + VY(i + 1, j + 1, k + 1)) * _dy;
            const double dzi = (-VZ(i + 1, j + 1, k) + VZ(i + 1, j + 1, k + 1)) * _dz;
            const double divV = dxi + dyi + dzi;
            if (a.obs) VST(a, a.f.divV[c], divV);
            const double _Kdt = 1.0 / (a.Kc[c] * a.dt), _Gdt = 1.0 / (a.Gc[c] * a.dt), _dt = 1.0 / a.dt;
            const double P = a.theta[c], P0 = a.f.P0[c];
            const double rhs = -divV + (a.f.Q[c] * _dt);
            if (a.obs) VST(a, a.f.RP[c], fma(-(P - P0), _Kdt, rhs));
            double et;
            if (ML) {
                m_next = plane_max(k + 1);
                et = m_prev;
                if (m_cur > et) et = m_cur;
                if (m_next > et) et = m_next;
                m_prev = m_cur; m_cur = m_next;
                const_cast<double *>(a.etatau)[c] = et;
            } else et = a.etatau[c];
            const double psi = 1.0 / (1.0 / et + _Gdt) * a.r / a.theta_dtau;
            a.theta[c] = (fma(P0, _Kdt, rhs) * psi + P) / (1.0 + _Kdt * psi);
            const double d3 = divV * (1.0 / 3.0);
            VST(a, a.f.exx[c], dxi - d3);
            VST(a, a.f.eyy[c], dyi - d3);
            VST(a, a.f.ezz[c], dzi - d3);
            if (RHO) a.f.fz[c] = mat_density_ratio(a.rh, a.f.phase_c + (i64)a.rh.nphase * c,
                                                   !a.f.T ? 0.0 : (a.tg ? a.f.T[i + (i64)(nx + 2) * (j + (i64)(ny + 2) * k)] : a.f.T[c]), a.f.P[c]) * a.rh.gravity;
        }
        if (i < nx) VST(a, EYZ(a.f.eyz, i, j, k), 0.5 * (_dz * (VY(i + 1, j, k + 1) - VY(i + 1, j, k)) + _dy * (VZ(i + 1, j + 1, k) - VZ(i + 1, j, k))));
        if (j < ny) VST(a, EXZ(a.f.exz, i, j, k), 0.5 * (_dz * (VX(i, j + 1, k + 1) - VX(i, j + 1, k)) + _dx * (VZ(i + 1, j + 1, k) - VZ(i, j + 1, k))));
        if (k < nz) VST(a, EXY(a.f.exy, i, j, k), 0.5 * (_dy * (VX(i, j + 1, k + 1) - VX(i, j, k + 1)) + _dx * (VY(i + 1, j, k + 1) - VY(i, j, k + 1))));
    }
}

// update_viscosity_τII! / compute_viscosity! for the table rheology (rheology/Viscosity.jl:282-300, 599-625)
// creep laws that read fields (Viscosity.jl:455-503): invariant of @stress / @strain with the shear components gathered from the cell's edges, eps() on the
// normal components when those vanish; args at the cell, T at I .+ 1 of the ghosted thermal.T (local_viscosity_args :513-523)
__device__ __forceinline__ double vep3_visc_fields(const Vep3Args &a, const i64 c, const bool tau)
{
    const int nx = a.nx, ny = a.ny;
    const int k = (int)(c / ((i64)nx * ny)), j = (int)((c - (i64)k * nx * ny) / nx), i = (int)(c - (i64)nx * (j + (i64)ny * k));
    double AII = 0.0;
    if (mat_viscosity_reads_invariant(&a.rh)) {
        const double *xx = tau ? a.f.txx : a.f.exx, *yy = tau ? a.f.tyy : a.f.eyy, *zz = tau ? a.f.tzz : a.f.ezz;
        const double *yz = tau ? a.f.tyz : a.f.eyz, *xz = tau ? a.f.txz : a.f.exz, *xy = tau ? a.f.txy : a.f.exy;
        const double a0 = (xx[c] == 0.0 && yy[c] == 0.0 && zz[c] == 0.0) ? 2.220446049250313e-16 : 0.0;
        const double x = xx[c] + a0, y = yy[c] + -a0 * 0.5, z = zz[c] + -a0 * 0.5;
        const double p0 = EYZ(yz, i, j, k), p1 = EYZ(yz, i, j + 1, k), p2 = EYZ(yz, i, j, k + 1), p3 = EYZ(yz, i, j + 1, k + 1);
        const double q0 = EXZ(xz, i, j, k), q1 = EXZ(xz, i + 1, j, k), q2 = EXZ(xz, i, j, k + 1), q3 = EXZ(xz, i + 1, j, k + 1);
        const double r0 = EXY(xy, i, j, k), r1 = EXY(xy, i + 1, j, k), r2 = EXY(xy, i, j + 1, k), r3 = EXY(xy, i + 1, j + 1, k);
        AII = sqrt(0.5 * (x * x + y * y + z * z) + 0.25 * (p0 * p0 + p1 * p1 + p2 * p2 + p3 * p3) + 0.25 * (q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3) +
                   0.25 * (r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3));
    }
    const double T = !a.f.T ? 0.0 : (a.tg ? a.f.T[(i + 1) + (i64)(nx + 2) * ((j + 1) + (i64)(ny + 2) * (k + 1))] : a.f.T[c]);
    return mat_phase_viscosity(a.rh, a.f.phase_c + (i64)a.rh.nphase * c, AII, T, a.f.P[c], tau);
}
// compute_viscosity of the linear laws: a pure cell takes its phase's η, a mixed one the harmonic phase average (Viscosity.jl:282-300 with constant η per phase)
__device__ __forceinline__ double vep3_eta_linear(const Vep3Args &a, const i64 c)
{
    const int np = a.rh.nphase;
    const double *r = a.f.phase_c + np * c;
    for (int q = 0; q < np; q++)
        if (r[q] > 0.999) return a.rh.eta[q];
    double s = 0.0;
    for (int q = 0; q < np; q++)
        if (r[q] != 0.0) s += (1.0 / a.rh.eta[q]) * r[q];
    return 1.0 / s;
}
template <bool FIELDS, bool TAU>
__global__ __launch_bounds__(256) void k_vep3_visc(const Vep3Args a, double nu)
{
    const i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (i64)a.nx * a.ny * a.nz) return;
    if (FIELDS) {
        const double e = vep3_visc_fields(a, c, TAU) * nu + a.f.eta[c] * (1.0 - nu);
        a.f.eta[c] = fmin(fmax(e, a.cut_lo), a.cut_hi);
        return;
    }
    double e = a.eta_lin ? a.eta_lin[c] : vep3_eta_linear(a, c);
    e = e * nu + a.f.eta[c] * (1.0 - nu);
    a.f.eta[c] = fmin(fmax(e, a.cut_lo), a.cut_hi);
}

// rho: also compute_ρg!(ρg, phase_ratios, rheology, args) (Stokes3D.jl:505)
__global__ __launch_bounds__(256) void k_vep3_phase_avg(double *__restrict__ Kc, double *__restrict__ Gc, const Vep3Args a, const bool rho, double *__restrict__ eta_lin = nullptr)
{
    const i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (i64)a.nx * a.ny * a.nz) return;
    if (eta_lin) eta_lin[c] = vep3_eta_linear(a, c);
    const double *r = a.f.phase_c + (i64)a.rh.nphase * c;
    Kc[c] = ratio_avg3(a.rh.Kb, r, a.rh.nphase);
    Gc[c] = ratio_avg3(a.rh.G, r, a.rh.nphase);
    if (rho) {
        const int k = (int)(c / ((i64)a.nx * a.ny)), j = (int)((c - (i64)k * a.nx * a.ny) / a.nx), i = (int)(c - (i64)k * a.nx * a.ny - (i64)j * a.nx);
        const double T = !a.f.T ? 0.0 : (a.tg ? a.f.T[i + (i64)(a.nx + 2) * (j + (i64)(a.ny + 2) * k)] : a.f.T[c]);
        a.f.fz[c] = mat_density_ratio(a.rh, r, T, a.f.P[c]) * a.rh.gravity;
    }
}

// Stencil tables of StressKernels.jl:604-668 for the edge families T = 0 (yz), 1 (xz), 2 (xy): entries pick the
// clamped index {0: n-1, 1: n, 2: n+1} per direction, in the reference's order of summation.  constexpr functions so that the
// unrolled loops fold every table entry into the address arithmetic.
__host__ __device__ constexpr int cen3(int t, int q, int d)
{
    constexpr int T[3][4][3] = {
        {{1, 0, 0}, {1, 1, 0}, {1, 0, 1}, {1, 1, 1}},
        {{0, 1, 0}, {1, 1, 0}, {0, 1, 1}, {1, 1, 1}},
        {{0, 0, 1}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}}};
    return T[t][q][d];
}
__host__ __device__ constexpr int oth3(int t, int s, int q, int d)
{
    constexpr int T[3][3][4][3] = {
        {{{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, {{1, 0, 1}, {2, 0, 1}, {1, 1, 1}, {2, 1, 1}}, {{1, 1, 0}, {2, 1, 0}, {1, 1, 1}, {2, 1, 1}}},
        {{{0, 1, 1}, {1, 1, 1}, {1, 2, 1}, {0, 2, 1}}, {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, {{1, 1, 0}, {1, 2, 0}, {1, 1, 1}, {1, 2, 1}}},
        {{{0, 1, 1}, {1, 1, 1}, {0, 1, 2}, {1, 1, 2}}, {{1, 0, 1}, {1, 1, 1}, {1, 0, 2}, {1, 1, 2}}, {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}}}};
    return T[t][s][q][d];
}

// return mapping of one edge node of family T (StressKernels.jl:724-779 and the two sibling blocks), in two steps so that the z-marching
// kernel can reduce its operands to the trial stress early: edge_mat (phase-ratio material values of the node), then vep3_edge_plastic on
// the trial stress tt = τ + dτ (order xx, yy, zz, yz, xz, xy), the family's own component before the increment (tij_own) and its
// increment (d_own), etav the harmonic viscosity, Pv the averaged θ, EIIv the averaged EII_pl (softening laws only)
struct EdgeMat { double _Gdt, Kv, eta_reg; bool is_pl; };
template <int NP = 0>
__device__ __forceinline__ EdgeMat edge_mat(const Vep3Args &a, const double *rv)
{
    EdgeMat m;
    const int np = NP > 0 ? NP : a.rh.nphase;
    plastic_params3<NP>(a.rh, rv, m.is_pl, m.eta_reg);
    m._Gdt = 1.0 / (ratio_avg3(a.rh.G, rv, np) * a.dt);
    m.Kv = ratio_avg3(a.rh.Kb, rv, np);
    return m;
}
// PRE: lam_pre is λv of the node, fetched by the caller ahead of the return mapping
template <int T, bool SOFT, int NP = 0, bool PRE = false>
__device__ __forceinline__ void vep3_edge_plastic(const Vep3Args &a, i64 v, const double *rv, const EdgeMat &m, const double tt[6], double tij_own, double d_own,
                                                  double etav, double Pv, double dtr, double EIIv, double lam_pre = 0.0)
{
    double *const eplsh[3] = {a.f.eplyz, a.f.eplxz, a.f.eplxy};
    const double tIIv = sinv3(tt);
    double dQdt[6], dQdP, dFdP;
    plastic_grad3<NP>(a.rh, rv, tt, dQdt, dQdP, dFdP);
    const double vol = isinf(m.Kv) ? 0.0 : m.Kv * a.dt * dFdP * dQdP;
    const double F = yield_F3<SOFT, NP>(a.rh, rv, Pv, tIIv, SOFT ? EIIv : 0.0);
    constexpr int own = 3 + T;
    if (m.is_pl && tIIv != 0.0 && F > 0) {
        const double l = (1.0 - a.rel) * (PRE ? lam_pre : a.lamv[T][v]) + a.rel * (fmax(F, 0.0) / (etav * dtr + m.eta_reg + vol));
        a.lamv[T][v] = l;
        const double epl = l * dQdt[own];
        VST(a, a.tnew[T][v], tij_own + fma(-(2.0 * etav * epl), dtr, d_own));
        if (a.obs) VST(a, eplsh[T][v], epl);
    } else {
        VST(a, a.tnew[T][v], tij_own + d_own);
        if (a.obs) VST(a, eplsh[T][v], 0.0);
    }
}
template <int T, bool SOFT, int NP = 0>
__device__ __forceinline__ void vep3_edge_finish(const Vep3Args &a, i64 v, const double eij[6], const double tij[6], const double toij[6], double etav, double Pv,
                                                 double EIIv)
{
    const double *const phsh[3] = {a.f.phase_yz, a.f.phase_xz, a.f.phase_xy};
    double rvv[NP > 0 ? NP : 1];
    if (NP > 0) {
#pragma unroll
        for (int q = 0; q < NP; q++) rvv[q] = phsh[T][(i64)NP * v + q];
    }
    const double *rv = NP > 0 ? rvv : phsh[T] + (i64)a.rh.nphase * v;
    const EdgeMat m = edge_mat<NP>(a, rv);
    const double dtr = 1.0 / (a.theta_dtau + etav * m._Gdt + 1.0);
    double d[6], tt[6];
#pragma unroll
    for (int s = 0; s < 6; s++) { d[s] = dev_stress_inc(tij[s], toij[s], etav, eij[s], m._Gdt, dtr); tt[s] = tij[s] + d[s]; }
    vep3_edge_plastic<T, SOFT, NP>(a, v, rv, m, tt, tij[3 + T], d[3 + T], etav, Pv, dtr, EIIv);
}

// update_stresses_center_vertex_ps! 3D -- one edge family at node (i, j, k) (StressKernels.jl:707-903).
// cen[s][T]: the clamped 4-cell averages of the normal components (s = 0..2: ε, 3..5: τ, 6..8: τ_o) for family T, etav / Pv: harmonic η
// and average θ, gathered once per node for the three families (vep3_gather_centres)
struct CenAvg {
    double v[9][3], etav[3], Pv[3], EIIv[3];      // EIIv: av_clamped_yz/xz/xy(EII_pl) (StressKernels.jl:710,783,854), only gathered for softening laws
};
template <int T, bool SOFT, int NP = 0>
__device__ __forceinline__ void vep3_edge_body(const Vep3Args &a, int i, int j, int k, const int ci[3], const int cj[3], const int ck[3], const CenAvg &C)
{
    const int nx = a.nx, ny = a.ny, nz = a.nz;
    const int n1 = nx + (T != 0), n2 = ny + (T != 1), n3 = nz + (T != 2);
    if (i >= n1 || j >= n2 || k >= n3) return;
    typedef unsigned int u32;
#define LB(p, off) (*(const double *)((const char *)(p) + (off)))
    const double etav = C.etav[T], Pv = C.Pv[T];
    const u32 vb = 8u * (u32)(i + n1 * (j + n2 * k));
    const i64 v = i + (i64)n1 * (j + (i64)n2 * k);
    double *const tsh[3] = {a.f.tyz, a.f.txz, a.f.txy};
    const double *const tosh[3] = {a.f.toyz, a.f.toxz, a.f.toxy};
    const double *const esh[3] = {a.f.eyz, a.f.exz, a.f.exy};
    double eij[6], tij[6], toij[6];
#pragma unroll
    for (int s = 0; s < 3; s++) { eij[s] = C.v[s][T]; tij[s] = C.v[3 + s][T]; toij[s] = C.v[6 + s][T]; }
#pragma unroll
    for (int s = 0; s < 3; s++) {
        if (s == T) { eij[3 + s] = LB(esh[s], vb); tij[3 + s] = LB(tsh[s], vb); toij[3 + s] = LB(tosh[s], vb); continue; }
        const int m1 = nx + (s != 0), m2 = ny + (s != 1);
        u32 o[4];
#pragma unroll
        for (int q = 0; q < 4; q++) o[q] = 8u * (u32)(ci[oth3(T, s, q, 0)] + m1 * (cj[oth3(T, s, q, 1)] + m2 * ck[oth3(T, s, q, 2)]));
        eij[3 + s] = 0.25 * (LB(esh[s], o[0]) + LB(esh[s], o[1]) + LB(esh[s], o[2]) + LB(esh[s], o[3]));
        tij[3 + s] = 0.25 * (LB(tsh[s], o[0]) + LB(tsh[s], o[1]) + LB(tsh[s], o[2]) + LB(tsh[s], o[3]));
        toij[3 + s] = 0.25 * (LB(tosh[s], o[0]) + LB(tosh[s], o[1]) + LB(tosh[s], o[2]) + LB(tosh[s], o[3]));
    }
#undef LB
    vep3_edge_finish<T, SOFT, NP>(a, v, eij, tij, toij, etav, Pv, C.EIIv[T]);
}

// The clamped centre stencils of the three families lie in the 2 x 2 x 2 cube of cells below the node (cen3: bit = 1 own index,
// 0 index - 1) and share 7 of its 8 cells: every centre array is read 7 times per node instead of 12, in the reference's order of
// summation per family.
template <bool SOFT>
__device__ __forceinline__ void vep3_gather_centres(const Vep3Args &a, const int ci[3], const int cj[3], const int ck[3], CenAvg &C)
{
    typedef unsigned int u32;
    const int nx = a.nx, ny = a.ny;
    u32 cb[8];
#pragma unroll
    for (int b = 1; b < 8; b++) cb[b] = 8u * (u32)(ci[b & 1] + nx * (cj[(b >> 1) & 1] + ny * ck[(b >> 2) & 1]));
#define LB(p, off) (*(const double *)((const char *)(p) + (off)))
#define CIDX(T, q) (cen3(T, q, 0) + 2 * cen3(T, q, 1) + 4 * cen3(T, q, 2))
    const double *const arr[9] = {a.f.exx, a.f.eyy, a.f.ezz, a.f.txx, a.f.tyy, a.f.tzz, a.f.toxx, a.f.toyy, a.f.tozz};
#pragma unroll
    for (int s = 0; s < 9; s++) {
        double v[8];
#pragma unroll
        for (int b = 1; b < 8; b++) v[b] = LB(arr[s], cb[b]);
#pragma unroll
        for (int T = 0; T < 3; T++) C.v[s][T] = 0.25 * (v[CIDX(T, 0)] + v[CIDX(T, 1)] + v[CIDX(T, 2)] + v[CIDX(T, 3)]);
    }
    {
        double v[8];
#pragma unroll
        for (int b = 1; b < 8; b++) v[b] = LB(a.theta, cb[b]);
#pragma unroll
        for (int T = 0; T < 3; T++) C.Pv[T] = 0.25 * (v[CIDX(T, 0)] + v[CIDX(T, 1)] + v[CIDX(T, 2)] + v[CIDX(T, 3)]);
#pragma unroll
        for (int b = 1; b < 8; b++) v[b] = 1 / LB(a.f.eta, cb[b]);
#pragma unroll
        for (int T = 0; T < 3; T++) C.etav[T] = 4 / (v[CIDX(T, 0)] + v[CIDX(T, 1)] + v[CIDX(T, 2)] + v[CIDX(T, 3)]);
        C.EIIv[0] = C.EIIv[1] = C.EIIv[2] = 0.0;
        if (SOFT) {
#pragma unroll
            for (int b = 1; b < 8; b++) v[b] = LB(a.f.EII_pl, cb[b]);
#pragma unroll
            for (int T = 0; T < 3; T++) C.EIIv[T] = 0.25 * (v[CIDX(T, 0)] + v[CIDX(T, 1)] + v[CIDX(T, 2)] + v[CIDX(T, 3)]);
        }
    }
#undef CIDX
#undef LB
}

// The three edge families of one node in one thread, as in the reference's kernel: they share the clamped centre stencils
// (η, θ, the normal components) and each other's shear components, so the second and third family mostly hit in L1/L2.
// New edge stresses go to a.tnew (committed by the caller), so every read sees last iteration's values.
// XS: blocks are dealt round-robin to the 8 XCDs; give XCD q the q-th eighth of the (flattened xy, z) block sequence instead, so that the
// rows j +- 1 and planes k +- 1 a block gathers from were fetched by the same L2
// i0, iw: the launch covers the node columns i0 .. i0 + iw - 1 (the whole box: 0, nx + 1)
template <bool P4, bool XS = false, bool SOFT = false, int NP = 0>
__global__ __launch_bounds__(256) void k_vep3_edges(const Vep3Args a, const int i0, const int iw)
{
    const int nx = a.nx, ny = a.ny;
    int i, j, k;
    if (P4) {
        unsigned bx = blockIdx.x, by = blockIdx.y;
        if (XS) {
            const unsigned L = by * gridDim.x + bx, T = gridDim.x * gridDim.y, per = T / 8;
            if (L < per * 8) {
                const unsigned Ln = (L & 7u) * per + (L >> 3);
                bx = Ln % gridDim.x; by = Ln / gridDim.x;
            }
        }
        const int t_ = bx * 64 + (threadIdx.x & 63);
        j = t_ / iw; i = i0 + t_ - j * iw; k = by * 4 + (threadIdx.x >> 6);
        if (j >= ny + 1 || k >= a.nz + 1) return;
    } else {
        const int t_ = blockIdx.x * blockDim.x + threadIdx.x;
        j = t_ / iw; i = i0 + t_ - j * iw; k = blockIdx.y;
        if (j >= ny + 1) return;
    }
    const int nz = a.nz;
    const int ci[3] = {clampi3(i - 1, 0, nx - 1), clampi3(i, 0, nx - 1), clampi3(i + 1, 0, nx - 1)};
    const int cj[3] = {clampi3(j - 1, 0, ny - 1), clampi3(j, 0, ny - 1), clampi3(j + 1, 0, ny - 1)};
    const int ck[3] = {clampi3(k - 1, 0, nz - 1), clampi3(k, 0, nz - 1), clampi3(k + 1, 0, nz - 1)};
    CenAvg C;
    vep3_gather_centres<SOFT>(a, ci, cj, ck, C);
    vep3_edge_body<0, SOFT, NP>(a, i, j, k, ci, cj, ck, C);
    vep3_edge_body<1, SOFT, NP>(a, i, j, k, ci, cj, ck, C);
    vep3_edge_body<2, SOFT, NP>(a, i, j, k, ci, cj, ck, C);
}

// ------------------------------------------------------------------------------------------------
// z-marching form of the edge pass (option "vep3_edges" = 1, the default when no phase has a softening law).  A wave owns the nodes
// (i0 .. i0+61, j) -- lane l sits on node i = i0 - 1 + l, lanes 0 and 63 only feed their neighbours -- and walks KZ node planes.  Per plane
// it loads its own cell column of the centre arrays in rows j-1 and j (2 loads per array instead of 7): the i-1 operands come from
// the neighbouring lane, the k-1 ones are carried as the partial sums the reference's order of summation starts with, ((a + b) + c) + d:
//   yz: a, b = (i, j-1, k-1), (i, j, k-1)   xz: a, b = (i-1, j, k-1), (i, j, k-1)   xy: all four cells in plane k
// The 4-point stencils of the other families' shear components work the same way: rows j-1 / j / j+1 are loaded, i +- 1 comes from the
// neighbouring lanes, plane k-1 (xy) is carried and plane k+1 (yz, xz) is loaded one step ahead.  Operands are reduced to the trial stress
// as soon as they arrive (centre phase, shear phase, then the return mapping), results are bit-identical to k_vep3_edges (same operands,
// same order of summation).  One wave updating all three families needs ~335 VGPRs (1 wave/SIMD: 2.63 ms at 256^3, no faster than the
// gathering kernel's 2.68 ms, profiles/r02_vep3_zmarching.txt), so a wave updates ONE family (~170 VGPRs, 3 waves/SIMD, ~34 loads per node
// and family instead of ~60) and the three waves of a tile are scheduled on the same XCD so that they share their operands in its L2.
// Index clamps are the reference's (cell range [0, n-1] for every array, StressKernels.jl:604-668): lane values are the clamped-own
// ones, so "i-1" / "i+1" degenerate to the own value on the domain faces.
// ------------------------------------------------------------------------------------------------
// FAM: bit T set = family T (0 yz, 1 xz, 2 xy) is updated by this launch; what the other families alone need (loads, lane exchanges, carried
// sums) is dead code then
// ilim: the node columns i >= ilim are left to another launch (the last, nearly empty lane segment of a row is given to the one-node-per-thread kernel)
// LDSC: the three family waves of a (row, lane segment) form ONE workgroup and share the centre operands of every plane through LDS: wave f loads the
// arrays s with s % 3 == f (rows j-1 and j of its own column), all three read the 2 x NC values back after a barrier (two buffers, one barrier per plane) --
// 8 instead of 22 centre loads per wave and plane, and every centre line is fetched exactly once per tile whatever the dispatcher does with the blocks.
// sh: [2][12][2][64] doubles of the workgroup; fidx: this wave's index 0..2 among the loaders.
// PROD (with LDSC and LDSS): 1 = a fourth wave of the workgroup publishes ALL operands of a plane step and the family waves only read them back; 2 = that fourth wave (FAM = 0: no family of
// its own).  Same barrier sequence as before -- publish, barrier, read back -- but the publishing wave has nothing else to do and so requests the operands of step t + 1 while the family waves
// are still in the arithmetic of step t: the memory latency the family waves used to sit out at the top of every plane is hidden by construction instead of by occupancy.
template <int KZ, int NP, int FAM, bool SOFT = false, bool LDSC = false, bool LDSS = false, int PROD = 0>
__device__ __forceinline__ void vep3_edges_z_tile(const Vep3Args &a, const int seg, const int j, const int zchunk, const int ilim, double *sh = nullptr, const int fidx = 0)
{
    constexpr int NC = SOFT ? 12 : 11;          // centre arrays averaged to the edges; softening laws add EII_pl (StressKernels.jl:710,783,854)
    const int nx = a.nx, ny = a.ny, nz = a.nz, np = NP;
    const int lane = threadIdx.x & 63;
    if (j > ny) return;                                  // whole waves; in the LDS-sharing forms (barriers!) j is the workgroup's row, so the whole workgroup leaves together
    const int i = seg * 62 - 1 + lane;
    const bool useful = lane >= 1 && lane <= 62 && i <= nx && i < ilim;
    const int kb = zchunk * KZ, ke = min(kb + KZ, nz + 1);
    const int ic = clampi3(i, 0, nx - 1), ir = clampi3(i, 0, nx);
    const int cj0 = clampi3(j - 1, 0, ny - 1), cj1 = clampi3(j, 0, ny - 1), cj2 = clampi3(j + 1, 0, ny - 1);
    const bool lo_i = i >= 1, hi_i = i < nx - 1;
    // value at clamp(i - 1) / clamp(i + 1): the neighbouring lane's, or the own one on the domain's faces and at the ends of the wave -- the clamp goes into the permute address
    // (as a select behind __shfl_up / __shfl_down it was two v_cndmask per double: 56 of the ~430 VALU instructions of a plane step)
    const int a_up = 4 * ((lo_i && lane > 0) ? lane - 1 : lane), a_dn = 4 * ((hi_i && lane < 63) ? lane + 1 : lane);
    auto perm = [&](int addr, double v) {
        const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
        return __hiloint2double(hi, lo);
    };
    auto up = [&](double v) { return perm(a_up, v); };
    auto dn = [&](double v) { return perm(a_dn, v); };
    // centre arrays in the order they are consumed: 1/η, θ, then the (ε, τ, τ_o) triple of each normal component
    const double *const cen[12] = {a.f.eta, a.theta, a.f.exx, a.f.txx, a.f.toxx, a.f.eyy, a.f.tyy, a.f.toyy, a.f.ezz, a.f.tzz, a.f.tozz, a.f.EII_pl};
    const double *const Yp[3] = {a.f.eyz, a.f.tyz, a.f.toyz}, *const Xp[3] = {a.f.exz, a.f.txz, a.f.toxz}, *const Zp[3] = {a.f.exy, a.f.txy, a.f.toxy};
    // row offsets inside one plane of each array family
    // byte offsets (32-bit: every array is below 4 GiB, check_vep3) = lane part inside one plane + uniform plane part
    typedef unsigned int u32;
#define LB(p, off) (*(const double *)((const char *)(p) + (off)))
    const u32 oc0 = 8u * (u32)(ic + nx * cj0), oc1 = 8u * (u32)(ic + nx * cj1);                     // centres (nx, ny)
    const u32 oy1 = oc1, oy2 = 8u * (u32)(ic + nx * cj2);                                           // yz (nx, ny+1): rows clamp(j), clamp(j+1)
    const u32 ox0 = 8u * (u32)(ic + (nx + 1) * cj0), ox1 = 8u * (u32)(ic + (nx + 1) * cj1);         // xz (nx+1, ny): rows clamp(j-1), clamp(j)
    const u32 oz1 = ox1, oz2 = 8u * (u32)(ic + (nx + 1) * cj2);                                     // xy (nx+1, ny+1): rows clamp(j), clamp(j+1)
    const u32 pc = 8u * (u32)(nx * ny), py = 8u * (u32)(nx * (ny + 1)), px = 8u * (u32)((nx + 1) * ny), pz = 8u * (u32)((nx + 1) * (ny + 1));
    const bool act0 = useful && i < nx, act1 = useful && j < ny;             // yz / xz edge exists at this (i, j); xy: useful && k < nz
    double pyz[NC], pxz[NC], Yn[3][2], Xn[3][2], Zc[3][2];
    // LDSC: publish the (row j-1, row j) values of the arrays this wave loads for plane kp into buffer b; everybody reads them back after the barrier
    // (every load of a publication is requested before the first LDS write: left to the scheduler, which sinks each load to its use, a publication becomes a chain of
    // load - wait - write round trips -- profiles/r04_vep3d_fused_pre_centre.txt)
    double *const shs = sh + 2 * NC * 2 * 64;
    // kp: plane of the centre operands; ky / kz (LDSS): planes of the yz, xz / xy shear operands of the plane step
    // kn >= 0 (publishing wave only): also the phase ratios and λv of the three families' nodes of plane kn -- the family waves then have no load left that the arithmetic of a
    // plane waits for (the ratios were requested behind the barrier and used ~100 instructions later, λv inside the yielding branch: two exposed memory round trips per plane)
    double *const shm = shs + 2 * 18 * 64;        // [2][3 * NP + 3][64]
    constexpr int NM = 3 * NP + 3;
    auto publish = [&](u32 kp, u32 ky, u32 kz, int b, int kn = -1) {
        double v0[NC], v1[NC], w[18], pm[NM];
        if constexpr (PROD == 2) {
            if (kn >= 0) {
                const bool ac[3] = {act0, act1, useful && kn < nz};
                const i64 vn[3] = {i + (i64)nx * (j + (i64)(ny + 1) * kn), i + (i64)(nx + 1) * (j + (i64)ny * kn), i + (i64)(nx + 1) * (j + (i64)(ny + 1) * kn)};
                const double *const phs[3] = {a.f.phase_yz, a.f.phase_xz, a.f.phase_xy};
#pragma unroll
                for (int T = 0; T < 3; T++) {
#pragma unroll
                    for (int q = 0; q < NP; q++) pm[T * NP + q] = phs[T][ac[T] ? NP * vn[T] + q : 0];
                    pm[3 * NP + T] = a.lamv[T][ac[T] ? vn[T] : 0];
                }
            }
        }
#pragma unroll
        for (int s = 0; s < NC; s++) {
            if (PROD == 1 || (PROD == 0 && s % 3 != fidx)) continue;
            v0[s] = LB(cen[s], oc0 + pc * kp); v1[s] = LB(cen[s], oc1 + pc * kp);
        }
        if constexpr (LDSS) {
#pragma unroll
            for (int q = 0; q < 3; q++) {
                if (PROD == 1) continue;
                if (PROD == 2 || (6 * q + 0) % 3 == fidx) w[6 * q + 0] = LB(Yp[q], oy1 + py * ky);
                if (PROD == 2 || (6 * q + 1) % 3 == fidx) w[6 * q + 1] = LB(Yp[q], oy2 + py * ky);
                if (PROD == 2 || (6 * q + 2) % 3 == fidx) w[6 * q + 2] = LB(Xp[q], ox0 + px * ky);
                if (PROD == 2 || (6 * q + 3) % 3 == fidx) w[6 * q + 3] = LB(Xp[q], ox1 + px * ky);
                if (PROD == 2 || (6 * q + 4) % 3 == fidx) w[6 * q + 4] = LB(Zp[q], oz1 + pz * kz);
                if (PROD == 2 || (6 * q + 5) % 3 == fidx) w[6 * q + 5] = LB(Zp[q], oz2 + pz * kz);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NC; s++) {
            if (PROD == 1 || (PROD == 0 && s % 3 != fidx)) continue;
            if (s == 0) { v0[s] = 1 / v0[s]; v1[s] = 1 / v1[s]; }
            sh[((b * NC + s) * 2 + 0) * 64 + lane] = v0[s];
            sh[((b * NC + s) * 2 + 1) * 64 + lane] = v1[s];
        }
        if constexpr (LDSS) {
#pragma unroll
            for (int r = 0; r < 18; r++)
                if (PROD == 2 || (PROD == 0 && r % 3 == fidx)) shs[(b * 18 + r) * 64 + lane] = w[r];
        }
        if constexpr (PROD == 2) {
            if (kn >= 0) {
#pragma unroll
                for (int r = 0; r < NM; r++) shm[(b * NM + r) * 64 + lane] = pm[r];
            }
        }
    };
    auto read_shear = [&](int b) {
#pragma unroll
        for (int q = 0; q < 3; q++) {
            Yn[q][0] = shs[(b * 18 + 6 * q + 0) * 64 + lane]; Yn[q][1] = shs[(b * 18 + 6 * q + 1) * 64 + lane];
            Xn[q][0] = shs[(b * 18 + 6 * q + 2) * 64 + lane]; Xn[q][1] = shs[(b * 18 + 6 * q + 3) * 64 + lane];
            Zc[q][0] = shs[(b * 18 + 6 * q + 4) * 64 + lane]; Zc[q][1] = shs[(b * 18 + 6 * q + 5) * 64 + lane];
        }
    };
    {
        const u32 kc = (u32)clampi3(kb - 1, 0, nz - 1);
        if constexpr (LDSC) { publish(kc, (u32)clampi3(kb, 0, nz - 1), kc, 0); __syncthreads(); }
        if constexpr (PROD == 2) {      // the publishing wave: one publication and one barrier per plane, in step with the family waves
#pragma unroll 1
            for (int k = kb; k < ke; k++) {
                const u32 k1 = (u32)clampi3(k, 0, nz - 1), k2 = (u32)clampi3(k + 1, 0, nz - 1);
                const int bsel = (k - kb + 1) & 1;
                publish(k1, k2, k1, bsel, k);
                __syncthreads();
            }
            return;
        }
#pragma unroll
        for (int s = 0; s < NC; s++) {
            double v0, v1, u1;
            if constexpr (LDSC) {     // (the left neighbour's value straight from the workgroup's slot: one LDS read instead of two lane permutes)
                v0 = sh[((0 * NC + s) * 2 + 0) * 64 + lane]; v1 = sh[((0 * NC + s) * 2 + 1) * 64 + lane];
                u1 = sh[((0 * NC + s) * 2 + 1) * 64 + (a_up >> 2)];
            } else {
                v0 = LB(cen[s], oc0 + pc * kc); v1 = LB(cen[s], oc1 + pc * kc);
                if (s == 0) { v0 = 1 / v0; v1 = 1 / v1; }
                u1 = up(v1);
            }
            pyz[s] = v0 + v1;
            pxz[s] = u1 + v1;
        }
        const u32 k1 = (u32)clampi3(kb, 0, nz - 1);
        if constexpr (LDSS) read_shear(0);
        else {
#pragma unroll
            for (int q = 0; q < 3; q++) {
                Yn[q][0] = LB(Yp[q], oy1 + py * k1); Yn[q][1] = LB(Yp[q], oy2 + py * k1);
                Xn[q][0] = LB(Xp[q], ox0 + px * k1); Xn[q][1] = LB(Xp[q], ox1 + px * k1);
                Zc[q][0] = LB(Zp[q], oz1 + pz * kc); Zc[q][1] = LB(Zp[q], oz2 + pz * kc);
            }
        }
    }
#pragma unroll 1
    for (int k = kb; k < ke; k++) {
        const u32 k1 = (u32)clampi3(k, 0, nz - 1), k2 = (u32)clampi3(k + 1, 0, nz - 1);
        const bool act2 = useful && k < nz;
        const i64 vi[3] = {i + (i64)nx * (j + (i64)(ny + 1) * k), i + (i64)(nx + 1) * (j + (i64)ny * k), i + (i64)(nx + 1) * (j + (i64)(ny + 1) * k)};
        double rvl[PROD == 1 ? NP : 1];          // PROD == 1: the own family's phase ratios, from the publishing wave's slots (read behind the barrier below)
        const double *const rv[3] = {PROD == 1 ? rvl : a.f.phase_yz + (act0 ? np * vi[0] : 0), PROD == 1 ? rvl : a.f.phase_xz + (act1 ? np * vi[1] : 0),
                                     PROD == 1 ? rvl : a.f.phase_xy + (act2 ? np * vi[2] : 0)};
        // the four cells of array s around the node, in plane clamp(k): sums of the three families, then the carried partial sums of the next plane
        const int bsel = (k - kb + 1) & 1;            // the prologue used buffer 0
        if constexpr (LDSC) { publish(k1, k2, k1, bsel); __syncthreads(); }
        double lam_pre = 0.0;
        if constexpr (PROD == 1) {
            constexpr int TF = FAM == 1 ? 0 : (FAM == 2 ? 1 : 2);
#pragma unroll
            for (int q = 0; q < NP; q++) rvl[q] = shm[(bsel * NM + TF * NP + q) * 64 + lane];
            lam_pre = shm[(bsel * NM + 3 * NP + TF) * 64 + lane];
        }
        auto sums = [&](int s, double S[3]) {
            double v0, v1, u0, u1;
            if constexpr (LDSC) {
                v0 = sh[((bsel * NC + s) * 2 + 0) * 64 + lane]; v1 = sh[((bsel * NC + s) * 2 + 1) * 64 + lane];
                u0 = sh[((bsel * NC + s) * 2 + 0) * 64 + (a_up >> 2)]; u1 = sh[((bsel * NC + s) * 2 + 1) * 64 + (a_up >> 2)];
            } else {
                v0 = LB(cen[s], oc0 + pc * k1); v1 = LB(cen[s], oc1 + pc * k1);
                if (s == 0) { v0 = 1 / v0; v1 = 1 / v1; }
                u0 = up(v0); u1 = up(v1);
            }
            S[0] = (pyz[s] + v0) + v1;
            S[1] = (pxz[s] + u1) + v1;
            S[2] = ((u0 + v0) + u1) + v1;
            pyz[s] = v0 + v1;
            pxz[s] = u1 + v1;
        };
        EdgeMat m[3];
        double etav[3], Pv[3], dtr[3], tt[3][6], S[3][3], EIIv[3] = {0.0, 0.0, 0.0};
        if constexpr (SOFT) {
            sums(11, S[0]);
#pragma unroll
            for (int T = 0; T < 3; T++) EIIv[T] = 0.25 * S[0][T];
        }
        sums(0, S[0]);
        sums(1, S[1]);
#pragma unroll
        for (int T = 0; T < 3; T++) {
            if (!((FAM >> T) & 1)) continue;
            m[T] = edge_mat<NP>(a, rv[T]);
            etav[T] = 4 / S[0][T];
            Pv[T] = 0.25 * S[1][T];
            dtr[T] = 1.0 / (a.theta_dtau + etav[T] * m[T]._Gdt + 1.0);
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            sums(2 + 3 * c, S[0]);
            sums(3 + 3 * c, S[1]);
            sums(4 + 3 * c, S[2]);
#pragma unroll
            for (int T = 0; T < 3; T++) {
                if (!((FAM >> T) & 1)) continue;
                const double t = 0.25 * S[1][T];
                tt[T][c] = t + dev_stress_inc(t, 0.25 * S[2][T], etav[T], 0.25 * S[0][T], m[T]._Gdt, dtr[T]);
            }
            }
        // shear components: yz, xz at planes clamp(k) (carried) and clamp(k+1) (loaded here); xy at clamp(k-1) (carried) and clamp(k)
        double Yc[3][2], Xc[3][2], Zq[3][2];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            Yc[q][0] = Yn[q][0]; Yc[q][1] = Yn[q][1]; Xc[q][0] = Xn[q][0]; Xc[q][1] = Xn[q][1]; Zq[q][0] = Zc[q][0]; Zq[q][1] = Zc[q][1];
            if constexpr (!LDSS) {
                Yn[q][0] = LB(Yp[q], oy1 + py * k2); Yn[q][1] = LB(Yp[q], oy2 + py * k2);
                Xn[q][0] = LB(Xp[q], ox0 + px * k2); Xn[q][1] = LB(Xp[q], ox1 + px * k2);
                Zc[q][0] = LB(Zp[q], oz1 + pz * k1); Zc[q][1] = LB(Zp[q], oz2 + pz * k1);
            }
        }
        if constexpr (LDSS) read_shear(bsel);
        double own_t[3], own_d[3];
        auto trial = [&](int T, const double o[3]) { return o[1] + dev_stress_inc(o[1], o[2], etav[T], o[0], m[T]._Gdt, dtr[T]); };
        if constexpr ((FAM & 1) != 0) {   // yz edge: own (i, j, k); xz at (i, i+1) x (j-1, j); xy at (i, i+1) x planes (k-1, k)
            double own[3], ox[3], oz[3];
#pragma unroll
            for (int q = 0; q < 3; q++) {
                own[q] = Yc[q][0];
                if (j == ny || k == nz) own[q] = LB(Yp[q], 8u * (u32)(ic + nx * j) + py * (u32)k);
                ox[q] = 0.25 * (((Xc[q][0] + dn(Xc[q][0])) + Xc[q][1]) + dn(Xc[q][1]));
                oz[q] = 0.25 * (((Zq[q][0] + dn(Zq[q][0])) + Zc[q][0]) + dn(Zc[q][0]));
            }
            own_t[0] = own[1];
            own_d[0] = dev_stress_inc(own[1], own[2], etav[0], own[0], m[0]._Gdt, dtr[0]);
            tt[0][3] = own_t[0] + own_d[0]; tt[0][4] = trial(0, ox); tt[0][5] = trial(0, oz);
        }
        if constexpr ((FAM & 2) != 0) {   // xz edge: own (i, j, k); yz at (i-1, i) x (j, j+1); xy at (j, j+1) x planes (k-1, k)
            double own[3], oy[3], oz[3];
#pragma unroll
            for (int q = 0; q < 3; q++) {
                own[q] = Xc[q][1];
                if (i >= nx || k == nz) own[q] = LB(Xp[q], 8u * (u32)(ir + (nx + 1) * cj1) + px * (u32)k);
                oy[q] = 0.25 * (((up(Yc[q][0]) + Yc[q][0]) + Yc[q][1]) + up(Yc[q][1]));
                oz[q] = 0.25 * (((Zq[q][0] + Zq[q][1]) + Zc[q][0]) + Zc[q][1]);
            }
            own_t[1] = own[1];
            own_d[1] = dev_stress_inc(own[1], own[2], etav[1], own[0], m[1]._Gdt, dtr[1]);
            tt[1][3] = trial(1, oy); tt[1][4] = own_t[1] + own_d[1]; tt[1][5] = trial(1, oz);
        }
        if constexpr ((FAM & 4) != 0) {   // xy edge: own (i, j, k); yz at (i-1, i) x planes (k, k+1); xz at (j-1, j) x planes (k, k+1)
            double own[3], oy[3], ox[3];
#pragma unroll
            for (int q = 0; q < 3; q++) {
                own[q] = Zc[q][0];
                if (i >= nx || j == ny) own[q] = LB(Zp[q], 8u * (u32)(ir + (nx + 1) * j) + pz * k1);
                oy[q] = 0.25 * (((up(Yc[q][0]) + Yc[q][0]) + up(Yn[q][0])) + Yn[q][0]);
                ox[q] = 0.25 * (((Xc[q][0] + Xc[q][1]) + Xn[q][0]) + Xn[q][1]);
            }
            own_t[2] = own[1];
            own_d[2] = dev_stress_inc(own[1], own[2], etav[2], own[0], m[2]._Gdt, dtr[2]);
            tt[2][3] = trial(2, oy); tt[2][4] = trial(2, ox); tt[2][5] = own_t[2] + own_d[2];
        }
        if constexpr ((FAM & 1) != 0) if (act0) vep3_edge_plastic<0, SOFT, NP, PROD == 1>(a, vi[0], rv[0], m[0], tt[0], own_t[0], own_d[0], etav[0], Pv[0], dtr[0], EIIv[0], lam_pre);
        if constexpr ((FAM & 2) != 0) if (act1) vep3_edge_plastic<1, SOFT, NP, PROD == 1>(a, vi[1], rv[1], m[1], tt[1], own_t[1], own_d[1], etav[1], Pv[1], dtr[1], EIIv[1], lam_pre);
        if constexpr ((FAM & 4) != 0) if (act2) vep3_edge_plastic<2, SOFT, NP, PROD == 1>(a, vi[2], rv[2], m[2], tt[2], own_t[2], own_d[2], etav[2], Pv[2], dtr[2], EIIv[2], lam_pre);
    }
}
#undef LB
template <int KZ, int NP, int FAM>
__global__ __launch_bounds__(256) void k_vep3_edges_z(const Vep3Args a, int nseg, int ilim)
{
    vep3_edges_z_tile<KZ, NP, FAM>(a, blockIdx.x % nseg, (blockIdx.x / nseg) * 4 + (int)(threadIdx.x >> 6), blockIdx.y, ilim);
}
// One launch, one family per block: the three blocks of a tile (same nodes, families yz / xz / xy) sit next to each other in the block sequence
// of ONE XCD (blocks are dealt round-robin to the 8 XCDs), so that the operands they share are fetched from HBM once and found in that XCD's L2 by
// the other two; per block only one family's state lives in registers.
template <int KZ, int NP, int MINB, bool SOFT = false>
__global__ __launch_bounds__(256, MINB) void k_vep3_edges_zf(const Vep3Args a, int nseg, int ntile_xy, int ntiles, int ilim)
{
    const unsigned L = blockIdx.x, xcd = L & 7u, q = L >> 3;            // q-th block of this XCD
    const unsigned per = ((unsigned)ntiles + 7u) / 8u;                  // XCD x works on the tiles [x * per, (x + 1) * per): a slab of z chunks
    const unsigned fam = q % 3u, t = xcd * per + q / 3u;                // tile index over (xy tiles fastest, z chunks)
    if (q / 3u >= per || t >= (unsigned)ntiles) return;
    const int txy = (int)(t % (unsigned)ntile_xy), zc = (int)(t / (unsigned)ntile_xy);
    const int j = (txy / nseg) * 4 + (int)(threadIdx.x >> 6);
    if (fam == 0) vep3_edges_z_tile<KZ, NP, 1, SOFT>(a, txy % nseg, j, zc, ilim);
    else if (fam == 1) vep3_edges_z_tile<KZ, NP, 2, SOFT>(a, txy % nseg, j, zc, ilim);
    else vep3_edges_z_tile<KZ, NP, 4, SOFT>(a, txy % nseg, j, zc, ilim);
}
// The LDS-sharing form (see LDSC above): a workgroup = the three family waves of one (row, lane segment, z chunk); tiles in the XCD slab order of k_vep3_edges_zf.
template <int KZ, int NP, bool SOFT = false, bool LDSS = false>
__global__ __launch_bounds__(192, SOFT ? 2 : 3) void k_vep3_edges_zl(const Vep3Args a, int nseg, int ntile_xy, int ntiles, int ilim)
{
    __shared__ double sh[2 * (SOFT ? 12 : 11) * 2 * 64 + (LDSS ? 2 * 18 * 64 : 0)];       // LDSS: 40 KB without softening laws, four workgroups (12 waves) per CU
    const unsigned L = blockIdx.x, xcd = L & 7u, q = L >> 3;
    const unsigned per = ((unsigned)ntiles + 7u) / 8u;
    const unsigned t = xcd * per + q;
    if (q >= per || t >= (unsigned)ntiles) return;                  // whole workgroups
    const int txy = (int)(t % (unsigned)ntile_xy), zc = (int)(t / (unsigned)ntile_xy);
    const int fam = (int)(threadIdx.x >> 6), j = txy / nseg;        // ntile_xy = nseg * (ny + 1): one row per workgroup
    // The three family waves run three instantiations of the tile function.  Each instantiation executes the same barrier sequence -- one in the prologue, one per plane of the
    // chunk, with loop bounds (kb, ke) that depend on the workgroup's chunk only -- and no instantiation returns early unless the whole workgroup does (row j, above): every
    // __syncthreads() is reached by all 192 threads the same number of times, although textually from different branches
    if (fam == 0) vep3_edges_z_tile<KZ, NP, 1, SOFT, true, LDSS>(a, txy % nseg, j, zc, ilim, sh, 0);
    else if (fam == 1) vep3_edges_z_tile<KZ, NP, 2, SOFT, true, LDSS>(a, txy % nseg, j, zc, ilim, sh, 1);
    else vep3_edges_z_tile<KZ, NP, 4, SOFT, true, LDSS>(a, txy % nseg, j, zc, ilim, sh, 2);
}
// The form with a publishing wave (see PROD above): a workgroup = the three family waves + the wave that loads for them; tiles as in k_vep3_edges_zl.
template <int KZ, int NP>
__global__ __launch_bounds__(256, 3) void k_vep3_edges_zp(const Vep3Args a, int nseg, int ntile_xy, int ntiles, int ilim)
{
    __shared__ double sh[2 * 11 * 2 * 64 + 2 * 18 * 64 + 2 * (3 * NP + 3) * 64];       // NP = 2: 50 KB, three workgroups per CU
    const unsigned L = blockIdx.x, xcd = L & 7u, q = L >> 3;
    const unsigned per = ((unsigned)ntiles + 7u) / 8u;
    const unsigned t = xcd * per + q;
    if (q >= per || t >= (unsigned)ntiles) return;                  // whole workgroups
    const int txy = (int)(t % (unsigned)ntile_xy), zc = (int)(t / (unsigned)ntile_xy);
    const int role = (int)(threadIdx.x >> 6), j = txy / nseg;
    // four instantiations, one barrier sequence (see k_vep3_edges_zl)
    if (role == 0) vep3_edges_z_tile<KZ, NP, 1, false, true, true, 1>(a, txy % nseg, j, zc, ilim, sh, 0);
    else if (role == 1) vep3_edges_z_tile<KZ, NP, 2, false, true, true, 1>(a, txy % nseg, j, zc, ilim, sh, 1);
    else if (role == 2) vep3_edges_z_tile<KZ, NP, 4, false, true, true, 1>(a, txy % nseg, j, zc, ilim, sh, 2);
    else vep3_edges_z_tile<KZ, NP, 0, false, true, true, 2>(a, txy % nseg, j, zc, ilim, sh, 3);
}
// update_stresses_center_vertex_ps! 3D -- centres (StressKernels.jl:906-985; cache_tensors StressUpdate.jl:269-301)
// NP > 0: the number of phases as a compile-time constant -- the cell's phase ratios are loaded once, in one batch, and the phase loops unroll (with a run-time count every
// loop iteration of every material function is a load the next instruction waits for: ~10 dependent memory round trips per cell)
template <bool SOFT, int NP = 0>
__global__ __launch_bounds__(256) void k_vep3_centre(const Vep3Args a)
{
    const int nx = a.nx, ny = a.ny, nz = a.nz, np = NP > 0 ? NP : a.rh.nphase;
    NODE_IJK_XS(nx, ny)
    if (k >= nz) return;
    const i64 c = i + (i64)nx * (j + (i64)ny * k);
    double rcv[NP > 0 ? NP : 1];
    if (NP > 0) {
#pragma unroll
        for (int q = 0; q < NP; q++) rcv[q] = a.f.phase_c[(i64)NP * c + q];
    }
    const double *rc = NP > 0 ? rcv : a.f.phase_c + (i64)np * c;
    const double _Gdt = 1.0 / (ratio_avg3(a.rh.G, rc, np) * a.dt);
    bool is_pl; double eta_reg;
    plastic_params3<NP>(a.rh, rc, is_pl, eta_reg);
    const double K = ratio_avg3(a.rh.Kb, rc, np);
    const double e = a.f.eta[c];
    const double dtr = 1.0 / (a.theta_dtau + e * _Gdt + 1.0);
    double eij[6] = {a.f.exx[c], a.f.eyy[c], a.f.ezz[c], 0, 0, 0};
    // _av_yz/_av_xz/_av_xy = 0.25 * mysum (MiniKernels.jl:116-121, 228-236): s = 0.0, then k-outer, j, i-inner adds
    eij[3] = 0.25 * ((((0.0 + EYZ(a.f.eyz, i, j, k)) + EYZ(a.f.eyz, i, j + 1, k)) + EYZ(a.f.eyz, i, j, k + 1)) + EYZ(a.f.eyz, i, j + 1, k + 1));
    eij[4] = 0.25 * ((((0.0 + EXZ(a.f.exz, i, j, k)) + EXZ(a.f.exz, i + 1, j, k)) + EXZ(a.f.exz, i, j, k + 1)) + EXZ(a.f.exz, i + 1, j, k + 1));
    eij[5] = 0.25 * ((((0.0 + EXY(a.f.exy, i, j, k)) + EXY(a.f.exy, i + 1, j, k)) + EXY(a.f.exy, i, j + 1, k)) + EXY(a.f.exy, i + 1, j + 1, k));
    double *const tc[6] = {a.f.txx, a.f.tyy, a.f.tzz, a.f.tyz_c, a.f.txz_c, a.f.txy_c};
    double *const tw[6] = {a.cnew[0] ? a.cnew[0] : a.f.txx, a.cnew[1] ? a.cnew[1] : a.f.tyy, a.cnew[2] ? a.cnew[2] : a.f.tzz, a.f.tyz_c, a.f.txz_c, a.f.txy_c};
    const double *const toc[6] = {a.f.toxx, a.f.toyy, a.f.tozz, a.f.toyz_c, a.f.toxz_c, a.f.toxy_c};
    double tij[6], d[6], tt[6];
#pragma unroll
    for (int s = 0; s < 6; s++) {
        tij[s] = tc[s][c];
        const double to = toc[s][c];
        d[s] = (-(tij[s] - to) * e * _Gdt - tij[s] + 2.0 * e * eij[s]) * dtr;       // :926, plain arithmetic
        tt[s] = tij[s] + d[s];
    }
    double tII;
    {
        double q6[6];
#pragma unroll
        for (int s = 0; s < 6; s++) q6[s] = d[s] + tij[s];
        tII = sinv3(q6);
    }
    double dQdt[6], dQdP, dFdP;
    plastic_grad3<NP>(a.rh, rc, tt, dQdt, dQdP, dFdP);
    const double vol = isinf(K) ? 0.0 : K * a.dt * dFdP * dQdP;
    const double Pr = a.theta[c];
    const double F = yield_F3<SOFT, NP>(a.rh, rc, Pr, tII, SOFT ? a.f.EII_pl[c] : 0.0);
    double l = a.lam[c];
    if (is_pl && tII != 0.0 && F > 0) {
        l = (1.0 - a.rel) * l + a.rel * (fmax(F, 0.0) / (e * dtr + eta_reg + vol));
        a.lam[c] = l;
        double epl[6];
#pragma unroll
        for (int s = 0; s < 6; s++) { epl[s] = l * dQdt[s]; d[s] = d[s] - 2.0 * e * epl[s] * dtr; tij[s] = d[s] + tij[s]; }
        if (a.obs) VST(a, a.f.evol_pl[c], -l * dQdP);
#pragma unroll
        for (int s = 0; s < 6; s++) VST(a, tw[s][c], tij[s]);
        if (a.obs) { VST(a, a.f.eplxx[c], epl[0]); VST(a, a.f.eplyy[c], epl[1]); VST(a, a.f.eplzz[c], epl[2]); }
        tII = sinv3(tij);
    } else {
        if (a.obs) VST(a, a.f.evol_pl[c], 0.0);
#pragma unroll
        for (int s = 0; s < 6; s++) VST(a, tw[s][c], d[s] + tij[s]);
        if (a.obs) { VST(a, a.f.eplxx[c], 0.0); VST(a, a.f.eplyy[c], 0.0); VST(a, a.f.eplzz[c], 0.0); }
    }
    if (a.obs) VST(a, a.f.tII[c], tII);
    if (a.obs) VST(a, a.f.eta_vep[c], tII * 0.5 * (1.0 / sinv3(eij)));
    VST(a, a.f.P[c], Pr - (isinf(K) ? 0.0 : K * a.dt * l * dQdP));
}

// ------------------------------------------------------------------------------------------------
// k_vep3_pre + k_vep3_visc (linear laws) + k_vep3_centre as ONE kernel (option "vep3_fuse_pc", single rank): the centre pass of update_stresses_center_vertex_ps!
// reads nothing the edge pass writes -- only the cell's own ε, τ, τ_o, θ, η, λ and the strain rates of its twelve edges -- so it can run before the edge pass, in the
// thread that has just produced the cell's ∇V, θ and ε.  The twelve edge strain rates are recomputed from the velocities (the expressions of k_vep3_pre, so the same bits
// as the stored arrays the edge pass reads); walking up its column the thread carries the four of plane k + 1 into the next step.  What the edge pass still has to find
// unchanged goes to second arrays: the relaxed η (compute_maxloc! of the neighbouring cells reads the old one) to eta_out, the new τxx, τyy, τzz (the edge pass averages the
// old ones) to a.cnew; the driver swaps the pointers.  Per iteration 16 written + ~27 fetched passes instead of (8.3 + 11.9) + (1 + 2) + (7.1 + 25.3).
// Same arithmetic, operation for operation, as the three kernels: bit-identical (tests/test_gpu_vep3d.py::test_vep3d_fused_pre_centre_equals_the_three_kernels).
// OBS: the launch stores the output-only arrays (a.obs as a compile-time constant: with the flag tested at run time every such store ends a basic block, and the
// scheduler, which works one block at a time, can no longer issue the plane's loads as one batch -- seven dependent memory round trips per plane instead of two)
// ML = false (ranks with neighbours): ητ arrives in a.etatau -- compute_maxloc! of the relaxed η and its update_halo! run on the halo stream beside the edge pass -- instead of being taken
// from the 3 x 3 x 3 window of η here
template <bool SOFT, bool RHO, int NP = 0, bool OBS = true, bool ML = true>
__global__ __launch_bounds__(256, 3) void k_vep3_prec(const Vep3Args a, double *__restrict__ eta_out, const int KZ, const int tile_ntx = 0)
{
    const int nx = a.nx, ny = a.ny, nz = a.nz, np = NP > 0 ? NP : a.rh.nphase;
    unsigned bx_ = blockIdx.x, by_ = blockIdx.y;
    {   // XCD slab order of the (flattened xy, z chunk) block sequence, see NODE_IJK_XS
        const unsigned L_ = by_ * gridDim.x + bx_, per_ = (gridDim.x * gridDim.y) / 8u;
        if (L_ < per_ * 8u) { const unsigned Ln_ = (L_ & 7u) * per_ + (L_ >> 3); bx_ = Ln_ % gridDim.x; by_ = Ln_ / gridDim.x; }
    }
    int i, j;
    if (tile_ntx > 0) {     // 64 x 4 tiles of node columns (one wave per row): the three rows of a velocity / η window that a block's rows share are requested by waves of the same block
        const int tyi = (int)bx_ / tile_ntx, txi = (int)bx_ - tyi * tile_ntx;
        i = txi * 64 + (int)(threadIdx.x & 63); j = tyi * 4 + (int)(threadIdx.x >> 6);
        if (i > nx || j > ny) return;
    } else {
        const int t_ = bx_ * blockDim.x + threadIdx.x;
        j = t_ / (nx + 1); i = t_ - j * (nx + 1);
        if (j >= ny + 1) return;
    }
    const int k0 = (int)by_ * KZ, k1 = min(k0 + KZ, nz + 1);
    const double *__restrict__ Vx = a.f.Vx, *__restrict__ Vy = a.f.Vy, *__restrict__ Vz = a.f.Vz;
    const double _dx = a._dx, _dy = a._dy, _dz = a._dz;
#define VX(i_, j_, k_) Vx[(i_) + (i64)(nx + 1) * ((j_) + (i64)(ny + 2) * (k_))]
#define VY(i_, j_, k_) Vy[(i_) + (i64)(nx + 2) * ((j_) + (i64)(ny + 1) * (k_))]
#define VZ(i_, j_, k_) Vz[(i_) + (i64)(nx + 2) * ((j_) + (i64)(ny + 2) * (k_))]
    const bool cellcol = i < nx && j < ny;
    if (!cellcol) {     // node columns on the high faces i = nx / j = ny: their own edge strain rates only, as k_vep3_pre
#pragma unroll 1
        for (int k = k0; k < k1; k++) {
            if (i < nx) VST(a, EYZ(a.f.eyz, i, j, k), 0.5 * (_dz * (VY(i + 1, j, k + 1) - VY(i + 1, j, k)) + _dy * (VZ(i + 1, j + 1, k) - VZ(i + 1, j, k))));
            if (j < ny) VST(a, EXZ(a.f.exz, i, j, k), 0.5 * (_dz * (VX(i, j + 1, k + 1) - VX(i, j + 1, k)) + _dx * (VZ(i + 1, j + 1, k) - VZ(i, j + 1, k))));
            if (k < nz) VST(a, EXY(a.f.exy, i, j, k), 0.5 * (_dy * (VX(i, j + 1, k + 1) - VX(i, j, k + 1)) + _dx * (VY(i + 1, j, k + 1) - VY(i, j, k + 1))));
        }
        return;
    }
    // Addressing: one 32-bit byte offset per array LAYOUT and thread (every array is below 4 GiB, check_vep3), advanced by the layout's plane stride per step; the row / plane
    // displacements of the stencils go into uniform base pointers (SGPRs) and the x displacements into the instruction's immediate offset.  Left to itself the compiler keeps a
    // 64-bit address per (array, displacement) pair alive across the loop: ~100 VGPRs, and with them the kernel's third wave per SIMD.
    typedef unsigned int u32;
#define LB(p, off) (*(const double *)((const char *)(p) + (off)))
#define SW(p, off) (*(double *)((char *)(p) + (off)))
    const u32 pVx = 8u * (u32)((nx + 1) * (ny + 2)), pVy = 8u * (u32)((nx + 2) * (ny + 1)), pVz = 8u * (u32)((nx + 2) * (ny + 2));
    const u32 pC = 8u * (u32)(nx * ny), pYZ = 8u * (u32)(nx * (ny + 1)), pXZ = 8u * (u32)((nx + 1) * ny), pXY = 8u * (u32)((nx + 1) * (ny + 1));
    u32 ovx = 8u * (u32)(i + (nx + 1) * j) + pVx * (u32)k0;          // Vx[i, j, k]
    u32 ovy = 8u * (u32)(i + (nx + 2) * j) + pVy * (u32)k0;          // Vy[i, j, k]
    u32 ovz = 8u * (u32)(i + (nx + 2) * j) + pVz * (u32)k0;          // Vz[i, j, k]
    u32 oc = 8u * (u32)(i + nx * j) + pC * (u32)k0;                  // cell (i, j, k)
    u32 oyz = 8u * (u32)(i + nx * j) + pYZ * (u32)k0, oxz = 8u * (u32)(i + (nx + 1) * j) + pXZ * (u32)k0, oxy = 8u * (u32)(i + (nx + 1) * j) + pXY * (u32)k0;
    // uniform base pointers: X[dj][dk] = Vx displaced by dj rows and dk planes, ...
    const i64 rX = nx + 1, qX = (i64)(nx + 1) * (ny + 2), rY = nx + 2, qY = (i64)(nx + 2) * (ny + 1), rZ = nx + 2, qZ = (i64)(nx + 2) * (ny + 2);
    const double *const X01 = Vx + qX, *const X11 = Vx + rX + qX, *const X21 = Vx + 2 * rX + qX, *const X12 = Vx + rX + 2 * qX, *const X10 = Vx + rX;
    const double *const Y01 = Vy + qY, *const Y11 = Vy + rY + qY, *const Y02 = Vy + 2 * qY, *const Y12 = Vy + rY + 2 * qY, *const Y00 = Vy, *const Y10 = Vy + rY;
    const double *const Z00 = Vz, *const Z10 = Vz + rZ, *const Z20 = Vz + 2 * rZ, *const Z01 = Vz + qZ, *const Z11 = Vz + rZ + qZ, *const Z21 = Vz + 2 * rZ + qZ;
    // clamped 3 x 3 window of η: three row offsets, the x displacements 0 on the faces
    const u32 er[3] = {8u * (u32)(i + nx * clampi3(j - 1, 0, ny - 1)), 8u * (u32)(i + nx * j), 8u * (u32)(i + nx * clampi3(j + 1, 0, ny - 1))};
    const u32 dxl = i > 0 ? 8u : 0u, dxr = i < nx - 1 ? 8u : 0u;
    auto plane_max = [&](int kk) {          // clamped 3 x 3 maximum of η in plane clamp(kk), the comparison order of k_maxloc within the plane
        const double *const ep = a.f.eta + (i64)nx * ny * clampi3(kk, 0, nz - 1);
        double m = -INFINITY;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const double v0 = LB(ep, er[q] - dxl), v1 = LB(ep, er[q]), v2 = LB(ep, er[q] + dxr);
            if (v0 > m) m = v0;
            if (v1 > m) m = v1;
            if (v2 > m) m = v2;
        }
        return m;
    };
    double m_prev = ML ? plane_max(k0 - 1) : 0.0, m_cur = ML ? plane_max(k0) : 0.0, m_next;
    // carried up the column: the velocities of plane k + 1 that plane k + 2's edges difference against, Vz of the cell's lower face, and the four edge strain rates of plane k
    double vx_a = LB(X11, ovx), vx_b = LB(X11, ovx + 8u), vy_a = LB(Y01, ovy + 8u), vy_b = LB(Y11, ovy + 8u), vz_c = LB(Z10, ovz + 8u);
    double e_yz0 = 0.5 * (_dz * (vy_a - LB(Y00, ovy + 8u)) + _dy * (vz_c - LB(Z00, ovz + 8u)));                      // eyz(i, j, k0)
    double e_yz1 = 0.5 * (_dz * (vy_b - LB(Y10, ovy + 8u)) + _dy * (LB(Z20, ovz + 8u) - vz_c));                      // eyz(i, j+1, k0)
    double e_xz0 = 0.5 * (_dz * (vx_a - LB(X10, ovx)) + _dx * (vz_c - LB(Z10, ovz)));                                // exz(i, j, k0)
    double e_xz1 = 0.5 * (_dz * (vx_b - LB(X10, ovx + 8u)) + _dx * (LB(Z10, ovz + 16u) - vz_c));                     // exz(i+1, j, k0)
    double *const tc[6] = {a.f.txx, a.f.tyy, a.f.tzz, a.f.tyz_c, a.f.txz_c, a.f.txy_c};
    double *const tw[6] = {a.cnew[0], a.cnew[1], a.cnew[2], a.f.tyz_c, a.f.txz_c, a.f.txy_c};
    const double *const toc[6] = {a.f.toxx, a.f.toyy, a.f.tozz, a.f.toyz_c, a.f.toxz_c, a.f.toxy_c};
    const double _dt = 1.0 / a.dt;
#pragma unroll 1
    for (int k = k0; k < k1; k++, ovx += pVx, ovy += pVy, ovz += pVz, oc += pC, oyz += pYZ, oxz += pXZ, oxy += pXY) {
        SW(a.f.eyz, oyz) = e_yz0;
        SW(a.f.exz, oxz) = e_xz0;
        if (k >= nz) break;               // the node plane above the last cells
        // ---- every operand of the plane, requested before the first of them is used
        const double z_n = LB(Z11, ovz + 8u), z_d = LB(Z01, ovz + 8u), z_u = LB(Z21, ovz + 8u), z_l = LB(Z11, ovz), z_r = LB(Z11, ovz + 16u);
        const double x00 = LB(X01, ovx), x02 = LB(X21, ovx), x10 = LB(X01, ovx + 8u), x12 = LB(X21, ovx + 8u);
        const double y00 = LB(Y01, ovy), y01 = LB(Y11, ovy), y20 = LB(Y01, ovy + 16u), y21 = LB(Y11, ovy + 16u);
        const double nx_a = LB(X12, ovx), nx_b = LB(X12, ovx + 8u), ny_a = LB(Y02, ovy + 8u), ny_b = LB(Y12, ovy + 8u);
        const double Kc_ = LB(a.Kc, oc), Gc_ = LB(a.Gc, oc), P = LB(a.theta, oc), P0 = LB(a.f.P0, oc), Q_ = LB(a.f.Q, oc);
        double w9[9], et_in = 0.0;
        if (ML) {
            const double *const ep = a.f.eta + (i64)nx * ny * clampi3(k + 1, 0, nz - 1);
#pragma unroll
            for (int q = 0; q < 3; q++) { w9[3 * q] = LB(ep, er[q] - dxl); w9[3 * q + 1] = LB(ep, er[q]); w9[3 * q + 2] = LB(ep, er[q] + dxr); }
        } else et_in = LB(a.etatau, oc);
        const double e_lin = LB(a.eta_lin, oc), e_old = LB(a.f.eta, oc), l_old = LB(a.lam, oc);
        const i64 c = (i64)(oc >> 3);
        double rcv[NP > 0 ? NP : 1];
        if (NP > 0) {
#pragma unroll
            for (int q = 0; q < NP; q++) rcv[q] = a.f.phase_c[(i64)NP * c + q];
        }
        double tij[6], toij[6];
#pragma unroll
        for (int s = 0; s < 6; s++) { tij[s] = LB(tc[s], oc); toij[s] = LB(toc[s], oc); }
        const double EII_ = SOFT ? LB(a.f.EII_pl, oc) : 0.0;
        __builtin_amdgcn_sched_barrier(0);      // (the scheduler, minimising register pressure, would sink every load to its first use again)
        // ---- compute_∇V!, compute_P!, compute_strain_rate! (k_vep3_pre)
        const double dxi = (-vx_a + vx_b) * _dx;
        const double dyi = (-vy_a + vy_b) * _dy;
        const double dzi = (-vz_c + z_n) * _dz;
        const double divV = dxi + dyi + dzi;
        if (OBS) SW(a.f.divV, oc) = divV;
        const double _Kdt = 1.0 / (Kc_ * a.dt), _Gdt0 = 1.0 / (Gc_ * a.dt);
        const double rhs = -divV + (Q_ * _dt);
        if (OBS) SW(a.f.RP, oc) = fma(-(P - P0), _Kdt, rhs);
        double et = et_in;
        if (ML) {
            m_next = -INFINITY;
#pragma unroll
            for (int q = 0; q < 9; q++)
                if (w9[q] > m_next) m_next = w9[q];         // the comparison order of plane_max
            et = m_prev;
            if (m_cur > et) et = m_cur;
            if (m_next > et) et = m_next;
            m_prev = m_cur; m_cur = m_next;
            SW(const_cast<double *>(a.etatau), oc) = et;
        }
        const double psi = 1.0 / (1.0 / et + _Gdt0) * a.r / a.theta_dtau;
        const double Pr = (fma(P0, _Kdt, rhs) * psi + P) / (1.0 + _Kdt * psi);
        SW(a.theta, oc) = Pr;
        const double d3 = divV * (1.0 / 3.0);
        double eij[6];
        eij[0] = dxi - d3; eij[1] = dyi - d3; eij[2] = dzi - d3;
        SW(a.f.exx, oc) = eij[0];
        SW(a.f.eyy, oc) = eij[1];
        SW(a.f.ezz, oc) = eij[2];
        if (RHO) a.f.fz[c] = mat_density_ratio(a.rh, a.f.phase_c + (i64)np * c,
                                               !a.f.T ? 0.0 : (a.tg ? a.f.T[i + (i64)(nx + 2) * (j + (i64)(ny + 2) * k)] : a.f.T[c]), a.f.P[c]) * a.rh.gravity;
        // the four xy edges of the cell (plane k) and the yz / xz edges of plane k + 1
        {
            const double exy00 = 0.5 * (_dy * (vx_a - x00) + _dx * (vy_a - y00));
            const double exy10 = 0.5 * (_dy * (vx_b - x10) + _dx * (y20 - vy_a));
            const double exy01 = 0.5 * (_dy * (x02 - vx_a) + _dx * (vy_b - y01));
            const double exy11 = 0.5 * (_dy * (x12 - vx_b) + _dx * (y21 - vy_b));
            SW(a.f.exy, oxy) = exy00;
            eij[5] = 0.25 * ((((0.0 + exy00) + exy10) + exy01) + exy11);
        }
        {
            const double n_yz0 = 0.5 * (_dz * (ny_a - vy_a) + _dy * (z_n - z_d));
            const double n_yz1 = 0.5 * (_dz * (ny_b - vy_b) + _dy * (z_u - z_n));
            const double n_xz0 = 0.5 * (_dz * (nx_a - vx_a) + _dx * (z_n - z_l));
            const double n_xz1 = 0.5 * (_dz * (nx_b - vx_b) + _dx * (z_r - z_n));
            // _av_yz/_av_xz/_av_xy = 0.25 * mysum (MiniKernels.jl:116-121, 228-236): s = 0.0, then k-outer, j, i-inner adds
            eij[3] = 0.25 * ((((0.0 + e_yz0) + e_yz1) + n_yz0) + n_yz1);
            eij[4] = 0.25 * ((((0.0 + e_xz0) + e_xz1) + n_xz0) + n_xz1);
            e_yz0 = n_yz0; e_yz1 = n_yz1; e_xz0 = n_xz0; e_xz1 = n_xz1;
            vx_a = nx_a; vx_b = nx_b; vy_a = ny_a; vy_b = ny_b; vz_c = z_n;
        }
        // ---- update_viscosity_τII! of the linear laws (k_vep3_visc<false, false>)
        double e = e_lin;
        e = e * a.nu + e_old * (1.0 - a.nu);
        e = fmin(fmax(e, a.cut_lo), a.cut_hi);
        SW(eta_out, oc) = e;
        // ---- update_stresses_center_vertex_ps!, centres (k_vep3_centre)
        const double *rc = NP > 0 ? rcv : a.f.phase_c + (i64)np * c;
        const double _Gdt = 1.0 / (ratio_avg3(a.rh.G, rc, np) * a.dt);
        bool is_pl; double eta_reg;
        plastic_params3<NP>(a.rh, rc, is_pl, eta_reg);
        const double K = ratio_avg3(a.rh.Kb, rc, np);
        const double dtr = 1.0 / (a.theta_dtau + e * _Gdt + 1.0);
        double d[6], tt[6];
#pragma unroll
        for (int s = 0; s < 6; s++) {
            d[s] = (-(tij[s] - toij[s]) * e * _Gdt - tij[s] + 2.0 * e * eij[s]) * dtr;       // :926, plain arithmetic
            tt[s] = tij[s] + d[s];
        }
        double tII;
        {
            double q6[6];
#pragma unroll
            for (int s = 0; s < 6; s++) q6[s] = d[s] + tij[s];
            tII = sinv3(q6);
        }
        double dQdt[6], dQdP, dFdP;
        plastic_grad3<NP>(a.rh, rc, tt, dQdt, dQdP, dFdP);
        const double vol = isinf(K) ? 0.0 : K * a.dt * dFdP * dQdP;
        const double F = yield_F3<SOFT, NP>(a.rh, rc, Pr, tII, EII_);
        double l = l_old;
        if (is_pl && tII != 0.0 && F > 0) {
            l = (1.0 - a.rel) * l + a.rel * (fmax(F, 0.0) / (e * dtr + eta_reg + vol));
            SW(a.lam, oc) = l;
            double epl[6];
#pragma unroll
            for (int s = 0; s < 6; s++) { epl[s] = l * dQdt[s]; d[s] = d[s] - 2.0 * e * epl[s] * dtr; tij[s] = d[s] + tij[s]; }
            if (OBS) SW(a.f.evol_pl, oc) = -l * dQdP;
#pragma unroll
            for (int s = 0; s < 6; s++) SW(tw[s], oc) = tij[s];
            if (OBS) { SW(a.f.eplxx, oc) = epl[0]; SW(a.f.eplyy, oc) = epl[1]; SW(a.f.eplzz, oc) = epl[2]; }
            tII = sinv3(tij);
        } else {
            if (OBS) SW(a.f.evol_pl, oc) = 0.0;
#pragma unroll
            for (int s = 0; s < 6; s++) SW(tw[s], oc) = d[s] + tij[s];
            if (OBS) { SW(a.f.eplxx, oc) = 0.0; SW(a.f.eplyy, oc) = 0.0; SW(a.f.eplzz, oc) = 0.0; }
        }
        if (OBS) SW(a.f.tII, oc) = tII;
        if (OBS) SW(a.f.eta_vep, oc) = tII * 0.5 * (1.0 / sinv3(eij));
        SW(a.f.P, oc) = Pr - (isinf(K) ? 0.0 : K * a.dt * l * dQdP);
    }
#undef LB
#undef SW
}

__device__ __forceinline__ double sinv_stag3(const double *xx, const double *yy, const double *zz, const double *yz, const double *xz, const double *xy,
                                             int nx, int ny, int i, int j, int k)
{   // second_invariant_staggered on the gathers of MiniKernels.jl:196-204 (mean of the squared edge values, as pinned in 2D)
    const i64 c = i + (i64)nx * (j + (i64)ny * k);
    const double a0 = EYZ(yz, i, j, k), a1 = EYZ(yz, i, j + 1, k), a2 = EYZ(yz, i, j, k + 1), a3 = EYZ(yz, i, j + 1, k + 1);
    const double b0 = EXZ(xz, i, j, k), b1 = EXZ(xz, i + 1, j, k), b2 = EXZ(xz, i, j, k + 1), b3 = EXZ(xz, i + 1, j, k + 1);
    const double c0 = EXY(xy, i, j, k), c1 = EXY(xy, i + 1, j, k), c2 = EXY(xy, i, j + 1, k), c3 = EXY(xy, i + 1, j + 1, k);
    const double syz = 0.25 * (a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3);
    const double sxz = 0.25 * (b0 * b0 + b1 * b1 + b2 * b2 + b3 * b3);
    const double sxy = 0.25 * (c0 * c0 + c1 * c1 + c2 * c2 + c3 * c3);
    return sqrt(0.5 * (xx[c] * xx[c] + yy[c] * yy[c] + zz[c] * zz[c]) + syz + sxz + sxy);
}

// tensor_invariant_kernel! 3D (StressKernels.jl:472-487)
__global__ __launch_bounds__(256) void k_tensor_invariant3d(double *__restrict__ II, const double *xx, const double *yy, const double *zz,
                                                            const double *yz, const double *xz, const double *xy, int nx, int ny, int nz)
{
    NODE_IJK(nx, ny)
    II[i + (i64)nx * (j + (i64)ny * k)] = sinv_stag3(xx, yy, zz, yz, xz, xy, nx, ny, i, j, k);
}

// shear2center_kernel! 3D (Interpolations.jl:314-323)
__global__ __launch_bounds__(256) void k_shear2center3d(double *__restrict__ yz_c, double *__restrict__ xz_c, double *__restrict__ xy_c,
                                                        const double *yz, const double *xz, const double *xy, int nx, int ny, int nz)
{
    NODE_IJK(nx, ny)
    const i64 c = i + (i64)nx * (j + (i64)ny * k);
    yz_c[c] = 0.25 * (EYZ(yz, i, j, k) + EYZ(yz, i, j + 1, k) + EYZ(yz, i, j, k + 1) + EYZ(yz, i, j + 1, k + 1));
    xz_c[c] = 0.25 * (EXZ(xz, i, j, k) + EXZ(xz, i + 1, j, k) + EXZ(xz, i, j, k + 1) + EXZ(xz, i + 1, j, k + 1));
    xy_c[c] = 0.25 * (EXY(xy, i, j, k) + EXY(xy, i + 1, j, k) + EXY(xy, i, j + 1, k) + EXY(xy, i + 1, j + 1, k));
}

// compute_vorticity!(ωyz, ωxz, ωxy, V..., _di) (stress_rotation_particles.jl:31-50) over the ni.+1 box
__global__ __launch_bounds__(256) void k_vorticity3d(double *__restrict__ wyz, double *__restrict__ wxz, double *__restrict__ wxy, const double *Vx,
                                                     const double *Vy, const double *Vz, int nx, int ny, int nz, double _dx, double _dy, double _dz)
{
    NODE_IJK(nx + 1, ny + 1)
    if (i < nx) EYZ(wyz, i, j, k) = 0.5 * ((-VZ(i, j, k) + VZ(i, j + 1, k)) * _dy - (-VY(i, j, k) + VY(i, j, k + 1)) * _dz);
    if (j < ny) EXZ(wxz, i, j, k) = 0.5 * ((-VX(i, j, k) + VX(i, j, k + 1)) * _dz - (-VZ(i, j, k) + VZ(i + 1, j, k)) * _dx);
    if (k < nz) EXY(wxy, i, j, k) = 0.5 * ((-VY(i, j, k) + VY(i + 1, j, k)) * _dx - (-VX(i, j, k) + VX(i, j + 1, k)) * _dy);
}

// accumulate_tensor_kernel! 3D alone (StressKernels.jl:394-408)
__global__ __launch_bounds__(256) void k_accumulate_tensor3d(double *__restrict__ II, const double *xx, const double *yy, const double *zz,
                                                             const double *yz, const double *xz, const double *xy, double dt, int nx, int ny, int nz)
{
    NODE_IJK(nx, ny)
    II[i + (i64)nx * (j + (i64)ny * k)] += sinv_stag3(xx, yy, zz, yz, xz, xy, nx, ny, i, j, k) * dt;
}

// accumulate_tensor! + accumulate_vol! (StressKernels.jl:394-431)
__global__ __launch_bounds__(256) void k_vep3_accumulate(const Vep3Args a)
{
    const int nx = a.nx, ny = a.ny;
    NODE_IJK(nx, ny)
    const i64 c = i + (i64)nx * (j + (i64)ny * k);
    a.f.EII_pl[c] += sinv_stag3(a.f.eplxx, a.f.eplyy, a.f.eplzz, a.f.eplyz, a.f.eplxz, a.f.eplxy, nx, ny, i, j, k) * a.dt;
    a.f.EVol_pl[c] += a.dt * a.f.evol_pl[c];
}
#undef VX
#undef VY
#undef VZ

void launch_vep3_visc(hipStream_t s, unsigned gc, const Vep3Args &a, double nu, bool tau)
{
    if (!mat_viscosity_reads_fields(&a.rh)) hipLaunchKernelGGL((k_vep3_visc<false, false>), dim3(gc), dim3(256), 0, s, a, nu);
    else if (tau) hipLaunchKernelGGL((k_vep3_visc<true, true>), dim3(gc), dim3(256), 0, s, a, nu);
    else hipLaunchKernelGGL((k_vep3_visc<true, false>), dim3(gc), dim3(256), 0, s, a, nu);
}

jrx_status check_vep3(jrx_handle *h, const jrx_vep3d_fields *f, const jrx_rheology *rh, const jrx_vep3d_params *p)
{
    if (!h) return JRX_ERR_ARG;
    if (!f || !rh || !p) return jrx_fail(h, JRX_ERR_ARG, "null VEP argument");
    JRX_TRY(jrx_check_device(h));
    if (p->nx < 3 || p->ny < 3 || p->nz < 3) return jrx_fail(h, JRX_ERR_ARG, "3D Stokes needs at least 3 cells per dimension");
    if ((double)(p->nx + 2) * (double)(p->ny + 2) * (double)(p->nz + 2) >= 536870912.0)
        return jrx_fail(h, JRX_ERR_UNSUPPORTED, "local block too large: every array must stay below 4 GiB (32-bit byte offsets)");
    if (rh->nphase < 1 || rh->nphase > JRX_MAXPHASE) return jrx_fail(h, JRX_ERR_ARG, "nphase must be in 1..%d", JRX_MAXPHASE);
    const void *req[] = {f->P, f->P0, f->divV, f->Q, f->Vx, f->Vy, f->Vz, f->Ux, f->Uy, f->Uz, f->exx, f->eyy, f->ezz, f->eyz, f->exz, f->exy,
                         f->eplxx, f->eplyy, f->eplzz, f->eplyz, f->eplxz, f->eplxy, f->txx, f->tyy, f->tzz, f->tyz, f->txz, f->txy,
                         f->tyz_c, f->txz_c, f->txy_c, f->tII, f->toxx, f->toyy, f->tozz, f->toyz, f->toxz, f->toxy, f->toyz_c, f->toxz_c,
                         f->toxy_c, f->eta, f->eta_vep, f->EII_pl, f->evol_pl, f->EVol_pl, f->fx, f->fy, f->fz, f->RP, f->Rx, f->Ry, f->Rz,
                         f->phase_c, f->phase_yz, f->phase_xz, f->phase_xy};
    for (const void *q : req)
        if (!q) return jrx_fail(h, JRX_ERR_ARG, "a required 3D VEP field pointer is NULL");
    return JRX_OK;
}

Vep3Args make_vep3(const jrx_vep3d_fields *f, const jrx_rheology *rh, const jrx_vep3d_params *p)
{
    Vep3Args a;
    memset(&a, 0, sizeof(a));
    a.f = *f; a.rh = *rh;
    a._dx = p->_dx; a._dy = p->_dy; a._dz = p->_dz; a.dt = p->dt; a.r = p->r; a.theta_dtau = p->theta_dtau; a.rel = p->lambda_relaxation;
    a.nu = p->viscosity_relaxation; a.cut_lo = p->cutoff_lo; a.cut_hi = p->cutoff_hi;
    a.nx = (int)p->nx; a.ny = (int)p->ny; a.nz = (int)p->nz;
    a.soft = mat_has_softening(rh);
    a.tg = p->T_ghosted != 0;
    a.nt = false;
    a.obs = true;
    return a;
}

struct EdgeN { i64 yz, xz, xy; };
EdgeN edge_counts(const jrx_vep3d_params *p)
{
    return EdgeN{(i64)p->nx * (p->ny + 1) * (p->nz + 1), (i64)(p->nx + 1) * p->ny * (p->nz + 1), (i64)(p->nx + 1) * (p->ny + 1) * p->nz};
}

// the three edge passes, the commit of the new edge stresses, then the centre pass
// commit = false: the caller adopts a.tnew as the current edge-stress arrays (pointer swap) instead of copying them back
// which: 0 = both passes, 1 = the edge pass alone, 2 = the centre pass alone (the multi-rank driver exchanges the new edge stresses in between)
// loop: called by the solve loop -- the form of the edge pass then follows the grid size (the kernel-level entry points keep the form the switch names)
jrx_status launch_vep3_stress(jrx_handle *h, hipStream_t s, const Vep3Args &a, const jrx_vep3d_params *p, bool commit = true, int which = 0, bool loop = false)
{
    const int nx = a.nx, ny = a.ny, nz = a.nz;
    // Small grids (the reference's own 3D tests run at 16^3 .. 32^3): the z-marching edge kernel has a few hundred blocks, each a chain of 16 dependent planes, and the
    // one-node-per-thread kernel is the faster form -- 16^3 14.1 k -> 19.4 k it/s, 32^3 13.0 k -> 18.1 k, 48^3 10.5 k -> 12.4 k; equal at 56^3, z-marching from there on
    // (scripts/bench_vep3d_sizes.py, profiles/r03_vep3d_small_grids.txt)
    const int edges = (loop && h->vep3_edges == 4 && (double)(nx + 1) * (ny + 1) * (nz + 1) <= 125000.0) ? 0 : h->vep3_edges;
    if (which != 2) {
    const bool p4 = h->vep3_map, xs = h->vep3_xcd;     // options "vep3_map", "vep3_xcd" (XCD slab order: +1-2 % measured)
    if (edges >= 1 && a.rh.nphase <= 4) {
        // option "vep3_edges": 0 one node per thread (k_vep3_edges; also the form more than 4 phases use), 1 (default) the z-marching
        // kernel with one family per block and the three blocks of a tile on one XCD, 2 the same kernel as one launch per family (A/B: the L2 sharing)
        const int cfg = h->vep3_cfg ? h->vep3_cfg : 162;           // option "vep3_cfg" = KZ * 10 + min blocks per CU (tuning); 0: KZ by the grid size (below)
        int kz = cfg / 10;
        const int mb = cfg % 10, np_ = a.rh.nphase;
        // lane segments of 62 node columns; a last segment that would be less than 40 % full (256^3: 257 = 4 x 62 + 9) is not launched -- its node columns
        // go to the one-node-per-thread kernel in a thin launch of their own (the two launches write disjoint nodes and read old values only)
        const int nfull = (nx + 1) / 62, rem = (nx + 1) - 62 * nfull;
        // (a thin last segment is only worth a launch of its own behind at least two full ones: 72^3 6.9 k -> 7.3 k, 80^3 5.5 k -> 6.4 k it/s without it, 128^3 2.03 k with against 1.98 k without)
        const bool peel = nfull >= 2 && rem > 0 && rem <= 24 && h->vep3_peel;
        const int nseg = peel ? nfull : (nx + 1 + 61) / 62, ilim = peel ? 62 * nfull : nx + 1;
        // chunk depth of the LDS-sharing form: 16 planes where that gives the chip enough blocks, halved (down to 4) while the launch has fewer than 3000 -- 56^3 12.6 k -> 15.6 k it/s,
        // 64^3 9.6 k -> 10.7 k, 96^3 4.12 k -> 4.33 k, 128^3 +1.4 %; from 160^3 on 16 planes are the best (profiles/r03_vep3d_small_grids.txt)
        if (!h->vep3_cfg && edges == 4 && !a.soft)
            while (kz > 4 && (i64)nseg * (ny + 1) * ((nz + kz) / kz) < 3000) kz /= 2;
        const int ntxy = nseg * ((ny + 1 + 3) / 4), nzc = (nz + 1 + kz - 1) / kz, nt = ntxy * nzc;
        bool ok = false;
        // the peeled columns are independent of the main launch (disjoint nodes written, only old values read); tuning switch "vep3_peel_fork" runs the thin,
        // load-instruction-bound node kernel (0.14 ms for 3.5 % of the nodes at 256^3) on the halo stream beside the main kernel -- measured 266.1 / 267.6 it/s
        // forked vs 269.2 / 269.8 in order (256^3, same box, alternating): off
        const bool fork = peel && h->vep3_peel_fork && s == h->stream && !jrx_comm_active(h);      // (with neighbours the halo stream carries the exchanges)
        hipStream_t ps = fork ? h->halo_stream : s;
        if (fork) {
            JRX_HIP(h, hipEventRecord(h->ev[5], s));
            JRX_HIP(h, hipStreamWaitEvent(ps, h->ev[5], 0));
        }
        if (peel) {
            const dim3 gp((unsigned)(((i64)rem * (ny + 1) + 63) / 64), (unsigned)((nz + 1 + 3) / 4));
            if (a.soft) hipLaunchKernelGGL((k_vep3_edges<true, false, true>), gp, dim3(256), 0, ps, a, ilim, rem);
            else switch (h->vep3_np_const ? np_ : 0) {
            case 1: hipLaunchKernelGGL((k_vep3_edges<true, false, false, 1>), gp, dim3(256), 0, ps, a, ilim, rem); break;
            case 2: hipLaunchKernelGGL((k_vep3_edges<true, false, false, 2>), gp, dim3(256), 0, ps, a, ilim, rem); break;
            case 3: hipLaunchKernelGGL((k_vep3_edges<true, false, false, 3>), gp, dim3(256), 0, ps, a, ilim, rem); break;
            case 4: hipLaunchKernelGGL((k_vep3_edges<true, false, false, 4>), gp, dim3(256), 0, ps, a, ilim, rem); break;
            default: hipLaunchKernelGGL((k_vep3_edges<true, false, false>), gp, dim3(256), 0, ps, a, ilim, rem);
            }
        }
        struct Join { jrx_handle *h; hipStream_t s, ps; bool on; ~Join() { if (on) { (void)hipEventRecord(h->ev[5], ps); (void)hipStreamWaitEvent(s, h->ev[5], 0); } } } join{h, s, ps, fork};
        if (a.soft && edges == 5) {       // softening laws: the yield function also reads the edge average of EII_pl (a twelfth shared centre array);
                                                  // measured at 256^3: 195.7 it/s through LDS vs 214.9 with one family per block (254 VGPRs either way): only on request
            const int ntxy_l = nseg * (ny + 1), nt_l = ntxy_l * nzc;
            const dim3 gl((unsigned)(((nt_l + 7) / 8) * 8));
#define EZLS(NP_) if (kz == 16 && np_ == NP_) { hipLaunchKernelGGL((k_vep3_edges_zl<16, NP_, true>), gl, dim3(192), 0, s, a, nseg, ntxy_l, nt_l, ilim); ok = true; }
            EZLS(1) EZLS(2) EZLS(3) EZLS(4)
#undef EZLS
        } else if (a.soft) {
            const dim3 gf((unsigned)(((nt + 7) / 8) * 8 * 3));
#define EZS(NP_) if (kz == 16 && np_ == NP_) { hipLaunchKernelGGL((k_vep3_edges_zf<16, NP_, 2, true>), gf, dim3(256), 0, s, a, nseg, ntxy, nt, ilim); ok = true; }
            EZS(1) EZS(2) EZS(3) EZS(4)
#undef EZS
        } else if (edges == 6) {
            const int ntxy_l = nseg * (ny + 1), nt_l = ntxy_l * nzc;
            const dim3 gl((unsigned)(((nt_l + 7) / 8) * 8));
#define EZP(KZ_, NP_) if (kz == KZ_ && np_ == NP_) { hipLaunchKernelGGL((k_vep3_edges_zp<KZ_, NP_>), gl, dim3(256), 0, s, a, nseg, ntxy_l, nt_l, ilim); ok = true; }
            EZP(16, 1) EZP(16, 2) EZP(16, 3) EZP(16, 4)
#undef EZP
        } else if (edges == 3 || edges == 4) {
            const int ntxy_l = nseg * (ny + 1), nt_l = ntxy_l * nzc;
            const dim3 gl((unsigned)(((nt_l + 7) / 8) * 8));
#define EZL(NP_) if (kz == 16 && np_ == NP_) { if (edges == 4) hipLaunchKernelGGL((k_vep3_edges_zl<16, NP_, false, true>), gl, dim3(192), 0, s, a, nseg, ntxy_l, nt_l, ilim); \
                else hipLaunchKernelGGL((k_vep3_edges_zl<16, NP_>), gl, dim3(192), 0, s, a, nseg, ntxy_l, nt_l, ilim); ok = true; }
            EZL(1) EZL(2) EZL(3) EZL(4)
#undef EZL
#define EZLK(KZ_, NP_) if (!ok && kz == KZ_ && np_ == NP_ && edges == 4) { hipLaunchKernelGGL((k_vep3_edges_zl<KZ_, NP_, false, true>), gl, dim3(192), 0, s, a, nseg, ntxy_l, nt_l, ilim); ok = true; }
            EZLK(8, 1) EZLK(8, 2) EZLK(8, 3) EZLK(8, 4) EZLK(4, 1) EZLK(4, 2) EZLK(4, 3) EZLK(4, 4)
#undef EZLK

        } else if (edges == 2) {
            const dim3 g((unsigned)ntxy, (unsigned)nzc);
#define EZ(NP_) if (kz == 16 && np_ == NP_) { hipLaunchKernelGGL((k_vep3_edges_z<16, NP_, 1>), g, dim3(256), 0, s, a, nseg, ilim); \
                hipLaunchKernelGGL((k_vep3_edges_z<16, NP_, 2>), g, dim3(256), 0, s, a, nseg, ilim); \
                hipLaunchKernelGGL((k_vep3_edges_z<16, NP_, 4>), g, dim3(256), 0, s, a, nseg, ilim); ok = true; }
            EZ(1) EZ(2) EZ(3) EZ(4)
#undef EZ
        } else {
            const dim3 gf((unsigned)(((nt + 7) / 8) * 8 * 3));
#define EZG(KZ_, NP_, MB_) if (kz == KZ_ && mb == MB_ && np_ == NP_) { hipLaunchKernelGGL((k_vep3_edges_zf<KZ_, NP_, MB_>), gf, dim3(256), 0, s, a, nseg, ntxy, nt, ilim); ok = true; }
#define EZN(KZ_, MB_) EZG(KZ_, 1, MB_) EZG(KZ_, 2, MB_) EZG(KZ_, 3, MB_) EZG(KZ_, 4, MB_)
            EZN(16, 2) EZN(16, 3) EZN(8, 2)
#undef EZN
#undef EZG
        }
        if (!ok) return jrx_fail(h, JRX_ERR_ARG, "vep3_cfg: no such configuration");
    } else if (a.soft) hipLaunchKernelGGL((k_vep3_edges<true, true, true>), GRID_IJK4(nx + 1, ny + 1, nz + 1), dim3(256), 0, s, a, 0, nx + 1);
    else if (p4 && xs) switch (h->vep3_np_const && a.rh.nphase <= 4 ? a.rh.nphase : 0) {
        case 1: hipLaunchKernelGGL((k_vep3_edges<true, true, false, 1>), GRID_IJK4(nx + 1, ny + 1, nz + 1), dim3(256), 0, s, a, 0, nx + 1); break;
        case 2: hipLaunchKernelGGL((k_vep3_edges<true, true, false, 2>), GRID_IJK4(nx + 1, ny + 1, nz + 1), dim3(256), 0, s, a, 0, nx + 1); break;
        case 3: hipLaunchKernelGGL((k_vep3_edges<true, true, false, 3>), GRID_IJK4(nx + 1, ny + 1, nz + 1), dim3(256), 0, s, a, 0, nx + 1); break;
        case 4: hipLaunchKernelGGL((k_vep3_edges<true, true, false, 4>), GRID_IJK4(nx + 1, ny + 1, nz + 1), dim3(256), 0, s, a, 0, nx + 1); break;
        default: hipLaunchKernelGGL((k_vep3_edges<true, true>), GRID_IJK4(nx + 1, ny + 1, nz + 1), dim3(256), 0, s, a, 0, nx + 1);
    }
    else if (p4) hipLaunchKernelGGL(k_vep3_edges<true>, GRID_IJK4(nx + 1, ny + 1, nz + 1), dim3(256), 0, s, a, 0, nx + 1);
    else hipLaunchKernelGGL(k_vep3_edges<false>, GRID_IJK(nx + 1, ny + 1, nz + 1), dim3(256), 0, s, a, 0, nx + 1);
    JRX_LAUNCH_CHECK(h);
    }
    if (which == 1) return JRX_OK;
    const EdgeN n = edge_counts(p);
    if (commit && which == 0) {
        hipLaunchKernelGGL(k_copy6, dim3(1024), dim3(256), 0, s, a.f.tyz, (const double *)a.tnew[0], n.yz, a.f.txz, (const double *)a.tnew[1], n.xz, a.f.txy,
                           (const double *)a.tnew[2], n.xy, (double *)nullptr, (const double *)nullptr, (i64)0, (double *)nullptr, (const double *)nullptr,
                           (i64)0, (double *)nullptr, (const double *)nullptr, (i64)0);
        JRX_LAUNCH_CHECK(h);
    }
    if (a.soft) hipLaunchKernelGGL(k_vep3_centre<true>, GRID_IJK(nx, ny, nz), dim3(256), 0, s, a);
    else switch (h->vep3_np_const ? a.rh.nphase : 0) {
    case 1: hipLaunchKernelGGL((k_vep3_centre<false, 1>), GRID_IJK(nx, ny, nz), dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL((k_vep3_centre<false, 2>), GRID_IJK(nx, ny, nz), dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL((k_vep3_centre<false, 3>), GRID_IJK(nx, ny, nz), dim3(256), 0, s, a); break;
    case 4: hipLaunchKernelGGL((k_vep3_centre<false, 4>), GRID_IJK(nx, ny, nz), dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL(k_vep3_centre<false>, GRID_IJK(nx, ny, nz), dim3(256), 0, s, a);
    }
    JRX_LAUNCH_CHECK(h);
    return JRX_OK;
}

jrx_stokes3d_fields view3d(const jrx_vep3d_fields *f)
{
    jrx_stokes3d_fields g;
    memset(&g, 0, sizeof(g));
    g.P = f->P; g.P0 = f->P0; g.divV = f->divV; g.Q = f->Q; g.Vx = f->Vx; g.Vy = f->Vy; g.Vz = f->Vz; g.Ux = f->Ux; g.Uy = f->Uy; g.Uz = f->Uz;
    g.txx = f->txx; g.tyy = f->tyy; g.tzz = f->tzz; g.tyz = f->tyz; g.txz = f->txz; g.txy = f->txy;
    g.toxx = f->toxx; g.toyy = f->toyy; g.tozz = f->tozz; g.toyz = f->toyz; g.toxz = f->toxz; g.toxy = f->toxy;
    g.exx = f->exx; g.eyy = f->eyy; g.ezz = f->ezz; g.eyz = f->eyz; g.exz = f->exz; g.exy = f->exy;
    g.eta = f->eta; g.K = f->eta; g.G = f->eta;      // K, G are not read by the velocity sweep
    g.fx = f->fx; g.fy = f->fy; g.fz = f->fz; g.RP = f->RP; g.Rx = f->Rx; g.Ry = f->Ry; g.Rz = f->Rz;
    return g;
}

}   // namespace

extern "C" {

jrx_status jrx_vep3d_update_stresses(jrx_handle *h, const jrx_vep3d_fields *f, const double *theta, double *lambda, double *const lambda_v[3],
                                     const jrx_rheology *rh, const jrx_vep3d_params *p)
{
    JRX_TRY(check_vep3(h, f, rh, p));
    if (!theta || !lambda || !lambda_v || !lambda_v[0] || !lambda_v[1] || !lambda_v[2]) return jrx_fail(h, JRX_ERR_ARG, "θ / λ / λv is NULL");
    const EdgeN n = edge_counts(p);
    JRX_TRY(jrx_ensure_etatau(h, (size_t)(n.yz + n.xz + n.xy)));
    Vep3Args a = make_vep3(f, rh, p);
    a.theta = const_cast<double *>(theta); a.lam = lambda;
    for (int t = 0; t < 3; t++) a.lamv[t] = lambda_v[t];
    a.tnew[0] = h->etatau; a.tnew[1] = a.tnew[0] + n.yz; a.tnew[2] = a.tnew[1] + n.xz;
    JRX_TRY(launch_vep3_stress(h, h->stream, a, p));
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

static jrx_status vep3_viscosity(jrx_handle *h, const jrx_vep3d_fields *f, const jrx_rheology *rh, const jrx_vep3d_params *p, double nu, bool tau)
{
    if (!h) return JRX_ERR_ARG;
    if (!f || !rh || !p || !f->eta || !f->phase_c) return jrx_fail(h, JRX_ERR_ARG, "compute_viscosity!: null argument");
    if (mat_viscosity_reads_invariant(rh)) {
        const void *need[] = {f->exx, f->eyy, f->ezz, f->eyz, f->exz, f->exy, f->txx, f->tyy, f->tzz, f->tyz, f->txz, f->txy, f->P};
        for (const void *q : need)
            if (!q) return jrx_fail(h, JRX_ERR_ARG, "compute_viscosity!: a power-law creep reads stokes.ε / stokes.τ and P");
    } else if (mat_viscosity_reads_fields(rh) && !f->P) return jrx_fail(h, JRX_ERR_ARG, "compute_viscosity!: the creep law reads P");
    Vep3Args a = make_vep3(f, rh, p);
    const i64 n = (i64)p->nx * p->ny * p->nz;
    launch_vep3_visc(h->stream, (unsigned)((n + 255) / 256), a, nu, tau);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}
jrx_status jrx_vep3d_compute_viscosity(jrx_handle *h, const jrx_vep3d_fields *f, const jrx_rheology *rh, const jrx_vep3d_params *p, double nu)
{
    return vep3_viscosity(h, f, rh, p, nu, false);
}
jrx_status jrx_vep3d_compute_viscosity_tauII(jrx_handle *h, const jrx_vep3d_fields *f, const jrx_rheology *rh, const jrx_vep3d_params *p, double nu)
{
    return vep3_viscosity(h, f, rh, p, nu, true);
}

jrx_status jrx_tensor_invariant3d(jrx_handle *h, double *II, const double *xx, const double *yy, const double *zz, const double *yz,
                                  const double *xz, const double *xy, int64_t nx, int64_t ny, int64_t nz)
{
    if (!h) return JRX_ERR_ARG;
    if (!II || !xx || !yy || !zz || !yz || !xz || !xy || nx < 1 || ny < 1 || nz < 1) return jrx_fail(h, JRX_ERR_ARG, "tensor_invariant!: bad argument");
    hipLaunchKernelGGL(k_tensor_invariant3d, GRID_IJK(nx, ny, nz), dim3(256), 0, h->stream, II, xx, yy, zz, yz, xz,
                       xy, (int)nx, (int)ny, (int)nz);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_shear2center3d(jrx_handle *h, double *yz_c, double *xz_c, double *xy_c, const double *yz, const double *xz, const double *xy,
                              int64_t nx, int64_t ny, int64_t nz)
{
    if (!h) return JRX_ERR_ARG;
    if (!yz_c || !xz_c || !xy_c || !yz || !xz || !xy || nx < 1 || ny < 1 || nz < 1) return jrx_fail(h, JRX_ERR_ARG, "shear2center!: bad argument");
    hipLaunchKernelGGL(k_shear2center3d, GRID_IJK(nx, ny, nz), dim3(256), 0, h->stream, yz_c, xz_c, xy_c, yz, xz, xy, (int)nx, (int)ny, (int)nz);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_accumulate_tensor3d(jrx_handle *h, double *II, const double *xx, const double *yy, const double *zz, const double *yz, const double *xz,
                                   const double *xy, double dt, int64_t nx, int64_t ny, int64_t nz)
{
    if (!h) return JRX_ERR_ARG;
    if (!II || !xx || !yy || !zz || !yz || !xz || !xy || nx < 1 || ny < 1 || nz < 1) return jrx_fail(h, JRX_ERR_ARG, "accumulate_tensor!: bad argument");
    hipLaunchKernelGGL(k_accumulate_tensor3d, GRID_IJK(nx, ny, nz), dim3(256), 0, h->stream, II, xx, yy, zz, yz, xz, xy, dt, (int)nx, (int)ny, (int)nz);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_compute_vorticity3d(jrx_handle *h, double *wyz, double *wxz, double *wxy, const double *Vx, const double *Vy, const double *Vz,
                                   int64_t nx, int64_t ny, int64_t nz, double _dx, double _dy, double _dz)
{
    if (!h) return JRX_ERR_ARG;
    if (!wyz || !wxz || !wxy || !Vx || !Vy || !Vz || nx < 1 || ny < 1 || nz < 1) return jrx_fail(h, JRX_ERR_ARG, "compute_vorticity!: bad argument");
    hipLaunchKernelGGL(k_vorticity3d, GRID_IJK(nx + 1, ny + 1, nz + 1), dim3(256), 0, h->stream, wyz, wxz, wxy, Vx, Vy, Vz, (int)nx, (int)ny, (int)nz, _dx, _dy,
                       _dz);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_stokes3d_vep_solve(jrx_handle *h, const jrx_vep3d_fields *f, const jrx_rheology *rh, const jrx_vep3d_params *p,
                                  jrx_solve_result *res)
{
    JRX_TRY(check_vep3(h, f, rh, p));
    if (!res) return jrx_fail(h, JRX_ERR_ARG, "null result");
    if (p->nout < 1) return jrx_fail(h, JRX_ERR_ARG, "nout must be >= 1");
    const int nx = (int)p->nx, ny = (int)p->ny, nz = (int)p->nz;
    const size_t n = (size_t)nx * ny * nz;
    const EdgeN ne = edge_counts(p);
    hipStream_t s = h->stream;
    // library scratch: ητ, θ, λ, K, G, a second ητ, the phase-averaged η of the linear laws (centres), λv and the new edge stresses (edges), carved out of one allocation
    JRX_TRY(jrx_ensure_etatau(h, 11 * n + 2 * (size_t)(ne.yz + ne.xz + ne.xy)));
    double *etatau = h->etatau, *theta = etatau + n, *lam = theta + n, *Kc = lam + n, *Gc = Kc + n, *etatau_next = Gc + n;
    Vep3Args a = make_vep3(f, rh, p);
    a.nt = h->vep3_nt;
    a.theta = theta; a.etatau = etatau; a.Kc = Kc; a.Gc = Gc; a.lam = lam;
    double *eta_lin = etatau_next + n;
    a.lamv[0] = eta_lin + n; a.lamv[1] = a.lamv[0] + ne.yz; a.lamv[2] = a.lamv[1] + ne.xz;
    a.tnew[0] = a.lamv[2] + ne.xy; a.tnew[1] = a.tnew[0] + ne.yz; a.tnew[2] = a.tnew[1] + ne.xz;
    double *const cset[3] = {a.tnew[2] + ne.xy, a.tnew[2] + ne.xy + n, a.tnew[2] + ne.xy + 2 * n};      // second set of τxx, τyy, τzz (see `fuse` / `fork` below)
    double *const eta2 = cset[2] + n;                                                                   // second η (see `fuse`)
    jrx_stokes3d_fields g = view3d(f);
    jrx_stokes3d_params q;
    memset(&q, 0, sizeof(q));
    q.nx = nx; q.ny = ny; q.nz = nz; q.nxg = p->nxg; q.nyg = p->nyg; q.nzg = p->nzg; q._dx = p->_dx; q._dy = p->_dy; q._dz = p->_dz;
    q.dt = p->dt; q.r = p->r; q.theta_dtau = p->theta_dtau; q.eta_dtau = p->eta_dtau;
    q.free_slip = p->free_slip; q.no_slip = p->no_slip; q.periodic = p->periodic;
    for (int d = 0; d < 3; d++) q.b_width[d] = p->b_width[d];
    const unsigned gc = (unsigned)((n + 255) / 256);
    const dim3 gv = GRID_IJK(nx + 1, ny + 1, nz + 1), g0 = GRID_IJK(nx, ny, nz);
    constexpr int PRE_KZ = 8;        // planes per thread of k_vep3_pre
    const dim3 gpre((unsigned)(((i64)(nx + 1) * (ny + 1) + 255) / 256), (unsigned)((nz + 1 + PRE_KZ - 1) / PRE_KZ));

    JRX_HIP(h, hipMemcpyAsync(f->P0, f->P, n * sizeof(double), hipMemcpyDeviceToDevice, s));        // @copy stokes.P0 stokes.P
    JRX_HIP(h, hipMemcpyAsync(theta, f->P, n * sizeof(double), hipMemcpyDeviceToDevice, s));        // θ = deepcopy(stokes.P)
    JRX_HIP(h, hipMemsetAsync(lam, 0, n * sizeof(double), s));
    JRX_HIP(h, hipMemsetAsync(a.lamv[0], 0, (size_t)(ne.yz + ne.xz + ne.xy) * sizeof(double), s));
    const bool lin = !mat_viscosity_reads_fields(rh);       // η of the laws depends on the phase ratios only: average it once per solve (update_viscosity_τII! then reads one array instead of the ratios)
    hipLaunchKernelGGL(k_vep3_phase_avg, dim3(gc), dim3(256), 0, s, Kc, Gc, a, rh->has_density != 0, lin ? eta_lin : (double *)nullptr);
    if (lin) a.eta_lin = eta_lin;
    launch_vep3_visc(s, gc, a, 1.0, false);                                                          // compute_viscosity! :507 (εII form)
    JRX_LAUNCH_CHECK(h);
    const bool upd_rho = rh->has_density && !mat_density_is_constant(rh);       // update_ρg!: a no-op for constant densities
    // body-force arrays that hold nothing but +0.0 (ρg_x, ρg_y of every model with gravity along z; all three in ShearBand3D.jl:114) are not loaded by the velocity sweep of
    // unobserved iterations (k_velocity3d_zb, NOF; x - (+0.0) = x: same bits).  ρg_z is the library's to write when the rheology carries densities (compute_ρg! above, update_ρg!)
    int vnof = 0;
    JRX_TRY(jrx3d_forces_zero(h, s, f->fx, f->fy, f->fz, (int64_t)n, &vnof));
    if (rh->has_density && vnof > 1) vnof = 1;
    const bool ubc = p->displacement_bcs != 0;
    if (ubc) {    // displacement2velocity!(stokes, dt, flow_bcs) (Stokes3D.jl:509): V = U * inv(dt)
        hipLaunchKernelGGL(k_scale3, dim3(2048), dim3(256), 0, s, f->Vx, (const double *)f->Ux, (i64)(nx + 1) * (ny + 2) * (nz + 2), f->Vy,
                           (const double *)f->Uy, (i64)(nx + 2) * (ny + 1) * (nz + 2), f->Vz, (const double *)f->Uz, (i64)(nx + 2) * (ny + 2) * (nz + 1),
                           1.0 / p->dt);
        JRX_LAUNCH_CHECK(h);
    }

    double err_it1 = 1.0, err = INFINITY;
    int64_t iter = 0, cont = 0;
    bool bcs_ordered = false;        // the reference's ordered flow_bcs! passes have run on V at least once
    res->iter = 0; res->nchecks = 0;
    const bool comm = jrx_comm_active(h);
    const int64_t nn[3] = {nx, ny, nz};
    const int rank = jrx_comm_rank(h);
    hipEvent_t t0 = h->ev[6], t1 = h->ev[7];
    JRX_HIP(h, hipEventRecord(t0, s));
    auto keep_going = [&](int64_t it) { return it < 2 || (((err / err_it1) > p->eps_rel && err > p->eps_abs) && it <= p->iterMax); };
    // one iteration without neighbours, enqueued on s; A / G: the kernel arguments and the view the velocity sweep takes (their edge-stress pointers swap)
    // planes per thread of k_vep3_pre: 8 where that gives the chip enough blocks, halved while the launch has fewer than 2048 (16^3: 19.3 k -> 26.0 k it/s with one plane per
    // thread, 32^3 18.1 k -> 23.3 k, 64^3 8.2 k -> 9.0 k, 128^3 +1 % with four; 256^3: eight planes stay the best, 318 against 304 it/s with one; profiles/r03_vep3d_small_grids.txt)
    int prekz = h->vep3_prekz;
    if (prekz == 0)
        for (prekz = 8; prekz > 1 && (i64)gpre.x * ((nz + prekz) / prekz) < 2048; prekz /= 2) {}
    // tuning switch "vep3_fork" (default off: measured equal, profiles/r04_vep3d_fork.txt): the centre pass beside the edge pass (see enqueue_iteration); not inside captured graphs (small grids), not with neighbours
    // (there the halo stream carries the exchanges and the centre pass already runs beside one)
    // "vep3_fuse_pc" (default): pre, viscosity relaxation and centre pass as one kernel AHEAD of the edge pass (k_vep3_prec).  The relaxed η and the new τxx, τyy, τzz go to second arrays -- the neighbours'
    // compute_maxloc! reads the old η, the edge pass averages the old normal stresses -- which are adopted by pointer swap like the edge stresses; every iteration swaps all of them, so the captured graphs of
    // an even number of iterations end where they began.  Laws whose η reads fields (invariants gathered from the neighbours) and ranks with neighbours (ητ is exchanged) keep the three kernels.
    const bool fuse = h->vep3_fuse_pc && !comm && lin && !a.soft;      // (softening laws: 255 VGPRs, one wave per SIMD -- they keep the three kernels)
    const bool fork = !fuse && h->vep3_fork && !comm && !((h->loop_graphs && !ubc && p->periodic == 0 && (double)n <= kGraphCells3D));
    double *const user_c[3] = {f->txx, f->tyy, f->tzz};
    // ordered_: flow_bcs! in the reference's pass order (always on observed iterations; also on the iteration BEFORE one, whose ghost edges and corners velocity2displacement! copies into U --
    // the one-launch form of the faces leaves those entries, which no stencil reads, to whichever thread wrote last)
    auto enqueue_iteration = [&](Vep3Args &A, jrx_stokes3d_fields &G, bool diag_, bool ordered_ = false) -> jrx_status {
        A.obs = diag_ || h->vep_store_all;
        if (fuse) {
            for (int c_ = 0; c_ < 3; c_++) A.cnew[c_] = (c_ == 0 ? A.f.txx : (c_ == 1 ? A.f.tyy : A.f.tzz)) == cset[c_] ? user_c[c_] : cset[c_];
            double *const eta_out = A.f.eta == f->eta ? eta2 : f->eta;
            // thread map of the fused kernel: 64 x 4 tiles of node columns where a plane has enough of them ("vep3_prec_tile" = 2, default: from 16,384 node columns per plane), else 256
            // consecutive nodes of the flattened plane.  With tiles the rows of a velocity / η window that a block's rows share are requested by waves of one block: 35.4 -> 30.1 fetched passes,
            // 256^3 350 -> 365 it/s (128^3 +3 %, 160^3 +4 %, 224^3 +5 %, 320^3 +2 %); on small planes the partly filled tiles cost more: 96^3 -7 %, 64^3 -8 %, 32^3 -6 % (gpurun_out/r04pt)
            const bool tiled = h->vep3_prec_tile == 1 || (h->vep3_prec_tile == 2 && (i64)(nx + 1) * (ny + 1) >= 16384);
            const int tntx = tiled ? (nx + 1 + 63) / 64 : 0;
            const dim3 gpc(tntx ? (unsigned)(tntx * ((ny + 1 + 3) / 4)) : gpre.x, (unsigned)((nz + 1 + prekz - 1) / prekz));
            const int npc = (h->vep3_np_const && !upd_rho) ? A.rh.nphase : 0;
            switch ((npc <= 4 ? npc : 0) * 2 + (A.obs ? 1 : 0) + (upd_rho ? 100 : 0)) {
#define PREC(NP_, OBS_) case NP_ * 2 + OBS_: hipLaunchKernelGGL((k_vep3_prec<false, false, NP_, OBS_ != 0>), gpc, dim3(256), 0, s, A, eta_out, prekz, tntx); break;
            PREC(0, 0) PREC(0, 1) PREC(1, 0) PREC(1, 1) PREC(2, 0) PREC(2, 1) PREC(3, 0) PREC(3, 1) PREC(4, 0) PREC(4, 1)
#undef PREC
            case 100: hipLaunchKernelGGL((k_vep3_prec<false, true, 0, false>), gpc, dim3(256), 0, s, A, eta_out, prekz); break;
            default: hipLaunchKernelGGL((k_vep3_prec<false, true, 0, true>), gpc, dim3(256), 0, s, A, eta_out, prekz);
            }
            JRX_LAUNCH_CHECK(h);
            A.f.eta = eta_out;
            JRX_TRY(launch_vep3_stress(h, s, A, p, false, 1, true));       // edge pass: the relaxed η, the OLD normal stresses
            A.f.txx = A.cnew[0]; A.f.tyy = A.cnew[1]; A.f.tzz = A.cnew[2];
            A.cnew[0] = A.cnew[1] = A.cnew[2] = nullptr;
            G.txx = A.f.txx; G.tyy = A.f.tyy; G.tzz = A.f.tzz;
            G.eta = G.K = G.G = A.f.eta;
            h->stat_vep3_fused++;
        } else {
        if (upd_rho) hipLaunchKernelGGL((k_vep3_pre<true, true, PRE_KZ>), gpre, dim3(256), 0, s, A);
        else if (prekz == 16) hipLaunchKernelGGL((k_vep3_pre<true, false, 16>), dim3(gpre.x, (unsigned)((nz + 1 + 15) / 16)), dim3(256), 0, s, A);
        else if (prekz == 32) hipLaunchKernelGGL((k_vep3_pre<true, false, 32>), dim3(gpre.x, (unsigned)((nz + 1 + 31) / 32)), dim3(256), 0, s, A);
        else if (prekz == 4) hipLaunchKernelGGL((k_vep3_pre<true, false, 4>), dim3(gpre.x, (unsigned)((nz + 1 + 3) / 4)), dim3(256), 0, s, A);
        else if (prekz == 2) hipLaunchKernelGGL((k_vep3_pre<true, false, 2>), dim3(gpre.x, (unsigned)((nz + 1 + 1) / 2)), dim3(256), 0, s, A);
        else if (prekz == 1) hipLaunchKernelGGL((k_vep3_pre<true, false, 1>), dim3(gpre.x, (unsigned)(nz + 1)), dim3(256), 0, s, A);
        else hipLaunchKernelGGL((k_vep3_pre<true, false, PRE_KZ>), gpre, dim3(256), 0, s, A);        // compute_maxloc! folded in
        launch_vep3_visc(s, gc, A, p->viscosity_relaxation, true);                               // update_viscosity_τII! :541
        JRX_LAUNCH_CHECK(h);
        if (fork) {
            // Edge pass and centre pass of update_stresses_center_vertex_ps! side by side on two streams.  They are independent except that the edge pass averages the OLD
            // τxx, τyy, τzz to its nodes while the centre pass overwrites them: the centre pass writes a second set, adopted by pointer swap like the edge stresses.  The
            // edge kernel is latency-bound (4.0 TB/s), the centre kernel streams (5.3 TB/s): the hope was that together they would fill the memory system -- they do not (profiles/r04_vep3d_fork.txt)
            hipStream_t hs = h->halo_stream;
            for (int c_ = 0; c_ < 3; c_++) A.cnew[c_] = (c_ == 0 ? A.f.txx : (c_ == 1 ? A.f.tyy : A.f.tzz)) == cset[c_] ? user_c[c_] : cset[c_];
            JRX_HIP(h, hipEventRecord(h->ev[3], s));
            JRX_HIP(h, hipStreamWaitEvent(hs, h->ev[3], 0));
            JRX_TRY(launch_vep3_stress(h, hs, A, p, false, 2, true));      // centre pass
            JRX_TRY(launch_vep3_stress(h, s, A, p, false, 1, true));       // edge pass
            JRX_HIP(h, hipEventRecord(h->ev[4], hs));
            JRX_HIP(h, hipStreamWaitEvent(s, h->ev[4], 0));
            A.f.txx = A.cnew[0]; A.f.tyy = A.cnew[1]; A.f.tzz = A.cnew[2];
            A.cnew[0] = A.cnew[1] = A.cnew[2] = nullptr;
            G.txx = A.f.txx; G.tyy = A.f.tyy; G.tzz = A.f.tzz;
        } else JRX_TRY(launch_vep3_stress(h, s, A, p, false, 0, true));
        }
        // the new edge stresses become the current ones: swap the pointers instead of copying three arrays back
        { double *t0_ = A.f.tyz; A.f.tyz = A.tnew[0]; A.tnew[0] = t0_; }
        { double *t1_ = A.f.txz; A.f.txz = A.tnew[1]; A.tnew[1] = t1_; }
        { double *t2_ = A.f.txy; A.f.txy = A.tnew[2]; A.tnew[2] = t2_; }
        G.tyz = A.f.tyz; G.txz = A.f.txz; G.txy = A.f.txy;
        JRX_TRY(jrx3d_velocity_sweep(h, s, &G, A.etatau, &q, diag_, vnof));
        if (diag_) JRX_TRY(jrx3d_scaleU(h, s, &G, &q));
        // flow_bcs!: on V, or -- DisplacementBoundaryConditions -- on U = V dt, which the next iteration overwrites (only observable when U is)
        if (!ubc && !diag_ && !ordered_ && bcs_ordered && p->periodic == 0) JRX_TRY(jrx3d_bcs_faces(h, s, f->Vx, f->Vy, f->Vz, nx, ny, nz, p->free_slip, p->no_slip));
        else if (!ubc) { JRX_TRY(jrx3d_bcs(h, s, f->Vx, f->Vy, f->Vz, nx, ny, nz, p->free_slip, p->no_slip, p->periodic)); bcs_ordered = true; }
        else if (diag_) JRX_TRY(jrx3d_bcs(h, s, f->Ux, f->Uy, f->Uz, nx, ny, nz, p->free_slip, p->no_slip, p->periodic));
        return JRX_OK;
    };
    // small grids are launch-bound (seven dependent launches per iteration): runs of unobserved iterations replay as a captured graph of GIT iterations
    // (an even count, so the swapped edge-stress pointers end where they began; one graph per parity; option "loop_graphs")
    constexpr int GIT = 16;
    GraphExecs gexec;
    bool graphs = h->loop_graphs && !comm && !ubc && p->periodic == 0 && (double)n <= kGraphCells3D;
    while (keep_going(iter)) {
        if (graphs && iter >= 2 && bcs_ordered) {
            int64_t nxt = ((iter / p->nout) + 1) * p->nout;         // observed iterations: the multiples of nout and iteration iterMax + 1
            if (nxt > p->iterMax + 1) nxt = p->iterMax + 1;
            int64_t run = nxt - 2 - iter;                           // (the iteration before an observed one runs its boundary conditions in the reference's pass order: not part of a replay)
            if (run >= GIT) {
                const int par = a.f.tyz == f->tyz ? 0 : 1;
                if (!gexec[par]) {
                    const int64_t stat0 = h->stat_vep3_fused;
                    JRX_TRY(jrx_capture_graph(s, &gexec[par], [&]() -> jrx_status {
                        Vep3Args A = a;
                        jrx_stokes3d_fields G = g;
                        for (int r_ = 0; r_ < GIT; r_++) JRX_TRY(enqueue_iteration(A, G, false));
                        return JRX_OK;
                    }));
                    h->stat_vep3_fused = stat0;
                    if (!gexec[par]) graphs = false;
                }
                if (gexec[par]) {
                    while (run >= GIT) {
                        JRX_HIP(h, hipGraphLaunch(gexec[par], s));
                        iter += GIT; run -= GIT;
                        h->stat_graph_replays++;
                    }
                    continue;
                }
            }
        }
        const int64_t it1 = iter + 1;
        const bool check = (it1 % p->nout == 0) && it1 > 1;
        const bool diag = check || !keep_going(it1);      // R and U are only observable after such an iteration
        a.obs = diag || h->vep_store_all;
        const bool pre_diag = ((it1 + 1) % p->nout == 0) || it1 + 1 > p->iterMax;       // the next iteration, if there is one, is observed
        if (comm) {
            // Hidden communication.  The reference hides update_halo!(V) behind compute_V! (@hide_communication, Stokes3D.jl:582-597);
            // its other two exchanges of the iteration block.  Here all three run on the halo stream beside kernels that do not depend on them:
            //   update_halo!(ητ) of the NEXT iteration (compute_maxloc! of the η that update_viscosity_τII! has just left, into a second ητ array)
            //                        beside the edge pass of the stress update,
            //   update_halo!(τ.yz, τ.xz, τ.xy) beside the centre pass (which reads no edge stress),
            //   update_halo!(V) behind the boundary slabs of compute_V!, beside its interior.
            // Same kernels, same operands: results are those of the serial order, bit for bit (tests/test_gpu_two_blocks.py, tests/test_gpu_halo.py).
            // tuning switch "vep3_hide_comm": 2 (default) as described; 1 = ητ and the edge stresses hidden, update_halo!(V) in order behind the whole sweep;
            // 0 = the same sequence on the compute stream alone (A/B)
            const bool hide = h->vep3_hide_comm >= 2;
            hipStream_t hs = h->vep3_hide_comm >= 1 ? h->halo_stream : s;
            if (iter == 0) {        // ητ of the first iteration: nothing to hide it behind
                double *cur = const_cast<double *>(a.etatau);
                hipLaunchKernelGGL(k_maxloc, dim3((unsigned)(((i64)nx * ny + 255) / 256), (unsigned)nz), dim3(256), 0, s, cur, (const double *)f->eta, nx, ny, nz);
                double *arrs[1] = {cur};
                const int64_t ext[1][3] = {{nx, ny, nz}};
                JRX_TRY(jrx_halo_exchange(h, s, 1, arrs, ext, nn));      // update_halo!(ητ) (Stokes3D.jl:515)
            }
            // "vep3_fuse_pc" with neighbours: pre, viscosity relaxation and centre pass as one kernel here too (k_vep3_prec<ML = false>: ητ is the exchanged array); the centre pass then no longer
            // runs beside update_halo!(τ.yz, τ.xz, τ.xy), which waits in line instead -- three thin planes against the ~0.3 ms the fusion saves at 256^3
            const bool fuse_c = h->vep3_fuse_pc && lin && !a.soft && !upd_rho && h->vep3_np_const && a.rh.nphase <= 4;
            if (fuse_c) {
                for (int c_ = 0; c_ < 3; c_++) a.cnew[c_] = (c_ == 0 ? a.f.txx : (c_ == 1 ? a.f.tyy : a.f.tzz)) == cset[c_] ? user_c[c_] : cset[c_];
                double *const eta_out = a.f.eta == f->eta ? eta2 : f->eta;
                // the same thread map as without neighbours: 64 x 4 tiles of node columns where a plane has enough of them ("vep3_prec_tile")
                const bool tiled_c = h->vep3_prec_tile == 1 || (h->vep3_prec_tile == 2 && (i64)(nx + 1) * (ny + 1) >= 16384);
                const int tntx_c = tiled_c ? (nx + 1 + 63) / 64 : 0;
                const dim3 gpc(tntx_c ? (unsigned)(tntx_c * ((ny + 1 + 3) / 4)) : gpre.x, (unsigned)((nz + 1 + PRE_KZ - 1) / PRE_KZ));
                switch (a.rh.nphase * 2 + (a.obs ? 1 : 0)) {
#define PRECC(NP_, OBS_) case NP_ * 2 + OBS_: hipLaunchKernelGGL((k_vep3_prec<false, false, NP_, OBS_ != 0, false>), gpc, dim3(256), 0, s, a, eta_out, PRE_KZ, tntx_c); break;
                PRECC(1, 0) PRECC(1, 1) PRECC(2, 0) PRECC(2, 1) PRECC(3, 0) PRECC(3, 1) PRECC(4, 0) PRECC(4, 1)
#undef PRECC
                }
                a.f.eta = eta_out;
                g.eta = g.K = g.G = a.f.eta;
                h->stat_vep3_fused++;
            } else {
            if (upd_rho) hipLaunchKernelGGL((k_vep3_pre<false, true, PRE_KZ>), gpre, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((k_vep3_pre<false, false, PRE_KZ>), gpre, dim3(256), 0, s, a);
            launch_vep3_visc(s, gc, a, p->viscosity_relaxation, true);                               // update_viscosity_τII! :541
            }
            JRX_LAUNCH_CHECK(h);
            JRX_HIP(h, hipEventRecord(h->ev[3], s));
            JRX_HIP(h, hipStreamWaitEvent(hs, h->ev[3], 0));
            // in-order pipeline ("vep3_hide_comm" = 0, the default): ητ of the next iteration is first read by that iteration's velocity sweep, so its planes travel with the
            // edge stresses below -- two exchanges (pack, copies, flags, unpack each) per iteration instead of three; same values in the same places
            const bool merged = h->vep3_hide_comm == 0;
            double *const nxt = a.etatau == etatau ? etatau_next : etatau;
            {
                hipLaunchKernelGGL(k_maxloc, dim3((unsigned)(((i64)nx * ny + 255) / 256), (unsigned)nz), dim3(256), 0, hs, nxt, (const double *)a.f.eta, nx, ny, nz);
                JRX_LAUNCH_CHECK(h);
                if (!merged) {
                    double *arrs[1] = {nxt};
                    const int64_t ext[1][3] = {{nx, ny, nz}};
                    JRX_TRY(jrx_halo_exchange(h, hs, 1, arrs, ext, nn));     // update_halo!(ητ) of iteration it1 + 1
                }
            }
            JRX_TRY(launch_vep3_stress(h, s, a, p, false, 1, true));     // edge pass
            { double *t0_ = a.f.tyz; a.f.tyz = a.tnew[0]; a.tnew[0] = t0_; }
            { double *t1_ = a.f.txz; a.f.txz = a.tnew[1]; a.tnew[1] = t1_; }
            { double *t2_ = a.f.txy; a.f.txy = a.tnew[2]; a.tnew[2] = t2_; }
            g.tyz = a.f.tyz; g.txz = a.f.txz; g.txy = a.f.txy;
            JRX_HIP(h, hipEventRecord(h->ev[3], s));
            JRX_HIP(h, hipStreamWaitEvent(hs, h->ev[3], 0));
            {   // update_halo!(τ.yz), (τ.xz), (τ.xy) (Stokes3D.jl:578-580) [+ ητ of the next iteration, see above]
                double *arrs[4] = {a.f.tyz, a.f.txz, a.f.txy, nxt};
                const int64_t ext[4][3] = {{nx, ny + 1, nz + 1}, {nx + 1, ny, nz + 1}, {nx + 1, ny + 1, nz}, {nx, ny, nz}};
                JRX_TRY(jrx_halo_exchange(h, hs, merged ? 4 : 3, arrs, ext, nn));
            }
            JRX_HIP(h, hipEventRecord(h->ev[4], hs));
            if (fuse_c) {       // the centre pass ran inside k_vep3_prec: adopt its normal stresses
                a.f.txx = a.cnew[0]; a.f.tyy = a.cnew[1]; a.f.tzz = a.cnew[2];
                a.cnew[0] = a.cnew[1] = a.cnew[2] = nullptr;
                g.txx = a.f.txx; g.tyy = a.f.tyy; g.tzz = a.f.tzz;
            } else JRX_TRY(launch_vep3_stress(h, s, a, p, false, 2, true));     // centre pass, beside the exchange
            JRX_HIP(h, hipStreamWaitEvent(s, h->ev[4], 0));
            int bc_kind = 3;
            if (!ubc && !diag && !pre_diag && bcs_ordered && p->periodic == 0) bc_kind = 1;
            else if (!ubc) { bc_kind = 0; bcs_ordered = true; }
            else if (diag) bc_kind = 2;
            if (hide) JRX_TRY(jrx3d_velocity_hidden(h, &g, a.etatau, &q, diag, bc_kind));      // joins the two streams
            else {
                JRX_TRY(jrx3d_velocity_sweep(h, s, &g, a.etatau, &q, diag, vnof));
                if (diag) JRX_TRY(jrx3d_scaleU(h, s, &g, &q));
                if (bc_kind == 1) JRX_TRY(jrx3d_bcs_faces(h, s, f->Vx, f->Vy, f->Vz, nx, ny, nz, p->free_slip, p->no_slip));
                else if (bc_kind == 0) JRX_TRY(jrx3d_bcs(h, s, f->Vx, f->Vy, f->Vz, nx, ny, nz, p->free_slip, p->no_slip, p->periodic));
                else if (bc_kind == 2) JRX_TRY(jrx3d_bcs(h, s, f->Ux, f->Uy, f->Uz, nx, ny, nz, p->free_slip, p->no_slip, p->periodic));
                double *arrs[3] = {f->Vx, f->Vy, f->Vz};
                const int64_t ext[3][3] = {{nx + 1, ny + 2, nz + 2}, {nx + 2, ny + 1, nz + 2}, {nx + 2, ny + 2, nz + 1}};
                JRX_TRY(jrx_halo_exchange(h, s, 3, arrs, ext, nn));       // update_halo!(@velocity(stokes)...) (Stokes3D.jl:596)
            }
            a.etatau = a.etatau == etatau ? etatau_next : etatau;
        } else JRX_TRY(enqueue_iteration(a, g, diag, pre_diag));
        iter = it1;
        if (check) {
            JRX_TRY(jrx3d_sumsq(h, s, &g, &q));
            JRX_HIP(h, hipMemcpyAsync(h->h_sums, h->d_sums, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
            JRX_HIP(h, hipStreamSynchronize(s));
            double ss[4] = {h->h_sums[0], h->h_sums[1], h->h_sums[2], h->h_sums[3]};
            JRX_TRY(jrx_allreduce_sum_host(h, ss, 4));                                                // norm_mpi
            const double den = (double)((p->nxg - 1) * (p->nyg - 1) * (p->nzg - 1));                  // Stokes3D.jl:607-612
            const double nRx = sqrt(ss[0]) / den, nRy = sqrt(ss[1]) / den, nRz = sqrt(ss[2]) / den;
            const double nDV = sqrt(ss[3]) / (double)n;                                               // norm_mpi(RP) / length(RP): local length
            err = fmax(fmax(nRx, nRy), fmax(nRz, nDV));
            if (std::isnan(nRx) || std::isnan(nRy) || std::isnan(nRz) || std::isnan(nDV)) err = NAN;
            if (cont < res->cap) {
                if (res->norm_Rx) res->norm_Rx[cont] = nRx;
                if (res->norm_Ry) res->norm_Ry[cont] = nRy;
                if (res->norm_Rz) res->norm_Rz[cont] = nRz;
                if (res->norm_divV) res->norm_divV[cont] = nDV;
                if (res->err_evo1) res->err_evo1[cont] = err;
                if (res->err_evo2) res->err_evo2[cont] = iter;
            }
            if (cont == 0) err_it1 = err;
            cont++;
            if (rank == 0 && ((p->verbose && (err / err_it1) > p->eps_rel && err > p->eps_abs) || iter == p->iterMax))
                printf("iter = %lld, abs_err = %1.3e, rel_err = %1.3e [norm_Rx=%1.3e, norm_Ry=%1.3e, norm_Rz=%1.3e, norm_∇V=%1.3e] \n", (long long)iter,
                       err, err / err_it1, nRx, nRy, nRz, nDV);
            if (std::isnan(err)) {
                // error("NaN(s)"): the current edge stresses may live in the second set -- leave them in the caller's arrays, drain the stream
                if (a.f.eta != f->eta) (void)hipMemcpyAsync(f->eta, a.f.eta, n * sizeof(double), hipMemcpyDeviceToDevice, s);
                if (a.f.txx != f->txx) {
                    (void)hipMemcpyAsync(f->txx, a.f.txx, n * sizeof(double), hipMemcpyDeviceToDevice, s);
                    (void)hipMemcpyAsync(f->tyy, a.f.tyy, n * sizeof(double), hipMemcpyDeviceToDevice, s);
                    (void)hipMemcpyAsync(f->tzz, a.f.tzz, n * sizeof(double), hipMemcpyDeviceToDevice, s);
                }
                if (a.f.tyz != f->tyz) {
                    (void)hipMemcpyAsync(f->tyz, a.f.tyz, (size_t)ne.yz * sizeof(double), hipMemcpyDeviceToDevice, s);
                    (void)hipMemcpyAsync(f->txz, a.f.txz, (size_t)ne.xz * sizeof(double), hipMemcpyDeviceToDevice, s);
                    (void)hipMemcpyAsync(f->txy, a.f.txy, (size_t)ne.xy * sizeof(double), hipMemcpyDeviceToDevice, s);
                }
                (void)hipEventRecord(t1, s);
                (void)hipStreamSynchronize(s);
                float msn = 0.f;
                (void)hipEventElapsedTime(&msn, t0, t1);
                res->iter = iter; res->nchecks = cont < res->cap ? cont : res->cap;
                res->time_s = msn * 1e-3; res->av_time_s = iter > 1 ? res->time_s / (double)(iter - 1) : res->time_s;
                return jrx_fail(h, JRX_ERR_NAN, "NaN(s)");
            }
        }
    }
    JRX_HIP(h, hipEventRecord(t1, s));
    if (a.f.eta != f->eta) {       // the relaxed η of the fused pre / centre kernel
        JRX_HIP(h, hipMemcpyAsync(f->eta, a.f.eta, n * sizeof(double), hipMemcpyDeviceToDevice, s));
        a.f.eta = f->eta; g.eta = g.K = g.G = f->eta;
    }
    if (a.f.txx != f->txx) {       // the same for the normal stresses of the fused / forked centre pass
        JRX_HIP(h, hipMemcpyAsync(f->txx, a.f.txx, n * sizeof(double), hipMemcpyDeviceToDevice, s));
        JRX_HIP(h, hipMemcpyAsync(f->tyy, a.f.tyy, n * sizeof(double), hipMemcpyDeviceToDevice, s));
        JRX_HIP(h, hipMemcpyAsync(f->tzz, a.f.tzz, n * sizeof(double), hipMemcpyDeviceToDevice, s));
        a.f.txx = f->txx; a.f.tyy = f->tyy; a.f.tzz = f->tzz;
        g.txx = f->txx; g.tyy = f->tyy; g.tzz = f->tzz;
    }
    if (a.f.tyz != f->tyz) {       // odd number of swaps: leave the edge stresses in the caller's arrays
        JRX_HIP(h, hipMemcpyAsync(f->tyz, a.f.tyz, (size_t)ne.yz * sizeof(double), hipMemcpyDeviceToDevice, s));
        JRX_HIP(h, hipMemcpyAsync(f->txz, a.f.txz, (size_t)ne.xz * sizeof(double), hipMemcpyDeviceToDevice, s));
        JRX_HIP(h, hipMemcpyAsync(f->txy, a.f.txy, (size_t)ne.xy * sizeof(double), hipMemcpyDeviceToDevice, s));
        a.tnew[0] = a.f.tyz; a.tnew[1] = a.f.txz; a.tnew[2] = a.f.txy;
        a.f.tyz = f->tyz; a.f.txz = f->txz; a.f.txy = f->txy;
        g.tyz = f->tyz; g.txz = f->txz; g.txy = f->txy;
    }
    // epilogue: vorticity, shear2center!, accumulate_tensor!/accumulate_vol!, τ -> τ_o (Stokes3D.jl:640-658)
    if (f->omega_yz && f->omega_xz && f->omega_xy)
        hipLaunchKernelGGL(k_vorticity3d, gv, dim3(256), 0, s, f->omega_yz, f->omega_xz, f->omega_xy, (const double *)f->Vx,
                           (const double *)f->Vy, (const double *)f->Vz, nx, ny, nz, p->_dx, p->_dy, p->_dz);
    if (f->eyz_c && f->exz_c && f->exy_c)
        hipLaunchKernelGGL(k_shear2center3d, g0, dim3(256), 0, s, f->eyz_c, f->exz_c, f->exy_c, (const double *)f->eyz, (const double *)f->exz,
                           (const double *)f->exy, nx, ny, nz);
    if (f->eplyz_c && f->eplxz_c && f->eplxy_c)
        hipLaunchKernelGGL(k_shear2center3d, g0, dim3(256), 0, s, f->eplyz_c, f->eplxz_c, f->eplxy_c, (const double *)f->eplyz,
                           (const double *)f->eplxz, (const double *)f->eplxy, nx, ny, nz);
    if (f->deyz_c && f->dexz_c && f->dexy_c && f->deyz && f->dexz && f->dexy)
        hipLaunchKernelGGL(k_shear2center3d, g0, dim3(256), 0, s, f->deyz_c, f->dexz_c, f->dexy_c, (const double *)f->deyz, (const double *)f->dexz,
                           (const double *)f->dexy, nx, ny, nz);
    hipLaunchKernelGGL(k_vep3_accumulate, g0, dim3(256), 0, s, a);
    JRX_LAUNCH_CHECK(h);
    const i64 nc = (i64)n;
    h->opv.valid = false;      // τ_o is written: a cached verdict of the 3D visco-elastic operand pass (option operand_cache) may describe these arrays
    hipLaunchKernelGGL(k_copy6, dim3(1024), dim3(256), 0, s, f->toxx, (const double *)f->txx, nc, f->toyy, (const double *)f->tyy, nc, f->tozz,
                       (const double *)f->tzz, nc, f->toyz, (const double *)f->tyz, ne.yz, f->toxz, (const double *)f->txz, ne.xz, f->toxy,
                       (const double *)f->txy, ne.xy);
    hipLaunchKernelGGL(k_copy6, dim3(1024), dim3(256), 0, s, f->toyz_c, (const double *)f->tyz_c, nc, f->toxz_c, (const double *)f->txz_c, nc, f->toxy_c,
                       (const double *)f->txy_c, nc, (double *)nullptr, (const double *)nullptr, (i64)0, (double *)nullptr, (const double *)nullptr, (i64)0,
                       (double *)nullptr, (const double *)nullptr, (i64)0);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(s));
    float ms = 0.f;
    JRX_HIP(h, hipEventElapsedTime(&ms, t0, t1));
    res->iter = iter;
    res->nchecks = cont < res->cap ? cont : res->cap;
    res->time_s = ms * 1e-3;
    res->av_time_s = iter > 1 ? res->time_s / (double)(iter - 1) : res->time_s;
    return JRX_OK;
}

}   // extern "C"
