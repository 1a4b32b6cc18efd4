// thermal3d.hip -- 3D pseudo-transient heat diffusion for gfx950.
//
// Reference being replaced: src/thermal_diffusion/DiffusionPT_solver.jl:34-149 (array-coefficient form) and :181-305
// (rheology form as test/test_diffusion3D.jl evaluates it: constant k and Cp, rho = rho0*(1 - alpha*(T - T0))), kernels
// DiffusionPT_kernels.jl:6-61 (compute_flux! 3D), :160-199 (update_T! 3D), :250-282 (check_res! 3D), :670-673 (update_ΔT!)
// and thermal_bcs! 3D (BoundaryConditions.jl:46-54; constant_value.jl:15-33; free_slip.jl:86-103; periodic.jl:42-60).
// 3D thermal face names: bot <-> k = 1, top <-> k = end.  HBM-bound fp64 stencils (20 array passes per iteration).
#include "jrx_internal.hpp"
#include "jrx_kernels.hpp"
#include "jrx_thermal_phases.hpp"

namespace {

enum { XL = 0, XR = 1, YF = 2, YB = 3, ZT = 4, ZB = 5 };

struct T3Args {
    jrx_thermal3d_fields t;
    jrx_thermal3d_params p;
    bool wpt = false;      // phase-ratio form: update_T! also writes next iteration's θr_dτ, dτ_ρ (update_pt_thermal_arrays! folded in)
    bool nt = false;       // fused one-launch kernels: non-temporal stores of the new (T, qT) set (nobody reads it before the next launch); tuning switch "thermal_nt"
};
// store of the fused kernels' outputs: streaming (non-temporal) when a.nt
#define TST(a_, lhs_, val_) do { double *q_ = &(lhs_); const double v_ = (val_); if ((a_).nt) __builtin_nontemporal_store(v_, q_); else *q_ = v_; } while (0)

__device__ __forceinline__ double rhoCp3_of(const jrx_thermal3d_params &p, const double *rhoCp, i64 c, double T)
{
    return p.rheology_form ? p.Cp * (p.rho0 * (1.0 - p.alpha * (T - p.T0))) : rhoCp[c];
}

#define NODE_IJK(n1_, n2_)                                              \
    const int t_ = blockIdx.x * blockDim.x + threadIdx.x;               \
    const int j = t_ / (n1_), i = t_ - j * (n1_), k = blockIdx.y;       \
    if (j >= (n2_)) return;
#define GRID_IJK(n1_, n2_, n3_) dim3((unsigned)(((i64)(n1_) * (n2_) + 255) / 256), (unsigned)(n3_))
// the same box in XCD slab order: block L of the launch takes position (L % 8) * (T / 8) + L / 8 of the (flattened xy, z) sequence, so that each of the
// 8 XCDs (blocks are dealt to them round-robin, each has its own L2) works on a contiguous slab of planes and finds the rows j +- 1 / planes k +- 1 of
// its stencils in its own L2 (k_flux3d, phase-ratio form at 256^3: 21.6 array passes fetched without, see profiles/r02_thermal3d_phases.txt)
#define NODE_IJK_XS(n1_, n2_)                                                         \
    unsigned bx_ = blockIdx.x, by_ = blockIdx.y;                                      \
    {                                                                                 \
        const unsigned L_ = by_ * gridDim.x + bx_, per_ = (gridDim.x * gridDim.y) / 8u; \
        if (L_ < per_ * 8u) { const unsigned Ln_ = (L_ & 7u) * per_ + (L_ >> 3); bx_ = Ln_ % gridDim.x; by_ = Ln_ / gridDim.x; } \
    }                                                                                 \
    const int t_ = bx_ * blockDim.x + threadIdx.x;                                    \
    const int j = t_ / (n1_), i = t_ - j * (n1_), k = by_;                            \
    if (j >= (n2_)) return;
#define T3_(i_, j_, k_) T[(i_) + (i64)(nx + 2) * ((j_) + (i64)(ny + 2) * (k_))]
#define CC_(A, i_, j_, k_) (A)[(i_) + (i64)nx * ((j_) + (i64)ny * (k_))]

// compute_flux! over (nx+1, ny+1, nz+1).  Q2 = false skips the stores of qT*2 (the un-relaxed flux is only read by check_res!,
// DiffusionPT_solver.jl:113-126, so it is written on the iterations a check follows)
template <bool Q2, class PHT>
__global__ __launch_bounds__(256) void k_flux3d(const T3Args a, const PHT ph)
{
    constexpr bool PH = is_tph<PHT>::value;
    const int nx = (int)a.p.nx, ny = (int)a.p.ny, nz = (int)a.p.nz;
    NODE_IJK_XS(nx + 1, ny + 1)
    const double *__restrict__ T = a.t.T, *__restrict__ th = a.t.thetar_dtau;
    const double kc = (a.p.k_const + a.p.k_const) * 0.5;
    if (j < ny && k < nz) {
        const i64 q = i + (i64)(nx + 1) * (j + (i64)ny * k);
        if (i == 0 && a.p.constant_flux_on[XL]) a.t.qTx[q] = a.p.constant_flux[XL];
        else if (i == nx && a.p.constant_flux_on[XR]) a.t.qTx[q] = a.p.constant_flux[XR];
        else {
            const int l = clampi(i - 1, 0, nx - 1), r = clampi(i, 0, nx - 1);
            double K;
            if constexpr (PH) K = (tph_cond<tph_np<PHT>::value>(ph.m, ph.f.phase_qx + tph_nph(ph) * (l + (i64)(nx + 1) * (j + (i64)ny * k))) +
                                   tph_cond<tph_np<PHT>::value>(ph.m, ph.f.phase_qx + tph_nph(ph) * (r + (i64)(nx + 1) * (j + (i64)ny * k)))) * 0.5;
            else K = a.p.rheology_form ? kc : (CC_(a.t.K, l, j, k) + CC_(a.t.K, r, j, k)) * 0.5;
            const double t = (CC_(th, l, j, k) + CC_(th, r, j, k)) * 0.5;
            const double qv = -K * (T3_(i + 1, j + 1, k + 1) - T3_(i, j + 1, k + 1)) * a.p._dx;
            if (Q2) a.t.qTx2[q] = qv;
            a.t.qTx[q] = (a.t.qTx[q] * t + qv) / (1.0 + t);
        }
    }
    if (i < nx && k < nz) {
        const i64 q = i + (i64)nx * (j + (i64)(ny + 1) * k);
        if (j == 0 && a.p.constant_flux_on[YF]) a.t.qTy[q] = a.p.constant_flux[YF];
        else if (j == ny && a.p.constant_flux_on[YB]) a.t.qTy[q] = a.p.constant_flux[YB];
        else {
            const int l = clampi(j - 1, 0, ny - 1), r = clampi(j, 0, ny - 1);
            double K;
            if constexpr (PH) K = (tph_cond<tph_np<PHT>::value>(ph.m, ph.f.phase_qy + tph_nph(ph) * (i + (i64)nx * (l + (i64)(ny + 1) * k))) +
                                   tph_cond<tph_np<PHT>::value>(ph.m, ph.f.phase_qy + tph_nph(ph) * (i + (i64)nx * (r + (i64)(ny + 1) * k)))) * 0.5;
            else K = a.p.rheology_form ? kc : (CC_(a.t.K, i, l, k) + CC_(a.t.K, i, r, k)) * 0.5;
            const double t = (CC_(th, i, l, k) + CC_(th, i, r, k)) * 0.5;
            const double qv = -K * (T3_(i + 1, j + 1, k + 1) - T3_(i + 1, j, k + 1)) * a.p._dy;
            if (Q2) a.t.qTy2[q] = qv;
            a.t.qTy[q] = (a.t.qTy[q] * t + qv) / (1.0 + t);
        }
    }
    if (i < nx && j < ny) {
        const i64 q = i + (i64)nx * (j + (i64)ny * k);
        if (k == 0 && a.p.constant_flux_on[ZB]) a.t.qTz[q] = a.p.constant_flux[ZB];
        else if (k == nz && a.p.constant_flux_on[ZT]) a.t.qTz[q] = a.p.constant_flux[ZT];
        else {
            const int l = clampi(k - 1, 0, nz - 1), r = clampi(k, 0, nz - 1);
            double K;
            if constexpr (PH) K = (tph_cond<tph_np<PHT>::value>(ph.m, ph.f.phase_qz + tph_nph(ph) * (i + (i64)nx * (j + (i64)ny * l))) +
                                   tph_cond<tph_np<PHT>::value>(ph.m, ph.f.phase_qz + tph_nph(ph) * (i + (i64)nx * (j + (i64)ny * r)))) * 0.5;
            else K = a.p.rheology_form ? kc : (CC_(a.t.K, i, j, l) + CC_(a.t.K, i, j, r)) * 0.5;
            const double t = (CC_(th, i, j, l) + CC_(th, i, j, r)) * 0.5;
            const double qv = -K * (T3_(i + 1, j + 1, k + 1) - T3_(i + 1, j + 1, k)) * a.p._dz;
            if (Q2) a.t.qTz2[q] = qv;
            a.t.qTz[q] = (a.t.qTz[q] * t + qv) / (1.0 + t);
        }
    }
}

// k_flux3d of the phase-ratio form with the phase count as a constant, every operand requested up front (option "thermal_np_const").  The control-flow form tests the constant-flux
// faces first and loads inside the branches: 25 loads in a dozen dependent groups at 22 VGPRs (4.4 TB/s of its 1.9 GB at 256^3, where k_updateT3d reaches 6.2).  Here: clamped
// unconditional loads, one batch, the face conditions as selects; the arithmetic of k_flux3d, expression for expression.
template <bool Q2, int NPH>
__global__ __launch_bounds__(256) void k_flux3d_b(const T3Args a, const TPhN<NPH> ph)
{
    const int nx = (int)a.p.nx, ny = (int)a.p.ny, nz = (int)a.p.nz;
    NODE_IJK_XS(nx + 1, ny + 1)
    const double *__restrict__ T = a.t.T, *__restrict__ th = a.t.thetar_dtau;
    const bool fx = j < ny && k < nz, fy = i < nx && k < nz, fz = i < nx && j < ny;
    const int ic = min(i, nx - 1), jc = min(j, ny - 1), kc = min(k, nz - 1);
    const int il = max(i - 1, 0), jl = max(j - 1, 0), kl = max(k - 1, 0);          // l = clamp(. - 1), r = clamp(.) = ic / jc / kc
    // face indices (clamped to an existing face for the threads that do not own one)
    const i64 qx = i + (i64)(nx + 1) * (jc + (i64)ny * kc), qy = ic + (i64)nx * (j + (i64)(ny + 1) * kc), qz = ic + (i64)nx * (jc + (i64)ny * k);
    // ---- every operand
    const double qox = a.t.qTx[qx], qoy = a.t.qTy[qy], qoz = a.t.qTz[qz];
    const double T111 = T3_(ic + 1, jc + 1, kc + 1);
    const double Tx1 = T3_(i + 1, jc + 1, kc + 1), Tx0 = T3_(i, jc + 1, kc + 1);
    const double Ty1 = T3_(ic + 1, j + 1, kc + 1), Ty0 = T3_(ic + 1, j, kc + 1);
    const double Tz1 = T3_(ic + 1, jc + 1, k + 1), Tz0 = T3_(ic + 1, jc + 1, k);
    const double txl = CC_(th, il, jc, kc), txr = CC_(th, ic, jc, kc), tyl = CC_(th, ic, jl, kc), tzl = CC_(th, ic, jc, kl);      // (the r operand is the same cell for the three faces)
    double rxl[NPH], rxr[NPH], ryl[NPH], ryr[NPH], rzl[NPH], rzr[NPH];
#pragma unroll
    for (int q = 0; q < NPH; q++) {
        rxl[q] = ph.f.phase_qx[NPH * (il + (i64)(nx + 1) * (jc + (i64)ny * kc)) + q];
        rxr[q] = ph.f.phase_qx[NPH * (ic + (i64)(nx + 1) * (jc + (i64)ny * kc)) + q];
        ryl[q] = ph.f.phase_qy[NPH * (ic + (i64)nx * (jl + (i64)(ny + 1) * kc)) + q];
        ryr[q] = ph.f.phase_qy[NPH * (ic + (i64)nx * (jc + (i64)(ny + 1) * kc)) + q];
        rzl[q] = ph.f.phase_qz[NPH * (ic + (i64)nx * (jc + (i64)ny * kl)) + q];
        rzr[q] = ph.f.phase_qz[NPH * (ic + (i64)nx * (jc + (i64)ny * kc)) + q];
    }
    (void)T111;
    __builtin_amdgcn_sched_barrier(0);
    auto cond = [&](const double *r) {          // tph_cond on ratios held in registers
        double x = 0.0;
#pragma unroll
        for (int q = 0; q < NPH; q++) {
            const double rq = r[q];
            if (rq == 1.0) return ph.m.k[q] * rq;
            x += (rq == 0.0) ? 0.0 : ph.m.k[q] * rq;
        }
        return x;
    };
    if (fx) {
        double qn;
        if (i == 0 && a.p.constant_flux_on[XL]) qn = a.p.constant_flux[XL];
        else if (i == nx && a.p.constant_flux_on[XR]) qn = a.p.constant_flux[XR];
        else {
            const double K = (cond(rxl) + cond(rxr)) * 0.5;
            const double t = (txl + txr) * 0.5;
            const double qv = -K * (Tx1 - Tx0) * a.p._dx;
            if (Q2) a.t.qTx2[qx] = qv;
            qn = (qox * t + qv) / (1.0 + t);
        }
        a.t.qTx[qx] = qn;
    }
    if (fy) {
        double qn;
        if (j == 0 && a.p.constant_flux_on[YF]) qn = a.p.constant_flux[YF];
        else if (j == ny && a.p.constant_flux_on[YB]) qn = a.p.constant_flux[YB];
        else {
            const double K = (cond(ryl) + cond(ryr)) * 0.5;
            const double t = (tyl + txr) * 0.5;
            const double qv = -K * (Ty1 - Ty0) * a.p._dy;
            if (Q2) a.t.qTy2[qy] = qv;
            qn = (qoy * t + qv) / (1.0 + t);
        }
        a.t.qTy[qy] = qn;
    }
    if (fz) {
        double qn;
        if (k == 0 && a.p.constant_flux_on[ZB]) qn = a.p.constant_flux[ZB];
        else if (k == nz && a.p.constant_flux_on[ZT]) qn = a.p.constant_flux[ZT];
        else {
            const double K = (cond(rzl) + cond(rzr)) * 0.5;
            const double t = (tzl + txr) * 0.5;
            const double qv = -K * (Tz1 - Tz0) * a.p._dz;
            if (Q2) a.t.qTz2[qz] = qv;
            qn = (qoz * t + qv) / (1.0 + t);
        }
        a.t.qTz[qz] = qn;
    }
}

// thermal_bcs! 3D restricted to the ghost cells around one interior cell that touches faces in the dimensions of `mask` (bit d): the
// reference's statement order (per BC type: z faces, x faces, y faces -- constant_value.jl:15-33, free_slip.jl:86-103) replayed on the
// 2 x 2 x 2 patch v[b], b = ghost bits (x, y, z); v[0] is the interior value.  side[d]: 0 low face, 1 high face.
__device__ __forceinline__ void thermal_ghosts3d(const jrx_thermal3d_params &p, double *__restrict__ T, const i64 st[3], i64 I1, int mask, const int side[3], double m)
{
    const int face[3] = {side[0] ? XR : XL, side[1] ? YB : YF, side[2] ? ZT : ZB};
    const i64 off[3] = {side[0] ? st[0] : -st[0], side[1] ? st[1] : -st[1], side[2] ? st[2] : -st[2]};
    double v[8];
    bool w[8];
#pragma unroll
    for (int b = 0; b < 8; b++) {
        w[b] = false;
        v[b] = m;
        if (b != 0 && (b & ~mask) == 0) v[b] = T[I1 + ((b & 1) ? off[0] : 0) + ((b & 2) ? off[1] : 0) + ((b & 4) ? off[2] : 0)];
    }
    const int order[3] = {2, 0, 1};
    for (int step = 0; step < 2; step++) {
        const int32_t *on = step == 0 ? p.constant_value_on : p.no_flux;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int d = order[q];
            if (!((mask >> d) & 1) || !on[face[d]]) continue;
            const double cv2 = 2 * p.constant_value[face[d]];
#pragma unroll
            for (int b = 0; b < 8; b++)
                if (((b >> d) & 1) && (b & ~mask) == 0) { const double src = v[b ^ (1 << d)]; v[b] = step == 0 ? cv2 - src : src; w[b] = true; }
        }
    }
#pragma unroll
    for (int b = 1; b < 8; b++)
        if (w[b]) T[I1 + ((b & 1) ? off[0] : 0) + ((b & 2) ? off[1] : 0) + ((b & 4) ? off[2] : 0)] = v[b];
}

// update_T! (RES=false) / check_res! (RES=true) over ni; BCF: cells next to a face also apply thermal_bcs! to their ghosts
template <bool RES, bool BCF, class PHT>
__global__ __launch_bounds__(256) void k_updateT3d(const T3Args a, const PHT ph)
{
    constexpr bool PH = is_tph<PHT>::value;
    const int nx = (int)a.p.nx, ny = (int)a.p.ny, nz = (int)a.p.nz;
    NODE_IJK_XS(nx, ny)
    if (k >= nz) return;
    const i64 c = i + (i64)nx * (j + (i64)ny * k), I1 = (i + 1) + (i64)(nx + 2) * ((j + 1) + (i64)(ny + 2) * (k + 1));
    const double _dt = 1.0 / a.p.dt;
    const double Tc = a.t.T[I1];
    double rcp, Hr = 0.0;
    if constexpr (PH) {
        const double *rc = ph.f.phase_c + tph_nph(ph) * c;
        rcp = tph_rhoCp<tph_np<PHT>::value>(ph.m, rc, Tc, ph.f.P[c]);
        Hr = tph_Hr<tph_np<PHT>::value>(ph.m, rc);
    } else rcp = rhoCp3_of(a.p, a.t.rhoCp, c, Tc);
    const double *qx = RES ? a.t.qTx2 : a.t.qTx, *qy = RES ? a.t.qTy2 : a.t.qTy, *qz = RES ? a.t.qTz2 : a.t.qTz;
    const double divq = (qx[(i + 1) + (i64)(nx + 1) * (j + (i64)ny * k)] - qx[i + (i64)(nx + 1) * (j + (i64)ny * k)]) * a.p._dx +
                        (qy[i + (i64)nx * ((j + 1) + (i64)(ny + 1) * k)] - qy[i + (i64)nx * (j + (i64)(ny + 1) * k)]) * a.p._dy +
                        (qz[i + (i64)nx * (j + (i64)ny * (k + 1))] - qz[c]) * a.p._dz;
    // optional terms: + adiabatic * T in the rheology forms; Dirichlet cells (mask != 0) take (1 - m) T + m value and have no residual
    const bool hasadi = a.p.rheology_form != 0 && a.t.adiabatic != nullptr;
    const double adi = hasadi ? a.t.adiabatic[c] * Tc : 0.0;
    const double dm = a.t.dirichlet_mask ? a.t.dirichlet_mask[I1] : 0.0;
    if (RES) {
        if (dm != 0.0) a.t.ResT[c] = 0.0;
        else if constexpr (PH) a.t.ResT[c] = hasadi ? -rcp * (Tc - a.t.Told[I1]) * _dt - divq + Hr + a.t.H[c] + a.t.shear_heating[c] + adi
                                                 : -rcp * (Tc - a.t.Told[I1]) * _dt - divq + Hr + a.t.H[c] + a.t.shear_heating[c];
        else a.t.ResT[c] = hasadi ? -rcp * (Tc - a.t.Told[I1]) * _dt - divq + a.t.H[c] + a.t.shear_heating[c] + adi
                                  : -rcp * (Tc - a.t.Told[I1]) * _dt - divq + a.t.H[c] + a.t.shear_heating[c];
    } else {
        const double dr = a.t.dtau_rho[c];
        double Tn;
        if (dm != 0.0) Tn = (1 - dm) * Tc + dm * (a.t.dirichlet_value ? a.t.dirichlet_value[I1] : a.p.dirichlet_const);
        else if constexpr (PH) Tn = hasadi ? (dr * (-divq + a.t.Told[I1] * rcp * _dt + Hr + a.t.H[c] + a.t.shear_heating[c] + adi) + Tc) / (1.0 + dr * rcp * _dt)
                                           : (dr * (-divq + a.t.Told[I1] * rcp * _dt + Hr + a.t.H[c] + a.t.shear_heating[c]) + Tc) / (1.0 + dr * rcp * _dt);
        else Tn = hasadi ? (dr * (-divq + a.t.Told[I1] * rcp * _dt + a.t.H[c] + a.t.shear_heating[c] + adi) + Tc) / (1.0 + dr * rcp * _dt)
                         : (dr * (-divq + a.t.Told[I1] * rcp * _dt + a.t.H[c] + a.t.shear_heating[c]) + Tc) / (1.0 + dr * rcp * _dt);
        a.t.T[I1] = Tn;
        if constexpr (PH) {
            if (a.wpt) {      // update_pt_thermal_arrays! of the next iteration (DiffusionPT_coefficients.jl:123-136) from the new T of this cell
                double th_, dr_;
                tph_pt_coeffs<tph_np<PHT>::value>(ph.m, ph.f.phase_c + tph_nph(ph) * c, Tn, ph.f.P[c], _dt, th_, dr_);
                const_cast<double *>(a.t.thetar_dtau)[c] = th_;
                const_cast<double *>(a.t.dtau_rho)[c] = dr_;
            }
        }
        if (BCF) {
            const int side[3] = {i == nx - 1, j == ny - 1, k == nz - 1};
            const int mask = ((i == 0 || i == nx - 1) ? 1 : 0) | ((j == 0 || j == ny - 1) ? 2 : 0) | ((k == 0 || k == nz - 1) ? 4 : 0);
            if (mask) {
                const i64 st[3] = {1, nx + 2, (i64)(nx + 2) * (ny + 2)};
                thermal_ghosts3d(a.p, a.t.T, st, I1, mask, side, Tn);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// One PT iteration in one launch: compute_flux! + update_T! + thermal_bcs! (DiffusionPT_solver.jl:104-111) for iterations nobody
// observes.  A block is a row segment of TX cells (whole waves) marching KZ planes; a thread owns cell (i, j, k) and stores the fluxes
// on its three low faces (plus the high face where that is a domain face).  For its own update it also needs the new fluxes on its
// high faces: x from the next lane (shuffle; the last lane of a wave computes it itself), y computed redundantly from the row above,
// z computed here and carried to the next plane.  Because neighbouring blocks recompute fluxes that another block stores, fluxes and
// T are read from one set (T, qT) and written to the other (ping-pong): 11 array reads + 4 writes per cell instead of 15 + 4
// (6 + 3 and 9 + 1 for the two kernels).  No LDS, no barrier.  Same arithmetic, in the same order, as k_flux3d / k_updateT3d.
// ------------------------------------------------------------------------------------------------
struct TSet { double *T, *qx, *qy, *qz; };
// R = rows per thread: the thread owns cells (i, j0 .. j0+R-1, k); the y fluxes between its rows are computed once, and only the
// two rows j0-1 and j0+R are re-read from what neighbouring blocks own ((R+2)/R row reads of T, K, θ per cell instead of 3).
// Shipped with R = 1: more rows per thread cost more in registers / occupancy than they save in re-reads (measured).
template <int TX, int KZ, int XG, int R>
__global__ __launch_bounds__(TX) void k_thermal3d_fused(const T3Args a, const TSet dst, int ntx, int nty)
{
    const int nx = (int)a.p.nx, ny = (int)a.p.ny, nz = (int)a.p.nz;
    int tile = blockIdx.x;
    {
        // XCD-banded order (blocks are dealt round-robin to the 8 XCDs): XCD q takes XG consecutive row groups of every 8*XG, so that the
        // rows j0-1 / j0+R a block re-reads were fetched by the same L2
        const int rows = (int)(gridDim.x / (unsigned)ntx);          // nty * number of z chunks
        const int full = (rows / (8 * XG)) * (8 * XG) * ntx;
        if (tile < full) {
            const int q = tile & 7, r = tile >> 3, r2 = r / ntx;
            tile = ((r2 / XG) * (8 * XG) + q * XG + r2 % XG) * ntx + r % ntx;
        }
    }
    const int tix = tile % ntx, tr = tile / ntx, j0 = (tr % nty) * R, tiz = tr / nty;
    const int i = tix * TX + (int)threadIdx.x;
    const int kb = tiz * KZ, kend = min(kb + KZ, nz);
    // whole waves beyond the row end leave; lanes beyond it inside a live wave idle but stay for the shuffles
    if (i >= nx && (i & ~63) >= nx) return;
    const bool cell = i < nx;
    const int ic = cell ? i : nx - 1;
    const double *__restrict__ T = a.t.T, *__restrict__ th = a.t.thetar_dtau, *__restrict__ Kk = a.t.K;
    const double kc = (a.p.k_const + a.p.k_const) * 0.5;
    const bool rf = a.p.rheology_form != 0;
    const double _dt = 1.0 / a.p.dt, _dx = a.p._dx, _dy = a.p._dy, _dz = a.p._dz;
    const i64 sT1 = nx + 2, sT2 = (i64)(nx + 2) * (ny + 2), sC2 = (i64)nx * ny;
    const int im = max(ic - 1, 0), ip = min(ic + 1, nx - 1);
    const bool edge = (threadIdx.x & 63) == 63 || i == nx - 1;      // no lane to the right holds cell i+1
    const bool cfxl = a.p.constant_flux_on[XL] != 0, cfxr = a.p.constant_flux_on[XR] != 0, cfyf = a.p.constant_flux_on[YF] != 0,
               cfyb = a.p.constant_flux_on[YB] != 0, cfzb = a.p.constant_flux_on[ZB] != 0, cfzt = a.p.constant_flux_on[ZT] != 0;
    // relaxed flux across a face: K and θ averaged over the two cells (clamped at the domain faces), T difference over the face
    auto relax = [&](double qold, double Kl, double Kr, double tl, double tr_, double Thi, double Tlo, double _d) -> double {
        const double K = rf ? kc : (Kl + Kr) * 0.5;
        const double t = (tl + tr_) * 0.5;
        const double qv = -K * (Thi - Tlo) * _d;
        return (qold * t + qv) / (1.0 + t);
    };
    // plane-k operands of the own cells, carried upward: T (ghost-indexed k+1), K, θ and the flux on the low z face
    double Tc[R], Kc_[R], tc[R], qz_lo[R];
    bool rok[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        rok[r] = j0 + r < ny;
        const int j = rok[r] ? j0 + r : ny - 1;
        const i64 c = ic + (i64)nx * j + sC2 * kb, I1 = (ic + 1) + sT1 * (j + 1) + sT2 * (kb + 1);
        Tc[r] = T[I1]; Kc_[r] = rf ? 0.0 : Kk[c]; tc[r] = th[c];
        if (kb == 0 && cfzb) qz_lo[r] = a.p.constant_flux[ZB];
        else {
            const i64 cl = kb > 0 ? c - sC2 : c;
            qz_lo[r] = relax(a.t.qTz[c], rf ? 0.0 : Kk[cl], Kc_[r], th[cl], tc[r], Tc[r], T[I1 - sT2], _dz);
        }
        if (kb == 0 && cell && rok[r]) TST(a, dst.qz[c], qz_lo[r]);        // face 0 has no chunk below that would own it
    }
    const int jlo = max(j0 - 1, 0);                                   // clamped row of the K / θ average on the lowest y face
    for (int k = kb; k < kend; ++k) {
        // ---- y fluxes on the faces j0 .. j0+R of this column
        double qy[R + 1];
        {
            // low halo row j0-1 (T ghost row when j0 = 0)
            const i64 cl = ic + (i64)nx * jlo + sC2 * k;
            const i64 q0 = ic + (i64)nx * (j0 + (i64)(ny + 1) * k);
            if (j0 == 0 && cfyf) qy[0] = a.p.constant_flux[YF];
            else qy[0] = relax(a.t.qTy[q0], rf ? 0.0 : Kk[cl], Kc_[0], th[cl], tc[0], Tc[0], T[(ic + 1) + sT1 * j0 + sT2 * (k + 1)], _dy);
            if (cell) TST(a, dst.qy[q0], qy[0]);
#pragma unroll
            for (int f = 1; f <= R; f++) {
                const int jf = j0 + f;                                // face index; exists while jf <= ny
                if (jf > ny) { qy[f] = 0.0; continue; }
                const i64 q = q0 + (i64)nx * f;
                if (jf == ny && cfyb) qy[f] = a.p.constant_flux[YB];
                else if (f < R && rok[f]) qy[f] = relax(a.t.qTy[q], Kc_[f - 1], Kc_[f], tc[f - 1], tc[f], Tc[f], Tc[f - 1], _dy);
                else {
                    // upper cell is the row above this thread's rows, or the T ghost row behind the back face (K, θ clamped)
                    const int ju = jf < ny ? jf : ny - 1;
                    const i64 cu = ic + (i64)nx * ju + sC2 * k;
                    qy[f] = relax(a.t.qTy[q], Kc_[f - 1], rf ? 0.0 : Kk[cu], tc[f - 1], th[cu], T[(ic + 1) + sT1 * (jf + 1) + sT2 * (k + 1)], Tc[f - 1], _dy);
                }
                if (cell && ((f < R && rok[f]) || jf == ny)) TST(a, dst.qy[q], qy[f]);
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (!rok[r]) continue;                                    // uniform over the block
            const int j = j0 + r;
            const i64 c = ic + (i64)nx * j + sC2 * k, I1 = (ic + 1) + sT1 * (j + 1) + sT2 * (k + 1);
            // ---- x: low face i (own), high face i+1 from the next lane
            double qx_lo;
            {
                const i64 q = ic + (i64)(nx + 1) * (j + (i64)ny * k);
                if (ic == 0 && cfxl) qx_lo = a.p.constant_flux[XL];
                else {
                    const i64 cl = c - (ic - im);
                    qx_lo = relax(a.t.qTx[q], rf ? 0.0 : Kk[cl], Kc_[r], th[cl], tc[r], Tc[r], T[I1 - 1], _dx);
                }
                if (cell) TST(a, dst.qx[q], qx_lo);
            }
            double qx_hi = __shfl_down(qx_lo, 1, 64);
            if (edge) {
                const i64 q = (ic + 1) + (i64)(nx + 1) * (j + (i64)ny * k);
                if (ic + 1 == nx && cfxr) qx_hi = a.p.constant_flux[XR];
                else {
                    const i64 cr = c + (ip - ic);
                    qx_hi = relax(a.t.qTx[q], Kc_[r], rf ? 0.0 : Kk[cr], tc[r], th[cr], T[I1 + 1], Tc[r], _dx);
                }
                if (cell && ic + 1 == nx) TST(a, dst.qx[q], qx_hi);           // the domain's right face belongs to no cell's low face
            }
            // ---- z: high face k+1 (owned here), becomes the low face of the next plane
            double qz_hi, Kn_ = 0.0;
            const i64 cr = k + 1 < nz ? c + sC2 : c;
            const double Tn_c = T[I1 + sT2], tn = th[cr];
            if (!rf) Kn_ = Kk[cr];
            if (k + 1 == nz && cfzt) qz_hi = a.p.constant_flux[ZT];
            else qz_hi = relax(a.t.qTz[c + sC2], Kc_[r], Kn_, tc[r], tn, Tn_c, Tc[r], _dz);
            if (cell) TST(a, dst.qz[c + sC2], qz_hi);
            if (cell) {
                const double rcp = rhoCp3_of(a.p, a.t.rhoCp, c, Tc[r]);
                const double divq = (qx_hi - qx_lo) * _dx + (qy[r + 1] - qy[r]) * _dy + (qz_hi - qz_lo[r]) * _dz;
                const double dr = a.t.dtau_rho[c];
                const double Tn = (dr * (-divq + a.t.Told[I1] * rcp * _dt + a.t.H[c] + a.t.shear_heating[c]) + Tc[r]) / (1.0 + dr * rcp * _dt);
                TST(a, dst.T[I1], Tn);
                const int side[3] = {i == nx - 1, j == ny - 1, k == nz - 1};
                const int mask = ((i == 0 || i == nx - 1) ? 1 : 0) | ((j == 0 || j == ny - 1) ? 2 : 0) | ((k == 0 || k == nz - 1) ? 4 : 0);
                if (mask) {
                    const i64 st[3] = {1, sT1, sT2};
                    thermal_ghosts3d(a.p, dst.T, st, I1, mask, side, Tn);
                }
            }
            qz_lo[r] = qz_hi; Tc[r] = Tn_c; Kc_[r] = Kn_; tc[r] = tn;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_thermal3d_fused with the rows j - 1 / j + 1 through LDS (round 6, tuning switch "thermal_tile" = TY): a block is a tile of 64 cells x TY rows (a row per wave) marching KZ
// planes.  The row-segment form above reads T, K, θ of the row below and T, K, θ, qTy of the row above from memory for every cell (served by L2 where the XCD band holds: 18 array
// passes moved for 15 needed); here every row publishes its (T, K, θ) of the plane, takes the row below's from LDS for its low y face, publishes the new flux of that face, and
// takes the row above's as the flux of its high face instead of recomputing it.  Only the bottom row of a tile loads the row below, only the top row (and the row on the domain's
// back face) computes its high face from loaded operands.  Two barriers per plane.  Operand for operand the arithmetic of k_thermal3d_fused<., ., ., 1>: same bits.
// ------------------------------------------------------------------------------------------------
template <int TY, int KZ, int XG>
__global__ __launch_bounds__(64 * TY) void k_thermal3d_fused_t(const T3Args a, const TSet dst, int ntx, int nty)
{
    __shared__ double sA[3][TY][64];      // T, K, θ of the own cell, plane k
    __shared__ double sQ[TY][64];         // new flux on the low y face of the row
    const int nx = (int)a.p.nx, ny = (int)a.p.ny, nz = (int)a.p.nz;
    int tile = blockIdx.x;
    {
        const int rows = (int)(gridDim.x / (unsigned)ntx);          // nty * number of z chunks
        const int full = (rows / (8 * XG)) * (8 * XG) * ntx;
        if (tile < full) {
            const int q = tile & 7, r = tile >> 3, r2 = r / ntx;
            tile = ((r2 / XG) * (8 * XG) + q * XG + r2 % XG) * ntx + r % ntx;
        }
    }
    const int lane = (int)(threadIdx.x & 63), ty = (int)(threadIdx.x >> 6);
    const int tix = tile % ntx, tr = tile / ntx, jt = (tr % nty) * TY, tiz = tr / nty;
    const int i = tix * 64 + lane;
    const int j = jt + ty;
    const int kb = tiz * KZ, kend = min(kb + KZ, nz);
    const bool rok = j < ny;                   // a whole wave
    const bool cell = i < nx && rok;
    const int ic = i < nx ? i : nx - 1, jc = rok ? j : ny - 1;
    const double *__restrict__ T = a.t.T, *__restrict__ th = a.t.thetar_dtau, *__restrict__ Kk = a.t.K;
    const double kc = (a.p.k_const + a.p.k_const) * 0.5;
    const bool rf = a.p.rheology_form != 0;
    const double _dt = 1.0 / a.p.dt, _dx = a.p._dx, _dy = a.p._dy, _dz = a.p._dz;
    const i64 sT1 = nx + 2, sT2 = (i64)(nx + 2) * (ny + 2), sC2 = (i64)nx * ny;
    const int im = max(ic - 1, 0), ip = min(ic + 1, nx - 1);
    const bool edge = lane == 63 || i == nx - 1;      // no lane to the right holds cell i+1
    const bool cfxl = a.p.constant_flux_on[XL] != 0, cfxr = a.p.constant_flux_on[XR] != 0, cfyf = a.p.constant_flux_on[YF] != 0,
               cfyb = a.p.constant_flux_on[YB] != 0, cfzb = a.p.constant_flux_on[ZB] != 0, cfzt = a.p.constant_flux_on[ZT] != 0;
    auto relax = [&](double qold, double Kl, double Kr, double tl, double tr_, double Thi, double Tlo, double _d) -> double {
        const double K = rf ? kc : (Kl + Kr) * 0.5;
        const double t = (tl + tr_) * 0.5;
        const double qv = -K * (Thi - Tlo) * _d;
        return (qold * t + qv) / (1.0 + t);
    };
    double Tc, Kc_, tc, qz_lo;
    {
        const i64 c = ic + (i64)nx * jc + sC2 * kb, I1 = (ic + 1) + sT1 * (jc + 1) + sT2 * (kb + 1);
        Tc = T[I1]; Kc_ = rf ? 0.0 : Kk[c]; tc = th[c];
        if (kb == 0 && cfzb) qz_lo = a.p.constant_flux[ZB];
        else {
            const i64 cl = kb > 0 ? c - sC2 : c;
            qz_lo = relax(a.t.qTz[c], rf ? 0.0 : Kk[cl], Kc_, th[cl], tc, Tc, T[I1 - sT2], _dz);
        }
        if (kb == 0 && cell) TST(a, dst.qz[c], qz_lo);
    }
    const int jlo = max(j - 1, 0);
    const bool below = ty > 0;                                       // the row below is a row of this tile
    const bool above = ty < TY - 1 && j + 1 < ny;                    // the row above is a live row of this tile
    for (int k = kb; k < kend; ++k) {
        sA[0][ty][lane] = Tc; sA[1][ty][lane] = Kc_; sA[2][ty][lane] = tc;
        __syncthreads();
        const i64 c = ic + (i64)nx * jc + sC2 * k, I1 = (ic + 1) + sT1 * (jc + 1) + sT2 * (k + 1);
        // ---- y: low face j of the own cell
        double qy_lo = 0.0, qy_hi = 0.0;
        const i64 q0 = ic + (i64)nx * (jc + (i64)(ny + 1) * k);
        if (rok) {
            if (j == 0 && cfyf) qy_lo = a.p.constant_flux[YF];
            else if (below) qy_lo = relax(a.t.qTy[q0], sA[1][ty - 1][lane], Kc_, sA[2][ty - 1][lane], tc, Tc, sA[0][ty - 1][lane], _dy);
            else {
                const i64 cl = ic + (i64)nx * jlo + sC2 * k;
                qy_lo = relax(a.t.qTy[q0], rf ? 0.0 : Kk[cl], Kc_, th[cl], tc, Tc, T[(ic + 1) + sT1 * j + sT2 * (k + 1)], _dy);
            }
            if (cell) TST(a, dst.qy[q0], qy_lo);
        }
        sQ[ty][lane] = qy_lo;
        __syncthreads();
        if (rok) {
            // ---- y: high face j + 1: the row above's low face, or (top row of the tile, back face of the domain) computed here from loaded operands
            if (above) qy_hi = sQ[ty + 1][lane];
            else {
                const int jf = j + 1;
                const i64 q = q0 + (i64)nx;
                if (jf == ny && cfyb) qy_hi = a.p.constant_flux[YB];
                else {
                    const int ju = jf < ny ? jf : ny - 1;
                    const i64 cu = ic + (i64)nx * ju + sC2 * k;
                    qy_hi = relax(a.t.qTy[q], Kc_, rf ? 0.0 : Kk[cu], tc, th[cu], T[(ic + 1) + sT1 * (jf + 1) + sT2 * (k + 1)], Tc, _dy);
                }
                if (cell && jf == ny) TST(a, dst.qy[q], qy_hi);
            }
            // ---- x: low face i (own), high face i+1 from the next lane
            double qx_lo;
            {
                const i64 q = ic + (i64)(nx + 1) * (j + (i64)ny * k);
                if (ic == 0 && cfxl) qx_lo = a.p.constant_flux[XL];
                else {
                    const i64 cl = c - (ic - im);
                    qx_lo = relax(a.t.qTx[q], rf ? 0.0 : Kk[cl], Kc_, th[cl], tc, Tc, T[I1 - 1], _dx);
                }
                if (cell) TST(a, dst.qx[q], qx_lo);
            }
            double qx_hi = __shfl_down(qx_lo, 1, 64);
            if (edge) {
                const i64 q = (ic + 1) + (i64)(nx + 1) * (j + (i64)ny * k);
                if (ic + 1 == nx && cfxr) qx_hi = a.p.constant_flux[XR];
                else {
                    const i64 cr = c + (ip - ic);
                    qx_hi = relax(a.t.qTx[q], Kc_, rf ? 0.0 : Kk[cr], tc, th[cr], T[I1 + 1], Tc, _dx);
                }
                if (cell && ic + 1 == nx) TST(a, dst.qx[q], qx_hi);
            }
            // ---- z: high face k+1 (owned here), becomes the low face of the next plane
            double qz_hi, Kn_ = 0.0;
            const i64 cr = k + 1 < nz ? c + sC2 : c;
            const double Tn_c = T[I1 + sT2], tn = th[cr];
            if (!rf) Kn_ = Kk[cr];
            if (k + 1 == nz && cfzt) qz_hi = a.p.constant_flux[ZT];
            else qz_hi = relax(a.t.qTz[c + sC2], Kc_, Kn_, tc, tn, Tn_c, Tc, _dz);
            if (cell) TST(a, dst.qz[c + sC2], qz_hi);
            if (cell) {
                const double rcp = rhoCp3_of(a.p, a.t.rhoCp, c, Tc);
                const double divq = (qx_hi - qx_lo) * _dx + (qy_hi - qy_lo) * _dy + (qz_hi - qz_lo) * _dz;
                const double dr = a.t.dtau_rho[c];
                const double Tn = (dr * (-divq + a.t.Told[I1] * rcp * _dt + a.t.H[c] + a.t.shear_heating[c]) + Tc) / (1.0 + dr * rcp * _dt);
                TST(a, dst.T[I1], Tn);
                const int side[3] = {i == nx - 1, j == ny - 1, k == nz - 1};
                const int mask = ((i == 0 || i == nx - 1) ? 1 : 0) | ((j == 0 || j == ny - 1) ? 2 : 0) | ((k == 0 || k == nz - 1) ? 4 : 0);
                if (mask) {
                    const i64 st[3] = {1, sT1, sT2};
                    thermal_ghosts3d(a.p, dst.T, st, I1, mask, side, Tn);
                }
            }
            qz_lo = qz_hi; Tc = Tn_c; Kc_ = Kn_; tc = tn;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_thermal3d_fused for the phase-ratio form (TPhN<NPH>: phase count as a constant), one row per thread (R = 1): compute_flux! with the conductivity of a face averaged from the
// phase ratios at the two clamped face positions (k_flux3d, DiffusionPT_kernels.jl:366-440), update_T! with ρCp and the radiogenic heat from the centre ratios (k_updateT3d),
// thermal_bcs! by rule, and update_pt_thermal_arrays! of the next iteration from the cell's new T (the `wpt` part of k_updateT3d).  θr_dτ is read at neighbouring cells and written
// for the own one, so it ping-pongs like T and the fluxes (dst.th); dτ_ρ is read and written at the own cell only and stays in place.  Same arithmetic, in the same order, as
// k_flux3d_b / k_updateT3d: 23 array passes per iteration instead of 18 + 16.
// ------------------------------------------------------------------------------------------------
struct TSetP { double *T, *qx, *qy, *qz, *th; };
template <int TX, int KZ, int XG, int NPH>
__global__ __launch_bounds__(TX) void k_thermal3d_fused_ph(const T3Args a, const TSetP dst, const TPhN<NPH> ph, int ntx, int nty)
{
    const int nx = (int)a.p.nx, ny = (int)a.p.ny, nz = (int)a.p.nz;
    int tile = blockIdx.x;
    {
        const int rows = (int)(gridDim.x / (unsigned)ntx);          // nty * number of z chunks
        const int full = (rows / (8 * XG)) * (8 * XG) * ntx;
        if (tile < full) {
            const int q = tile & 7, r = tile >> 3, r2 = r / ntx;
            tile = ((r2 / XG) * (8 * XG) + q * XG + r2 % XG) * ntx + r % ntx;
        }
    }
    const int tix = tile % ntx, tr = tile / ntx, j = tr % nty, tiz = tr / nty;
    const int i = tix * TX + (int)threadIdx.x;
    const int kb = tiz * KZ, kend = min(kb + KZ, nz);
    if (i >= nx && (i & ~63) >= nx) return;
    const bool cell = i < nx;
    const int ic = cell ? i : nx - 1;
    const double *__restrict__ T = a.t.T, *__restrict__ th = a.t.thetar_dtau;
    const double _dt = 1.0 / a.p.dt, _dx = a.p._dx, _dy = a.p._dy, _dz = a.p._dz;
    const i64 sT1 = nx + 2, sT2 = (i64)(nx + 2) * (ny + 2), sC2 = (i64)nx * ny;
    const int im = max(ic - 1, 0), ip = min(ic + 1, nx - 1), jl = max(j - 1, 0), ju = min(j + 1, ny - 1);
    const bool edge = (threadIdx.x & 63) == 63 || i == nx - 1;      // no lane to the right holds cell i+1
    const bool cfxl = a.p.constant_flux_on[XL] != 0, cfxr = a.p.constant_flux_on[XR] != 0, cfyf = a.p.constant_flux_on[YF] != 0,
               cfyb = a.p.constant_flux_on[YB] != 0, cfzb = a.p.constant_flux_on[ZB] != 0, cfzt = a.p.constant_flux_on[ZT] != 0;
    // tph_cond on ratios held in registers, without the early return (a select instead: same value -- the sum stops where a pure phase is found)
    struct R { double v[NPH]; };
    auto ld = [&](const double *__restrict__ base, i64 pos) { R r; _Pragma("unroll") for (int q = 0; q < NPH; q++) r.v[q] = base[NPH * pos + q]; return r; };
    auto cond = [&](const R &r) {
        double x = 0.0, pure = 0.0;
        bool done = false;
#pragma unroll
        for (int q = 0; q < NPH; q++) {
            const double rq = r.v[q], term = ph.m.k[q] * rq;
            if (!done && rq == 1.0) { pure = term; done = true; }
            if (!done) x += (rq == 0.0) ? 0.0 : term;
        }
        return done ? pure : x;
    };
    // relaxed flux across a face (k_flux3d): K = the mean of the conductivities at the two clamped face positions, θ the mean over the two cells
    auto relax = [&](double qold, double Kl, double Kr, double tl, double tr_, double Thi, double Tlo, double _d) -> double {
        const double K = (Kl + Kr) * 0.5;
        const double t = (tl + tr_) * 0.5;
        const double qv = -K * (Thi - Tlo) * _d;
        return (qold * t + qv) / (1.0 + t);
    };
    // plane-kb operands of the own cell, carried upward: T (ghost-indexed k+1), θ, the conductivity at the z-face position k and the flux on the low z face
    double Tc, tc, cz, qz_lo;
    {
        const i64 c = ic + (i64)nx * j + sC2 * kb, I1 = (ic + 1) + sT1 * (j + 1) + sT2 * (kb + 1);
        Tc = T[I1]; tc = th[c]; cz = cond(ld(ph.f.phase_qz, c));
        if (kb == 0 && cfzb) qz_lo = a.p.constant_flux[ZB];
        else {
            const i64 cl = kb > 0 ? c - sC2 : c;
            qz_lo = relax(a.t.qTz[c], cond(ld(ph.f.phase_qz, cl)), cz, th[cl], tc, Tc, T[I1 - sT2], _dz);
        }
        if (kb == 0 && cell) TST(a, dst.qz[c], qz_lo);
    }
    for (int k = kb; k < kend; ++k) {
        const i64 c = ic + (i64)nx * j + sC2 * k, I1 = (ic + 1) + sT1 * (j + 1) + sT2 * (k + 1);
        const i64 q0 = ic + (i64)nx * (j + (i64)(ny + 1) * k), q1 = q0 + nx;          // y faces j, j + 1 (= positions j, j + 1 of phase_qy)
        const i64 cl = ic + (i64)nx * jl + sC2 * k, cu = ic + (i64)nx * ju + sC2 * k, cr = k + 1 < nz ? c + sC2 : c;
        const i64 qx = ic + (i64)(nx + 1) * (j + (i64)ny * k), qxe = qx + 1;
        // ---- every operand of the plane, requested before the first of them is used
        const R ry0 = ld(ph.f.phase_qy, q0), ryl = ld(ph.f.phase_qy, ic + (i64)nx * (jl + (i64)(ny + 1) * k)), ryu = ld(ph.f.phase_qy, ic + (i64)nx * (ju + (i64)(ny + 1) * k));
        const R rxl = ld(ph.f.phase_qx, qx - (ic - im)), rx0 = ld(ph.f.phase_qx, qx), rzn = ld(ph.f.phase_qz, cr), rcc = ld(ph.f.phase_c, c);
        const double qoy0 = a.t.qTy[q0], qoy1 = a.t.qTy[q1], qox = a.t.qTx[qx], qoz = a.t.qTz[c + sC2];
        const double thl = th[cl], thu = th[cu], thx = th[c - (ic - im)], tn = th[cr];
        const double Tyl = T[(ic + 1) + sT1 * j + sT2 * (k + 1)], Tyu = T[(ic + 1) + sT1 * (j + 2) + sT2 * (k + 1)], Txl = T[I1 - 1], Tn_c = T[I1 + sT2];
        const double Pc = ph.f.P[c], dr = a.t.dtau_rho[c], Told = a.t.Told[I1], Hc = a.t.H[c], shc = a.t.shear_heating[c];
        R rxe = rx0;
        double qoxe = 0.0, thxe = 0.0, Txe = 0.0;
        if (edge) { rxe = ld(ph.f.phase_qx, qxe - 1 + (ip - ic)); qoxe = a.t.qTx[qxe]; thxe = th[c + (ip - ic)]; Txe = T[I1 + 1]; }
        __builtin_amdgcn_sched_barrier(0);
        // ---- y: low face j, high face j + 1 (computed here as the row above computes its low face)
        double qy_lo, qy_hi;
        {
            const double cy0 = cond(ry0);
            if (j == 0 && cfyf) qy_lo = a.p.constant_flux[YF];
            else qy_lo = relax(qoy0, cond(ryl), cy0, thl, tc, Tc, Tyl, _dy);
            if (cell) TST(a, dst.qy[q0], qy_lo);
            if (j + 1 == ny && cfyb) qy_hi = a.p.constant_flux[YB];
            else qy_hi = relax(qoy1, cy0, cond(ryu), tc, thu, Tyu, Tc, _dy);
            if (cell && j + 1 == ny) TST(a, dst.qy[q1], qy_hi);                              // the back face belongs to no cell's low face
        }
        // ---- x: low face i (own), high face i + 1 from the next lane
        const double cx0 = cond(rx0);
        double qx_lo;
        if (ic == 0 && cfxl) qx_lo = a.p.constant_flux[XL];
        else qx_lo = relax(qox, cond(rxl), cx0, thx, tc, Tc, Txl, _dx);
        if (cell) TST(a, dst.qx[qx], qx_lo);
        double qx_hi = __shfl_down(qx_lo, 1, 64);
        if (edge) {
            if (ic + 1 == nx && cfxr) qx_hi = a.p.constant_flux[XR];
            else qx_hi = relax(qoxe, cx0, cond(rxe), tc, thxe, Txe, Tc, _dx);
            if (cell && ic + 1 == nx) TST(a, dst.qx[qxe], qx_hi);
        }
        // ---- z: high face k + 1 (owned here), becomes the low face of the next plane
        double qz_hi;
        const double czn = cond(rzn);
        if (k + 1 == nz && cfzt) qz_hi = a.p.constant_flux[ZT];
        else qz_hi = relax(qoz, cz, czn, tc, tn, Tn_c, Tc, _dz);
        if (cell) TST(a, dst.qz[c + sC2], qz_hi);
        if (cell) {
            const double rcp = tph_rhoCp<NPH>(ph.m, rcc.v, Tc, Pc);
            const double Hr = tph_Hr<NPH>(ph.m, rcc.v);
            const double divq = (qx_hi - qx_lo) * _dx + (qy_hi - qy_lo) * _dy + (qz_hi - qz_lo) * _dz;
            const double Tn = (dr * (-divq + Told * rcp * _dt + Hr + Hc + shc) + Tc) / (1.0 + dr * rcp * _dt);
            TST(a, dst.T[I1], Tn);
            double th_, dr_;      // update_pt_thermal_arrays! of the next iteration
            tph_pt_coeffs<NPH>(ph.m, rcc.v, Tn, Pc, _dt, th_, dr_);
            dst.th[c] = th_;
            const_cast<double *>(a.t.dtau_rho)[c] = dr_;
            const int side[3] = {i == nx - 1, j == ny - 1, k == nz - 1};
            const int mask = ((i == 0 || i == nx - 1) ? 1 : 0) | ((j == 0 || j == ny - 1) ? 2 : 0) | ((k == 0 || k == nz - 1) ? 4 : 0);
            if (mask) {
                const i64 st[3] = {1, sT1, sT2};
                thermal_ghosts3d(a.p, dst.T, st, I1, mask, side, Tn);
            }
        }
        qz_lo = qz_hi; Tc = Tn_c; tc = tn; cz = czn;
    }
}

// thermal_bcs! 3D: one launch per (step, dim); step 0 constant_value, 1 no_flux, 2 periodic; dim = direction normal to the face pair
__global__ __launch_bounds__(256) void k_tbc3d(double *__restrict__ T, int nx, int ny, int nz, int step, int dim, int lo_on, int hi_on, double lo_val,
                                               double hi_val)
{
    const int n[3] = {nx + 2, ny + 2, nz + 2};
    const int d1 = dim == 0 ? 1 : 0, d2 = dim == 2 ? 1 : 2;
    const int u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y;
    if (u >= n[d1] || v >= n[d2]) return;
    const i64 s[3] = {1, n[0], (i64)n[0] * n[1]};
    const i64 base = u * s[d1] + v * s[d2];
    const int m = n[dim];
    double *p0 = T + base, *p1 = T + base + s[dim], *pm2 = T + base + (i64)(m - 2) * s[dim], *pm1 = T + base + (i64)(m - 1) * s[dim];
    if (step == 0) {
        if (lo_on) *p0 = 2 * lo_val - *p1;
        if (hi_on) *pm1 = 2 * hi_val - *pm2;
    } else if (step == 1) {
        if (lo_on) *p0 = *p1;
        if (hi_on) *pm1 = *pm2;
    } else {
        if (lo_on) *p0 = *pm2;
        if (hi_on) *pm1 = *p1;
    }
}
#undef T3_
#undef CC_

__global__ __launch_bounds__(256) void k_sub3(double *__restrict__ d, const double *__restrict__ a, const double *__restrict__ b, i64 n)
{
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x) d[t] = a[t] - b[t];
}

jrx_status checkT3(jrx_handle *h, const jrx_thermal3d_fields *t, const jrx_thermal3d_params *p)
{
    if (!h) return JRX_ERR_ARG;
    if (!t || !p) return jrx_fail(h, JRX_ERR_ARG, "null thermal fields/params");
    JRX_TRY(jrx_check_device(h));
    if (p->nx < 2 || p->ny < 2 || p->nz < 2) return jrx_fail(h, JRX_ERR_ARG, "thermal grid too small");
    if ((double)(p->nx + 2) * (double)(p->ny + 2) * (double)(p->nz + 2) >= 2147483647.0)
        return jrx_fail(h, JRX_ERR_UNSUPPORTED, "local block too large for 32-bit plane indices");
    const void *req[] = {t->T, t->Told, t->dT, t->qTx, t->qTx2, t->qTy, t->qTy2, t->qTz, t->qTz2, t->H, t->shear_heating, t->ResT, t->thetar_dtau, t->dtau_rho};
    for (const void *q : req)
        if (!q) return jrx_fail(h, JRX_ERR_ARG, "a required thermal field pointer is NULL");
    if (!p->rheology_form && (!t->K || !t->rhoCp)) return jrx_fail(h, JRX_ERR_ARG, "K / rhoCp arrays required in the array-coefficient form");
    if (p->rheology_form != 0 && p->rheology_form != 1) return jrx_fail(h, JRX_ERR_ARG, "rheology_form must be 0 or 1 (phase-ratio form: jrx_heatdiffusion_PT3d_phases)");
    return JRX_OK;
}

jrx_status launch_tbcs3(jrx_handle *h, hipStream_t s, double *T, const jrx_thermal3d_params *p)
{
    const int nx = (int)p->nx, ny = (int)p->ny, nz = (int)p->nz;
    const int n[3] = {nx + 2, ny + 2, nz + 2};
    // per BC type: z faces (bot/top), then x faces (left/right), then y faces (front/back) -- the statement order of the kernels
    const int dims[3] = {2, 0, 1};
    const int lo[3] = {ZB, XL, YF}, hi[3] = {ZT, XR, YB};
    for (int step = 0; step < 3; step++) {
        const int32_t *on = step == 0 ? p->constant_value_on : (step == 1 ? p->no_flux : p->periodic);
        for (int q = 0; q < 3; q++) {
            if (!(on[lo[q]] | on[hi[q]])) continue;
            const int dim = dims[q], d1 = dim == 0 ? 1 : 0, d2 = dim == 2 ? 1 : 2;
            hipLaunchKernelGGL(k_tbc3d, dim3((unsigned)((n[d1] + 255) / 256), (unsigned)n[d2]), dim3(256), 0, s, T, nx, ny, nz, step, dim, on[lo[q]], on[hi[q]],
                               p->constant_value[lo[q]], p->constant_value[hi[q]]);
            JRX_LAUNCH_CHECK(h);
        }
    }
    return JRX_OK;
}

static jrx_status ensure_tscratch(jrx_handle *h, int nx, int ny, int nz)
{
    if (h->tscratch[0] && h->tscratch_dims[0] == nx && h->tscratch_dims[1] == ny && h->tscratch_dims[2] == nz) return JRX_OK;
    for (int q = 0; q < 4; q++) {
        if (h->tscratch[q]) JRX_TRY(jrx_dev_free(h, h->tscratch[q]));
        h->tscratch[q] = nullptr;
    }
    h->tscratch_dims[0] = h->tscratch_dims[1] = h->tscratch_dims[2] = 0;
    const size_t n[4] = {(size_t)(nx + 2) * (ny + 2) * (nz + 2), (size_t)(nx + 1) * ny * nz, (size_t)nx * (ny + 1) * nz, (size_t)nx * ny * (nz + 1)};
    for (int q = 0; q < 4; q++) JRX_TRY(jrx_dev_alloc(h, n[q] * sizeof(double), (void **)&h->tscratch[q], 1));
    h->tscratch_dims[0] = nx; h->tscratch_dims[1] = ny; h->tscratch_dims[2] = nz;
    return JRX_OK;
}

template <class PHT>
jrx_status enqueue_titer3(jrx_handle *h, const jrx_thermal3d_fields *t, const jrx_thermal3d_params *p, const PHT &ph, bool q2 = true, bool fuse_bc = false, bool wpt = false)
{
    T3Args a;
    a.t = *t; a.p = *p; a.wpt = wpt;
    const int nx = (int)p->nx, ny = (int)p->ny, nz = (int)p->nz;
    hipStream_t s = h->stream;
    if constexpr (tph_np<PHT>::value > 0) {       // phase count as a constant: the form with batched loads
        if (q2) hipLaunchKernelGGL((k_flux3d_b<true, tph_np<PHT>::value>), GRID_IJK(nx + 1, ny + 1, nz + 1), dim3(256), 0, s, a, ph);
        else hipLaunchKernelGGL((k_flux3d_b<false, tph_np<PHT>::value>), GRID_IJK(nx + 1, ny + 1, nz + 1), dim3(256), 0, s, a, ph);
    } else if (q2) hipLaunchKernelGGL((k_flux3d<true, PHT>), GRID_IJK(nx + 1, ny + 1, nz + 1), dim3(256), 0, s, a, ph);
    else hipLaunchKernelGGL((k_flux3d<false, PHT>), GRID_IJK(nx + 1, ny + 1, nz + 1), dim3(256), 0, s, a, ph);
    JRX_LAUNCH_CHECK(h);
    bool any_periodic = false;
    for (int q = 0; q < 6; q++) any_periodic |= p->periodic[q] != 0;
    if (fuse_bc && !any_periodic && !jrx_comm_active(h)) {        // thermal_bcs! refreshed by the update kernel itself
        hipLaunchKernelGGL((k_updateT3d<false, true, PHT>), GRID_IJK(nx, ny, nz), dim3(256), 0, s, a, ph);
        JRX_LAUNCH_CHECK(h);
        return JRX_OK;
    }
    hipLaunchKernelGGL((k_updateT3d<false, false, PHT>), GRID_IJK(nx, ny, nz), dim3(256), 0, s, a, ph);
    JRX_LAUNCH_CHECK(h);
    JRX_TRY(launch_tbcs3(h, s, t->T, p));
    if (jrx_comm_active(h)) {
        double *arrs[1] = {t->T};
        const int64_t ext[1][3] = {{nx + 2, ny + 2, nz + 2}};
        const int64_t n[3] = {nx, ny, nz};
        JRX_TRY(jrx_halo_exchange(h, s, 1, arrs, ext, n));
    }
    return JRX_OK;
}

}   // namespace

extern "C" {

jrx_status jrx_thermal_bcs3d(jrx_handle *h, double *T, const jrx_thermal3d_params *p)
{
    if (!h) return JRX_ERR_ARG;
    if (!T || !p) return jrx_fail(h, JRX_ERR_ARG, "thermal_bcs!: null argument");
    JRX_TRY(launch_tbcs3(h, h->stream, T, p));
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_thermal3d_iteration(jrx_handle *h, const jrx_thermal3d_fields *t, const jrx_thermal3d_params *p)
{
    JRX_TRY(checkT3(h, t, p));
    JRX_TRY(enqueue_titer3(h, t, p, NoPh{}));
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_thermal3d_check_res(jrx_handle *h, const jrx_thermal3d_fields *t, const jrx_thermal3d_params *p)
{
    JRX_TRY(checkT3(h, t, p));
    T3Args a;
    a.t = *t; a.p = *p;
    hipLaunchKernelGGL((k_updateT3d<true, false, NoPh>), GRID_IJK(p->nx, p->ny, p->nz), dim3(256), 0, h->stream, a, NoPh{});
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

}   // extern "C"

namespace {
template <class PHT>
jrx_status heat3d(jrx_handle *h, const jrx_thermal3d_fields *t, const jrx_thermal3d_params *p, const PHT &ph, int64_t *iter_count, double *norm_ResT,
                  int64_t cap, int64_t *nnorms)
{
    constexpr bool PH = is_tph<PHT>::value;
    if (p->nout < 1) return jrx_fail(h, JRX_ERR_ARG, "nout must be >= 1");
    const int nx = (int)p->nx, ny = (int)p->ny, nz = (int)p->nz;
    const i64 nT = (i64)(nx + 2) * (ny + 2) * (nz + 2), n = (i64)nx * ny * nz;
    hipStream_t s = h->stream;
    const double sq = 1.0 / sqrt((double)n);
    JRX_HIP(h, hipMemcpyAsync(t->Told, t->T, (size_t)nT * sizeof(double), hipMemcpyDeviceToDevice, s));   // @copy thermal.Told thermal.T
    int64_t iter = 0, cnt = 0;
    double err = 2 * p->eps;
    bool pt_fresh = false;
    T3Args a;
    a.t = *t; a.p = *p;
    a.nt = h->thermal_nt;
    // Iterations nobody observes run as one fused launch that ping-pongs (T, qT) between the caller's arrays and a library-owned set
    // (option "thermal_fused" = 0: always the two kernels); observed ones (check / last) run the two kernels in place on the current set.
    bool any_periodic = false;
    for (int q = 0; q < 6; q++) any_periodic |= p->periodic[q] != 0;
    // the phase-ratio form fuses, too, where the phase count is a template constant (k_thermal3d_fused_ph; its default tile shape only): θr_dτ then ping-pongs with T and the fluxes
    constexpr int NPHC = tph_np<PHT>::value;
    const bool fusable = (!PH || (NPHC > 0 && h->thermal_cfg == 0 && h->thermal_xg == 8 && h->thermal_fused_ph)) && !t->adiabatic && !t->dirichlet_mask && h->thermal_fused &&
                         h->scratch_sets && !any_periodic && !jrx_comm_active(h);
    const TSet user = {t->T, t->qTx, t->qTy, t->qTz};
    TSet cur = user, oth = user;
    double *const th_user = const_cast<double *>(t->thetar_dtau);
    double *th_cur = th_user, *th_oth = th_user;
    if (fusable) {
        JRX_TRY(ensure_tscratch(h, nx, ny, nz));
        oth = TSet{h->tscratch[0], h->tscratch[1], h->tscratch[2], h->tscratch[3]};
        // ghosts that no BC rewrites (prescribed values) must exist in both sets
        JRX_HIP(h, hipMemcpyAsync(oth.T, t->T, (size_t)nT * sizeof(double), hipMemcpyDeviceToDevice, s));
        if (PH) { JRX_TRY(jrx_ensure_etatau(h, (size_t)n)); th_oth = h->etatau; }
    }
    // tile = TX cells of R rows, KZ planes deep; option "thermal_cfg" = R*10000 + (TX/64)*100 + KZ overrides (tuning)
    const int cfg = h->thermal_cfg;
    const int FR = cfg ? cfg / 10000 : 1, FTX = cfg ? ((cfg / 100) % 100) * 64 : (nx > 128 ? 256 : (nx > 64 ? 128 : 64));
    // planes per block: 4, halved while the launch has fewer than 4096 waves -- a block marches its planes one after the other, and a small grid in 4-plane chunks leaves most of the
    // chip idle (64^3: 1024 single-wave blocks): 16^3 74.7 k -> 175 k it/s, 32^3 72.4 k -> 159 k, 48^3 64.3 k -> 112 k, 64^3 52.5 k -> 101 k with one plane per block; from 96^3 on
    // 4 planes are the better depth again (35.5 k against 33.2 k) (profiles/r03_small_grids_graphs.txt)
    int FKZ = cfg ? cfg % 100 : 4;
    const int FXG = h->thermal_xg;
    // (the shallower chunks are instantiated for the default XCD band width only: with the tuning switch "thermal_xg" != 8 the depth stays 4)
    if (!cfg && FXG == 8)
        while (FKZ > 1 && (i64)((nx + FTX - 1) / FTX) * (FTX / 64) * ny * ((nz + FKZ - 1) / FKZ) < 4096) FKZ /= 2;
    const int ntx = (nx + FTX - 1) / FTX, nty = (ny + FR - 1) / FR, ntz = (nz + FKZ - 1) / FKZ;
    // launch_fused: one unobserved iteration from set c into set o (the caller swaps)
    auto launch_fused = [&](const TSet &c, const TSet &o, double *thc = nullptr, double *tho = nullptr) -> jrx_status {
        T3Args b = a;
        b.t.T = c.T; b.t.qTx = c.qx; b.t.qTy = c.qy; b.t.qTz = c.qz;
        if constexpr (NPHC > 0) {
            b.t.thetar_dtau = thc;
            const TSetP op = {o.T, o.qx, o.qy, o.qz, tho};
            bool ok = true;
#define THP(TX_, KZ_) if (FTX == TX_ && FKZ == KZ_) hipLaunchKernelGGL((k_thermal3d_fused_ph<TX_, KZ_, 8, NPHC>), dim3((unsigned)(ntx * nty * ntz)), dim3(TX_), 0, s, b, op, ph, ntx, nty); else
            THP(256, 4) THP(128, 4) THP(64, 4) THP(256, 2) THP(128, 2) THP(64, 2) THP(256, 1) THP(128, 1) THP(64, 1) ok = false;
#undef THP
            if (!ok) return jrx_fail(h, JRX_ERR_ARG, "internal: no fused phase-ratio instantiation for this tile shape");
            h->stat_thermal_fused++;
            JRX_LAUNCH_CHECK(h);
            return JRX_OK;
        }
#define THL(TX_, KZ_, R_, XG_)                                                                                                      \
    if (FTX == TX_ && FKZ == KZ_ && FR == R_ && FXG == XG_) {                                                                       \
        hipLaunchKernelGGL((k_thermal3d_fused<TX_, KZ_, XG_, R_>), dim3((unsigned)(ntx * nty * ntz)), dim3(TX_), 0, s, b, o, ntx, nty); \
        launched = true; h->stat_thermal_fused++;                                                                                                         \
    }
        bool launched = false;
        if (h->thermal_tile == 4 || h->thermal_tile == 8) {        // round 6: 64 x TY tiles, y neighbours through LDS (k_thermal3d_fused_t)
            const int TYv = h->thermal_tile, ntx_t = (nx + 63) / 64, nty_t = (ny + TYv - 1) / TYv, ntz_t = (nz + 3) / 4;
            const unsigned nb = (unsigned)(ntx_t * nty_t * ntz_t);
            const int xg = h->thermal_xg == 8 ? (TYv == 4 ? 2 : 1) : h->thermal_xg;      // default band: 8 rows per XCD, as the row-segment form has
#define THT(TY_, XG_) if (TYv == TY_ && xg == XG_) { hipLaunchKernelGGL((k_thermal3d_fused_t<TY_, 4, XG_>), dim3(nb), dim3(64 * TY_), 0, s, b, o, ntx_t, nty_t); launched = true; }
            THT(4, 1) THT(4, 2) THT(4, 4) THT(8, 1) THT(8, 2) THT(8, 4)
#undef THT
            if (launched) { h->stat_thermal_fused++; JRX_LAUNCH_CHECK(h); return JRX_OK; }
        }
        // measured at 256^3 (profiles/r01_thermal3d_fused_sweep.txt): one row per thread 2289 it/s, two rows 1526, four rows 1301
        THL(256, 4, 1, 8) THL(128, 4, 1, 8) THL(64, 4, 1, 8) THL(256, 8, 1, 8) THL(256, 4, 2, 8)
        THL(256, 4, 1, 1) THL(256, 4, 1, 2) THL(256, 4, 1, 4) THL(256, 2, 1, 1) THL(256, 2, 1, 2) THL(256, 8, 1, 1) THL(128, 4, 1, 1) THL(64, 4, 1, 1)
        THL(64, 2, 1, 8) THL(64, 1, 1, 8) THL(128, 2, 1, 8) THL(128, 1, 1, 8) THL(256, 2, 1, 8) THL(256, 1, 1, 8)
        if (!launched) return jrx_fail(h, JRX_ERR_ARG, "JRX_TH_CFG: no such configuration");
#undef THL
        JRX_LAUNCH_CHECK(h);
        return JRX_OK;
    };
    // small grids: runs of unobserved one-launch iterations replay as captured graphs of GIT iterations (an even count: the ping-pong sets end where they
    // began; one graph per parity), as the 2D loop does (option "loop_graphs"; profiles/r03_small_grids_graphs.txt)
    constexpr int GIT = 32;
    GraphExecs gexec;
    bool graphs = h->loop_graphs && fusable && (double)n <= kGraphCells3D;
    while (err > p->eps && iter < p->iterMax) {
        if (graphs) {
            // observed iterations (1-based number it1 = iter + 1): the multiples of nout and every it1 >= iterMax
            int64_t nxt = ((iter / p->nout) + 1) * p->nout;
            if (nxt > p->iterMax) nxt = p->iterMax;
            int64_t run = nxt - 1 - iter;
            if (run >= GIT) {
                const int par = cur.T == user.T ? 0 : 1;
                if (!gexec[par]) {
                    JRX_TRY(jrx_capture_graph(s, &gexec[par], [&]() -> jrx_status {
                        TSet c = cur, o = oth;
                        double *tc_ = th_cur, *to_ = th_oth;
                        for (int q = 0; q < GIT; q++) {
                            JRX_TRY(launch_fused(c, o, tc_, to_));
                            const TSet tmp = c; c = o; o = tmp;
                            double *tt_ = tc_; tc_ = to_; to_ = tt_;
                        }
                        return JRX_OK;
                    }));
                    if (!gexec[par]) graphs = false;
                    else h->stat_thermal_fused -= GIT;          // the capture counted launches that have not run
                }
                if (gexec[par]) {
                    while (run >= GIT) {
                        JRX_HIP(h, hipGraphLaunch(gexec[par], s));
                        iter += GIT; run -= GIT;
                        h->stat_thermal_fused += GIT; h->stat_graph_replays++;
                    }
                    continue;
                }
            }
        }
        // qT*2 is observable after the loop as well (the arrays belong to the caller): written on check iterations and on the last one
        const bool q2 = ((iter + 1) % p->nout == 0) || (iter + 1 >= p->iterMax);
        a.t.T = cur.T; a.t.qTx = cur.qx; a.t.qTy = cur.qy; a.t.qTz = cur.qz;
        a.t.thetar_dtau = th_cur;
        if constexpr (PH) {    // update_pt_thermal_arrays!(pt_thermal, phase, rheology, args, _dt) -- DiffusionPT_solver.jl:233-234
            // on unobserved iterations update_T! writes the coefficients of the next iteration itself (same values: they depend on the cell's own new T
            // only); observed iterations leave pt_thermal as the reference does, and the stand-alone kernel runs before the following iteration
            if (!pt_fresh) JRX_TRY(jrx_enqueue_pt_thermal_arrays(h, s, th_cur, const_cast<double *>(t->dtau_rho), cur.T, nx, ny, nz, 3, 1.0 / p->dt, ph));
            pt_fresh = !q2;
            a.wpt = pt_fresh;
        }
        if (fusable && !q2) {
            JRX_TRY(launch_fused(cur, oth, th_cur, th_oth));
            const TSet tmp = cur; cur = oth; oth = tmp;
            if (PH) { double *tt_ = th_cur; th_cur = th_oth; th_oth = tt_; }
        } else {
            JRX_TRY(enqueue_titer3(h, &a.t, p, ph, q2, true, a.wpt));
        }
        iter++;
        if (iter % p->nout == 0) {
            hipLaunchKernelGGL((k_updateT3d<true, false, PHT>), GRID_IJK(nx, ny, nz), dim3(256), 0, s, a, ph);
            JRX_LAUNCH_CHECK(h);
            RedArr Z = {nullptr, {0, 0, 0}, 0}, A3 = {t->ResT, {nx, ny, nz}, 0};
            int nb = (int)((n + 2047) / 2048);
            nb = nb < 1 ? 1 : (nb > kMaxRedBlocks ? kMaxRedBlocks : nb);
            hipLaunchKernelGGL(k_sumsq_partial, dim3(nb), dim3(256), 0, s, Z, Z, Z, A3, h->d_partials);
            JRX_LAUNCH_CHECK(h);
            hipLaunchKernelGGL(k_sumsq_final, dim3(1), dim3(256), 0, s, h->d_partials, nb, h->d_sums);
            JRX_LAUNCH_CHECK(h);
            JRX_HIP(h, hipMemcpyAsync(h->h_sums, h->d_sums, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
            JRX_HIP(h, hipStreamSynchronize(s));
            err = sqrt(h->h_sums[3]) * sq;          // norm(ResT) * _sq_len_RT (local norm, DiffusionPT_solver.jl:131)
            // The reference stops each rank on its own local norm (no MPI reduction at :131); with an exchange in every iteration ranks that
            // cross ϵ at different checks would then wait for each other for ever.  Deviation: with more than one rank the loop test uses the
            // maximum of the local norms, so that all ranks leave together; the reported norm_ResT stays the local one.
            double err_stop = err;
            if (jrx_comm_active(h)) JRX_TRY(jrx_allreduce_host(h, &err_stop, 1, 1));
            if (cnt < cap) {
                if (norm_ResT) norm_ResT[cnt] = err;
                if (iter_count) iter_count[cnt] = iter;
            }
            cnt++;
            if (p->verbose && jrx_comm_rank(h) == 0) printf("iter = %lld, err = %1.3e \n", (long long)iter, err);
            err = err_stop;
        }
    }
    if (th_cur != th_user) JRX_HIP(h, hipMemcpyAsync(th_user, th_cur, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));      // ... and the PT coefficients of the phase-ratio form
    if (cur.T != user.T) {      // leave the results in the caller's arrays
        JRX_HIP(h, hipMemcpyAsync(user.T, cur.T, (size_t)nT * sizeof(double), hipMemcpyDeviceToDevice, s));
        JRX_HIP(h, hipMemcpyAsync(user.qx, cur.qx, (size_t)(nx + 1) * ny * nz * sizeof(double), hipMemcpyDeviceToDevice, s));
        JRX_HIP(h, hipMemcpyAsync(user.qy, cur.qy, (size_t)nx * (ny + 1) * nz * sizeof(double), hipMemcpyDeviceToDevice, s));
        JRX_HIP(h, hipMemcpyAsync(user.qz, cur.qz, (size_t)nx * ny * (nz + 1) * sizeof(double), hipMemcpyDeviceToDevice, s));
    }
    hipLaunchKernelGGL(k_sub3, dim3(1024), dim3(256), 0, s, t->dT, (const double *)t->T, (const double *)t->Told, nT);   // update_ΔT!
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(s));
    if (nnorms) *nnorms = cnt < cap ? cnt : cap;
    return JRX_OK;
}
}   // namespace

extern "C" {

jrx_status jrx_heatdiffusion_PT3d(jrx_handle *h, const jrx_thermal3d_fields *t, const jrx_thermal3d_params *p, int64_t *iter_count, double *norm_ResT,
                                  int64_t cap, int64_t *nnorms)
{
    JRX_TRY(checkT3(h, t, p));
    return heat3d(h, t, p, NoPh{}, iter_count, norm_ResT, cap, nnorms);
}

jrx_status jrx_heatdiffusion_PT3d_phases(jrx_handle *h, const jrx_thermal3d_fields *t, const jrx_thermal3d_params *p, const jrx_thermal_phases *ph,
                                         const jrx_thermal_phase_fields *pf, int64_t *iter_count, double *norm_ResT, int64_t cap, int64_t *nnorms)
{
    if (!h) return JRX_ERR_ARG;
    if (!p) return jrx_fail(h, JRX_ERR_ARG, "null thermal params");
    jrx_thermal3d_params q = *p;
    q.rheology_form = 1;                        // K / rhoCp arrays are not read
    JRX_TRY(checkT3(h, t, &q));
    JRX_TRY(tph_check(h, ph, pf, true));
    q.rheology_form = 2;
    TPh x;
    x.m = *ph; x.f = *pf;
    // option "thermal_np_const" (default): the instantiations with the phase count as a constant
    switch (h->thermal_np_const ? ph->nphase : 0) {
#define TPN(N_) case N_: { TPhN<N_> y; y.m = *ph; y.f = *pf; return heat3d(h, t, &q, y, iter_count, norm_ResT, cap, nnorms); }
    TPN(1) TPN(2) TPN(3) TPN(4)
#undef TPN
    default: break;
    }
    return heat3d(h, t, &q, x, iter_count, norm_ResT, cap, nnorms);
}

}   // extern "C"
