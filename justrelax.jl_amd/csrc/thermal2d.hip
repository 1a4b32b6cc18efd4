// thermal2d.hip -- 2D pseudo-transient heat diffusion for gfx950.
//
// Reference being replaced: src/thermal_diffusion/DiffusionPT_solver.jl:34-149 (array-coefficient
// form) and :181-305 (rheology form as test/test_diffusion2D.jl evaluates it: constant k, constant
// Cp, rho = rho0*(1 - alpha*(T - T0)) -- the PT_Density form of the un-vendored GeoParams.jl is an
// assumption), kernels DiffusionPT_kernels.jl:327-440,519-601,603-673 and
// thermal_bcs! (BoundaryConditions.jl:39-53; constant_value.jl:1-13; free_slip.jl:72-84; periodic.jl:1-13).
#include "jrx_internal.hpp"
#include "jrx_kernels.hpp"
#include "jrx_thermal_phases.hpp"

namespace {

enum { TL = 0, TR = 1, TT = 2, TB = 3 };

struct TArgs {
    jrx_thermal2d_fields t;
    jrx_thermal2d_params p;
    bool wpt = false;      // phase-ratio form: update_T! also writes next iteration's θr_dτ, dτ_ρ (update_pt_thermal_arrays! folded in)
};

__device__ __forceinline__ double rhoCp_of(const jrx_thermal2d_params &p, const double *rhoCp, i64 c, double T)
{
    return p.rheology_form ? p.Cp * (p.rho0 * (1.0 - p.alpha * (T - p.T0))) : rhoCp[c];
}

// blocks are dealt round-robin to the 8 XCDs (own L2 each): block L works on position (L % 8) * (T / 8) + L / 8 of the flattened node sequence, so that
// an XCD owns a contiguous band of rows and the rows j +- 1 of its stencils are in its own L2
__device__ __forceinline__ unsigned xcd_slab_block()
{
    const unsigned L = blockIdx.x, per = gridDim.x / 8u;
    return L < per * 8u ? (L & 7u) * per + (L >> 3) : L;
}

// compute_flux! over (nx+1, ny+1); PHT = TPh: conductivity from the face phase ratios
template <class PHT>
__global__ __launch_bounds__(256) void k_flux2d(const TArgs a, const PHT ph)
{
    constexpr bool PH = is_tph<PHT>::value;
    const int nx = (int)a.p.nx, ny = (int)a.p.ny;
    const int t = xcd_slab_block() * blockDim.x + threadIdx.x;
    const int j = t / (nx + 1), i = t - j * (nx + 1);
    if (j > ny) return;
    const double *__restrict__ T = a.t.T, *__restrict__ th = a.t.thetar_dtau;
#define TT_(i_, j_) T[(i_) + (i64)(nx + 2) * (j_)]
    if (j < ny) {
        const i64 q = i + (i64)(nx + 1) * j;
        if (i == 0 && a.p.constant_flux_on[TL]) a.t.qTx[q] = a.p.constant_flux[TL];
        else if (i == nx && a.p.constant_flux_on[TR]) a.t.qTx[q] = a.p.constant_flux[TR];
        else {
            const int iL = clampi(i - 1, 0, nx - 1), iR = clampi(i, 0, nx - 1);
            double Kx;
            if constexpr (PH) Kx = (tph_cond<tph_np<PHT>::value>(ph.m, ph.f.phase_qx + tph_nph(ph) * (iL + (i64)(nx + 1) * j)) + tph_cond<tph_np<PHT>::value>(ph.m, ph.f.phase_qx + tph_nph(ph) * (iR + (i64)(nx + 1) * j))) * 0.5;
            else Kx = a.p.rheology_form ? (a.p.k_const + a.p.k_const) * 0.5 : (a.t.K[iL + (i64)nx * j] + a.t.K[iR + (i64)nx * j]) * 0.5;
            const double thx = (th[iL + (i64)nx * j] + th[iR + (i64)nx * j]) * 0.5;
            const double qx = -Kx * (TT_(i + 1, j + 1) - TT_(i, j + 1)) * (a.p.inv_spacing[0] ? a.p.inv_spacing[0][clampi(i, 0, nx - 2)] : a.p._dx);
            a.t.qTx2[q] = qx;
            a.t.qTx[q] = (a.t.qTx[q] * thx + qx) / (1.0 + thx);
        }
    }
    if (i < nx) {
        const i64 q = i + (i64)nx * j;
        if (j == 0 && a.p.constant_flux_on[TB]) a.t.qTy[q] = a.p.constant_flux[TB];
        else if (j == ny && a.p.constant_flux_on[TT]) a.t.qTy[q] = a.p.constant_flux[TT];
        else {
            const int jB = clampi(j - 1, 0, ny - 1), jT = clampi(j, 0, ny - 1);
            double Ky;
            if constexpr (PH) Ky = (tph_cond<tph_np<PHT>::value>(ph.m, ph.f.phase_qy + tph_nph(ph) * (i + (i64)nx * jB)) + tph_cond<tph_np<PHT>::value>(ph.m, ph.f.phase_qy + tph_nph(ph) * (i + (i64)nx * jT))) * 0.5;
            else Ky = a.p.rheology_form ? (a.p.k_const + a.p.k_const) * 0.5 : (a.t.K[i + (i64)nx * jB] + a.t.K[i + (i64)nx * jT]) * 0.5;
            const double thy = (th[i + (i64)nx * jB] + th[i + (i64)nx * jT]) * 0.5;
            const double qy = -Ky * (TT_(i + 1, j + 1) - TT_(i + 1, j)) * (a.p.inv_spacing[1] ? a.p.inv_spacing[1][clampi(j, 0, ny - 2)] : a.p._dy);
            a.t.qTy2[q] = qy;
            a.t.qTy[q] = (a.t.qTy[q] * thy + qy) / (1.0 + thy);
        }
    }
}

// thermal_bcs! restricted to the ghost cells next to one interior cell: the reference's statement order (constant_value rows, columns;
// no_flux rows, columns -- BoundaryConditions.jl:46-54, constant_value.jl:1-13, free_slip.jl:72-84) replayed on the 2 x 2 patch
// {corner ghost c, row ghost r, column ghost l, interior m}.  xs/ys: which column / row face the cell touches (0 low, 1 high, -1 none).
__device__ __forceinline__ void thermal_ghosts2d(const jrx_thermal2d_params &p, double *__restrict__ T, int n1, i64 I1, int xs, int ys, double m)
{
    const int fx = xs < 0 ? -1 : (xs == 0 ? TL : TR), fy = ys < 0 ? -1 : (ys == 0 ? TB : TT);
    const i64 dl = xs == 0 ? -1 : 1, dr = ys == 0 ? -(i64)n1 : (i64)n1;      // offsets to the column ghost / row ghost
    double l = fx >= 0 ? T[I1 + dl] : 0.0, r = fy >= 0 ? T[I1 + dr] : 0.0, c = (fx >= 0 && fy >= 0) ? T[I1 + dl + dr] : 0.0;
    bool wl = false, wr = false, wc = false;
    for (int step = 0; step < 2; step++) {
        const int32_t *on = step == 0 ? p.constant_value_on : p.no_flux;
        if (fy >= 0 && on[fy]) {         // rows (bot / top): T[i, ghost] for every i, including the ghost column
            if (fx >= 0) { c = step == 0 ? 2 * p.constant_value[fy] - l : l; wc = true; }
            r = step == 0 ? 2 * p.constant_value[fy] - m : m; wr = true;
        }
        if (fx >= 0 && on[fx]) {         // columns (left / right): T[ghost, j] for every j, including the ghost row
            if (fy >= 0) { c = step == 0 ? 2 * p.constant_value[fx] - r : r; wc = true; }
            l = step == 0 ? 2 * p.constant_value[fx] - m : m; wl = true;
        }
    }
    if (wl) T[I1 + dl] = l;
    if (wr) T[I1 + dr] = r;
    if (wc) T[I1 + dl + dr] = c;
}

// update_T! (RES=false) / check_res! (RES=true) over ni; BCF: cells next to a face also apply thermal_bcs! to their ghosts
template <bool RES, bool BCF, class PHT>
__global__ __launch_bounds__(256) void k_updateT2d(const TArgs a, const PHT ph)
{
    constexpr bool PH = is_tph<PHT>::value;
    const int nx = (int)a.p.nx, ny = (int)a.p.ny;
    const int t = xcd_slab_block() * blockDim.x + threadIdx.x;
    const int j = t / nx, i = t - j * nx;
    if (j >= ny) return;
    const i64 c = i + (i64)nx * j, I1 = (i + 1) + (i64)(nx + 2) * (j + 1);
    const double _dt = 1.0 / a.p.dt;
    const double _dx = a.p.inv_spacing[2] ? a.p.inv_spacing[2][i] : a.p._dx, _dy = a.p.inv_spacing[3] ? a.p.inv_spacing[3][j] : a.p._dy;      // _di.vertex on a non-uniform grid
    const double Tij = a.t.T[I1];
    double rcp, Hr = 0.0;
    if constexpr (PH) {
        const double *rc = ph.f.phase_c + tph_nph(ph) * c;
        rcp = tph_rhoCp<tph_np<PHT>::value>(ph.m, rc, Tij, ph.f.P[c]);
        Hr = tph_Hr<tph_np<PHT>::value>(ph.m, rc);
    } else rcp = rhoCp_of(a.p, a.t.rhoCp, c, Tij);
    // optional terms: + adiabatic * T in the rheology forms; Dirichlet cells (mask != 0) take (1 - m) T + m value and have no residual
    const bool hasadi = a.p.rheology_form != 0 && a.t.adiabatic != nullptr;
    const double adi = hasadi ? a.t.adiabatic[c] * Tij : 0.0;
    const double dm = a.t.dirichlet_mask ? a.t.dirichlet_mask[I1] : 0.0;
    if (RES) {
        const double dq = (a.t.qTx2[(i + 1) + (i64)(nx + 1) * j] - a.t.qTx2[i + (i64)(nx + 1) * j]) * _dx +
                          (a.t.qTy2[i + (i64)nx * (j + 1)] - a.t.qTy2[c]) * _dy;
        if (dm != 0.0) a.t.ResT[c] = 0.0;
        else if constexpr (PH) a.t.ResT[c] = hasadi ? -rcp * (Tij - a.t.Told[I1]) * _dt - dq + Hr + a.t.H[c] + a.t.shear_heating[c] + adi
                                                 : -rcp * (Tij - a.t.Told[I1]) * _dt - dq + Hr + a.t.H[c] + a.t.shear_heating[c];
        else a.t.ResT[c] = hasadi ? -rcp * (Tij - a.t.Told[I1]) * _dt - dq + a.t.H[c] + a.t.shear_heating[c] + adi
                                  : -rcp * (Tij - a.t.Told[I1]) * _dt - dq + a.t.H[c] + a.t.shear_heating[c];
    } else {
        const double dr = a.t.dtau_rho[c];
        const double divq = (a.t.qTx[(i + 1) + (i64)(nx + 1) * j] - a.t.qTx[i + (i64)(nx + 1) * j]) * _dx +
                            (a.t.qTy[i + (i64)nx * (j + 1)] - a.t.qTy[c]) * _dy;
        double Tn;
        if (dm != 0.0) Tn = (1 - dm) * Tij + dm * (a.t.dirichlet_value ? a.t.dirichlet_value[I1] : a.p.dirichlet_const);
        else if constexpr (PH) Tn = hasadi ? (dr * (-divq + a.t.Told[I1] * rcp * _dt + Hr + a.t.H[c] + a.t.shear_heating[c] + adi) + Tij) / (1.0 + dr * rcp * _dt)
                                           : (dr * (-divq + a.t.Told[I1] * rcp * _dt + Hr + a.t.H[c] + a.t.shear_heating[c]) + Tij) / (1.0 + dr * rcp * _dt);
        else Tn = hasadi ? (dr * (-divq + a.t.Told[I1] * rcp * _dt + a.t.H[c] + a.t.shear_heating[c] + adi) + Tij) / (1.0 + dr * rcp * _dt)
                         : (dr * (-divq + a.t.Told[I1] * rcp * _dt + a.t.H[c] + a.t.shear_heating[c]) + Tij) / (1.0 + dr * rcp * _dt);
        a.t.T[I1] = Tn;
        if constexpr (PH) {
            if (a.wpt) {      // update_pt_thermal_arrays! of the next iteration (DiffusionPT_coefficients.jl:123-136) from the new T of this cell
                double th_, dr_;
                tph_pt_coeffs<tph_np<PHT>::value>(ph.m, ph.f.phase_c + tph_nph(ph) * c, Tn, ph.f.P[c], _dt, th_, dr_);
                a.t.thetar_dtau[c] = th_;
                a.t.dtau_rho[c] = dr_;
            }
        }
        if (BCF) {
            const int xs = i == 0 ? 0 : (i == nx - 1 ? 1 : -1), ys = j == 0 ? 0 : (j == ny - 1 ? 1 : -1);
            if (xs >= 0 || ys >= 0) thermal_ghosts2d(a.p, a.t.T, nx + 2, I1, xs, ys, Tn);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// One PT iteration in one launch (compute_flux! + update_T! + thermal_bcs!, DiffusionPT_solver.jl:104-111) for iterations nobody
// observes -- the 2D loop is launch-bound, so this halves its cost.  A thread owns cell (i, j): it stores the fluxes on its two low
// faces (and on the domain's high faces), takes the x flux of its high face from the next lane (wave-edge lanes compute it), recomputes
// the y one; (T, qT) are read from one set and written to the other because neighbouring threads recompute what another thread
// stores.  The un-relaxed fluxes qT*2 are not written (only check_res! reads them: observed iterations run the two kernels).
// Same arithmetic, in the same order, as k_flux2d / k_updateT2d.
// ------------------------------------------------------------------------------------------------
struct TSet2 { double *T, *qx, *qy; };
template <int TX>
__global__ __launch_bounds__(TX) void k_thermal2d_fused(const TArgs a, const TSet2 dst, int ntx)
{
    const int nx = (int)a.p.nx, ny = (int)a.p.ny;
    const unsigned bid = xcd_slab_block();
    const int tix = bid % ntx, j = bid / ntx;
    const int i = tix * TX + (int)threadIdx.x;
    if (i >= nx && (i & ~63) >= nx) return;            // whole waves beyond the row end; idle lanes of a live wave stay for the shuffle
    const bool cell = i < nx;
    const int ic = cell ? i : nx - 1;
    const double *__restrict__ T = a.t.T, *__restrict__ th = a.t.thetar_dtau, *__restrict__ Kk = a.t.K;
    const bool rf = a.p.rheology_form != 0;
    const double kc = (a.p.k_const + a.p.k_const) * 0.5;
    const i64 n1 = nx + 2;
    const i64 c = ic + (i64)nx * j, I1 = (ic + 1) + n1 * (j + 1);
    const int im = max(ic - 1, 0), ip = min(ic + 1, nx - 1), jm = max(j - 1, 0), jp = min(j + 1, ny - 1);
    auto relax = [&](double qold, double Kl, double Kr, double tl, double tr_, double Thi, double Tlo, double _d) -> double {
        const double K = rf ? kc : (Kl + Kr) * 0.5;
        const double t = (tl + tr_) * 0.5;
        const double qv = -K * (Thi - Tlo) * _d;
        return (qold * t + qv) / (1.0 + t);
    };
    const double Tc = T[I1], Kc_ = rf ? 0.0 : Kk[c], tc = th[c];
    // x: low face i (own), high face i+1 from the next lane
    double qx_lo;
    {
        const i64 q = ic + (i64)(nx + 1) * j;
        if (ic == 0 && a.p.constant_flux_on[TL]) qx_lo = a.p.constant_flux[TL];
        else {
            const i64 cl = c - (ic - im);
            qx_lo = relax(a.t.qTx[q], rf ? 0.0 : Kk[cl], Kc_, th[cl], tc, Tc, T[I1 - 1], a.p._dx);
        }
        if (cell) dst.qx[q] = qx_lo;
    }
    double qx_hi = __shfl_down(qx_lo, 1, 64);
    if ((threadIdx.x & 63) == 63 || i == nx - 1) {
        const i64 q = (ic + 1) + (i64)(nx + 1) * j;
        if (ic + 1 == nx && a.p.constant_flux_on[TR]) qx_hi = a.p.constant_flux[TR];
        else {
            const i64 cr = c + (ip - ic);
            qx_hi = relax(a.t.qTx[q], Kc_, rf ? 0.0 : Kk[cr], tc, th[cr], T[I1 + 1], Tc, a.p._dx);
        }
        if (cell && ic + 1 == nx) dst.qx[q] = qx_hi;
    }
    // y: low face j (own), high face j+1 recomputed (owned by the row above, or by this row on the top face)
    double qy_lo, qy_hi;
    {
        const i64 q = ic + (i64)nx * j;
        if (j == 0 && a.p.constant_flux_on[TB]) qy_lo = a.p.constant_flux[TB];
        else {
            const i64 cl = c - (i64)nx * (j - jm);
            qy_lo = relax(a.t.qTy[q], rf ? 0.0 : Kk[cl], Kc_, th[cl], tc, Tc, T[I1 - n1], a.p._dy);
        }
        if (cell) dst.qy[q] = qy_lo;
        if (j + 1 == ny && a.p.constant_flux_on[TT]) qy_hi = a.p.constant_flux[TT];
        else {
            const i64 cr = c + (i64)nx * (jp - j);
            qy_hi = relax(a.t.qTy[q + nx], Kc_, rf ? 0.0 : Kk[cr], tc, th[cr], T[I1 + n1], Tc, a.p._dy);
        }
        if (cell && j + 1 == ny) dst.qy[q + nx] = qy_hi;
    }
    if (cell) {
        const double _dt = 1.0 / a.p.dt;
        const double rcp = rhoCp_of(a.p, a.t.rhoCp, c, Tc);
        const double dr = a.t.dtau_rho[c];
        const double divq = (qx_hi - qx_lo) * a.p._dx + (qy_hi - qy_lo) * a.p._dy;
        const double Tn = (dr * (-divq + a.t.Told[I1] * rcp * _dt + a.t.H[c] + a.t.shear_heating[c]) + Tc) / (1.0 + dr * rcp * _dt);
        dst.T[I1] = Tn;
        const int xs = i == 0 ? 0 : (i == nx - 1 ? 1 : -1), ys = j == 0 ? 0 : (j == ny - 1 ? 1 : -1);
        if (xs >= 0 || ys >= 0) thermal_ghosts2d(a.p, dst.T, nx + 2, I1, xs, ys, Tn);
    }
}

// thermal_bcs!: step 0 constant_value, 1 no_flux, 2 periodic; dimy=1: rows j=0 / j=end (bot/top), else columns (left/right)
__global__ __launch_bounds__(256) void k_tbc2d(double *__restrict__ T, int nx, int ny, int step, int dimy,
                                               int lo_on, int hi_on, double lo_val, double hi_val)
{
    const int n1 = nx + 2, n2 = ny + 2;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (dimy) {
        if (t >= n1) return;
#define T2(i_, j_) T[(i_) + (i64)n1 * (j_)]
        if (step == 0) {
            if (lo_on) T2(t, 0) = 2 * lo_val - T2(t, 1);
            if (hi_on) T2(t, n2 - 1) = 2 * hi_val - T2(t, n2 - 2);
        } else if (step == 1) {
            if (lo_on) T2(t, 0) = T2(t, 1);
            if (hi_on) T2(t, n2 - 1) = T2(t, n2 - 2);
        } else {
            if (lo_on) T2(t, 0) = T2(t, n2 - 2);
            if (hi_on) T2(t, n2 - 1) = T2(t, 1);
        }
    } else {
        if (t >= n2) return;
        if (step == 0) {
            if (lo_on) T2(0, t) = 2 * lo_val - T2(1, t);
            if (hi_on) T2(n1 - 1, t) = 2 * hi_val - T2(n1 - 2, t);
        } else if (step == 1) {
            if (lo_on) T2(0, t) = T2(1, t);
            if (hi_on) T2(n1 - 1, t) = T2(n1 - 2, t);
        } else {
            if (lo_on) T2(0, t) = T2(n1 - 2, t);
            if (hi_on) T2(n1 - 1, t) = T2(1, t);
        }
#undef T2
    }
}
#undef TT_

__global__ __launch_bounds__(256) void k_sub(double *__restrict__ d, const double *__restrict__ a, const double *__restrict__ b, i64 n)
{
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x) d[t] = a[t] - b[t];
}

// update_pt_thermal_arrays! (DiffusionPT_coefficients.jl:105-136) over ni of a 2D or 3D grid (nz = 1 in 2D): T read at Idx .+ 1
__global__ __launch_bounds__(256) void k_pt_thermal_arrays(double *__restrict__ th, double *__restrict__ dr, const double *__restrict__ T, int nx, int ny, int nz,
                                                           int ndim, double _dt, const TPh ph)
{
    const i64 n = (i64)nx * ny * nz;
    for (i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (i64)gridDim.x * blockDim.x) {
        const int k = (int)(c / ((i64)nx * ny)), j = (int)((c - (i64)k * nx * ny) / nx), i = (int)(c - (i64)k * nx * ny - (i64)j * nx);
        const i64 I1 = ndim == 3 ? (i + 1) + (i64)(nx + 2) * ((j + 1) + (i64)(ny + 2) * (k + 1)) : (i + 1) + (i64)(nx + 2) * (j + 1);
        double a, b;
        tph_pt_coeffs(ph.m, ph.f.phase_c + ph.m.nphase * c, T[I1], ph.f.P[c], _dt, a, b);
        th[c] = a;
        dr[c] = b;
    }
}

// adiabatic_heating! (DiffusionPT_kernels.jl:720-729): A = (P - P0) * α * _dt, α = phase-weighted expansivity of the density laws
__global__ __launch_bounds__(256) void k_adiabatic(double *__restrict__ A, const double *__restrict__ P, const double *__restrict__ P0, i64 n, double _dt, const TPh ph)
{
    for (i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (i64)gridDim.x * blockDim.x) {
        double al = 0.0;
        for (int q = 0; q < ph.m.nphase; q++) {
            const double r = ph.f.phase_c ? ph.f.phase_c[(i64)ph.m.nphase * c + q] : (q == 0 ? 1.0 : 0.0);
            const double aq = (ph.m.rho_kind[q] == 1 || ph.m.rho_kind[q] == 2) ? ph.m.alpha[q] : 0.0;
            al += (r == 0.0) ? 0.0 : aq * r;
        }
        A[c] = (P[c] - P0[c]) * al * _dt;
    }
}

jrx_status checkT(jrx_handle *h, const jrx_thermal2d_fields *t, const jrx_thermal2d_params *p)
{
    if (!h) return JRX_ERR_ARG;
    if (!t || !p) return jrx_fail(h, JRX_ERR_ARG, "null thermal fields/params");
    JRX_TRY(jrx_check_device(h));
    if (p->nx < 2 || p->ny < 2) return jrx_fail(h, JRX_ERR_ARG, "thermal grid too small");
    const void *req[] = {t->T, t->Told, t->dT, t->qTx, t->qTx2, t->qTy, t->qTy2, t->H, t->shear_heating, t->ResT, t->thetar_dtau, t->dtau_rho};
    for (const void *q : req)
        if (!q) return jrx_fail(h, JRX_ERR_ARG, "a required thermal field pointer is NULL");
    if (!p->rheology_form && (!t->K || !t->rhoCp)) return jrx_fail(h, JRX_ERR_ARG, "K / rhoCp arrays required in the array-coefficient form");
    {
        int nsp = 0;
        for (int q = 0; q < 4; q++) nsp += p->inv_spacing[q] != nullptr;
        if (nsp != 0 && nsp != 4) return jrx_fail(h, JRX_ERR_ARG, "non-uniform grid: all four inverse-spacing arrays are required");
        if (nsp && !p->rheology_form)
            return jrx_fail(h, JRX_ERR_UNSUPPORTED, "non-uniform grid with the array form (K, ρCp): the reference's update_T! indexes _di.center beyond its extent there; use the rheology form");
    }
    if (p->rheology_form != 0 && p->rheology_form != 1) return jrx_fail(h, JRX_ERR_ARG, "rheology_form must be 0 or 1 (phase-ratio form: jrx_heatdiffusion_PT2d_phases)");
    return JRX_OK;
}

jrx_status launch_tbcs(jrx_handle *h, hipStream_t s, double *T, const jrx_thermal2d_params *p)
{
    const int nx = (int)p->nx, ny = (int)p->ny;
    const unsigned gy = (unsigned)((nx + 2 + 255) / 256), gx = (unsigned)((ny + 2 + 255) / 256);
    auto any = [](const int32_t *m) { return m[0] | m[1] | m[2] | m[3]; };
    // each reference kernel does rows (bot/top) then columns (left/right) per thread: rows first, then columns
    if (any(p->constant_value_on)) {
        if (p->constant_value_on[TB] | p->constant_value_on[TT]) {
            hipLaunchKernelGGL(k_tbc2d, dim3(gy), dim3(256), 0, s, T, nx, ny, 0, 1, p->constant_value_on[TB], p->constant_value_on[TT], p->constant_value[TB], p->constant_value[TT]);
            JRX_LAUNCH_CHECK(h);
        }
        if (p->constant_value_on[TL] | p->constant_value_on[TR]) {
            hipLaunchKernelGGL(k_tbc2d, dim3(gx), dim3(256), 0, s, T, nx, ny, 0, 0, p->constant_value_on[TL], p->constant_value_on[TR], p->constant_value[TL], p->constant_value[TR]);
            JRX_LAUNCH_CHECK(h);
        }
    }
    if (any(p->no_flux)) {
        if (p->no_flux[TB] | p->no_flux[TT]) {
            hipLaunchKernelGGL(k_tbc2d, dim3(gy), dim3(256), 0, s, T, nx, ny, 1, 1, p->no_flux[TB], p->no_flux[TT], 0.0, 0.0);
            JRX_LAUNCH_CHECK(h);
        }
        if (p->no_flux[TL] | p->no_flux[TR]) {
            hipLaunchKernelGGL(k_tbc2d, dim3(gx), dim3(256), 0, s, T, nx, ny, 1, 0, p->no_flux[TL], p->no_flux[TR], 0.0, 0.0);
            JRX_LAUNCH_CHECK(h);
        }
    }
    if (any(p->periodic)) {
        if (p->periodic[TB] | p->periodic[TT]) {
            hipLaunchKernelGGL(k_tbc2d, dim3(gy), dim3(256), 0, s, T, nx, ny, 2, 1, p->periodic[TB], p->periodic[TT], 0.0, 0.0);
            JRX_LAUNCH_CHECK(h);
        }
        if (p->periodic[TL] | p->periodic[TR]) {
            hipLaunchKernelGGL(k_tbc2d, dim3(gx), dim3(256), 0, s, T, nx, ny, 2, 0, p->periodic[TL], p->periodic[TR], 0.0, 0.0);
            JRX_LAUNCH_CHECK(h);
        }
    }
    return JRX_OK;
}

// fuse_bc: thermal_bcs! refreshed by the update kernel itself (no periodic face, no neighbour rank, grid at least 2 cells wide)
template <class PHT>
jrx_status enqueue_titer(jrx_handle *h, const jrx_thermal2d_fields *t, const jrx_thermal2d_params *p, const PHT &ph, bool fuse_bc = false, bool wpt = false)
{
    TArgs a;
    a.t = *t; a.p = *p; a.wpt = wpt;
    const int nx = (int)p->nx, ny = (int)p->ny;
    hipStream_t s = h->stream;
    hipLaunchKernelGGL(k_flux2d<PHT>, dim3((unsigned)(((i64)(nx + 1) * (ny + 1) + 255) / 256)), dim3(256), 0, s, a, ph);
    JRX_LAUNCH_CHECK(h);
    const bool any_periodic = p->periodic[0] | p->periodic[1] | p->periodic[2] | p->periodic[3];
    if (fuse_bc && !any_periodic && !jrx_comm_active(h) && nx >= 2 && ny >= 2) {
        hipLaunchKernelGGL((k_updateT2d<false, true, PHT>), dim3((unsigned)(((i64)nx * ny + 255) / 256)), dim3(256), 0, s, a, ph);
        JRX_LAUNCH_CHECK(h);
        return JRX_OK;
    }
    hipLaunchKernelGGL((k_updateT2d<false, false, PHT>), dim3((unsigned)(((i64)nx * ny + 255) / 256)), dim3(256), 0, s, a, ph);
    JRX_LAUNCH_CHECK(h);
    JRX_TRY(launch_tbcs(h, s, t->T, p));
    if (jrx_comm_active(h)) {
        double *arrs[1] = {t->T};
        const int64_t ext[1][3] = {{nx + 2, ny + 2, 1}};
        const int64_t n[3] = {nx, ny, 1};
        JRX_TRY(jrx_halo_exchange(h, s, 1, arrs, ext, n));
    }
    return JRX_OK;
}

}   // namespace

jrx_status jrx_enqueue_pt_thermal_arrays(jrx_handle *h, hipStream_t s, double *th, double *dr, const double *T, int nx, int ny, int nz, int ndim, double _dt, const TPh &ph)
{
    const i64 n = (i64)nx * ny * nz;
    const i64 nb = (n + 255) / 256;
    hipLaunchKernelGGL(k_pt_thermal_arrays, dim3((unsigned)(nb > 8192 ? 8192 : nb)), dim3(256), 0, s, th, dr, T, nx, ny, nz, ndim, _dt, ph);
    JRX_LAUNCH_CHECK(h);
    return JRX_OK;
}

extern "C" {

jrx_status jrx_thermal_bcs2d(jrx_handle *h, double *T, const jrx_thermal2d_params *p)
{
    if (!h) return JRX_ERR_ARG;
    if (!T || !p) return jrx_fail(h, JRX_ERR_ARG, "thermal_bcs!: null argument");
    JRX_TRY(launch_tbcs(h, h->stream, T, p));
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_thermal2d_iteration(jrx_handle *h, const jrx_thermal2d_fields *t, const jrx_thermal2d_params *p)
{
    JRX_TRY(checkT(h, t, p));
    JRX_TRY(enqueue_titer(h, t, p, NoPh{}));
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_thermal2d_check_res(jrx_handle *h, const jrx_thermal2d_fields *t, const jrx_thermal2d_params *p)
{
    JRX_TRY(checkT(h, t, p));
    TArgs a;
    a.t = *t; a.p = *p;
    hipLaunchKernelGGL((k_updateT2d<true, false, NoPh>), dim3((unsigned)(((i64)p->nx * p->ny + 255) / 256)), dim3(256), 0, h->stream, a, NoPh{});
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

}   // extern "C"

namespace {
template <class PHT>
jrx_status heat2d(jrx_handle *h, const jrx_thermal2d_fields *t, const jrx_thermal2d_params *p, const PHT &ph,
                  int64_t *iter_count, double *norm_ResT, int64_t cap, int64_t *nnorms)
{
    constexpr bool PH = is_tph<PHT>::value;
    if (p->nout < 1) return jrx_fail(h, JRX_ERR_ARG, "nout must be >= 1");
    const int nx = (int)p->nx, ny = (int)p->ny;
    const i64 nT = (i64)(nx + 2) * (ny + 2), n = (i64)nx * ny;
    hipStream_t s = h->stream;
    const double sq = 1.0 / sqrt((double)n);
    JRX_HIP(h, hipMemcpyAsync(t->Told, t->T, (size_t)nT * sizeof(double), hipMemcpyDeviceToDevice, s));   // @copy thermal.Told thermal.T
    int64_t iter = 0, cnt = 0;
    double err = 2 * p->eps;
    bool pt_fresh = false;
    TArgs a;
    a.t = *t; a.p = *p;
    // iterations nobody observes: one fused launch, ping-pong between the caller's (T, qT) and a library-owned set (option thermal_fused)
    const bool any_periodic = p->periodic[0] | p->periodic[1] | p->periodic[2] | p->periodic[3];
    const bool fusable = !PH && !t->adiabatic && !t->dirichlet_mask && !p->inv_spacing[0] && h->thermal_fused && h->scratch_sets && !any_periodic && !jrx_comm_active(h) && nx >= 2 && ny >= 2;
    const TSet2 user = {t->T, t->qTx, t->qTy};
    TSet2 cur = user, oth = user;
    if (fusable) {
        if (!(h->tscratch2[0] && h->tscratch2_dims[0] == nx && h->tscratch2_dims[1] == ny)) {
            for (int q = 0; q < 3; q++) { if (h->tscratch2[q]) JRX_HIP(h, hipFree(h->tscratch2[q])); h->tscratch2[q] = nullptr; }
            h->tscratch2_dims[0] = h->tscratch2_dims[1] = 0;
            const size_t sz[3] = {(size_t)nT, (size_t)(nx + 1) * ny, (size_t)nx * (ny + 1)};
            for (int q = 0; q < 3; q++) JRX_HIP(h, hipMalloc(&h->tscratch2[q], sz[q] * sizeof(double)));
            h->tscratch2_dims[0] = nx; h->tscratch2_dims[1] = ny;
        }
        oth = TSet2{h->tscratch2[0], h->tscratch2[1], h->tscratch2[2]};
        JRX_HIP(h, hipMemcpyAsync(oth.T, t->T, (size_t)nT * sizeof(double), hipMemcpyDeviceToDevice, s));    // ghosts no BC rewrites
    }
    const int FTX = nx > 128 ? 256 : (nx > 64 ? 128 : 64), ntx = (nx + FTX - 1) / FTX;
    jrx_thermal2d_fields tc = *t;
    auto fused_launch = [&](const TArgs &aa, const TSet2 &dst) {
        if (FTX == 256) hipLaunchKernelGGL(k_thermal2d_fused<256>, dim3((unsigned)(ntx * ny)), dim3(256), 0, s, aa, dst, ntx);
        else if (FTX == 128) hipLaunchKernelGGL(k_thermal2d_fused<128>, dim3((unsigned)(ntx * ny)), dim3(128), 0, s, aa, dst, ntx);
        else hipLaunchKernelGGL(k_thermal2d_fused<64>, dim3((unsigned)(ntx * ny)), dim3(64), 0, s, aa, dst, ntx);
    };
    // Runs of unobserved fused iterations replay as a captured graph of GIT iterations (an even count: the ping-pong sets end where they started): on these
    // launch-bound grids the gap between dependent launches is shorter inside a graph (scripts/graph_probe.hip: 4.6 vs 5.7 - 6.1 us per pair of short kernels).
    // One graph per parity of the current set, built on first use, destroyed at the end; option "loop_graphs" = 0 keeps plain launches.
    constexpr int GIT = 32;
    GraphExecs gexec;        // released on every exit path
    bool graphs = fusable && h->loop_graphs;
    // the forms that keep the two kernels (phase ratios, adiabatic term, Dirichlet cells) replay compute_flux! + update_T! pairs the same way, in place
    bool graphs2 = !fusable && h->loop_graphs && !any_periodic && !jrx_comm_active(h) && nx >= 2 && ny >= 2 && n <= 200000;
    auto destroy_graphs = [&]() { gexec.reset(); };
    while (err > p->eps && iter < p->iterMax) {
        if (graphs) {
            const int64_t nxt = std::min<int64_t>(((iter / p->nout) + 1) * p->nout, p->iterMax);      // 1-based number of the next observed iteration
            int64_t run = nxt - 1 - iter;                                                              // unobserved iterations from here
            if (run >= GIT) {
                const int par = cur.T == user.T ? 0 : 1;
                if (!gexec[par]) {
                    hipGraph_t g = nullptr;
                    bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
                    if (ok) {
                        TSet2 c = cur, o = oth;
                        for (int q = 0; q < GIT; q++) {
                            TArgs aa = a;
                            aa.t.T = c.T; aa.t.qTx = c.qx; aa.t.qTy = c.qy;
                            fused_launch(aa, o);
                            const TSet2 tmp = c; c = o; o = tmp;
                        }
                        ok = hipStreamEndCapture(s, &g) == hipSuccess && g != nullptr;
                    }
                    if (ok) ok = hipGraphInstantiate(&gexec[par], g, nullptr, nullptr, 0) == hipSuccess;
                    if (g) (void)hipGraphDestroy(g);
                    if (!ok) { (void)hipGetLastError(); gexec[par] = nullptr; graphs = false; }
                }
                if (gexec[par]) {
                    while (run >= GIT) {
                        JRX_HIP(h, hipGraphLaunch(gexec[par], s));
                        iter += GIT; run -= GIT;
                        h->stat_thermal_fused += GIT;
                    }
                    continue;
                }
            }
        }
        if (graphs2 && (!PH || pt_fresh)) {
            const int64_t nxt = std::min<int64_t>(((iter / p->nout) + 1) * p->nout, p->iterMax);
            int64_t run = nxt - 1 - iter;
            if (run >= GIT) {
                if (!gexec[0]) {
                    hipGraph_t g = nullptr;
                    bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
                    if (ok) {
                        TArgs aa = a;
                        aa.t.T = cur.T; aa.t.qTx = cur.qx; aa.t.qTy = cur.qy; aa.wpt = PH;
                        for (int q = 0; q < GIT; q++) {
                            hipLaunchKernelGGL(k_flux2d<PHT>, dim3((unsigned)(((i64)(nx + 1) * (ny + 1) + 255) / 256)), dim3(256), 0, s, aa, ph);
                            hipLaunchKernelGGL((k_updateT2d<false, true, PHT>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, aa, ph);
                        }
                        ok = hipStreamEndCapture(s, &g) == hipSuccess && g != nullptr;
                    }
                    if (ok) ok = hipGraphInstantiate(&gexec[0], g, nullptr, nullptr, 0) == hipSuccess;
                    if (g) (void)hipGraphDestroy(g);
                    if (!ok) { (void)hipGetLastError(); gexec[0] = nullptr; graphs2 = false; }
                }
                if (gexec[0]) {
                    while (run >= GIT) {
                        JRX_HIP(h, hipGraphLaunch(gexec[0], s));
                        iter += GIT; run -= GIT;
                    }
                    continue;
                }
            }
        }
        const bool observed = ((iter + 1) % p->nout == 0) || (iter + 1 >= p->iterMax);
        a.t.T = cur.T; a.t.qTx = cur.qx; a.t.qTy = cur.qy;
        if constexpr (PH) {      // update_pt_thermal_arrays!(pt_thermal, phase, rheology, args, _dt) -- DiffusionPT_solver.jl:233-234
            // on unobserved iterations update_T! writes the coefficients of the next iteration itself (same values: they depend on the cell's own new T
            // only); observed iterations leave pt_thermal as the reference does, and the stand-alone kernel runs before the following iteration
            if (!pt_fresh) JRX_TRY(jrx_enqueue_pt_thermal_arrays(h, s, t->thetar_dtau, t->dtau_rho, cur.T, nx, ny, 1, 2, 1.0 / p->dt, ph));
            pt_fresh = !observed;
        }
        if (fusable && !observed) {
            fused_launch(a, oth);
            h->stat_thermal_fused++;
            JRX_LAUNCH_CHECK(h);
            const TSet2 tmp = cur; cur = oth; oth = tmp;
        } else {
            tc.T = cur.T; tc.qTx = cur.qx; tc.qTy = cur.qy;
            JRX_TRY(enqueue_titer(h, &tc, p, ph, true, PH && pt_fresh));
        }
        iter++;
        if (iter % p->nout == 0) {
            hipLaunchKernelGGL((k_updateT2d<true, false, PHT>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, ph);
            JRX_LAUNCH_CHECK(h);
            RedArr Z = {nullptr, {0, 0, 0}, 0}, A3 = {t->ResT, {nx, ny, 1}, 0};
            int nb = (int)((n + 2047) / 2048);
            nb = nb < 1 ? 1 : (nb > kMaxRedBlocks ? kMaxRedBlocks : nb);
            hipLaunchKernelGGL(k_sumsq_partial, dim3(nb), dim3(256), 0, s, Z, Z, Z, A3, h->d_partials);
            JRX_LAUNCH_CHECK(h);
            hipLaunchKernelGGL(k_sumsq_final, dim3(1), dim3(256), 0, s, h->d_partials, nb, h->d_sums);
            JRX_LAUNCH_CHECK(h);
            JRX_HIP(h, hipMemcpyAsync(h->h_sums, h->d_sums, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
            JRX_HIP(h, hipStreamSynchronize(s));
            err = sqrt(h->h_sums[3]) * sq;          // norm(ResT) * _sq_len_RT : local norm, no reduction across ranks (:131)
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
    destroy_graphs();
    if (cur.T != user.T) {      // leave the results in the caller's arrays
        JRX_HIP(h, hipMemcpyAsync(user.T, cur.T, (size_t)nT * sizeof(double), hipMemcpyDeviceToDevice, s));
        JRX_HIP(h, hipMemcpyAsync(user.qx, cur.qx, (size_t)(nx + 1) * ny * sizeof(double), hipMemcpyDeviceToDevice, s));
        JRX_HIP(h, hipMemcpyAsync(user.qy, cur.qy, (size_t)nx * (ny + 1) * sizeof(double), hipMemcpyDeviceToDevice, s));
    }
    hipLaunchKernelGGL(k_sub, dim3(256), dim3(256), 0, s, t->dT, (const double *)t->T, (const double *)t->Told, nT);   // update_ΔT!
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(s));
    if (nnorms) *nnorms = cnt < cap ? cnt : cap;
    return JRX_OK;
}
}   // namespace

extern "C" {

jrx_status jrx_heatdiffusion_PT2d(jrx_handle *h, const jrx_thermal2d_fields *t, const jrx_thermal2d_params *p,
                                  int64_t *iter_count, double *norm_ResT, int64_t cap, int64_t *nnorms)
{
    JRX_TRY(checkT(h, t, p));
    return heat2d(h, t, p, NoPh{}, iter_count, norm_ResT, cap, nnorms);
}

jrx_status jrx_heatdiffusion_PT2d_phases(jrx_handle *h, const jrx_thermal2d_fields *t, const jrx_thermal2d_params *p, const jrx_thermal_phases *ph,
                                         const jrx_thermal_phase_fields *pf, int64_t *iter_count, double *norm_ResT, int64_t cap, int64_t *nnorms)
{
    if (!h) return JRX_ERR_ARG;
    if (!p) return jrx_fail(h, JRX_ERR_ARG, "null thermal params");
    jrx_thermal2d_params q = *p;
    q.rheology_form = 1;                        // K / rhoCp arrays are not read
    JRX_TRY(checkT(h, t, &q));
    JRX_TRY(tph_check(h, ph, pf, false));
    q.rheology_form = 2;
    TPh x;
    x.m = *ph; x.f = *pf;
    switch (h->thermal_np_const ? ph->nphase : 0) {       // option "thermal_np_const" (default): the instantiations with the phase count as a constant
#define TPN(N_) case N_: { TPhN<N_> y; y.m = *ph; y.f = *pf; return heat2d(h, t, &q, y, iter_count, norm_ResT, cap, nnorms); }
    TPN(1) TPN(2) TPN(3) TPN(4)
#undef TPN
    default: break;
    }
    return heat2d(h, t, &q, x, iter_count, norm_ResT, cap, nnorms);
}

jrx_status jrx_adiabatic_heating(jrx_handle *h, double *adiabatic, const double *P, const double *P0, int64_t ncells, double dt, const jrx_thermal_phases *ph,
                                 const double *phase_c)
{
    if (!h) return JRX_ERR_ARG;
    if (!adiabatic || !P || !P0 || !ph || ncells < 1) return jrx_fail(h, JRX_ERR_ARG, "adiabatic_heating!: bad argument");
    if (ph->nphase < 1 || ph->nphase > JRX_MAXPHASE) return jrx_fail(h, JRX_ERR_ARG, "adiabatic_heating!: nphase out of range");
    JRX_TRY(jrx_check_device(h));
    TPh x;
    memset(&x, 0, sizeof(x));
    x.m = *ph; x.f.phase_c = phase_c;
    const i64 nb = (ncells + 255) / 256;
    hipLaunchKernelGGL(k_adiabatic, dim3((unsigned)(nb > 8192 ? 8192 : nb)), dim3(256), 0, h->stream, adiabatic, P, P0, (i64)ncells, 1.0 / dt, x);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_update_pt_thermal_arrays(jrx_handle *h, double *thetar_dtau, double *dtau_rho, const double *T, const int64_t n[3], int32_t ndim, double dt,
                                        const jrx_thermal_phases *ph, const jrx_thermal_phase_fields *pf)
{
    if (!h) return JRX_ERR_ARG;
    if (!thetar_dtau || !dtau_rho || !T || !n || (ndim != 2 && ndim != 3)) return jrx_fail(h, JRX_ERR_ARG, "update_pt_thermal_arrays!: bad argument");
    JRX_TRY(jrx_check_device(h));
    if (!ph || !pf || !pf->P || !pf->phase_c) return jrx_fail(h, JRX_ERR_ARG, "update_pt_thermal_arrays!: phases, args.P and the centre ratios are required");
    TPh x;
    x.m = *ph; x.f = *pf;
    JRX_TRY(jrx_enqueue_pt_thermal_arrays(h, h->stream, thetar_dtau, dtau_rho, T, (int)n[0], (int)n[1], ndim == 3 ? (int)n[2] : 1, (int)ndim, 1.0 / dt, x));
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

}   // extern "C"
