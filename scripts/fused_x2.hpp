// fused_x2.hpp -- TWO PT iterations per launch (temporal blocking) for the viscous-limit form of the fused 3D Stokes kernel.
// Development copy (scripts/kbench_x2.hip validates it bit for bit against two launches of k_fused3d + the boundary-layer launch); the library includes the same file.
//
// One launch takes the state (P, τ)_m, V_{m-1} of the source set to (P, τ)_{m+2}, V_{m+1} in the destination set: velocity update m (V1), stress update m+1 (S1),
// velocity update m+1 (V2), stress update m+2 (S2) -- the reference's iterations m and m+1 (src/stokes/Stokes3D.jl:78-121) with nothing observed in between, so that
// V1 and S1 never leave the chip: 15 array reads + 10 writes per TWO iterations.
//
// Tile: TX = 64 lanes (a wave per row) x TY rows x KZ planes.  Every stage loses one cell on the low side (stress at a cell needs the new velocities of its lower
// neighbours) or on the high side (velocity at a cell needs the new stresses of its upper neighbours), so of the TX x TY lanes the tile owns (TX - 4) x (TY - 3) cell
// columns: lanes tx in [2, TX-3], ty in [2, TY-2] (lane TX-1 only feeds its left neighbour, as in k_fused3d).  In z the second iteration runs one plane behind the
// first: step t computes V1(t), S1(t), V2(t-1), S2(t-1); a chunk of KZ planes walks KZ + 3 steps (two prologue planes below, one above).
// Values travel between lanes through LDS (one plane set of V1, two of S1 -- the velocity update needs the node planes t-1 and t of τxz, τyz --, one of V2) and
// between planes in registers.  The boundary entries of V follow the flow_bcs! rules exactly as in k_fused3d / stress3d_node<.., GH> (free slip: copy, no slip:
// negate / zero, none: the prescribed value in memory), between the two iterations too; the stress nodes on the high faces i = nx, j = ny, k = nz are updated by the
// threads of the last cell column / row / plane (as k_fused3d<..., HIF>), for S1 as well, because V2 of the last cells reads them.
// Arithmetic: the same expressions in the same order as k_fused3d<..., VISC>, i.e. as the reference's kernels in the limit dt = Inf.
#pragma once
#include "stokes3d_kernels.hpp"

namespace {

struct X2Stress { double P, txx, tyy, tzz, txy, txz, tyz; };

// compute_P! + compute_τ! at cell (i, j, k) in the viscous limit (PressureKernels.jl:186-195, StressKernels.jl:149-230, VelocityKernels.jl:59-104):
// va..vcy are the new velocities around the cell in plane k + 1 of V, a_p..cy_p the ones in plane k, e.. the viscosities of the clamped shear averages
__device__ __forceinline__ X2Stress x2_stress(const double va, const double vax, const double vay, const double vb, const double vby, const double vbx, const double vc,
                                              const double a_p, const double b_p, const double c_p, const double cx_p, const double cy_p, const double e, const double ex,
                                              const double ey, const double exy_, const double e_p, const double ex_p, const double ey_p, const double P_k,
                                              const double txx, const double tyy, const double tzz, const double txy, const double txz, const double tyz,
                                              const double _dx, const double _dy, const double _dz, const double th, const double rr)
{
    X2Stress o;
    const double dxi = (-va + vax) * _dx;
    const double dyi = (-vb + vby) * _dy;
    const double dzi = (-c_p + vc) * _dz;
    const double divV = dxi + dyi + dzi;
    const double rhs = -divV + (0.0 * (1.0 / INFINITY));
    const double psi = 1.0 / (1.0 / e + 0.0) * rr / th;
    o.P = (fma(0.0, 0.0, rhs) * psi + P_k) / (1.0 + 0.0 * psi);
    const double d3 = divV * (1.0 / 3.0);
    const double exx = dxi - d3, eyy = dyi - d3, ezz = dzi - d3;
    const double dtr = dev_dtau_r(th, e, 0.0);
    o.txx = txx + dev_stress_inc(txx, 0.0, e, exx, 0.0, dtr);
    o.tyy = tyy + dev_stress_inc(tyy, 0.0, e, eyy, 0.0, dtr);
    o.tzz = tzz + dev_stress_inc(tzz, 0.0, e, ezz, 0.0, dtr);
    {
        const double s_ = 0.5 * (_dy * (va - vay) + _dx * (vb - vbx));
        const double ee = 0.25 * (exy_ + ey + ex + e);
        o.txy = txy + dev_stress_inc(txy, 0.0, ee, s_, 0.0, dev_dtau_r(th, ee, 0.0));
    }
    {
        const double s_ = 0.5 * (_dz * (va - a_p) + _dx * (c_p - cx_p));
        const double ee = 0.25 * (ex_p + e_p + ex + e);
        o.txz = txz + dev_stress_inc(txz, 0.0, ee, s_, 0.0, dev_dtau_r(th, ee, 0.0));
    }
    {
        const double s_ = 0.5 * (_dz * (vb - b_p) + _dy * (c_p - cy_p));
        const double ee = 0.25 * (ey_p + e_p + ey + e);
        o.tyz = tyz + dev_stress_inc(tyz, 0.0, ee, s_, 0.0, dev_dtau_r(th, ee, 0.0));
    }
    return o;
}
// one shear node: τ + Δτ(ε, η) with the strain rate s_ and the averaged viscosity ee
__device__ __forceinline__ double x2_shear(const double t0, const double s_, const double ee, const double th)
{
    return t0 + dev_stress_inc(t0, 0.0, ee, s_, 0.0, dev_dtau_r(th, ee, 0.0));
}

template <int TX, int TY, int KZ, int XG>
__global__ __launch_bounds__(TX *TY, 1) void k_fused3d_x2(const SweepArgs a, const FusedBC bc, int ntx, int nty)
{
    static_assert(TX == 64, "a row per wave");
    __shared__ double sV[2][3][TY][TX];      // V1 of the planes t-1 and t (slot = plane & 1): the previous plane is re-read instead of carried in registers
    __shared__ double sS[2][7][TY][TX];      // S1: P, τxx, τyy, τzz, τxy, τxz, τyz of the planes t-1 and t (slot = plane & 1); the slot of plane t first carries the
                                             // y-neighbour operands of the velocity update of plane t (P, ητ, τyy, fy, τxy, τyz, η)
    __shared__ double sW[2][3][TY][TX];      // V2 of the planes k-1 and k (k = t-1)
    __shared__ double sHx[2][TY], sHy[2][TX];   // S1's τxy on the high faces: (nx, j, ·) per row, (i, ny, ·) per column (slot = plane & 1)
    const Lay3 &L = a.L;
    const int nx = L.nx, ny = L.ny, nz = L.nz;
    const jrx_stokes3d_fields &f = a.f;
    const double *et = a.etatau;
    const int tx = (int)(threadIdx.x % TX), ty = (int)(threadIdx.x / TX);
    int tile = blockIdx.x;
    if (XG > 0) {
        const int full = ((nty * (int)(gridDim.x / (unsigned)(ntx * nty))) / (8 * XG)) * (8 * XG) * ntx;
        if (tile < full) {
            const int q = tile & 7, r = tile >> 3, r2 = r / ntx;
            tile = ((r2 / XG) * (8 * XG) + q * XG + r2 % XG) * ntx + r % ntx;
        }
    }
    const int tr = tile / ntx, tix = tile % ntx, tiy = tr % nty, tiz = tr / nty;
    const int i = tix * (TX - 4) - 2 + tx;
    const int j = tiy * (TY - 3) - 2 + ty;
    const int kb = tiz * KZ;
    const int kend = min(kb + KZ, nz);
    const bool bvalid = i >= 0 && j >= 0 && i < nx && j < ny;
    const bool a1 = bvalid && tx >= 1 && ty >= 1 && tx <= TX - 2;                  // S1
    const bool b2 = a1 && tx <= TX - 3 && ty <= TY - 2;                              // V2
    const bool own = b2 && tx >= 2 && ty >= 2;                                       // S2 and the stores
    const bool hx = i < nx - 1, hy = j < ny - 1;
    const bool xl = i == nx - 1, yl = j == ny - 1;
    const double _dx = a._dx, _dy = a._dy, _dz = a._dz, th = a.theta_dtau, rr = a.r, edt = a.eta_dtau;

    const u32 sc = (u32)L.cp * 8u, svx = (u32)L.vxp * 8u, svy = (u32)L.vyp * 8u, svz = (u32)L.vzp * 8u;
    const u32 sxy = (u32)L.xyp * 8u, sxz = (u32)L.xzp * 8u, syz = (u32)L.yzp * 8u;
    const u32 rc = (u32)nx * 8u, rxy = (u32)L.xy1 * 8u, ryz = (u32)L.yz1 * 8u;
    const u32 rvx = (u32)L.vx1 * 8u, rvy = (u32)L.vy1 * 8u, rvz = (u32)L.vz1 * 8u;
    const int ic = bvalid ? i : 0, jc = bvalid ? j : 0;
    const int t0 = kb >= 2 ? kb - 2 : 0;
    // byte offsets of plane t (cells) / t + 1 (V, node planes of τxz, τyz), advanced once per step
    u32 oc = 8u * (u32)(ic + nx * jc) + sc * (u32)t0;
    u32 oxy = 8u * (u32)(ic + L.xy1 * jc) + sxy * (u32)t0;
    u32 oxz = 8u * (u32)(ic + L.xz1 * jc) + sxz * (u32)(t0 + 1);
    u32 oyz = 8u * (u32)(ic + L.yz1 * jc) + syz * (u32)(t0 + 1);
    u32 ovx = 8u * (u32)((ic + 1) + L.vx1 * (jc + 1)) + svx * (u32)(t0 + 1);
    u32 ovy = 8u * (u32)((ic + 1) + L.vy1 * (jc + 1)) + svy * (u32)(t0 + 1);
    u32 ovz = 8u * (u32)((ic + 1) + L.vz1 * (jc + 1)) + svz * (u32)(t0 + 1);

    // ---- carries of the first iteration (as k_fused3d)
    double Pc = 0, ec = 0, tzz_c = 0, fz_c = 0, s10 = 0, r10 = 0, s01p = 0, r01p = 0;
    if (bvalid) {
        Pc = LDB(f.P, oc); ec = LDB(et, oc); tzz_c = LDB(f.tzz, oc); fz_c = LDB(f.fz, oc);
        s10 = LDB(f.txz, oxz + 8u - sxz); r10 = LDB(f.tyz, oyz + ryz - syz);
        s01p = LDB(f.txz, oxz - sxz); r01p = LDB(f.tyz, oyz - syz);
    }
    double e_p = 0, ex_p = 0, ey_p = 0;                         // η of plane t - 1 and its clamped x / y neighbours
    // ---- what the second iteration keeps of plane t - 1: the velocity update's own operands (the i + 1 ones come back by lane shuffle) ...
    double fxc_b = 0, fyc_b = 0, fyy_b = 0, fzc_b = 0, ec_b = 0, eyb_b = 0, exy_b = 0;
    double e_q = 0, ex_q = 0, ey_q = 0;                         // ... and η of plane t - 2
    double hxz1 = 0, hyz1 = 0;                                  // S1 on the high faces, node plane t - 1: τxz (nx, j, ·) / τyz (i, ny, ·) of this thread's column / row

    const int tlast = kend;                                     // the step that finishes plane kend - 1 of the second iteration
    for (int t = t0; t <= tlast; ++t) {
        const bool top = t == nz;                               // the step above the last cell plane: S1 is the node plane k = nz only
        const bool have1 = t > t0 || t0 == 0;                   // V1 (t - 1) is in the carries (or t = 0: the boundary rule stands in)
        const int slot = t & 1, pslot = slot ^ 1;
        const bool hz = t < nz - 1;
        double vxn = 0, vyn = 0, vzn = 0, txx_c = 0, tyy_c = 0, txy_own = 0;
        const double P_k = Pc, tzz_k = tzz_c, s01k = s01p, r01k = r01p, s10k = s10, r10k = r10;
        double e = 0, ex = 0, ey = 0, exy_ = 0;
        double fx_c = 0, fx_x = 0, fy_c = 0, fy_y = 0, fz_z = 0, ecx = 0, eyb = 0, ez = 0;
        const double ec_t = ec, fz_t = fz_c;
        // ================================================================ V1 (t): compute_V! of iteration m on plane t
        if (!top) {
            double q01 = 0, s01 = 0, r11 = 0, r01 = 0, Pz = 0, tzz_z = 0, Py = 0, tyy_y = 0, vx = 0, vy = 0, vz = 0;
            const bool yrow = ty < TY - 1 && hy;
            if (bvalid) {
                const u32 dz1 = hz ? sc : 0u;
                e = LDB(f.eta, oc);
                tyy_c = LDB(f.tyy, oc); fy_c = LDB(f.fy, oc); txy_own = LDB(f.txy, oxy); r01 = LDB(f.tyz, oyz);
                if (!yrow) {
                    q01 = LDB(f.txy, oxy + rxy); r11 = LDB(f.tyz, oyz + ryz);
                    if (hy) { Py = LDB(f.P, oc + rc); eyb = LDB(et, oc + rc); tyy_y = LDB(f.tyy, oc + rc); fy_y = LDB(f.fy, oc + rc); }
                }
                s01 = LDB(f.txz, oxz);
                Pz = LDB(f.P, oc + dz1); ez = LDB(et, oc + dz1); tzz_z = LDB(f.tzz, oc + dz1); fz_z = LDB(f.fz, oc + dz1);
                txx_c = LDB(f.txx, oc); fx_c = LDB(f.fx, oc);
                vx = LDB(f.Vx, ovx); vy = LDB(f.Vy, ovy); vz = LDB(f.Vz, ovz);
                sS[slot][0][ty][tx] = Pc; sS[slot][1][ty][tx] = ec; sS[slot][2][ty][tx] = tyy_c; sS[slot][3][ty][tx] = fy_c; sS[slot][4][ty][tx] = txy_own;
                sS[slot][5][ty][tx] = r01; sS[slot][6][ty][tx] = e;
            }
            __syncthreads();
            if (bvalid) {
                if (yrow) {
                    Py = sS[slot][0][ty + 1][tx]; eyb = sS[slot][1][ty + 1][tx]; tyy_y = sS[slot][2][ty + 1][tx]; fy_y = sS[slot][3][ty + 1][tx];
                    q01 = sS[slot][4][ty + 1][tx]; r11 = sS[slot][5][ty + 1][tx];
                }
                if (ty > 0 && j > 0) ey = sS[slot][6][ty - 1][tx]; else ey = e;
                const double e_l = __shfl_up(e, 1, TX), ey_l = __shfl_up(ey, 1, TX);
                ex = i > 0 ? e_l : e; exy_ = i > 0 ? ey_l : ey;
                double q11 = __shfl_down(q01, 1, TX), s11 = __shfl_down(s01, 1, TX);
                const double q10 = __shfl_down(txy_own, 1, TX);
                const double Px = __shfl_down(Pc, 1, TX), txx_x = __shfl_down(txx_c, 1, TX);
                ecx = __shfl_down(ec, 1, TX); fx_x = __shfl_down(fx_c, 1, TX);
                if (!hx) { q11 = LDB(f.txy, oxy + 8u + rxy); s11 = LDB(f.txz, oxz + 8u); }
                if (hx) {
                    const double R = (-txx_c + txx_x) * _dx + _dy * (q11 - q10) + _dz * (s11 - s10) - (-Pc + Px) * _dx - 0.5 * (fx_c + fx_x);
                    vxn = vx + R * edt / (0.5 * (ec + ecx));
                } else vxn = bc.nsR ? 0.0 : vx;
                if (hy) {
                    const double R = _dx * (q11 - q01) + _dy * (tyy_y - tyy_c) + _dz * (r11 - r10) - (-Pc + Py) * _dy - 0.5 * (fy_c + fy_y);
                    vyn = vy + R * edt / (0.5 * (ec + eyb));
                } else vyn = bc.nsBk ? 0.0 : vy;
                if (hz) {
                    const double R = _dx * (s11 - s01) + _dy * (r11 - r01) + (-tzz_c + tzz_z) * _dz - (-Pc + Pz) * _dz - 0.5 * (fz_c + fz_z);
                    vzn = vz + R * edt / (0.5 * (ec + ez));
                } else vzn = bc.nsK1 ? 0.0 : vz;
                Pc = Pz; ec = ez; tzz_c = tzz_z; fz_c = fz_z; s10 = s11; r10 = r11; s01p = s01; r01p = r01;
                sV[slot][0][ty][tx] = vxn; sV[slot][1][ty][tx] = vyn; sV[slot][2][ty][tx] = vzn;
            }
        }
        __syncthreads();
        // ================================================================ S1 (t): compute_P! / compute_τ! of iteration m + 1 on plane t (top: the node plane k = nz)
        X2Stress S = {0, 0, 0, 0, 0, 0, 0};
        double hxy_x = 0, hxy_y = 0, hxz_t = 0, hyz_t = 0;         // S1 on the high faces of plane t
        if (a1) {
            // the new velocities of plane t - 1 around the cell (plane t of V): re-read from the other slot, or the flow_bcs! rule of the low z face at t = 0
            double a_p = 0, b_p = 0, c_p = 0, cx_p = 0, cy_p = 0, vax_p = 0, vby_p = 0;
            const u32 gvx = ovx - 8u, gvy = ovy - rvy, gvz = ovz;
            if (t > 0 && have1) {
                a_p = i > 0 ? sV[pslot][0][ty][tx - 1] : (bc.nsL ? 0.0 : LDB(f.Vx, gvx - svx));
                b_p = j > 0 ? sV[pslot][1][ty - 1][tx] : (bc.nsF ? 0.0 : LDB(f.Vy, gvy - svy));
                c_p = sV[pslot][2][ty][tx];
                cx_p = i > 0 ? sV[pslot][2][ty][tx - 1] : (bc.fsL ? c_p : (bc.nsL ? -c_p : LDB(f.Vz, gvz - svz - 8u)));
                cy_p = j > 0 ? sV[pslot][2][ty - 1][tx] : (bc.fsF ? c_p : (bc.nsF ? -c_p : LDB(f.Vz, gvz - svz - rvz)));
                if (xl) vax_p = sV[pslot][0][ty][tx];
                if (yl) vby_p = sV[pslot][1][ty][tx];
            }
            if (!top) {
                const double vax = vxn, vby = vyn, vc = vzn;
                const double va = i > 0 ? sV[slot][0][ty][tx - 1] : (bc.nsL ? 0.0 : LDB(f.Vx, gvx));
                double vay, vbx;
                if (j > 0) vay = i > 0 ? sV[slot][0][ty - 1][tx - 1] : (bc.nsL ? 0.0 : LDB(f.Vx, gvx - rvx));
                else vay = bc.fsF ? va : (bc.nsF ? -va : LDB(f.Vx, gvx - rvx));
                const double vb = j > 0 ? sV[slot][1][ty - 1][tx] : (bc.nsF ? 0.0 : LDB(f.Vy, gvy));
                if (i > 0) vbx = j > 0 ? sV[slot][1][ty - 1][tx - 1] : (bc.nsF ? 0.0 : LDB(f.Vy, gvy - 8u));
                else vbx = bc.fsL ? vb : (bc.nsL ? -vb : LDB(f.Vy, gvy - 8u));
                if (t == 0) {
                    a_p = bc.fsK0 ? va : (bc.nsK0 ? -va : LDB(f.Vx, gvx - svx));
                    b_p = bc.fsK0 ? vb : (bc.nsK0 ? -vb : LDB(f.Vy, gvy - svy));
                    c_p = bc.nsK0 ? 0.0 : LDB(f.Vz, gvz - svz);
                    cx_p = i > 0 ? (bc.nsK0 ? 0.0 : LDB(f.Vz, gvz - svz - 8u)) : (bc.fsL ? c_p : (bc.nsL ? -c_p : LDB(f.Vz, gvz - svz - 8u)));
                    cy_p = j > 0 ? (bc.nsK0 ? 0.0 : LDB(f.Vz, gvz - svz - rvz)) : (bc.fsF ? c_p : (bc.nsF ? -c_p : LDB(f.Vz, gvz - svz - rvz)));
                    e_p = e; ex_p = ex; ey_p = ey;
                    if (xl) vax_p = bc.nsR ? 0.0 : (bc.fsK0 ? vax : (bc.nsK0 ? -vax : LDB(f.Vx, ovx - svx)));
                    if (yl) vby_p = bc.nsBk ? 0.0 : (bc.fsK0 ? vby : (bc.nsK0 ? -vby : LDB(f.Vy, ovy - svy)));
                }
                if (have1) {
                    S = x2_stress(va, vax, vay, vb, vby, vbx, vc, a_p, b_p, c_p, cx_p, cy_p, e, ex, ey, exy_, e_p, ex_p, ey_p, P_k, txx_c, tyy_c, tzz_k, txy_own, s01k, r01k,
                                  _dx, _dy, _dz, th, rr);
                    if (xl) {
                        const double vxl = j > 0 ? sV[slot][0][ty - 1][tx] : (bc.nsR ? 0.0 : (bc.fsF ? vax : (bc.nsF ? -vax : LDB(f.Vx, ovx - rvx))));
                        const double vyg = (j == 0 && bc.nsF) ? 0.0 : (bc.fsR ? vb : (bc.nsR ? -vb : LDB(f.Vy, ovy - rvy + 8u)));
                        hxy_x = x2_shear(LDB(f.txy, oxy + 8u), 0.5 * (_dy * (vax - vxl) + _dx * (vyg - vb)), 0.25 * (ey + ey + e + e), th);
                        const double vzg = (t == 0 && bc.nsK0) ? 0.0 : (bc.fsR ? c_p : (bc.nsR ? -c_p : LDB(f.Vz, ovz - svz + 8u)));
                        hxz_t = x2_shear(s10k, 0.5 * (_dz * (vax - vax_p) + _dx * (vzg - c_p)), 0.25 * (e_p + e_p + e + e), th);
                    }
                    if (yl) {
                        const double vxg = (i == 0 && bc.nsL) ? 0.0 : (bc.fsBk ? va : (bc.nsBk ? -va : LDB(f.Vx, ovx - 8u + rvx)));
                        const double vyl = i > 0 ? sV[slot][1][ty][tx - 1] : (bc.nsBk ? 0.0 : (bc.fsL ? vby : (bc.nsL ? -vby : LDB(f.Vy, ovy - 8u))));
                        hxy_y = x2_shear(LDB(f.txy, oxy + rxy), 0.5 * (_dy * (vxg - va) + _dx * (vby - vyl)), 0.25 * (ex + e + ex + e), th);
                        const double vzg = (t == 0 && bc.nsK0) ? 0.0 : (bc.fsBk ? c_p : (bc.nsBk ? -c_p : LDB(f.Vz, ovz - svz + rvz)));
                        hyz_t = x2_shear(r10k, 0.5 * (_dz * (vby - vby_p) + _dy * (vzg - c_p)), 0.25 * (e_p + e_p + e + e), th);
                    }
                }
            } else if (have1) {
                // node plane k = nz of S1: only τxz, τyz exist there; a_p.. are plane nz of V1 (the boundary plane of Vz included), e_p.. plane nz - 1 of η, the
                // offsets stand at plane nz + 1 of V
                const double vxg = (i == 0 && bc.nsL) ? 0.0 : (bc.fsK1 ? a_p : (bc.nsK1 ? -a_p : LDB(f.Vx, ovx - 8u)));
                S.txz = x2_shear(s01k, 0.5 * (_dz * (vxg - a_p) + _dx * (c_p - cx_p)), 0.25 * (ex_p + e_p + ex_p + e_p), th);
                const double vyg = (j == 0 && bc.nsF) ? 0.0 : (bc.fsK1 ? b_p : (bc.nsK1 ? -b_p : LDB(f.Vy, ovy - rvy)));
                S.tyz = x2_shear(r01k, 0.5 * (_dz * (vyg - b_p) + _dy * (c_p - cy_p)), 0.25 * (ey_p + e_p + ey_p + e_p), th);
                if (xl) {
                    const double vxh = bc.nsR ? 0.0 : (bc.fsK1 ? vax_p : (bc.nsK1 ? -vax_p : LDB(f.Vx, ovx)));
                    const double vzg = bc.nsK1 ? 0.0 : (bc.fsR ? c_p : (bc.nsR ? -c_p : LDB(f.Vz, ovz - svz + 8u)));
                    hxz_t = x2_shear(s10k, 0.5 * (_dz * (vxh - vax_p) + _dx * (vzg - c_p)), 0.25 * (e_p + e_p + e_p + e_p), th);
                }
                if (yl) {
                    const double vyh = bc.nsBk ? 0.0 : (bc.fsK1 ? vby_p : (bc.nsK1 ? -vby_p : LDB(f.Vy, ovy)));
                    const double vzg = bc.nsK1 ? 0.0 : (bc.fsBk ? c_p : (bc.nsBk ? -c_p : LDB(f.Vz, ovz - svz + rvz)));
                    hyz_t = x2_shear(r10k, 0.5 * (_dz * (vyh - vby_p) + _dy * (vzg - c_p)), 0.25 * (e_p + e_p + e_p + e_p), th);
                }
            }
        }
        // publish S1 (t) for the second iteration's velocity update (the y-neighbour operands in this slot have been consumed before the last barrier)
        {
            sS[slot][0][ty][tx] = S.P; sS[slot][1][ty][tx] = S.txx; sS[slot][2][ty][tx] = S.tyy; sS[slot][3][ty][tx] = S.tzz; sS[slot][4][ty][tx] = S.txy;
            sS[slot][5][ty][tx] = S.txz; sS[slot][6][ty][tx] = S.tyz;
            if (xl) sHx[slot][ty] = hxy_x;
            if (yl) sHy[slot][tx] = hxy_y;
        }
        __syncthreads();
        // ================================================================ V2 (t - 1): compute_V! of iteration m + 1 on plane k = t - 1, from S1 (t - 1) and S1 (t)
        const int k = t - 1;
        const bool s1_both = k >= 0 && (k > t0 || t0 == 0);      // S1 (k) exists in the other slot (k = t0 has no S1 unless the chunk starts at the bottom)
        double wx = 0, wy = 0, wz = 0;
        double P1 = 0, txx1 = 0, tyy1 = 0, tzz1 = 0, txy1 = 0, txz1 = 0, tyz1 = 0;      // this cell's S1 (k)
        const double fxx_b = __shfl_down(fxc_b, 1, TX), ecx_b = __shfl_down(ec_b, 1, TX);      // fx, ητ of the cell to the right, plane k
        if (b2 && s1_both) {
            P1 = sS[pslot][0][ty][tx]; txx1 = sS[pslot][1][ty][tx]; tyy1 = sS[pslot][2][ty][tx]; tzz1 = sS[pslot][3][ty][tx]; txy1 = sS[pslot][4][ty][tx];
            txz1 = sS[pslot][5][ty][tx]; tyz1 = sS[pslot][6][ty][tx];
            const bool hz2 = k < nz - 1;
            // upper neighbours in x (lane tx + 1), y (row ty + 1) and z (slot of plane t); the high-face nodes come from the boundary column / row themselves
            const double Px = sS[pslot][0][ty][tx + 1], txx_x = sS[pslot][1][ty][tx + 1];
            const double Py = sS[pslot][0][ty + 1][tx], tyy_y = sS[pslot][2][ty + 1][tx];
            const double Pz = S.P, tzz_z = S.tzz;
            const double q10 = sS[pslot][4][ty][tx + 1];                                                      // τxy (i+1, j, k)
            const double q01 = yl ? sHy[pslot][tx] : sS[pslot][4][ty + 1][tx];                                // τxy (i, j+1, k)
            double q11;                                                                                        // τxy (i+1, j+1, k)
            if (xl) q11 = (ty + 1 < TY && !yl) ? sHx[pslot][ty + 1] : 0.0;
            else q11 = yl ? sHy[pslot][tx + 1] : sS[pslot][4][ty + 1][tx + 1];
            const double s10b = xl ? hxz1 : sS[pslot][5][ty][tx + 1];                                          // τxz (i+1, j, k)
            const double s11b = xl ? hxz_t : sS[slot][5][ty][tx + 1];                                          // τxz (i+1, j, k+1)
            const double s01b = S.txz;                                                                         // τxz (i, j, k+1)
            const double r10b = yl ? hyz1 : sS[pslot][6][ty + 1][tx];                                          // τyz (i, j+1, k)
            const double r11b = yl ? hyz_t : sS[slot][6][ty + 1][tx];                                          // τyz (i, j+1, k+1)
            const double r01b = S.tyz;                                                                         // τyz (i, j, k+1)
            const double v1x = sV[pslot][0][ty][tx], v1y = sV[pslot][1][ty][tx], v1z = sV[pslot][2][ty][tx];      // this cell's V1 (k)
            if (hx) {
                const double R = (-txx1 + txx_x) * _dx + _dy * (q11 - q10) + _dz * (s11b - s10b) - (-P1 + Px) * _dx - 0.5 * (fxc_b + fxx_b);
                wx = v1x + R * edt / (0.5 * (ec_b + ecx_b));
            } else wx = v1x;
            if (hy) {
                const double R = _dx * (q11 - q01) + _dy * (tyy_y - tyy1) + _dz * (r11b - r10b) - (-P1 + Py) * _dy - 0.5 * (fyc_b + fyy_b);
                wy = v1y + R * edt / (0.5 * (ec_b + eyb_b));
            } else wy = v1y;
            if (hz2) {
                const double R = _dx * (s11b - s01b) + _dy * (r11b - r01b) + (-tzz1 + tzz_z) * _dz - (-P1 + Pz) * _dz - 0.5 * (fzc_b + fz_t);
                wz = v1z + R * edt / (0.5 * (ec_b + ec_t));
            } else wz = v1z;
        }
        const int ks = k & 1, kp = ks ^ 1;       // slots of V2 (k) and V2 (k - 1)
        sW[ks][0][ty][tx] = wx; sW[ks][1][ty][tx] = wy; sW[ks][2][ty][tx] = wz;
        __syncthreads();
        // ================================================================ S2 (t - 1): compute_P! / compute_τ! of iteration m + 2 on plane k; the results of the launch
        if (own && s1_both) {
            const bool live = k >= kb;
            // byte offsets of plane k: the running offsets stand one plane further
            const u32 oc2 = oc - sc, oxy2 = oxy - sxy, oxz2 = oxz - sxz, oyz2 = oyz - syz, ovx2 = ovx - svx, ovy2 = ovy - svy, ovz2 = ovz - svz;
            const u32 gvx = ovx2 - 8u, gvy = ovy2 - rvy, gvz = ovz2;
            const double wax = wx, wby = wy, wc = wz;
            const double wa = i > 0 ? sW[ks][0][ty][tx - 1] : (bc.nsL ? 0.0 : LDB(f.Vx, gvx));
            double way, wbx;
            if (j > 0) way = i > 0 ? sW[ks][0][ty - 1][tx - 1] : (bc.nsL ? 0.0 : LDB(f.Vx, gvx - rvx));
            else way = bc.fsF ? wa : (bc.nsF ? -wa : LDB(f.Vx, gvx - rvx));
            const double wb = j > 0 ? sW[ks][1][ty - 1][tx] : (bc.nsF ? 0.0 : LDB(f.Vy, gvy));
            if (i > 0) wbx = j > 0 ? sW[ks][1][ty - 1][tx - 1] : (bc.nsF ? 0.0 : LDB(f.Vy, gvy - 8u));
            else wbx = bc.fsL ? wb : (bc.nsL ? -wb : LDB(f.Vy, gvy - 8u));
            const double wcx = i > 0 ? sW[ks][2][ty][tx - 1] : (bc.fsL ? wc : (bc.nsL ? -wc : LDB(f.Vz, gvz - 8u)));
            const double wcy = j > 0 ? sW[ks][2][ty - 1][tx] : (bc.fsF ? wc : (bc.nsF ? -wc : LDB(f.Vz, gvz - rvz)));
            if (live) {
                // V2 of plane k - 1 around the cell (plane k of V): the other slot, or the flow_bcs! rule of the low z face
                double a_q, b_q, c_q, cx_q, cy_q, wax_q = 0, wby_q = 0;
                if (k > 0) {
                    a_q = i > 0 ? sW[kp][0][ty][tx - 1] : (bc.nsL ? 0.0 : LDB(f.Vx, gvx - svx));
                    b_q = j > 0 ? sW[kp][1][ty - 1][tx] : (bc.nsF ? 0.0 : LDB(f.Vy, gvy - svy));
                    c_q = sW[kp][2][ty][tx];
                    cx_q = i > 0 ? sW[kp][2][ty][tx - 1] : (bc.fsL ? c_q : (bc.nsL ? -c_q : LDB(f.Vz, gvz - svz - 8u)));
                    cy_q = j > 0 ? sW[kp][2][ty - 1][tx] : (bc.fsF ? c_q : (bc.nsF ? -c_q : LDB(f.Vz, gvz - svz - rvz)));
                    if (xl) wax_q = sW[kp][0][ty][tx];
                    if (yl) wby_q = sW[kp][1][ty][tx];
                } else {
                    a_q = bc.fsK0 ? wa : (bc.nsK0 ? -wa : LDB(f.Vx, gvx - svx));
                    b_q = bc.fsK0 ? wb : (bc.nsK0 ? -wb : LDB(f.Vy, gvy - svy));
                    c_q = bc.nsK0 ? 0.0 : LDB(f.Vz, gvz - svz);
                    cx_q = i > 0 ? (bc.nsK0 ? 0.0 : LDB(f.Vz, gvz - svz - 8u)) : (bc.fsL ? c_q : (bc.nsL ? -c_q : LDB(f.Vz, gvz - svz - 8u)));
                    cy_q = j > 0 ? (bc.nsK0 ? 0.0 : LDB(f.Vz, gvz - svz - rvz)) : (bc.fsF ? c_q : (bc.nsF ? -c_q : LDB(f.Vz, gvz - svz - rvz)));
                    e_q = e_p; ex_q = ex_p; ey_q = ey_p;
                    if (xl) wax_q = bc.nsR ? 0.0 : (bc.fsK0 ? wax : (bc.nsK0 ? -wax : LDB(f.Vx, ovx2 - svx)));
                    if (yl) wby_q = bc.nsBk ? 0.0 : (bc.fsK0 ? wby : (bc.nsK0 ? -wby : LDB(f.Vy, ovy2 - svy)));
                }
                const bool hz2 = k < nz - 1;
                if (hx) STN<true>(a.o.Vx, ovx2, wx);
                if (hy) STN<true>(a.o.Vy, ovy2, wy);
                if (hz2) STN<true>(a.o.Vz, ovz2, wz);
                // η of plane k is still in e_p, ex_p, ey_p (the first iteration moves them on at the end of the step), its diagonal neighbour in exy_b
                const X2Stress R2 = x2_stress(wa, wax, way, wb, wby, wbx, wc, a_q, b_q, c_q, cx_q, cy_q, e_p, ex_p, ey_p, exy_b, e_q, ex_q, ey_q, P1, txx1, tyy1, tzz1, txy1, txz1,
                                              tyz1, _dx, _dy, _dz, th, rr);
                STN<true>(a.o.P, oc2, R2.P); STN<true>(a.o.txx, oc2, R2.txx); STN<true>(a.o.tyy, oc2, R2.tyy); STN<true>(a.o.tzz, oc2, R2.tzz);
                STN<true>(a.o.txy, oxy2, R2.txy); STN<true>(a.o.txz, oxz2 - sxz, R2.txz); STN<true>(a.o.tyz, oyz2 - syz, R2.tyz);
                if (xl) {
                    const double vxl = j > 0 ? sW[ks][0][ty - 1][tx] : (bc.nsR ? 0.0 : (bc.fsF ? wax : (bc.nsF ? -wax : LDB(f.Vx, ovx2 - rvx))));
                    const double vyg = (j == 0 && bc.nsF) ? 0.0 : (bc.fsR ? wb : (bc.nsR ? -wb : LDB(f.Vy, ovy2 - rvy + 8u)));
                    STN<true>(a.o.txy, oxy2 + 8u, x2_shear(sHx[pslot][ty], 0.5 * (_dy * (wax - vxl) + _dx * (vyg - wb)), 0.25 * (ey_p + ey_p + e_p + e_p), th));
                    const double vzg = (k == 0 && bc.nsK0) ? 0.0 : (bc.fsR ? c_q : (bc.nsR ? -c_q : LDB(f.Vz, ovz2 - svz + 8u)));
                    STN<true>(a.o.txz, oxz2 - sxz + 8u, x2_shear(hxz1, 0.5 * (_dz * (wax - wax_q) + _dx * (vzg - c_q)), 0.25 * (e_q + e_q + e_p + e_p), th));
                }
                if (yl) {
                    const double vxg = (i == 0 && bc.nsL) ? 0.0 : (bc.fsBk ? wa : (bc.nsBk ? -wa : LDB(f.Vx, ovx2 - 8u + rvx)));
                    const double vyl = i > 0 ? sW[ks][1][ty][tx - 1] : (bc.nsBk ? 0.0 : (bc.fsL ? wby : (bc.nsL ? -wby : LDB(f.Vy, ovy2 - 8u))));
                    STN<true>(a.o.txy, oxy2 + rxy, x2_shear(sHy[pslot][tx], 0.5 * (_dy * (vxg - wa) + _dx * (wby - vyl)), 0.25 * (ex_p + e_p + ex_p + e_p), th));
                    const double vzg = (k == 0 && bc.nsK0) ? 0.0 : (bc.fsBk ? c_q : (bc.nsBk ? -c_q : LDB(f.Vz, ovz2 - svz + rvz)));
                    STN<true>(a.o.tyz, oyz2 - syz + ryz, x2_shear(hyz1, 0.5 * (_dz * (wby - wby_q) + _dy * (vzg - c_q)), 0.25 * (e_q + e_q + e_p + e_p), th));
                }
                if (xl && yl) {
                    // τxy (nx, ny, k): S1 there first (nobody needed it before), then S2
                    const double v1x = sV[pslot][0][ty][tx], v1y = sV[pslot][1][ty][tx];
                    const double vxg1 = bc.nsR ? 0.0 : (bc.fsBk ? v1x : (bc.nsBk ? -v1x : LDB(f.Vx, ovx2 + rvx)));
                    const double vyg1 = bc.nsBk ? 0.0 : (bc.fsR ? v1y : (bc.nsR ? -v1y : LDB(f.Vy, ovy2 + 8u)));
                    const double c1 = x2_shear(LDB(f.txy, oxy2 + 8u + rxy), 0.5 * (_dy * (vxg1 - v1x) + _dx * (vyg1 - v1y)), 0.25 * (e_p + e_p + e_p + e_p), th);
                    const double vxg = bc.nsR ? 0.0 : (bc.fsBk ? wax : (bc.nsBk ? -wax : LDB(f.Vx, ovx2 + rvx)));
                    const double vyg = bc.nsBk ? 0.0 : (bc.fsR ? wby : (bc.nsR ? -wby : LDB(f.Vy, ovy2 + 8u)));
                    STN<true>(a.o.txy, oxy2 + 8u + rxy, x2_shear(c1, 0.5 * (_dy * (vxg - wax) + _dx * (vyg - wby)), 0.25 * (e_p + e_p + e_p + e_p), th));
                }
                if (k == nz - 1) {
                    // node plane k = nz of S2, from V2 (nz - 1) (boundary plane of Vz included) and S1 (nz) published in this step
                    const double vxg = (i == 0 && bc.nsL) ? 0.0 : (bc.fsK1 ? wa : (bc.nsK1 ? -wa : LDB(f.Vx, ovx - 8u)));
                    STN<true>(a.o.txz, oxz2, x2_shear(S.txz, 0.5 * (_dz * (vxg - wa) + _dx * (wc - wcx)), 0.25 * (ex_p + e_p + ex_p + e_p), th));
                    const double vyg = (j == 0 && bc.nsF) ? 0.0 : (bc.fsK1 ? wb : (bc.nsK1 ? -wb : LDB(f.Vy, ovy - rvy)));
                    STN<true>(a.o.tyz, oyz2, x2_shear(S.tyz, 0.5 * (_dz * (vyg - wb) + _dy * (wc - wcy)), 0.25 * (ey_p + e_p + ey_p + e_p), th));
                    if (xl) {
                        const double vxh = bc.nsR ? 0.0 : (bc.fsK1 ? wax : (bc.nsK1 ? -wax : LDB(f.Vx, ovx)));
                        const double vzg = bc.nsK1 ? 0.0 : (bc.fsR ? wc : (bc.nsR ? -wc : LDB(f.Vz, ovz2 + 8u)));
                        STN<true>(a.o.txz, oxz2 + 8u, x2_shear(hxz_t, 0.5 * (_dz * (vxh - wax) + _dx * (vzg - wc)), 0.25 * (e_p + e_p + e_p + e_p), th));
                    }
                    if (yl) {
                        const double vyh = bc.nsBk ? 0.0 : (bc.fsK1 ? wby : (bc.nsK1 ? -wby : LDB(f.Vy, ovy)));
                        const double vzg = bc.nsK1 ? 0.0 : (bc.fsBk ? wc : (bc.nsBk ? -wc : LDB(f.Vz, ovz2 + rvz)));
                        STN<true>(a.o.tyz, oyz2 + ryz, x2_shear(hyz_t, 0.5 * (_dz * (vyh - wby) + _dy * (vzg - wc)), 0.25 * (e_p + e_p + e_p + e_p), th));
                    }
                }
            }
        }
        // ---- carries for the next step: η of plane t - 1 becomes plane t - 2, plane t becomes plane t - 1
        if (k >= 0) { e_q = e_p; ex_q = ex_p; ey_q = ey_p; }
        if (!top) { e_p = e; ex_p = ex; ey_p = ey; exy_b = exy_; }
        hxz1 = hxz_t; hyz1 = hyz_t;
        fxc_b = fx_c; fyc_b = fy_c; fyy_b = fy_y; fzc_b = fz_t; ec_b = ec_t; eyb_b = eyb;
        oc += sc; oxy += sxy; oxz += sxz; oyz += syz; ovx += svx; ovy += svy; ovz += svz;
    }
}

}   // namespace
