// fused_ws.hpp -- experiment kept for the record (kbench only, not part of the library): producer/consumer wave specialisation of
// the fused iteration kernel.  Measured slower than the single-role kernel.  Include after stokes3d_kernels.hpp.
#pragma once
namespace {

// ------------------------------------------------------------------------------------------------
// Wave-specialised form of the fused iteration kernel: the block has 2 x (TX x TY) threads; the first
// half ("producers") runs the velocity sweep of plane k+1 while the second half ("consumers") runs
// the stress sweep of plane k from the velocities the producers left in LDS one step earlier.  Each
// role keeps only its own operands in registers (about the budget of the separate sweeps instead of
// their sum: the single-role form above needs ~150 VGPRs and runs at 3 waves/SIMD), and the two
// memory round trips of a plane overlap.  One s_barrier per plane, two LDS slots.
// Consumers re-load P, τxx, τyy, τzz, τxz, τyz of their own cell (L1/L2 hits: the producer wave of the
// same cell loaded the same lines one step earlier).
// ------------------------------------------------------------------------------------------------
template <int TX, int TY, int KZ, int MINW>
__global__ __launch_bounds__(2 * TX *TY, MINW) void k_fused3d_ws(const SweepArgs a, const FusedBC bc, int ntx, int nty)
{
    __shared__ double sV[2][3][TY][TX];
    const Lay3 &L = a.L;
    const int nx = L.nx, ny = L.ny, nz = L.nz;
    const jrx_stokes3d_fields &f = a.f;
    const int role = (int)(threadIdx.x / (TX * TY));          // wave-uniform: TX*TY is a multiple of 64
    const int lt = (int)(threadIdx.x % (TX * TY));
    const int tx = lt % TX, ty = lt / TX;
    const int tile = blockIdx.x;
    const int tix = tile % ntx, tr = tile / ntx, tiy = tr % nty, tiz = tr / nty;
    const int i = tix * (TX - 1) - 1 + tx;
    const int j = tiy * (TY - 1) - 1 + ty;
    const int kb = tiz * KZ;
    const int kend = min(kb + KZ, nz);
    const bool bvalid = i >= 0 && j >= 0 && i < nx && j < ny;
    const bool avalid = bvalid && tx >= 1 && ty >= 1;
    const int kfirst = kb > 0 ? kb - 1 : 0;
    const int nplanes = kend - kfirst;
    const int ic = bvalid ? i : 0, jc = bvalid ? j : 0;

    const u32 sc = (u32)L.cp * 8u, svx = (u32)L.vxp * 8u, svy = (u32)L.vyp * 8u, svz = (u32)L.vzp * 8u;
    const u32 sxy = (u32)L.xyp * 8u, sxz = (u32)L.xzp * 8u, syz = (u32)L.yzp * 8u;
    const u32 rc = (u32)nx * 8u, rxy = (u32)L.xy1 * 8u, ryz = (u32)L.yz1 * 8u;
    const u32 rvx = (u32)L.vx1 * 8u, rvy = (u32)L.vy1 * 8u, rvz = (u32)L.vz1 * 8u;
    u32 oc = 8u * (u32)(ic + nx * jc) + sc * (u32)kfirst;
    u32 oxy = 8u * (u32)(ic + L.xy1 * jc) + sxy * (u32)kfirst;
    u32 oxz = 8u * (u32)(ic + L.xz1 * jc) + sxz * (u32)(kfirst + 1);
    u32 oyz = 8u * (u32)(ic + L.yz1 * jc) + syz * (u32)(kfirst + 1);
    u32 ovx = 8u * (u32)((ic + 1) + L.vx1 * (jc + 1)) + svx * (u32)(kfirst + 1);
    u32 ovy = 8u * (u32)((ic + 1) + L.vy1 * (jc + 1)) + svy * (u32)(kfirst + 1);
    u32 ovz = 8u * (u32)((ic + 1) + L.vz1 * (jc + 1)) + svz * (u32)(kfirst + 1);

    if (role == 0) {
        // ------------------------------------------------------------------ producers: velocity sweep
        const double *et = a.etatau;
        const double _dx = a._dx, _dy = a._dy, _dz = a._dz, edt = a.eta_dtau;
        const bool hx = i < nx - 1, hy = j < ny - 1;
        const u32 dx1 = hx ? 8u : 0u, dy1 = hy ? rc : 0u;
        double Pc = 0, ec = 0, tzz_c = 0, fz_c = 0, s10 = 0, r10 = 0;
        if (bvalid) {
            Pc = LDB(f.P, oc); ec = LDB(et, oc); tzz_c = LDB(f.tzz, oc); fz_c = LDB(f.fz, oc);
            s10 = LDB(f.txz, oxz + 8u - sxz); r10 = LDB(f.tyz, oyz + ryz - syz);
        }
        for (int st = 0; st <= nplanes; ++st) {
            if (st < nplanes && bvalid) {
                const int k = kfirst + st;
                const bool hz = k < nz - 1;
                const bool own = avalid && k >= kb;
                const u32 dz1 = hz ? sc : 0u;
                const double q11 = LDB(f.txy, oxy + 8u + rxy), q10 = LDB(f.txy, oxy + 8u), q01 = LDB(f.txy, oxy + rxy);
                const double s11 = LDB(f.txz, oxz + 8u), s01 = LDB(f.txz, oxz);
                const double r11 = LDB(f.tyz, oyz + ryz), r01 = LDB(f.tyz, oyz);
                const double Pz = LDB(f.P, oc + dz1), ez = LDB(et, oc + dz1), tzz_z = LDB(f.tzz, oc + dz1), fz_z = LDB(f.fz, oc + dz1);
                const double Px = LDB(f.P, oc + dx1), Py = LDB(f.P, oc + dy1), ex = LDB(et, oc + dx1), ey = LDB(et, oc + dy1);
                const double txx_c = LDB(f.txx, oc), txx_x = LDB(f.txx, oc + dx1), tyy_c = LDB(f.tyy, oc), tyy_y = LDB(f.tyy, oc + dy1);
                const double fx_c = LDB(f.fx, oc), fx_x = LDB(f.fx, oc + dx1), fy_c = LDB(f.fy, oc), fy_y = LDB(f.fy, oc + dy1);
                const double vx = LDB(f.Vx, ovx), vy = LDB(f.Vy, ovy), vz = LDB(f.Vz, ovz);
                double vxn, vyn, vzn;
                if (hx) {
                    const double R = (-txx_c + txx_x) * _dx + _dy * (q11 - q10) + _dz * (s11 - s10) - (-Pc + Px) * _dx - 0.5 * (fx_c + fx_x);
                    vxn = vx + R * edt / (0.5 * (ec + ex));
                    if (own) STB(a.o.Vx, ovx, vxn);
                } else vxn = bc.nsR ? 0.0 : vx;
                if (hy) {
                    const double R = _dx * (q11 - q01) + _dy * (tyy_y - tyy_c) + _dz * (r11 - r10) - (-Pc + Py) * _dy - 0.5 * (fy_c + fy_y);
                    vyn = vy + R * edt / (0.5 * (ec + ey));
                    if (own) STB(a.o.Vy, ovy, vyn);
                } else vyn = bc.nsBk ? 0.0 : vy;
                if (hz) {
                    const double R = _dx * (s11 - s01) + _dy * (r11 - r01) + (-tzz_c + tzz_z) * _dz - (-Pc + Pz) * _dz - 0.5 * (fz_c + fz_z);
                    vzn = vz + R * edt / (0.5 * (ec + ez));
                    if (own) STB(a.o.Vz, ovz, vzn);
                } else vzn = bc.nsK1 ? 0.0 : vz;
                Pc = Pz; ec = ez; tzz_c = tzz_z; fz_c = fz_z; s10 = s11; r10 = r11;
                const int slot = st & 1;
                sV[slot][0][ty][tx] = vxn; sV[slot][1][ty][tx] = vyn; sV[slot][2][ty][tx] = vzn;
                oc += sc; oxy += sxy; oxz += sxz; oyz += syz; ovx += svx; ovy += svy; ovz += svz;
            }
            __syncthreads();
        }
    } else {
        // ------------------------------------------------------------------ consumers: stress sweep
        const double _dx = a._dx, _dy = a._dy, _dz = a._dz, dt = a.dt, th = a.theta_dtau, _dt = 1.0 / dt, rr = a.r;
        const int im = max(ic - 1, 0), jm = max(jc - 1, 0);
        const u32 dcx = 8u * (u32)(ic - im), dcy = rc * (u32)(jc - jm);
        double a_p = 0, b_p = 0, c_p = 0, cx_p = 0, cy_p = 0, e_p = 0, ex_p = 0, ey_p = 0, g_p = 0, gx_p = 0, gy_p = 0;
        for (int st = 0; st <= nplanes; ++st) {
            if (st >= 1 && avalid) {
                const int k = kfirst + st - 1;
                const bool live = k >= kb;
                const int slot = (st - 1) & 1;
                // operands that do not depend on the new velocities first (in flight while LDS is read)
                const double e = LDB(f.eta, oc), ex = LDB(f.eta, oc - dcx), ey = LDB(f.eta, oc - dcy);
                const double g = LDB(f.G, oc), gx = LDB(f.G, oc - dcx), gy = LDB(f.G, oc - dcy);
                double exy_ = 0, gxy = 0, P0 = 0, Kc = 0, Qc = 0, P_k = 0, txx_c = 0, tyy_c = 0, tzz_k = 0;
                double toxx = 0, toyy = 0, tozz = 0, txy = 0, toxy = 0, txz = 0, toxz = 0, tyz = 0, toyz = 0;
                if (live) {
                    exy_ = LDB(f.eta, oc - dcx - dcy); gxy = LDB(f.G, oc - dcx - dcy);
                    P0 = LDB(f.P0, oc); Kc = LDB(f.K, oc); Qc = LDB(f.Q, oc); P_k = LDB(f.P, oc);
                    txx_c = LDB(f.txx, oc); tyy_c = LDB(f.tyy, oc); tzz_k = LDB(f.tzz, oc);
                    toxx = LDB(f.toxx, oc); toyy = LDB(f.toyy, oc); tozz = LDB(f.tozz, oc);
                    txy = LDB(f.txy, oxy); toxy = LDB(f.toxy, oxy);
                    txz = LDB(f.txz, oxz - sxz); toxz = LDB(f.toxz, oxz - sxz);
                    tyz = LDB(f.tyz, oyz - syz); toyz = LDB(f.toyz, oyz - syz);
                }
                const double vax = sV[slot][0][ty][tx], vby = sV[slot][1][ty][tx], vc = sV[slot][2][ty][tx];
                double va, vay, vb, vbx, vcx, vcy;
                const u32 gvx = ovx - 8u, gvy = ovy - rvy, gvz = ovz;
                va = i > 0 ? sV[slot][0][ty][tx - 1] : (bc.nsL ? 0.0 : LDB(f.Vx, gvx));
                if (j > 0) vay = i > 0 ? sV[slot][0][ty - 1][tx - 1] : (bc.nsL ? 0.0 : LDB(f.Vx, gvx - rvx));
                else vay = bc.fsF ? va : (bc.nsF ? -va : LDB(f.Vx, gvx - rvx));
                vb = j > 0 ? sV[slot][1][ty - 1][tx] : (bc.nsF ? 0.0 : LDB(f.Vy, gvy));
                if (i > 0) vbx = j > 0 ? sV[slot][1][ty - 1][tx - 1] : (bc.nsF ? 0.0 : LDB(f.Vy, gvy - 8u));
                else vbx = bc.fsL ? vb : (bc.nsL ? -vb : LDB(f.Vy, gvy - 8u));
                vcx = i > 0 ? sV[slot][2][ty][tx - 1] : (bc.fsL ? vc : (bc.nsL ? -vc : LDB(f.Vz, gvz - 8u)));
                vcy = j > 0 ? sV[slot][2][ty - 1][tx] : (bc.fsF ? vc : (bc.nsF ? -vc : LDB(f.Vz, gvz - rvz)));
                if (k == 0) {
                    a_p = bc.fsK0 ? va : (bc.nsK0 ? -va : LDB(f.Vx, gvx - svx));
                    b_p = bc.fsK0 ? vb : (bc.nsK0 ? -vb : LDB(f.Vy, gvy - svy));
                    c_p = bc.nsK0 ? 0.0 : LDB(f.Vz, gvz - svz);
                    cx_p = i > 0 ? (bc.nsK0 ? 0.0 : LDB(f.Vz, gvz - svz - 8u)) : (bc.fsL ? c_p : (bc.nsL ? -c_p : LDB(f.Vz, gvz - svz - 8u)));
                    cy_p = j > 0 ? (bc.nsK0 ? 0.0 : LDB(f.Vz, gvz - svz - rvz)) : (bc.fsF ? c_p : (bc.nsF ? -c_p : LDB(f.Vz, gvz - svz - rvz)));
                    e_p = e; ex_p = ex; ey_p = ey; g_p = g; gx_p = gx; gy_p = gy;
                }
                if (live) {
                    {   // centre
                        const double dxi = (-va + vax) * _dx;
                        const double dyi = (-vb + vby) * _dy;
                        const double dzi = (-c_p + vc) * _dz;
                        const double divV = dxi + dyi + dzi;
                        const double _Gdt = 1.0 / (g * dt);
                        const double _Kdt = 1.0 / (Kc * dt);
                        const double rhs = -divV + (Qc * _dt);
                        const double psi = 1.0 / (1.0 / e + _Gdt) * rr / th;
                        STB(a.o.P, oc, (fma(P0, _Kdt, rhs) * psi + P_k) / (1.0 + _Kdt * psi));
                        const double d3 = divV * (1.0 / 3.0);
                        const double exx = dxi - d3, eyy = dyi - d3, ezz = dzi - d3;
                        const double dtr = dev_dtau_r(th, e, _Gdt);
                        STB(a.o.txx, oc, txx_c + dev_stress_inc(txx_c, toxx, e, exx, _Gdt, dtr));
                        STB(a.o.tyy, oc, tyy_c + dev_stress_inc(tyy_c, toyy, e, eyy, _Gdt, dtr));
                        STB(a.o.tzz, oc, tzz_k + dev_stress_inc(tzz_k, tozz, e, ezz, _Gdt, dtr));
                    }
                    {   // τxy (i,j,k)
                        const double s_ = 0.5 * (_dy * (va - vay) + _dx * (vb - vbx));
                        const double ee = 0.25 * (exy_ + ey + ex + e);
                        const double gg = 0.25 * (gxy + gy + gx + g);
                        const double _Gdt = 1.0 / (gg * dt);
                        const double dtr = dev_dtau_r(th, ee, _Gdt);
                        STB(a.o.txy, oxy, txy + dev_stress_inc(txy, toxy, ee, s_, _Gdt, dtr));
                    }
                    {   // τxz (i,j,k)
                        const double s_ = 0.5 * (_dz * (va - a_p) + _dx * (c_p - cx_p));
                        const double ee = 0.25 * (ex_p + e_p + ex + e);
                        const double gg = 0.25 * (gx_p + g_p + gx + g);
                        const double _Gdt = 1.0 / (gg * dt);
                        const double dtr = dev_dtau_r(th, ee, _Gdt);
                        STB(a.o.txz, oxz - sxz, txz + dev_stress_inc(txz, toxz, ee, s_, _Gdt, dtr));
                    }
                    {   // τyz (i,j,k)
                        const double s_ = 0.5 * (_dz * (vb - b_p) + _dy * (c_p - cy_p));
                        const double ee = 0.25 * (ey_p + e_p + ey + e);
                        const double gg = 0.25 * (gy_p + g_p + gy + g);
                        const double _Gdt = 1.0 / (gg * dt);
                        const double dtr = dev_dtau_r(th, ee, _Gdt);
                        STB(a.o.tyz, oyz - syz, tyz + dev_stress_inc(tyz, toyz, ee, s_, _Gdt, dtr));
                    }
                }
                a_p = va; b_p = vb; c_p = vc; cx_p = vcx; cy_p = vcy;
                e_p = e; ex_p = ex; ey_p = ey; g_p = g; gx_p = gx; gy_p = gy;
                oc += sc; oxy += sxy; oxz += sxz; oyz += syz; ovx += svx; ovy += svy; ovz += svz;
            }
            __syncthreads();
        }
    }
}


}   // namespace
