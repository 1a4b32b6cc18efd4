// fused_tb.hpp -- TIMING PROTOTYPE of two PT iterations per launch (temporal blocking) for the viscous-limit form of k_fused3d.
//
// NOT a solver kernel: the boundary rules are stubbed and the second iteration takes its k + 1 operands from the current plane (the real kernel would run the second
// iteration one plane behind the first), so the numbers it stores are not the iteration's.  What it reproduces faithfully is what decides the question "can two
// iterations per launch be >= 1.25 x faster per iteration?": per block and plane the same global loads as k_fused3d<..., VISC> (15 arrays with the halo of a
// 64 x 8 tile), the stores of the 60 x 5 cells a two-iteration tile owns (two halo columns / rows a side are lost to the second iteration's dependency cone), BOTH
// iterations' arithmetic (velocity update, stress update, twice, with the second one's operands coming from LDS and lane shuffles exactly as the first one's do), the
// second iteration's carried registers, four barriers per plane and ~104 KB of LDS per 512-thread block (one block per CU: 8 waves, 2 per SIMD).  Whatever the real
// kernel would add (skewed z pipeline, boundary layers inside the kernel, two prologue planes per chunk instead of one) only makes it slower.
#pragma once
#include "stokes3d_kernels.hpp"

namespace {

template <int TX, int TY, int KZ, int XG>
__global__ __launch_bounds__(TX *TY, 1) void k_fused3d_tb(const SweepArgs a, const FusedBC bc, int ntx, int nty)
{
    static_assert(TX == 64, "a row per wave");
    __shared__ double sV[2][3][TY][TX];      // iteration 1 velocities (two plane slots)
    __shared__ double sY[7][TY][TX];         // y-neighbour operands of iteration 1 (as k_fused3d: P, ητ, τyy, fy, τxy, τyz, η)
    __shared__ double sS[7][TY][TX];         // iteration 1 stresses for iteration 2's velocity update (P, τxx, τyy, τzz, τxy, τxz, τyz)
    __shared__ double sW[2][3][TY][TX];      // iteration 2 velocities
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
    const int i = tix * (TX - 4) - 2 + tx;          // two halo columns on the left, one + the feeder lane on the right
    const int j = tiy * (TY - 3) - 2 + ty;          // two halo rows below, one above
    const int kb = tiz * KZ;
    const int kend = min(kb + KZ, nz);
    const bool bvalid = i >= 0 && j >= 0 && i < nx && j < ny;
    const bool a1 = bvalid && tx >= 1 && ty >= 1 && tx < TX - 1;                       // iteration 1 stresses
    const bool b2 = a1 && tx < TX - 2 && ty < TY - 1;                                    // iteration 2 velocities
    const bool own = b2 && tx >= 2 && ty >= 2;                                           // iteration 2 stresses: the 60 x 5 cells this tile stores
    const bool hx = i < nx - 1, hy = j < ny - 1;
    const double _dx = a._dx, _dy = a._dy, _dz = a._dz, th = a.theta_dtau, rr = a.r, edt = a.eta_dtau;

    const u32 sc = (u32)L.cp * 8u, svx = (u32)L.vxp * 8u, svy = (u32)L.vyp * 8u, svz = (u32)L.vzp * 8u;
    const u32 sxy = (u32)L.xyp * 8u, sxz = (u32)L.xzp * 8u, syz = (u32)L.yzp * 8u;
    const u32 rc = (u32)nx * 8u, rxy = (u32)L.xy1 * 8u, ryz = (u32)L.yz1 * 8u;
    const int ic = bvalid ? i : 0, jc = bvalid ? j : 0;
    const int kfirst = kb > 0 ? kb - 1 : 0;
    u32 oc = 8u * (u32)(ic + nx * jc) + sc * (u32)kfirst;
    u32 oxy = 8u * (u32)(ic + L.xy1 * jc) + sxy * (u32)kfirst;
    u32 oxz = 8u * (u32)(ic + L.xz1 * jc) + sxz * (u32)(kfirst + 1);
    u32 oyz = 8u * (u32)(ic + L.yz1 * jc) + syz * (u32)(kfirst + 1);
    u32 ovx = 8u * (u32)((ic + 1) + L.vx1 * (jc + 1)) + svx * (u32)(kfirst + 1);
    u32 ovy = 8u * (u32)((ic + 1) + L.vy1 * (jc + 1)) + svy * (u32)(kfirst + 1);
    u32 ovz = 8u * (u32)((ic + 1) + L.vz1 * (jc + 1)) + svz * (u32)(kfirst + 1);

    // iteration 1 carries (as k_fused3d)
    double Pc = 0, ec = 0, tzz_c = 0, fz_c = 0, s10 = 0, r10 = 0, s01p = 0, r01p = 0;
    if (bvalid) {
        Pc = LDB(f.P, oc); ec = LDB(et, oc); tzz_c = LDB(f.tzz, oc); fz_c = LDB(f.fz, oc);
        s10 = LDB(f.txz, oxz + 8u - sxz); r10 = LDB(f.tyz, oyz + ryz - syz);
        s01p = LDB(f.txz, oxz - sxz); r01p = LDB(f.tyz, oyz - syz);
    }
    double a_p = 0, b_p = 0, c_p = 0, cx_p = 0, cy_p = 0, e_p = 0, ex_p = 0, ey_p = 0;
    // iteration 2 carries
    double P2c = 0, tzz2c = 0, s10_2 = 0, r10_2 = 0, s01p2 = 0, r01p2 = 0;
    double a_q = 0, b_q = 0, c_q = 0, cx_q = 0, cy_q = 0;

    for (int k = kfirst; k < kend; ++k) {
        const bool hz = k < nz - 1;
        const bool live = k >= kb;
        const int slot = k & 1;
        double vxn = 0, vyn = 0, vzn = 0, txx_c = 0, tyy_c = 0, P_k = Pc, tzz_k = tzz_c, s01k = s01p, r01k = r01p;
        double e = 0, ex = 0, ey = 0, exy_ = 0;
        if (bvalid) e = LDB(f.eta, oc);
        double q01 = 0, s01 = 0, r11 = 0, r01 = 0, Pz = 0, ez = 0, tzz_z = 0, fz_z = 0, Py = 0, eyb = 0, tyy_y = 0;
        double fx_c = 0, fy_c = 0, fy_y = 0, vx = 0, vy = 0, vz = 0, txy_own = 0;
        const bool yrow = ty < TY - 1 && hy;
        if (bvalid) {
            const u32 dz1 = hz ? sc : 0u;
            tyy_c = LDB(f.tyy, oc); fy_c = LDB(f.fy, oc); txy_own = LDB(f.txy, oxy); r01 = LDB(f.tyz, oyz);
            if (!yrow) {
                q01 = LDB(f.txy, oxy + rxy); r11 = LDB(f.tyz, oyz + ryz);
                if (hy) { Py = LDB(f.P, oc + rc); eyb = LDB(et, oc + rc); tyy_y = LDB(f.tyy, oc + rc); fy_y = LDB(f.fy, oc + rc); }
            }
            s01 = LDB(f.txz, oxz);
            Pz = LDB(f.P, oc + dz1); ez = LDB(et, oc + dz1); tzz_z = LDB(f.tzz, oc + dz1); fz_z = LDB(f.fz, oc + dz1);
            txx_c = LDB(f.txx, oc); fx_c = LDB(f.fx, oc);
            vx = LDB(f.Vx, ovx); vy = LDB(f.Vy, ovy); vz = LDB(f.Vz, ovz);
            sY[0][ty][tx] = Pc; sY[1][ty][tx] = ec; sY[2][ty][tx] = tyy_c; sY[3][ty][tx] = fy_c; sY[4][ty][tx] = txy_own; sY[5][ty][tx] = r01; sY[6][ty][tx] = e;
        }
        __syncthreads();
        double q11 = 0, q10 = 0, s11 = 0, Px = 0, ecx = 0, txx_x = 0, fx_x = 0;
        if (bvalid) {
            if (yrow) {
                Py = sY[0][ty + 1][tx]; eyb = sY[1][ty + 1][tx]; tyy_y = sY[2][ty + 1][tx]; fy_y = sY[3][ty + 1][tx];
                q01 = sY[4][ty + 1][tx]; r11 = sY[5][ty + 1][tx];
            }
            if (ty > 0 && j > 0) ey = sY[6][ty - 1][tx]; else ey = e;
            const double e_l = __shfl_up(e, 1, TX), ey_l = __shfl_up(ey, 1, TX);
            ex = i > 0 ? e_l : e; exy_ = i > 0 ? ey_l : ey;
            q11 = __shfl_down(q01, 1, TX); q10 = __shfl_down(txy_own, 1, TX); s11 = __shfl_down(s01, 1, TX);
            Px = __shfl_down(Pc, 1, TX); ecx = __shfl_down(ec, 1, TX); txx_x = __shfl_down(txx_c, 1, TX); fx_x = __shfl_down(fx_c, 1, TX);
            if (!hx) { q11 = LDB(f.txy, oxy + 8u + rxy); s11 = LDB(f.txz, oxz + 8u); }
            if (hx) {
                const double R = (-txx_c + txx_x) * _dx + _dy * (q11 - q10) + _dz * (s11 - s10) - (-Pc + Px) * _dx - 0.5 * (fx_c + fx_x);
                vxn = vx + R * edt / (0.5 * (ec + ecx));
            } else vxn = vx;
            if (hy) {
                const double R = _dx * (q11 - q01) + _dy * (tyy_y - tyy_c) + _dz * (r11 - r10) - (-Pc + Py) * _dy - 0.5 * (fy_c + fy_y);
                vyn = vy + R * edt / (0.5 * (ec + eyb));
            } else vyn = vy;
            if (hz) {
                const double R = _dx * (s11 - s01) + _dy * (r11 - r01) + (-tzz_c + tzz_z) * _dz - (-Pc + Pz) * _dz - 0.5 * (fz_c + fz_z);
                vzn = vz + R * edt / (0.5 * (ec + ez));
            } else vzn = vz;
            s10 = s11; r10 = r11; s01p = s01; r01p = r01;
            sV[slot][0][ty][tx] = vxn; sV[slot][1][ty][tx] = vyn; sV[slot][2][ty][tx] = vzn;
        }
        __syncthreads();
        // ---- iteration 1 stresses at (i, j, k)
        double Pn = 0, txxn = 0, tyyn = 0, tzzn = 0, txyn = 0, txzn = 0, tyzn = 0;
        if (a1) {
            double va, vay, vb, vbx, vcx, vcy;
            const double vax = vxn, vby = vyn, vc = vzn;
            va = sV[slot][0][ty][tx - 1]; vay = sV[slot][0][ty - 1][tx - 1];
            vb = sV[slot][1][ty - 1][tx]; vbx = sV[slot][1][ty - 1][tx - 1];
            vcx = sV[slot][2][ty][tx - 1]; vcy = sV[slot][2][ty - 1][tx];
            if (k == 0) { a_p = bc.fsK0 ? va : -va; b_p = bc.fsK0 ? vb : -vb; c_p = 0.0; cx_p = 0.0; cy_p = 0.0; e_p = e; ex_p = ex; ey_p = ey; }
            const double dxi = (-va + vax) * _dx, dyi = (-vb + vby) * _dy, dzi = (-c_p + vc) * _dz;
            const double divV = dxi + dyi + dzi;
            const double psi = 1.0 / (1.0 / e + 0.0) * rr / th;
            Pn = (fma(0.0, 0.0, -divV) * psi + P_k) / (1.0 + 0.0 * psi);
            const double d3 = divV * (1.0 / 3.0);
            const double dtr = dev_dtau_r(th, e, 0.0);
            txxn = txx_c + dev_stress_inc(txx_c, 0.0, e, dxi - d3, 0.0, dtr);
            tyyn = tyy_c + dev_stress_inc(tyy_c, 0.0, e, dyi - d3, 0.0, dtr);
            tzzn = tzz_k + dev_stress_inc(tzz_k, 0.0, e, dzi - d3, 0.0, dtr);
            {
                const double s_ = 0.5 * (_dy * (va - vay) + _dx * (vb - vbx)), ee = 0.25 * (exy_ + ey + ex + e);
                txyn = txy_own + dev_stress_inc(txy_own, 0.0, ee, s_, 0.0, dev_dtau_r(th, ee, 0.0));
            }
            {
                const double s_ = 0.5 * (_dz * (va - a_p) + _dx * (c_p - cx_p)), ee = 0.25 * (ex_p + e_p + ex + e);
                txzn = s01k + dev_stress_inc(s01k, 0.0, ee, s_, 0.0, dev_dtau_r(th, ee, 0.0));
            }
            {
                const double s_ = 0.5 * (_dz * (vb - b_p) + _dy * (c_p - cy_p)), ee = 0.25 * (ey_p + e_p + ey + e);
                tyzn = r01k + dev_stress_inc(r01k, 0.0, ee, s_, 0.0, dev_dtau_r(th, ee, 0.0));
            }
            a_p = va; b_p = vb; c_p = vc; cx_p = vcx; cy_p = vcy;
        }
        // publish iteration 1's stresses for iteration 2's velocity update
        sS[0][ty][tx] = Pn; sS[1][ty][tx] = txxn; sS[2][ty][tx] = tyyn; sS[3][ty][tx] = tzzn; sS[4][ty][tx] = txyn; sS[5][ty][tx] = txzn; sS[6][ty][tx] = tyzn;
        __syncthreads();
        // ---- iteration 2 velocities (prototype: the k + 1 operands are the current plane's)
        double wx = 0, wy = 0, wz = 0;
        if (b2) {
            const double Py2 = sS[0][ty + 1][tx], tyy2y = sS[2][ty + 1][tx], q01b = sS[4][ty + 1][tx], r11b = sS[6][ty + 1][tx];
            const double q11b = __shfl_down(q01b, 1, TX), q10b = __shfl_down(txyn, 1, TX), s11b = __shfl_down(txzn, 1, TX);
            const double Px2 = __shfl_down(Pn, 1, TX), txx2x = __shfl_down(txxn, 1, TX);
            {
                const double R = (-txxn + txx2x) * _dx + _dy * (q11b - q10b) + _dz * (s11b - s10_2) - (-Pn + Px2) * _dx - 0.5 * (fx_c + fx_x);
                wx = vxn + R * edt / (0.5 * (ec + ecx));
            }
            {
                const double R = _dx * (q11b - q01b) + _dy * (tyy2y - tyyn) + _dz * (r11b - r10_2) - (-Pn + Py2) * _dy - 0.5 * (fy_c + fy_y);
                wy = vyn + R * edt / (0.5 * (ec + eyb));
            }
            {
                const double R = _dx * (s11b - txzn) + _dy * (r11b - tyzn) + (-tzz2c + tzzn) * _dz - (-P2c + Pn) * _dz - 0.5 * (fz_c + fz_z);
                wz = vzn + R * edt / (0.5 * (ec + ez));
            }
            s10_2 = s11b; r10_2 = r11b; s01p2 = txzn; r01p2 = tyzn; P2c = Pn; tzz2c = tzzn;
        }
        sW[slot][0][ty][tx] = wx; sW[slot][1][ty][tx] = wy; sW[slot][2][ty][tx] = wz;
        __syncthreads();
        // ---- iteration 2 stresses: what the launch stores
        if (own && live) {
            const double va = sW[slot][0][ty][tx - 1], vay = sW[slot][0][ty - 1][tx - 1];
            const double vb = sW[slot][1][ty - 1][tx], vbx = sW[slot][1][ty - 1][tx - 1];
            const double vcx = sW[slot][2][ty][tx - 1], vcy = sW[slot][2][ty - 1][tx];
            const double dxi = (-va + wx) * _dx, dyi = (-vb + wy) * _dy, dzi = (-c_q + wz) * _dz;
            const double divV = dxi + dyi + dzi;
            const double psi = 1.0 / (1.0 / e + 0.0) * rr / th;
            STN<true>(a.o.Vx, ovx, wx); STN<true>(a.o.Vy, ovy, wy); STN<true>(a.o.Vz, ovz, wz);
            STN<true>(a.o.P, oc, (fma(0.0, 0.0, -divV) * psi + Pn) / (1.0 + 0.0 * psi));
            const double d3 = divV * (1.0 / 3.0);
            const double dtr = dev_dtau_r(th, e, 0.0);
            STN<true>(a.o.txx, oc, txxn + dev_stress_inc(txxn, 0.0, e, dxi - d3, 0.0, dtr));
            STN<true>(a.o.tyy, oc, tyyn + dev_stress_inc(tyyn, 0.0, e, dyi - d3, 0.0, dtr));
            STN<true>(a.o.tzz, oc, tzzn + dev_stress_inc(tzzn, 0.0, e, dzi - d3, 0.0, dtr));
            {
                const double s_ = 0.5 * (_dy * (va - vay) + _dx * (vb - vbx)), ee = 0.25 * (exy_ + ey + ex + e);
                STN<true>(a.o.txy, oxy, txyn + dev_stress_inc(txyn, 0.0, ee, s_, 0.0, dev_dtau_r(th, ee, 0.0)));
            }
            {
                const double s_ = 0.5 * (_dz * (va - a_q) + _dx * (c_q - cx_q)), ee = 0.25 * (ex_p + e_p + ex + e);
                STN<true>(a.o.txz, oxz - sxz, s01p2 + dev_stress_inc(s01p2, 0.0, ee, s_, 0.0, dev_dtau_r(th, ee, 0.0)));
            }
            {
                const double s_ = 0.5 * (_dz * (vb - b_q) + _dy * (c_q - cy_q)), ee = 0.25 * (ey_p + e_p + ey + e);
                STN<true>(a.o.tyz, oyz - syz, r01p2 + dev_stress_inc(r01p2, 0.0, ee, s_, 0.0, dev_dtau_r(th, ee, 0.0)));
            }
            a_q = va; b_q = vb; c_q = wz; cx_q = vcx; cy_q = vcy;
        }
        if (a1) { e_p = e; ex_p = ex; ey_p = ey; }
        Pc = Pz; ec = ez; tzz_c = tzz_z; fz_c = fz_z;
        oc += sc; oxy += sxy; oxz += sxz; oyz += syz; ovx += svx; ovy += svy; ovz += svz;
    }
}

}   // namespace
