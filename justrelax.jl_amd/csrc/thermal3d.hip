// thermal3d.hip -- 3D pseudo-transient heat diffusion for gfx950.
//
// Reference being replaced: src/thermal_diffusion/DiffusionPT_solver.jl:34-149 (array-coefficient form) and :181-305
// (rheology form as test/test_diffusion3D.jl evaluates it: constant k and Cp, rho = rho0*(1 - alpha*(T - T0))), kernels
// DiffusionPT_kernels.jl:6-61 (compute_flux! 3D), :160-199 (update_T! 3D), :250-282 (check_res! 3D), :670-673 (update_ΔT!)
// and thermal_bcs! 3D (BoundaryConditions.jl:46-54; constant_value.jl:15-33; free_slip.jl:86-103; periodic.jl:42-60).
// 3D thermal face names: bot <-> k = 1, top <-> k = end.  HBM-bound fp64 stencils (20 array passes per iteration).
#include "jrx_internal.hpp"
#include "jrx_kernels.hpp"

namespace {

enum { XL = 0, XR = 1, YF = 2, YB = 3, ZT = 4, ZB = 5 };

struct T3Args {
    jrx_thermal3d_fields t;
    jrx_thermal3d_params p;
};

__device__ __forceinline__ double rhoCp3_of(const jrx_thermal3d_params &p, const double *rhoCp, i64 c, double T)
{
    return p.rheology_form ? p.Cp * (p.rho0 * (1.0 - p.alpha * (T - p.T0))) : rhoCp[c];
}

#define NODE_IJK(n1_, n2_)                                              \
    const int t_ = blockIdx.x * blockDim.x + threadIdx.x;               \
    const int j = t_ / (n1_), i = t_ - j * (n1_), k = blockIdx.y;       \
    if (j >= (n2_)) return;
#define GRID_IJK(n1_, n2_, n3_) dim3((unsigned)(((i64)(n1_) * (n2_) + 255) / 256), (unsigned)(n3_))
#define T3_(i_, j_, k_) T[(i_) + (i64)(nx + 2) * ((j_) + (i64)(ny + 2) * (k_))]
#define CC_(A, i_, j_, k_) (A)[(i_) + (i64)nx * ((j_) + (i64)ny * (k_))]

// compute_flux! over (nx+1, ny+1, nz+1).  Q2 = false skips the stores of qT*2 (the un-relaxed flux is only read by check_res!,
// DiffusionPT_solver.jl:113-126, so it is written on the iterations a check follows)
template <bool Q2>
__global__ __launch_bounds__(256) void k_flux3d(const T3Args a)
{
    const int nx = (int)a.p.nx, ny = (int)a.p.ny, nz = (int)a.p.nz;
    NODE_IJK(nx + 1, ny + 1)
    const double *__restrict__ T = a.t.T, *__restrict__ th = a.t.thetar_dtau;
    const double kc = (a.p.k_const + a.p.k_const) * 0.5;
    if (j < ny && k < nz) {
        const i64 q = i + (i64)(nx + 1) * (j + (i64)ny * k);
        if (i == 0 && a.p.constant_flux_on[XL]) a.t.qTx[q] = a.p.constant_flux[XL];
        else if (i == nx && a.p.constant_flux_on[XR]) a.t.qTx[q] = a.p.constant_flux[XR];
        else {
            const int l = clampi(i - 1, 0, nx - 1), r = clampi(i, 0, nx - 1);
            const double K = a.p.rheology_form ? kc : (CC_(a.t.K, l, j, k) + CC_(a.t.K, r, j, k)) * 0.5;
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
            const double K = a.p.rheology_form ? kc : (CC_(a.t.K, i, l, k) + CC_(a.t.K, i, r, k)) * 0.5;
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
            const double K = a.p.rheology_form ? kc : (CC_(a.t.K, i, j, l) + CC_(a.t.K, i, j, r)) * 0.5;
            const double t = (CC_(th, i, j, l) + CC_(th, i, j, r)) * 0.5;
            const double qv = -K * (T3_(i + 1, j + 1, k + 1) - T3_(i + 1, j + 1, k)) * a.p._dz;
            if (Q2) a.t.qTz2[q] = qv;
            a.t.qTz[q] = (a.t.qTz[q] * t + qv) / (1.0 + t);
        }
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
template <bool RES, bool BCF = false>
__global__ __launch_bounds__(256) void k_updateT3d(const T3Args a)
{
    const int nx = (int)a.p.nx, ny = (int)a.p.ny, nz = (int)a.p.nz;
    NODE_IJK(nx, ny)
    if (k >= nz) return;
    const i64 c = i + (i64)nx * (j + (i64)ny * k), I1 = (i + 1) + (i64)(nx + 2) * ((j + 1) + (i64)(ny + 2) * (k + 1));
    const double _dt = 1.0 / a.p.dt;
    const double Tc = a.t.T[I1];
    const double rcp = rhoCp3_of(a.p, a.t.rhoCp, c, Tc);
    const double *qx = RES ? a.t.qTx2 : a.t.qTx, *qy = RES ? a.t.qTy2 : a.t.qTy, *qz = RES ? a.t.qTz2 : a.t.qTz;
    const double divq = (qx[(i + 1) + (i64)(nx + 1) * (j + (i64)ny * k)] - qx[i + (i64)(nx + 1) * (j + (i64)ny * k)]) * a.p._dx +
                        (qy[i + (i64)nx * ((j + 1) + (i64)(ny + 1) * k)] - qy[i + (i64)nx * (j + (i64)(ny + 1) * k)]) * a.p._dy +
                        (qz[i + (i64)nx * (j + (i64)ny * (k + 1))] - qz[c]) * a.p._dz;
    if (RES) {
        a.t.ResT[c] = -rcp * (Tc - a.t.Told[I1]) * _dt - divq + a.t.H[c] + a.t.shear_heating[c];
    } else {
        const double dr = a.t.dtau_rho[c];
        const double Tn = (dr * (-divq + a.t.Told[I1] * rcp * _dt + a.t.H[c] + a.t.shear_heating[c]) + Tc) / (1.0 + dr * rcp * _dt);
        a.t.T[I1] = Tn;
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
    if (p->nx < 2 || p->ny < 2 || p->nz < 2) return jrx_fail(h, JRX_ERR_ARG, "thermal grid too small");
    if ((double)(p->nx + 2) * (double)(p->ny + 2) * (double)(p->nz + 2) >= 2147483647.0)
        return jrx_fail(h, JRX_ERR_UNSUPPORTED, "local block too large for 32-bit plane indices");
    const void *req[] = {t->T, t->Told, t->dT, t->qTx, t->qTx2, t->qTy, t->qTy2, t->qTz, t->qTz2, t->H, t->shear_heating, t->ResT, t->thetar_dtau, t->dtau_rho};
    for (const void *q : req)
        if (!q) return jrx_fail(h, JRX_ERR_ARG, "a required thermal field pointer is NULL");
    if (!p->rheology_form && (!t->K || !t->rhoCp)) return jrx_fail(h, JRX_ERR_ARG, "K / rhoCp arrays required in the array-coefficient form");
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

jrx_status enqueue_titer3(jrx_handle *h, const jrx_thermal3d_fields *t, const jrx_thermal3d_params *p, bool q2 = true, bool fuse_bc = false)
{
    T3Args a;
    a.t = *t; a.p = *p;
    const int nx = (int)p->nx, ny = (int)p->ny, nz = (int)p->nz;
    hipStream_t s = h->stream;
    if (q2) hipLaunchKernelGGL(k_flux3d<true>, GRID_IJK(nx + 1, ny + 1, nz + 1), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_flux3d<false>, GRID_IJK(nx + 1, ny + 1, nz + 1), dim3(256), 0, s, a);
    JRX_LAUNCH_CHECK(h);
    bool any_periodic = false;
    for (int q = 0; q < 6; q++) any_periodic |= p->periodic[q] != 0;
    if (fuse_bc && !any_periodic && !jrx_comm_active(h)) {        // thermal_bcs! refreshed by the update kernel itself
        hipLaunchKernelGGL((k_updateT3d<false, true>), GRID_IJK(nx, ny, nz), dim3(256), 0, s, a);
        JRX_LAUNCH_CHECK(h);
        return JRX_OK;
    }
    hipLaunchKernelGGL(k_updateT3d<false>, GRID_IJK(nx, ny, nz), dim3(256), 0, s, a);
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
    JRX_TRY(enqueue_titer3(h, t, p));
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_thermal3d_check_res(jrx_handle *h, const jrx_thermal3d_fields *t, const jrx_thermal3d_params *p)
{
    JRX_TRY(checkT3(h, t, p));
    T3Args a;
    a.t = *t; a.p = *p;
    hipLaunchKernelGGL(k_updateT3d<true>, GRID_IJK(p->nx, p->ny, p->nz), dim3(256), 0, h->stream, a);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_heatdiffusion_PT3d(jrx_handle *h, const jrx_thermal3d_fields *t, const jrx_thermal3d_params *p, int64_t *iter_count, double *norm_ResT,
                                  int64_t cap, int64_t *nnorms)
{
    JRX_TRY(checkT3(h, t, p));
    if (p->nout < 1) return jrx_fail(h, JRX_ERR_ARG, "nout must be >= 1");
    const int nx = (int)p->nx, ny = (int)p->ny, nz = (int)p->nz;
    const i64 nT = (i64)(nx + 2) * (ny + 2) * (nz + 2), n = (i64)nx * ny * nz;
    hipStream_t s = h->stream;
    const double sq = 1.0 / sqrt((double)n);
    JRX_HIP(h, hipMemcpyAsync(t->Told, t->T, (size_t)nT * sizeof(double), hipMemcpyDeviceToDevice, s));   // @copy thermal.Told thermal.T
    int64_t iter = 0, cnt = 0;
    double err = 2 * p->eps;
    T3Args a;
    a.t = *t; a.p = *p;
    while (err > p->eps && iter < p->iterMax) {
        // qT*2 is observable after the loop as well (the arrays belong to the caller): written on check iterations and on the last one
        const bool q2 = ((iter + 1) % p->nout == 0) || (iter + 1 >= p->iterMax);
        JRX_TRY(enqueue_titer3(h, t, p, q2, true));
        iter++;
        if (iter % p->nout == 0) {
            hipLaunchKernelGGL(k_updateT3d<true>, GRID_IJK(nx, ny, nz), dim3(256), 0, s, a);
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
            if (cnt < cap) {
                if (norm_ResT) norm_ResT[cnt] = err;
                if (iter_count) iter_count[cnt] = iter;
            }
            cnt++;
            if (p->verbose) printf("iter = %lld, err = %1.3e \n", (long long)iter, err);
        }
    }
    hipLaunchKernelGGL(k_sub3, dim3(1024), dim3(256), 0, s, t->dT, (const double *)t->T, (const double *)t->Told, nT);   // update_ΔT!
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(s));
    if (nnorms) *nnorms = cnt < cap ? cnt : cap;
    return JRX_OK;
}

}   // extern "C"
