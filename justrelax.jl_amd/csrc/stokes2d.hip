// stokes2d.hip -- 2D visco-elastic pseudo-transient Stokes path for gfx950.
//
// Reference being replaced: src/stokes/Stokes2D.jl:181-325 and its kernels
// (VelocityKernels.jl:3-6,10-44,108-131,246-269; PressureKernels.jl:10-15,186-195;
// StressKernels.jl:63-91; MiniKernels.jl:76-80; boundaryconditions/*.jl).  Same two-sweep fusion as
// the 3D path; at the reference's 2D sizes (<= 1024^2) the working set is Infinity-Cache resident,
// so these kernels are launch/latency bound rather than HBM bound.
#include "jrx_internal.hpp"
#include "jrx_kernels.hpp"

namespace {

struct Args2 {
    jrx_stokes2d_fields f;
    const double *etatau;
    double _dx, _dy, dt, r, theta_dtau, eta_dtau;
    int nx, ny;
};

#define VX(i_, j_) Vx[(i_) + (i64)(nx + 1) * (j_)]
#define VY(i_, j_) Vy[(i_) + (i64)(nx + 2) * (j_)]
#define CC(i_, j_) ((i_) + (i64)nx * (j_))

template <bool DIAG>
__global__ __launch_bounds__(256) void k_stress2d(const Args2 a)
{
    const int nx = a.nx, ny = a.ny;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = t / (nx + 1), i = t - j * (nx + 1);
    if (j > ny) return;
    const double *__restrict__ Vx = a.f.Vx, *__restrict__ Vy = a.f.Vy, *__restrict__ eta = a.f.eta, *__restrict__ G = a.f.G;
    const double _dx = a._dx, _dy = a._dy, dt = a.dt, th = a.theta_dtau;
    if (i < nx && j < ny) {
        const i64 c = CC(i, j);
        const double dxi = (-VX(i, j + 1) + VX(i + 1, j + 1)) * _dx;
        const double dyi = (-VY(i + 1, j) + VY(i + 1, j + 1)) * _dy;
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
        const double exy = 0.5 * (_dy * (VX(i, j + 1) - VX(i, j)) + _dx * (VY(i + 1, j) - VY(i, j)));
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

// compute_V! (VelocityKernels.jl:108-131); RES additionally stores compute_Res! (:246-269) values
template <bool RES_ONLY>
__global__ __launch_bounds__(256) void k_velocity2d(const Args2 a)
{
    const int nx = a.nx, ny = a.ny;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = t / nx, i = t - j * nx;
    if (j >= ny) return;
    const double _dx = a._dx, _dy = a._dy, edt = a.eta_dtau;
    const double *__restrict__ P = a.f.P, *__restrict__ txy = a.f.txy, *__restrict__ et = a.etatau;
#define TXY(i_, j_) txy[(i_) + (i64)(nx + 1) * (j_)]
    const i64 c = CC(i, j);
    if (i < nx - 1) {
        const double dP = (-P[c] + P[c + 1]) * _dx, dT = (-a.f.txx[c] + a.f.txx[c + 1]) * _dx;
        const double dS = (-TXY(i + 1, j) + TXY(i + 1, j + 1)) * _dy, av = (a.f.fx[c] + a.f.fx[c + 1]) * 0.5;
        if (RES_ONLY) a.f.Rx[i + (i64)(nx - 1) * j] = dT + dS - dP - av;
        else a.f.Vx[(i + 1) + (i64)(nx + 1) * (j + 1)] += (-dP + dT + dS - av) * edt / ((et[c] + et[c + 1]) * 0.5);
    }
    if (j < ny - 1) {
        const double dP = (-P[c] + P[c + nx]) * _dy, dT = (-a.f.tyy[c] + a.f.tyy[c + nx]) * _dy;
        const double dS = (-TXY(i, j + 1) + TXY(i + 1, j + 1)) * _dx, av = (a.f.fy[c] + a.f.fy[c + nx]) * 0.5;
        if (RES_ONLY) a.f.Ry[c] = dT + dS - dP - av;
        else a.f.Vy[(i + 1) + (i64)(nx + 2) * (j + 1)] += (-dP + dT + dS - av) * edt / ((et[c] + et[c + nx]) * 0.5);
    }
#undef TXY
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
    return a;
}

jrx_status check2(jrx_handle *h, const jrx_stokes2d_fields *f, const jrx_stokes2d_params *p)
{
    if (!h) return JRX_ERR_ARG;
    if (!f || !p) return jrx_fail(h, JRX_ERR_ARG, "null fields/params");
    if (p->nx < 3 || p->ny < 3) return jrx_fail(h, JRX_ERR_ARG, "2D Stokes needs at least 3 cells per dimension");
    if ((double)(p->nx + 2) * (double)(p->ny + 2) >= 2147483647.0) return jrx_fail(h, JRX_ERR_UNSUPPORTED, "grid too large");
    const void *req[] = {f->P, f->P0, f->divV, f->Q, f->Vx, f->Vy, f->Ux, f->Uy, f->txx, f->tyy, f->txy, f->toxx, f->toyy, f->toxy,
                         f->exx, f->eyy, f->exy, f->eta, f->K, f->G, f->fx, f->fy, f->RP, f->Rx, f->Ry};
    for (const void *q : req)
        if (!q) return jrx_fail(h, JRX_ERR_ARG, "a required 2D field pointer is NULL");
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

jrx_status enqueue_iteration2(jrx_handle *h, const jrx_stokes2d_fields *f, const double *etatau, const jrx_stokes2d_params *p, bool diag)
{
    const int nx = (int)p->nx, ny = (int)p->ny;
    Args2 a = make_args2(f, etatau, p);
    hipStream_t s = h->stream;
    const unsigned gA = (unsigned)(((i64)(nx + 1) * (ny + 1) + 255) / 256), gB = (unsigned)(((i64)nx * ny + 255) / 256);
    if (diag) hipLaunchKernelGGL(k_stress2d<true>, dim3(gA), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_stress2d<false>, dim3(gA), dim3(256), 0, s, a);
    JRX_LAUNCH_CHECK(h);
    hipLaunchKernelGGL(k_velocity2d<false>, dim3(gB), dim3(256), 0, s, a);
    JRX_LAUNCH_CHECK(h);
    if (diag) {
        hipLaunchKernelGGL(k_scale3, dim3(256), dim3(256), 0, s, f->Ux, f->Vx, (i64)(nx + 1) * (ny + 2), f->Uy, f->Vy, (i64)(nx + 2) * (ny + 1),
                           (double *)nullptr, (const double *)nullptr, (i64)0, p->dt);
        JRX_LAUNCH_CHECK(h);
    }
    JRX_TRY(launch_bcs2(h, s, f->Vx, f->Vy, nx, ny, p->free_slip, p->no_slip, p->periodic));
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
    double err_it1 = 1.0, err = 1.0;
    int64_t iter = 0, cont = 0;
    const int rank = jrx_comm_rank(h);
    JRX_HIP(h, hipEventRecord(h->ev[6], s));
    auto keep_going = [&](int64_t it) { return it < 2 || (((err / err_it1) > p->eps_rel && err > p->eps_abs) && it <= p->iterMax); };
    Args2 a = make_args2(f, h->etatau, p);
    while (keep_going(iter)) {
        const int64_t it1 = iter + 1;
        const bool check = (it1 % p->nout == 0) && it1 > 1;
        const bool diag = check || !keep_going(it1);
        JRX_TRY(enqueue_iteration2(h, f, h->etatau, p, diag));
        iter = it1;
        if (check) {
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
    JRX_HIP(h, hipEventRecord(h->ev[7], s));
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
