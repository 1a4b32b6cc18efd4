// stokes3d.hip -- 3D isoviscous visco-elastic pseudo-transient Stokes path for gfx950.
//
// Reference being replaced (PTsolvers/JustRelax.jl v0.7.1): src/stokes/Stokes3D.jl:25-186 and the
// kernels it launches (VelocityKernels.jl:3-6,59-104,182-242; PressureKernels.jl:10-15,186-195;
// StressKernels.jl:2-5,149-230; types/displacement.jl:17-28; boundaryconditions/*.jl;
// Utils.jl:409-461,62-72).  The reference runs six separate launches per iteration; here the
// iteration is two fused sweeps:
//   stress sweep   (V -> ∇V, P, ε, τ)   reads V(3) P P0 Q η K G τ(6) τ_o(6), writes P τ(6)   = 28 passes
//   velocity sweep (τ, P -> R, V)        reads P τ(6) f(3) ητ V(3),           writes V(3)      = 17 passes
// i.e. 45 array passes * 8 B = 360 B per cell per iteration (the two-sweep floor); the diagnostic
// outputs ∇V, ε, RP, R, U are only written on iterations whose results can be observed (norm
// checks and the last iteration).  Pure HBM-bandwidth-bound fp64 stencils: no MFMA.
#include <vector>
#include <cstdlib>
#include "jrx_internal.hpp"
#include "jrx_kernels.hpp"
#include "stokes3d_kernels.hpp"
#include "jrx_tuning.h"

namespace {

SweepArgs make_args(const jrx_stokes3d_fields *f, const double *etatau, const jrx_stokes3d_params *p)
{
    SweepArgs a;
    a.f = *f;
    a.o = Out10{f->P, f->txx, f->tyy, f->tzz, f->tyz, f->txz, f->txy, f->Vx, f->Vy, f->Vz};
    a.etatau = etatau;
    a._dx = p->_dx; a._dy = p->_dy; a._dz = p->_dz;
    a.dt = p->dt; a.r = p->r; a.theta_dtau = p->theta_dtau; a.eta_dtau = p->eta_dtau;
    a.L = make_lay((int)p->nx, (int)p->ny, (int)p->nz);
    a.i0 = a.j0 = a.k0 = 0;
    a.i1 = a.j1 = a.k1 = 0;
    return a;
}

jrx_status check_params(jrx_handle *h, const jrx_stokes3d_fields *f, const jrx_stokes3d_params *p)
{
    if (!h) return JRX_ERR_ARG;
    if (!f || !p) return jrx_fail(h, JRX_ERR_ARG, "null fields/params");
    JRX_TRY(jrx_check_device(h));
    if (p->nx < 3 || p->ny < 3 || p->nz < 3) return jrx_fail(h, JRX_ERR_ARG, "3D Stokes needs at least 3 cells per dimension");
    const double cells = (double)(p->nx + 2) * (double)(p->ny + 2) * (double)(p->nz + 2);
    if (cells >= 2147483647.0) return jrx_fail(h, JRX_ERR_UNSUPPORTED, "local block too large for 32-bit plane indices");
    const void *req[] = {f->P, f->P0, f->Q, f->Vx, f->Vy, f->Vz, f->txx, f->tyy, f->tzz, f->tyz, f->txz, f->txy,
                         f->toxx, f->toyy, f->tozz, f->toyz, f->toxz, f->toxy, f->eta, f->K, f->G, f->fx, f->fy, f->fz};
    for (const void *q : req)
        if (!q) return jrx_fail(h, JRX_ERR_ARG, "a required field pointer is NULL");
    return JRX_OK;
}

jrx_status check_diag(jrx_handle *h, const jrx_stokes3d_fields *f)
{
    const void *req[] = {f->divV, f->exx, f->eyy, f->ezz, f->eyz, f->exz, f->exy, f->RP, f->Rx, f->Ry, f->Rz, f->Ux, f->Uy, f->Uz};
    for (const void *q : req)
        if (!q) return jrx_fail(h, JRX_ERR_ARG, "a diagnostic field pointer (∇V, ε, R, U) is NULL");
    return JRX_OK;
}

// launch the stress sweep over the sub-box [i0,i1) x [j0,j1) x [k0,k1) of the ni.+1 box
jrx_status launch_stress_v1(jrx_handle *h, hipStream_t s, SweepArgs a, bool diag, int i0, int i1, int j0, int j1, int k0, int k1)
{
    if (i1 <= i0 || j1 <= j0 || k1 <= k0) return JRX_OK;
    a.i0 = i0; a.i1 = i1; a.j0 = j0; a.j1 = j1; a.k0 = k0; a.k1 = k1;
    const i64 plane = (i64)(i1 - i0) * (j1 - j0);
    dim3 grid((unsigned)((plane + 255) / 256), (unsigned)(k1 - k0));
    if (diag) hipLaunchKernelGGL(k_stress3d<true>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_stress3d<false>, grid, dim3(256), 0, s, a);
    JRX_LAUNCH_CHECK(h);
    return JRX_OK;
}

jrx_status launch_velocity_v1(jrx_handle *h, hipStream_t s, SweepArgs a, bool diag, int i0, int i1, int j0, int j1, int k0, int k1)
{
    if (i1 <= i0 || j1 <= j0 || k1 <= k0) return JRX_OK;
    a.i0 = i0; a.i1 = i1; a.j0 = j0; a.j1 = j1; a.k0 = k0; a.k1 = k1;
    const i64 plane = (i64)(i1 - i0) * (j1 - j0);
    dim3 grid((unsigned)((plane + 255) / 256), (unsigned)(k1 - k0));
    if (diag) hipLaunchKernelGGL(k_velocity3d<true>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_velocity3d<false>, grid, dim3(256), 0, s, a);
    JRX_LAUNCH_CHECK(h);
    return JRX_OK;
}

// every array of the block addressable with a 32-bit byte offset (needed by the z-marching kernels)
bool fits_u32(const Lay3 &L)
{
    const double m = (double)(L.nx + 2) * (double)(L.ny + 2) * (double)(L.nz + 2) * 8.0;
    return m < 4294967296.0;
}

// Viscous-limit guard.  With dt = Inf the reference still multiplies τ_o, P0, Q by an exact 0 and divides by K dt, G dt (StressKernels.jl:2-5, PressureKernels.jl:186-195):
// a NaN / Inf in τ_o, P0 or Q, or a K or G that is NaN or 0 (0 * Inf), turns the result into NaN and the driver raises error("NaN(s)") (Stokes3D.jl:162).  The
// viscous-limit kernels never read those ten arrays, so before they are selected one streaming pass checks that every entry is harmless; if not, the general
// kernels run (and produce the reference's NaNs).  η must be finite, too: the viscous-limit form of the fused kernel folds η · 0 = 0 (k_fused3d, VFOLD).  sets h->visc_ok.
// (round 6: every array is streamed on its own with 16-byte loads -- 14 interleaved 8-byte streams per thread ran at 4.9 TB/s, 3.08 ms at 512^3)
template <class F>
__device__ __forceinline__ void scan_pairs(const double *__restrict__ p, i64 n, F f)
{
    if (n <= 0) return;
    const i64 stride = (i64)gridDim.x * blockDim.x, t0 = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 head = (reinterpret_cast<uintptr_t>(p) & 15) ? 1 : 0;            // any device pointer may be handed in: an array that starts 8 bytes off a 16-byte boundary gives up its first entry to one thread
    const double2 *__restrict__ q = reinterpret_cast<const double2 *>(p + head);
    const i64 m = n - head;
    for (i64 t = t0; t < (m >> 1); t += stride) { const double2 v = q[t]; f(v.x); f(v.y); }
    if (t0 == 0) { if (head) f(p[0]); if (m & 1) f(p[n - 1]); }
}
__global__ __launch_bounds__(256) void k_visc_operands_ok(const double *__restrict__ c0, const double *__restrict__ c1, const double *__restrict__ c2, const double *__restrict__ c3,
                                                          const double *__restrict__ c4, const double *__restrict__ eta, i64 nc, const double *__restrict__ K, const double *__restrict__ G,
                                                          const double *__restrict__ yz, i64 nyz, const double *__restrict__ xz, i64 nxz, const double *__restrict__ xy, i64 nxy, int *bad,
                                                          const unsigned long long *__restrict__ fx, const unsigned long long *__restrict__ fy, const unsigned long long *__restrict__ fz)
{
    bool b = false;
    unsigned long long bxy = 0, bz = 0;        // fx != nullptr: the same pass ORs the bits of the body forces (bad |= 2: ρg_x or ρg_y has an entry that is not +0.0, |= 4: ρg_z has)
    auto fin = [&](double v) { b |= !isfinite(v); };
    auto mod = [&](double v) { b |= (v != v) || v == 0.0; };
    scan_pairs(c0, nc, fin); scan_pairs(c1, nc, fin); scan_pairs(c2, nc, fin); scan_pairs(c3, nc, fin); scan_pairs(c4, nc, fin); scan_pairs(eta, nc, fin);
    scan_pairs(K, nc, mod); scan_pairs(G, nc, mod);
    scan_pairs(yz, nyz, fin); scan_pairs(xz, nxz, fin); scan_pairs(xy, nxy, fin);
    if (fx) {
        scan_pairs(reinterpret_cast<const double *>(fx), nc, [&](double v) { bxy |= (unsigned long long)__double_as_longlong(v); });
        scan_pairs(reinterpret_cast<const double *>(fy), nc, [&](double v) { bxy |= (unsigned long long)__double_as_longlong(v); });
        scan_pairs(reinterpret_cast<const double *>(fz), nc, [&](double v) { bz |= (unsigned long long)__double_as_longlong(v); });
    }
    if (__any(b) && (threadIdx.x & 63) == 0) atomicOr(bad, 1);
    if (fx) {
        const int m = (__any(bxy != 0) ? 2 : 0) | (__any(bz != 0) ? 4 : 0);
        if (m && (threadIdx.x & 63) == 0) atomicOr(bad, m);
    }
}

// finite dt: only the bits of the three body-force arrays are looked at (bad |= 2: ρg_x or ρg_y has an entry that is not +0.0, |= 4: ρg_z has)
__global__ __launch_bounds__(256) void k_forces_zero(const unsigned long long *__restrict__ fx, const unsigned long long *__restrict__ fy, const unsigned long long *__restrict__ fz, i64 nc, int *bad)
{
    const i64 stride = (i64)gridDim.x * blockDim.x;
    unsigned long long bxy = 0, bz = 0;
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < nc; t += stride) { bxy |= fx[t] | fy[t]; bz |= fz[t]; }
    const int m = (__any(bxy != 0) ? 2 : 0) | (__any(bz != 0) ? 4 : 0);
    if (m && (threadIdx.x & 63) == 0) atomicOr(bad, m);
}

// forces: also look at the body forces (the drivers whose fused kernels read them; the stand-alone stress sweep does not)
static jrx_status visc_operands_local(jrx_handle *h, const jrx_stokes3d_fields *f, const jrx_stokes3d_params *p, bool forces);
// ... and with neighbours every rank must pick the same kernel forms (the in-kernel neighbour faces exist for the viscous-limit form only: ranks that disagreed would run different
// pipelines inside one exchange): the verdicts are combined over the ranks -- the viscous-limit form only if every rank's operands allow it, body-force loads dropped only where every rank's are zero
jrx_status visc_operands_check(jrx_handle *h, const jrx_stokes3d_fields *f, const jrx_stokes3d_params *p, bool forces = true)
{
    JRX_TRY(visc_operands_local(h, f, p, forces));
    if (jrx_comm_active(h)) {
        double v[2] = {h->visc_ok ? 0.0 : 1.0, (double)(2 - h->nof)};
        JRX_TRY(jrx_allreduce_host(h, v, 2, 1));
        h->visc_ok = v[0] == 0.0;
        h->nof = 2 - (int)v[1];
    }
    return JRX_OK;
}
static jrx_status visc_operands_local(jrx_handle *h, const jrx_stokes3d_fields *f, const jrx_stokes3d_params *p, bool forces)
{
    h->visc_ok = false;
    h->nof = 0;
    forces = forces && h->zero_forces && f->fx && f->fy && f->fz;
    // option "operand_cache" = 1: the same operand arrays, extents and dt as at the last pass, and the caller has not declared them changed (jrx_fields_dirty): the verdict stands
    // (the pass streams up to 17 arrays and ends in a host synchronisation: 3.0 + 0.6 ms at 512^3 -- nothing inside a solve!, 3 % of a 20-iteration batch)
    const void *key[14] = {f->P0, f->Q, f->toxx, f->toyy, f->tozz, f->toyz, f->toxz, f->toxy, f->eta, f->K, f->G, f->fx, f->fy, f->fz};
    const int kflags = (forces ? 1 : 0) | (h->viscous_limit ? 2 : 0) | (p->displacement_bcs ? 4 : 0) | (h->kernel_variant << 3);
    if (h->operand_cache && h->opv.valid && h->opv.dt == p->dt && h->opv.flags == kflags && h->opv.n[0] == p->nx && h->opv.n[1] == p->ny && h->opv.n[2] == p->nz &&
        memcmp(h->opv.ptr, key, sizeof(key)) == 0) {
        h->visc_ok = h->opv.visc_ok;
        h->nof = h->opv.nof;
        h->stat_operand_cache_hits++;
        return JRX_OK;
    }
    struct Remember {      // whatever way the pass ends, its verdict is what the next call may reuse
        jrx_handle *h; const void *const *key; const jrx_stokes3d_params *p; int kflags; bool ok = false;
        ~Remember() {
            h->opv.valid = ok;
            if (!ok) return;
            memcpy(h->opv.ptr, key, sizeof(h->opv.ptr));
            h->opv.n[0] = p->nx; h->opv.n[1] = p->ny; h->opv.n[2] = p->nz; h->opv.dt = p->dt; h->opv.flags = kflags; h->opv.visc_ok = h->visc_ok; h->opv.nof = h->nof;
        }
    } remember{h, key, p, kflags};
    if ((!h->viscous_limit || p->dt != INFINITY) && forces && !p->displacement_bcs && (h->kernel_variant == 0 || h->kernel_variant == 3) && p->nx >= 48) {
        // the general form of the fused kernel has the instantiations without body-force loads, too (every 3D model of the reference has gravity along z: ρg_x = ρg_y = 0)
        const i64 nc = (i64)p->nx * p->ny * p->nz;
        int *d_bad = reinterpret_cast<int *>(h->d_sums + 6), *h_bad = reinterpret_cast<int *>(h->h_sums + 6);
        hipStream_t s = h->stream;
        JRX_HIP(h, hipMemsetAsync(d_bad, 0, sizeof(double), s));
        hipLaunchKernelGGL(k_forces_zero, dim3(2048), dim3(256), 0, s, reinterpret_cast<const unsigned long long *>(f->fx), reinterpret_cast<const unsigned long long *>(f->fy),
                           reinterpret_cast<const unsigned long long *>(f->fz), nc, d_bad);
        JRX_LAUNCH_CHECK(h);
        JRX_HIP(h, hipMemcpyAsync(h_bad, d_bad, sizeof(double), hipMemcpyDeviceToHost, s));
        JRX_HIP(h, hipStreamSynchronize(s));
        if (!(*h_bad & 2)) h->nof = (*h_bad & 4) ? 1 : 2;
        remember.ok = true;
        return JRX_OK;
    }
    if (!h->viscous_limit || p->dt != INFINITY) { remember.ok = true; return JRX_OK; }
    const i64 nx = p->nx, ny = p->ny, nz = p->nz;
    int *d_bad = reinterpret_cast<int *>(h->d_sums + 6), *h_bad = reinterpret_cast<int *>(h->h_sums + 6);
    hipStream_t s = h->stream;
    JRX_HIP(h, hipMemsetAsync(d_bad, 0, sizeof(double), s));
    hipLaunchKernelGGL(k_visc_operands_ok, dim3(4096), dim3(256), 0, s, f->P0, f->Q, f->toxx, f->toyy, f->tozz, f->eta, nx * ny * nz, f->K, f->G, f->toyz, nx * (ny + 1) * (nz + 1),
                       f->toxz, (nx + 1) * ny * (nz + 1), f->toxy, (nx + 1) * (ny + 1) * nz, d_bad,
                       forces ? reinterpret_cast<const unsigned long long *>(f->fx) : nullptr, reinterpret_cast<const unsigned long long *>(f->fy), reinterpret_cast<const unsigned long long *>(f->fz));
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipMemcpyAsync(h_bad, d_bad, sizeof(double), hipMemcpyDeviceToHost, s));
    JRX_HIP(h, hipStreamSynchronize(s));
    h->visc_ok = ((*h_bad & 1) == 0);
    // body forces that are +0.0 throughout need not be streamed (k_fused3d, NOF): the usual 3D model has gravity along z only, SolVi3D has none
    if (h->visc_ok && forces && !(*h_bad & 2)) h->nof = (*h_bad & 4) ? 1 : 2;
    h->stat_visc_checks++;
    if (!h->visc_ok) h->stat_visc_fallbacks++;
    remember.ok = true;
    return JRX_OK;
}

template <int TX, int TY, int KZ>
jrx_status launch_stress_zb(jrx_handle *h, hipStream_t s, const SweepArgs &a, bool diag)
{
    const TileMap tm = make_tilemap(a.L.nx, a.L.ny, a.L.nz, TX, TY, KZ);
    if (diag) hipLaunchKernelGGL((k_stress3d_zb<true, TX, TY, KZ, 4, false, 8, true>), dim3(tm.ntiles), dim3(TX * TY), 0, s, a, tm);
    else if (h->viscous_limit && h->visc_ok && a.dt == INFINITY)      // dt = Inf: τ_o, P0, K, G, Q only meet factors that are exactly 0 and are not loaded (see k_fused3d)
        hipLaunchKernelGGL((k_stress3d_zb<false, TX, TY, KZ, 4, false, 8, true, true>), dim3(tm.ntiles), dim3(TX * TY), 0, s, a, tm);
    else hipLaunchKernelGGL((k_stress3d_zb<false, TX, TY, KZ, 4, false, 8, true>), dim3(tm.ntiles), dim3(TX * TY), 0, s, a, tm);
    JRX_LAUNCH_CHECK(h);
    return JRX_OK;
}

template <int TX, int TY, int KZ>
jrx_status launch_velocity_zb(jrx_handle *h, hipStream_t s, const SweepArgs &a, bool diag, int nof = 0)
{
    // nof: body-force arrays the caller has just seen to hold only +0.0 (1: fx, fy; 2: all three) are not loaded by the unobserved form
    const TileMap tm = make_tilemap(a.i1 - a.i0, a.j1 - a.j0, a.k1 - a.k0, TX, TY, KZ);
    if (!diag && nof == 2) hipLaunchKernelGGL((k_velocity3d_zb<false, TX, TY, KZ, 4, 8, true, 2>), dim3(tm.ntiles), dim3(TX * TY), 0, s, a, tm);
    else if (!diag && nof == 1) hipLaunchKernelGGL((k_velocity3d_zb<false, TX, TY, KZ, 4, 8, true, 1>), dim3(tm.ntiles), dim3(TX * TY), 0, s, a, tm);
    else if (diag) hipLaunchKernelGGL((k_velocity3d_zb<true, TX, TY, KZ, 4, 8, true>), dim3(tm.ntiles), dim3(TX * TY), 0, s, a, tm);
    else hipLaunchKernelGGL((k_velocity3d_zb<false, TX, TY, KZ, 4, 8, true>), dim3(tm.ntiles), dim3(TX * TY), 0, s, a, tm);
    JRX_LAUNCH_CHECK(h);
    return JRX_OK;
}

static jrx_status launch_stress_boxes(jrx_handle *h, hipStream_t s, const SweepArgs &a, const int (*box)[6], int nbox, bool diag,
                                      const GhostRule *rule = nullptr);

// Stress sweep over the whole ni.+1 node box.  Wide grids: z-marching kernel over the cell box
// (tile = one or two full-width row segments: measured best for DRAM page locality)
// plus three thin launches of the per-node kernel for the upper boundary planes i = nx, j = ny,
// k = nz that no cell column owns.
jrx_status launch_stress(jrx_handle *h, hipStream_t s, SweepArgs a, bool diag, int i0, int i1, int j0, int j1, int k0, int k1)
{
    const int nx = a.L.nx, ny = a.L.ny, nz = a.L.nz;
    const bool full = i0 == 0 && j0 == 0 && k0 == 0 && i1 == nx + 1 && j1 == ny + 1 && k1 == nz + 1;
    // small blocks: the z-marching kernel would have a few hundred blocks, each a chain of dependent planes; one node per thread is faster up to ~88^3 (two sweeps per iteration:
    // 48^3 28.9 k against 19.5 k it/s, 64^3 20.5 k / 17.5 k, 80^3 14.4 k / 13.1 k; 96^3 10.0 k / 11.2 k, 128^3 5.4 k / 8.0 k)
    const bool small = (double)nx * ny * nz <= 681472.0;
    if (h->kernel_variant == 1 || !full || !fits_u32(a.L) || nx < 48 || small) return launch_stress_v1(h, s, a, diag, i0, i1, j0, j1, k0, k1);
    a.i0 = a.j0 = a.k0 = 0; a.i1 = nx; a.j1 = ny; a.k1 = nz;
    if (nx > 384) JRX_TRY((launch_stress_zb<512, 1, 4>(h, s, a, diag)));
    else if (nx > 192) JRX_TRY((launch_stress_zb<256, 1, 8>(h, s, a, diag)));
    else if (nx > 96) JRX_TRY((launch_stress_zb<128, 2, 8>(h, s, a, diag)));
    else JRX_TRY((launch_stress_zb<64, 4, 8>(h, s, a, diag)));
    // the upper boundary planes i = nx, j = ny, k = nz in one launch of the per-node kernel over three disjoint boxes
    const int planes[3][6] = {{nx, nx + 1, 0, ny + 1, 0, nz + 1}, {0, nx, ny, ny + 1, 0, nz + 1}, {0, nx, 0, ny, nz, nz + 1}};
    return launch_stress_boxes(h, s, a, planes, 3, diag);
}

jrx_status launch_velocity(jrx_handle *h, hipStream_t s, SweepArgs a, bool diag, int i0, int i1, int j0, int j1, int k0, int k1, int nof = 0)
{
    if (i1 <= i0 || j1 <= j0 || k1 <= k0) return JRX_OK;
    const int w = i1 - i0;
    const bool small = (double)w * (j1 - j0) * (k1 - k0) <= 681472.0;          // see launch_stress
    if (h->kernel_variant == 1 || !fits_u32(a.L) || w < 48 || (k1 - k0) < 4 || small) return launch_velocity_v1(h, s, a, diag, i0, i1, j0, j1, k0, k1);
    a.i0 = i0; a.i1 = i1; a.j0 = j0; a.j1 = j1; a.k0 = k0; a.k1 = k1;
    if (w > 384) return launch_velocity_zb<512, 1, 4>(h, s, a, diag, nof);
    if (w > 192) return launch_velocity_zb<256, 1, 8>(h, s, a, diag, nof);
    if (w > 96) return launch_velocity_zb<128, 2, 8>(h, s, a, diag, nof);
    return launch_velocity_zb<64, 4, 8>(h, s, a, diag, nof);
}

jrx_status launch_scaleU(jrx_handle *h, hipStream_t s, const jrx_stokes3d_fields *f, const jrx_stokes3d_params *p)
{
    const i64 n0 = (i64)(p->nx + 1) * (p->ny + 2) * (p->nz + 2), n1 = (i64)(p->nx + 2) * (p->ny + 1) * (p->nz + 2),
              n2 = (i64)(p->nx + 2) * (p->ny + 2) * (p->nz + 1);
    hipLaunchKernelGGL(k_scale3, dim3(2048), dim3(256), 0, s, f->Ux, f->Vx, n0, f->Uy, f->Vy, n1, f->Uz, f->Vz, n2, p->dt);
    JRX_LAUNCH_CHECK(h);
    return JRX_OK;
}

jrx_status launch_bcs(jrx_handle *h, hipStream_t s, double *Vx, double *Vy, double *Vz, int nx, int ny, int nz,
                      uint32_t fs, uint32_t ns, uint32_t pe)
{
    BcArr A[3] = {{Vx, {nx + 1, ny + 2, nz + 2}}, {Vy, {nx + 2, ny + 1, nz + 2}}, {Vz, {nx + 2, ny + 2, nz + 1}}};
    auto run = [&](int type, int dim, bool lo, bool hi) -> jrx_status {
        if (!lo && !hi) return JRX_OK;
        const int d1 = dim == 0 ? 1 : 0, d2 = dim == 2 ? 1 : 2;
        int na = 0, nb = 0;
        for (int c = 0; c < 3; c++) { na = A[c].n[d1] > na ? A[c].n[d1] : na; nb = A[c].n[d2] > nb ? A[c].n[d2] : nb; }
        hipLaunchKernelGGL(k_bc3d, dim3((na + 255) / 256, nb), dim3(256), 0, s, A[0], A[1], A[2], type, dim, (int)lo, (int)hi);
        JRX_LAUNCH_CHECK(h);
        return JRX_OK;
    };
    if (ns) {   // no_slip.jl:20-54 : left,right ; front,back ; bot (k=1), top (k=end)
        JRX_TRY(run(1, 0, ns & JRX_FACE_LEFT, ns & JRX_FACE_RIGHT));
        JRX_TRY(run(1, 1, ns & JRX_FACE_FRONT, ns & JRX_FACE_BACK));
        JRX_TRY(run(1, 2, ns & JRX_FACE_BOT, ns & JRX_FACE_TOP));
    }
    if (fs) {   // free_slip.jl:15-70 : front,back ; top (k=1), bot (k=end) ; left,right
        JRX_TRY(run(0, 1, fs & JRX_FACE_FRONT, fs & JRX_FACE_BACK));
        JRX_TRY(run(0, 2, fs & JRX_FACE_TOP, fs & JRX_FACE_BOT));
        JRX_TRY(run(0, 0, fs & JRX_FACE_LEFT, fs & JRX_FACE_RIGHT));
    }
    if (pe) {   // periodic.jl:56-98 : left,right ; front,back ; bot (k=1), top (k=end)
        JRX_TRY(run(2, 0, pe & JRX_FACE_LEFT, pe & JRX_FACE_RIGHT));
        JRX_TRY(run(2, 1, pe & JRX_FACE_FRONT, pe & JRX_FACE_BACK));
        JRX_TRY(run(2, 2, pe & JRX_FACE_BOT, pe & JRX_FACE_TOP));
    }
    return JRX_OK;
}

// free_slip / no_slip of all faces in one launch: equal to launch_bcs on every entry a Stokes stencil reads once the ordered passes
// have run at least once on the same arrays (normal planes of no-slip faces are then already zero); see k_bc3d_faces
static jrx_status launch_bcs_faces(jrx_handle *h, hipStream_t s, double *Vx, double *Vy, double *Vz, int nx, int ny, int nz, uint32_t fs, uint32_t ns)
{
    BcArr A[3] = {{Vx, {nx + 1, ny + 2, nz + 2}}, {Vy, {nx + 2, ny + 1, nz + 2}}, {Vz, {nx + 2, ny + 2, nz + 1}}};
    // 3D naming of the reference: free_slip `top` <-> k = 1, no_slip `bot` <-> k = 1 (App. C.4)
    auto ty = [&](uint32_t fsbit, uint32_t nsbit) { return (ns & nsbit) ? 2 : ((fs & fsbit) ? 1 : 0); };
    const int t[6] = {ty(JRX_FACE_LEFT, JRX_FACE_LEFT), ty(JRX_FACE_RIGHT, JRX_FACE_RIGHT), ty(JRX_FACE_FRONT, JRX_FACE_FRONT),
                      ty(JRX_FACE_BACK, JRX_FACE_BACK), ty(JRX_FACE_TOP, JRX_FACE_BOT), ty(JRX_FACE_BOT, JRX_FACE_TOP)};
    if (!(t[0] | t[1] | t[2] | t[3] | t[4] | t[5])) return JRX_OK;
    const int na = (nx > ny ? nx : ny) + 2, nb = (ny > nz ? ny : nz) + 2;      // in-plane extents never exceed these
    hipLaunchKernelGGL(k_bc3d_faces, dim3((na + 255) / 256, nb, 3), dim3(256), 0, s, A[0], A[1], A[2], t[0], t[1], t[2], t[3], t[4], t[5]);
    JRX_LAUNCH_CHECK(h);
    return JRX_OK;
}

// the stress sweep over up to six disjoint node boxes in one launch
// rule != nullptr: the boundary entries of V are derived by the flow_bcs! rules instead of being read (no flow_bcs! launch needed before)
static jrx_status launch_stress_boxes(jrx_handle *h, hipStream_t s, const SweepArgs &a, const int (*box)[6], int nbox, bool diag, const GhostRule *rule)
{
    StressBoxes B = {};
    int tot = 0;
    for (int q = 0; q < nbox; q++) {
        const int *b = box[q];
        if (b[1] <= b[0] || b[3] <= b[2] || b[5] <= b[4]) continue;
        const i64 plane = (i64)(b[1] - b[0]) * (b[3] - b[2]);
        for (int c = 0; c < 6; c++) B.box[B.n][c] = b[c];
        B.per_plane[B.n] = (int)((plane + 255) / 256);
        B.start[B.n] = tot;
        tot += B.per_plane[B.n] * (b[5] - b[4]);
        B.n++;
    }
    if (!B.n) return JRX_OK;
    B.start[B.n] = tot;
    const GhostRule none = {{0, 0, 0, 0, 0, 0}};
    const bool visc = !diag && h->viscous_limit && h->visc_ok && a.dt == INFINITY;
    if (rule && visc) hipLaunchKernelGGL((k_stress3d_boxes<false, true, true>), dim3((unsigned)tot), dim3(256), 0, s, a, B, *rule);
    else if (rule) hipLaunchKernelGGL((k_stress3d_boxes<false, true>), dim3((unsigned)tot), dim3(256), 0, s, a, B, *rule);
    else if (visc) hipLaunchKernelGGL((k_stress3d_boxes<false, false, true>), dim3((unsigned)tot), dim3(256), 0, s, a, B, none);
    else if (diag) hipLaunchKernelGGL(k_stress3d_boxes<true>, dim3((unsigned)tot), dim3(256), 0, s, a, B, none);
    else hipLaunchKernelGGL(k_stress3d_boxes<false>, dim3((unsigned)tot), dim3(256), 0, s, a, B, none);
    JRX_LAUNCH_CHECK(h);
    return JRX_OK;
}

jrx_status launch_sumsq(jrx_handle *h, hipStream_t s, const jrx_stokes3d_fields *f, const jrx_stokes3d_params *p)
{
    const int nx = (int)p->nx, ny = (int)p->ny, nz = (int)p->nz;
    RedArr A0 = {f->Rx, {nx - 1, ny, nz}, 1}, A1 = {f->Ry, {nx, ny - 1, nz}, 1}, A2 = {f->Rz, {nx, ny, nz - 1}, 1},
           A3 = {f->RP, {nx, ny, nz}, 0};
    const i64 n = (i64)nx * ny * nz;
    int nb = (int)((n + 256 * 8 - 1) / (256 * 8));
    nb = nb < 1 ? 1 : (nb > kMaxRedBlocks ? kMaxRedBlocks : nb);
    hipLaunchKernelGGL(k_sumsq_partial, dim3(nb), dim3(256), 0, s, A0, A1, A2, A3, h->d_partials);
    JRX_LAUNCH_CHECK(h);
    hipLaunchKernelGGL(k_sumsq_final, dim3(1), dim3(256), 0, s, h->d_partials, nb, h->d_sums);
    JRX_LAUNCH_CHECK(h);
    return JRX_OK;
}

}   // namespace

jrx_status jrx3d_velocity_sweep(jrx_handle *h, hipStream_t s, const jrx_stokes3d_fields *f, const double *etatau, const jrx_stokes3d_params *p, bool diag, int nof)
{
    if (diag && (!f->Rx || !f->Ry || !f->Rz)) return jrx_fail(h, JRX_ERR_ARG, "residual arrays are NULL");
    return launch_velocity(h, s, make_args(f, etatau, p), diag, 0, (int)p->nx, 0, (int)p->ny, 0, (int)p->nz, nof);
}
// which body-force arrays hold nothing but +0.0 right now: 0 none (or the switch zero_forces is off), 1 fx and fy, 2 all three.  One streaming pass and a host synchronisation
jrx_status jrx3d_forces_zero(jrx_handle *h, hipStream_t s, const double *fx, const double *fy, const double *fz, int64_t nc, int *nof)
{
    *nof = 0;
    if (!h->zero_forces || !fx || !fy || !fz) return JRX_OK;
    int *d_bad = reinterpret_cast<int *>(h->d_sums + 6), *h_bad = reinterpret_cast<int *>(h->h_sums + 6);
    JRX_HIP(h, hipMemsetAsync(d_bad, 0, sizeof(double), s));
    hipLaunchKernelGGL(k_forces_zero, dim3(2048), dim3(256), 0, s, reinterpret_cast<const unsigned long long *>(fx), reinterpret_cast<const unsigned long long *>(fy),
                       reinterpret_cast<const unsigned long long *>(fz), (i64)nc, d_bad);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipMemcpyAsync(h_bad, d_bad, sizeof(double), hipMemcpyDeviceToHost, s));
    JRX_HIP(h, hipStreamSynchronize(s));
    if (!(*h_bad & 2)) *nof = (*h_bad & 4) ? 1 : 2;
    return JRX_OK;
}
jrx_status jrx3d_scaleU(jrx_handle *h, hipStream_t s, const jrx_stokes3d_fields *f, const jrx_stokes3d_params *p) { return launch_scaleU(h, s, f, p); }
jrx_status jrx3d_bcs(jrx_handle *h, hipStream_t s, double *Vx, double *Vy, double *Vz, int nx, int ny, int nz, uint32_t fs, uint32_t ns, uint32_t pe)
{
    return launch_bcs(h, s, Vx, Vy, Vz, nx, ny, nz, fs, ns, pe);
}
jrx_status jrx3d_sumsq(jrx_handle *h, hipStream_t s, const jrx_stokes3d_fields *f, const jrx_stokes3d_params *p) { return launch_sumsq(h, s, f, p); }
jrx_status jrx3d_bcs_faces(jrx_handle *h, hipStream_t s, double *Vx, double *Vy, double *Vz, int nx, int ny, int nz, uint32_t fs, uint32_t ns)
{
    return launch_bcs_faces(h, s, Vx, Vy, Vz, nx, ny, nz, fs, ns);
}

// @hide_communication b_width (Stokes3D.jl:104-121, 582-597): compute_V! over the six boundary slabs of width b first, on the halo stream, followed
// there by velocity2displacement! (observable iterations), flow_bcs! and update_halo!(V), while the interior runs on the compute stream; the two
// streams are joined on return.  bc_kind: 0 = flow_bcs! on V in the reference's pass order, 1 = all faces of V in one launch (equal wherever a stencil
// reads once the ordered passes have run), 2 = flow_bcs! on U (DisplacementBoundaryConditions, observable iterations), 3 = none.
jrx_status jrx3d_velocity_hidden(jrx_handle *h, const jrx_stokes3d_fields *f, const double *etatau, const jrx_stokes3d_params *p, bool diag, int bc_kind)
{
    if (diag && (!f->Rx || !f->Ry || !f->Rz)) return jrx_fail(h, JRX_ERR_ARG, "residual arrays are NULL");
    const int nx = (int)p->nx, ny = (int)p->ny, nz = (int)p->nz;
    hipStream_t s = h->stream;
    const SweepArgs a = make_args(f, etatau, p);
    int bx = p->b_width[0] > 0 ? p->b_width[0] : 4, by = p->b_width[1] > 0 ? p->b_width[1] : 4,
        bz = p->b_width[2] > 0 ? p->b_width[2] : 4;
    if (h->b_width_opt[0] > 0) bx = h->b_width_opt[0];       // tuning switches "b_width_x/y/z" (the split does not change results)
    if (h->b_width_opt[1] > 0) by = h->b_width_opt[1];
    if (h->b_width_opt[2] > 0) bz = h->b_width_opt[2];
    const int xa = bx < nx / 2 ? bx : nx / 2, ya = by < ny / 2 ? by : ny / 2, za = bz < nz / 2 ? bz : nz / 2;
    hipStream_t hs = h->halo_stream;
    JRX_HIP(h, hipEventRecord(h->ev[0], s));
    JRX_HIP(h, hipStreamWaitEvent(hs, h->ev[0], 0));
    // six slabs (z-lo, z-hi, y-lo, y-hi, x-lo, x-hi), disjoint
    JRX_TRY(launch_velocity(h, hs, a, diag, 0, nx, 0, ny, 0, za));
    JRX_TRY(launch_velocity(h, hs, a, diag, 0, nx, 0, ny, nz - za, nz));
    JRX_TRY(launch_velocity(h, hs, a, diag, 0, nx, 0, ya, za, nz - za));
    JRX_TRY(launch_velocity(h, hs, a, diag, 0, nx, ny - ya, ny, za, nz - za));
    JRX_TRY(launch_velocity(h, hs, a, diag, 0, xa, ya, ny - ya, za, nz - za));
    JRX_TRY(launch_velocity(h, hs, a, diag, nx - xa, nx, ya, ny - ya, za, nz - za));
    // interior on the compute stream, concurrently
    JRX_TRY(launch_velocity(h, s, a, diag, xa, nx - xa, ya, ny - ya, za, nz - za));
    if (diag) {
        // U = V*dt needs the whole updated V: join first
        JRX_HIP(h, hipEventRecord(h->ev[1], s));
        JRX_HIP(h, hipStreamWaitEvent(hs, h->ev[1], 0));
        JRX_TRY(launch_scaleU(h, hs, f, p));
    }
    if (bc_kind == 0) JRX_TRY(launch_bcs(h, hs, f->Vx, f->Vy, f->Vz, nx, ny, nz, p->free_slip, p->no_slip, p->periodic));
    else if (bc_kind == 1) JRX_TRY(launch_bcs_faces(h, hs, f->Vx, f->Vy, f->Vz, nx, ny, nz, p->free_slip, p->no_slip));
    else if (bc_kind == 2) JRX_TRY(launch_bcs(h, hs, f->Ux, f->Uy, f->Uz, nx, ny, nz, p->free_slip, p->no_slip, p->periodic));
    double *arrs[3] = {f->Vx, f->Vy, f->Vz};
    const int64_t ext[3][3] = {{nx + 1, ny + 2, nz + 2}, {nx + 2, ny + 1, nz + 2}, {nx + 2, ny + 2, nz + 1}};
    const int64_t n[3] = {nx, ny, nz};
    JRX_TRY(jrx_halo_exchange(h, hs, 3, arrs, ext, n));
    JRX_HIP(h, hipEventRecord(h->ev[2], hs));
    JRX_HIP(h, hipStreamWaitEvent(s, h->ev[2], 0));
    return JRX_OK;
}

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

jrx_status jrx_stokes3d_sweep_stress(jrx_handle *h, const jrx_stokes3d_fields *f, const jrx_stokes3d_params *p, int32_t flags)
{
    JRX_TRY(check_params(h, f, p));
    const bool diag = flags & JRX_OUT_DIAG;
    if (diag) JRX_TRY(check_diag(h, f));
    SweepArgs a = make_args(f, nullptr, p);
    // one sweep: a pass over the ten operand arrays to find out whether they may stay unread costs more than reading them -- the general kernel runs, unless a verdict is
    // already at hand (option "operand_cache")
    if (h->operand_cache) JRX_TRY(visc_operands_check(h, f, p, false));
    else { h->visc_ok = false; h->nof = 0; }
    JRX_TRY(launch_stress(h, h->stream, a, diag, 0, (int)p->nx + 1, 0, (int)p->ny + 1, 0, (int)p->nz + 1));
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_stokes3d_sweep_velocity(jrx_handle *h, const jrx_stokes3d_fields *f, const double *etatau,
                                       const jrx_stokes3d_params *p, int32_t flags)
{
    JRX_TRY(check_params(h, f, p));
    if (!etatau) return jrx_fail(h, JRX_ERR_ARG, "etatau is NULL");
    const bool diag = flags & JRX_OUT_DIAG;
    if (diag) JRX_TRY(check_diag(h, f));
    SweepArgs a = make_args(f, etatau, p);
    JRX_TRY(launch_velocity(h, h->stream, a, diag, 0, (int)p->nx, 0, (int)p->ny, 0, (int)p->nz));
    if (diag) JRX_TRY(launch_scaleU(h, h->stream, f, p));
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_flow_bcs3d(jrx_handle *h, double *Vx, double *Vy, double *Vz, int64_t nx, int64_t ny, int64_t nz,
                          uint32_t free_slip, uint32_t no_slip, uint32_t periodic)
{
    if (!h) return JRX_ERR_ARG;
    if (!Vx || !Vy || !Vz) return jrx_fail(h, JRX_ERR_ARG, "null velocity pointer");
    JRX_TRY(launch_bcs(h, h->stream, Vx, Vy, Vz, (int)nx, (int)ny, (int)nz, free_slip, no_slip, periodic));
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

jrx_status jrx_stokes3d_residual_sumsq(jrx_handle *h, const jrx_stokes3d_fields *f, const jrx_stokes3d_params *p, double out[4])
{
    JRX_TRY(check_params(h, f, p));
    JRX_TRY(check_diag(h, f));
    JRX_TRY(launch_sumsq(h, h->stream, f, p));
    JRX_HIP(h, hipMemcpyAsync(h->h_sums, h->d_sums, 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    for (int c = 0; c < 4; c++) out[c] = h->h_sums[c];
    return JRX_OK;
}

// velocity2displacement!(stokes, dt): U = V * dt ; displacement2velocity!(stokes, dt): V = U * inv(dt)   (types/displacement.jl:2-60)
jrx_status jrx_velocity2displacement(jrx_handle *h, double *const U[3], const double *const V[3], const int64_t n[3], double dt)
{
    if (!h) return JRX_ERR_ARG;
    if (!U || !V || !n || !U[0] || !U[1] || !V[0] || !V[1]) return jrx_fail(h, JRX_ERR_ARG, "velocity2displacement!: null argument");
    hipLaunchKernelGGL(k_scale3, dim3(2048), dim3(256), 0, h->stream, U[0], V[0], (i64)n[0], U[1], V[1], (i64)n[1], U[2], V[2], (i64)(U[2] ? n[2] : 0), dt);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}
jrx_status jrx_displacement2velocity(jrx_handle *h, double *const V[3], const double *const U[3], const int64_t n[3], double dt)
{
    return jrx_velocity2displacement(h, V, U, n, 1.0 / dt);
}

// compute_dt(stokes, di[, dt_diff][, igg]) = min(dt_diff, 0.9 * min_d(di[d] * inv(max|V_d|)))   (Utils.jl:492-519); the maximum is
// all-reduced over the ranks when a communicator is active (maximum_mpi)
jrx_status jrx_compute_dt(jrx_handle *h, const double *const V[3], const int64_t n[3], const double di[3], int32_t ndim, double dt_diff, double *dt_out)
{
    if (!h) return JRX_ERR_ARG;
    if (!V || !n || !di || !dt_out || ndim < 2 || ndim > 3 || !V[0] || !V[1] || (ndim == 3 && !V[2])) return jrx_fail(h, JRX_ERR_ARG, "compute_dt: bad argument");
    i64 nmax = 0;
    for (int d = 0; d < ndim; d++) nmax = n[d] > nmax ? (i64)n[d] : nmax;
    int nb = (int)((nmax + 2047) / 2048);
    nb = nb < 1 ? 1 : (nb > kMaxRedBlocks ? kMaxRedBlocks : nb);
    hipStream_t s = h->stream;
    hipLaunchKernelGGL(k_maxabs_partial, dim3(nb), dim3(256), 0, s, V[0], (i64)n[0], V[1], (i64)n[1], ndim == 3 ? V[2] : (const double *)nullptr,
                       (i64)(ndim == 3 ? n[2] : 0), h->d_partials);
    hipLaunchKernelGGL(k_maxabs_final, dim3(1), dim3(256), 0, s, h->d_partials, nb, h->d_sums);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipMemcpyAsync(h->h_sums, h->d_sums, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
    JRX_HIP(h, hipStreamSynchronize(s));
    double m[3] = {h->h_sums[0], h->h_sums[1], h->h_sums[2]};
    JRX_TRY(jrx_allreduce_host(h, m, ndim, 1));
    double dt_adv = INFINITY;
    for (int d = 0; d < ndim; d++) dt_adv = fmin(dt_adv, di[d] * (1.0 / m[d]));
    *dt_out = fmin(dt_diff, dt_adv * 0.9);
    return JRX_OK;
}

jrx_status jrx_compute_maxloc(jrx_handle *h, double *B, const double *A, int64_t nx, int64_t ny, int64_t nz)
{
    if (!h) return JRX_ERR_ARG;
    if (!A || !B || nx < 1 || ny < 1 || nz < 1) return jrx_fail(h, JRX_ERR_ARG, "bad compute_maxloc arguments");
    if (A == B) return jrx_fail(h, JRX_ERR_ARG, "compute_maxloc!: B must not alias A");
    const i64 plane = (i64)nx * ny;
    hipLaunchKernelGGL(k_maxloc, dim3((unsigned)((plane + 255) / 256), (unsigned)nz), dim3(256), 0, h->stream, B, A, (int)nx, (int)ny, (int)nz);
    JRX_LAUNCH_CHECK(h);
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

}   // extern "C"

// Tile shape of the fused kernel (lane-shuffle form): TX - 2 stress columns per TX-lane tile row (one halo lane on the left, one feeder
// lane on the right), TY - 1 stress rows, KZ planes per chunk.  Option "fused_tile": 0 = 64 x 4 (a row per wave), 1 = 32 x 8 (two rows per wave).
struct FusedShape { int tx, ty, kz; };
static void fused_tiles(const Lay3 &L, const FusedShape S, int nt[3])
{
    nt[0] = (L.nx + S.tx - 3) / (S.tx - 2); nt[1] = (L.ny + S.ty - 2) / (S.ty - 1); nt[2] = (L.nz + S.kz - 1) / S.kz;
}
// Chunk depth: 8 planes where that gives the chip enough blocks; on small grids a block's chain of dependent planes is what takes the time (64^3 in 8-plane chunks is
// 352 blocks on 256 CUs: 33 us per launch, 24 us in 4-plane chunks; 48^3: 26 -> 12 us in 2-plane chunks; from 96^3 on 8 planes are best; scripts/kbench_visc.hip,
// profiles/r03_small_grid_chunks.txt), so the depth is halved while the launch has fewer than 512 blocks
static FusedShape fused_shape(const jrx_handle *h, const Lay3 &L, double dt)
{
    // tile: 64 x 4 threads (a row per wave); 32 x 8 (two rows per wave, 30 stress columns per tile) costs ~20 % more per lane but quantises nx in steps of 30:
    // it is the better shape where three 32-lane tiles replace two 64-lane ones, nx = 63 .. 90 (64^3: 25.4 k -> 30.4 k it/s, 80^3: 15.9 k -> 22.1 k, 90^3: 14.6 k -> 17.8 k;
    // from nx = 91 on -- four 32-lane tiles -- and on every larger grid 64 x 4 is faster again: 96^3 14.3 k against 13.1 k, 130^3 6.9 k against 6.4 k, 192^3 2.6 k against 2.2 k).
    // tuning switch "fused_tile": 2 = this rule (default), 0 / 1 force a shape
    const bool narrow = h->fused_tile == 1 || (h->fused_tile == 2 && L.nx > 62 && L.nx <= 90);
    FusedShape S = narrow ? FusedShape{32, 8, 8} : FusedShape{64, 4, 8};
    int nt[3];
    // round 5: 64 x 8 threads (seven stress rows per tile instead of three: the y halo costs 8/7 instead of 4/3 of the velocity-phase operands; two 8-wave blocks per CU, tile rows
    // dealt to the XCDs in bands of four) where the launch still has >= 4,096 blocks: 512^3 -2.5 .. -5 % kernel time in every physical backing of the arrays (hipMalloc, shuffled
    // 2 MiB chunks, contiguous), 256^3 -1 .. -4 % (scripts/kbench_place.hip, profiles/r05_placement_ab.txt).  "fused_tile" = 3 forces it, 0 keeps 64 x 4
    {
        // chunk depth of the tall tile: 12 planes from nz = 384 on (a third fewer prologue planes: 512^3 4.877 -> 4.838 ms, three rounds alike, with 16 planes 4.865, with 32 4.958;
        // 256^3 is best at 8: 0.801 against 0.813 / 0.835; scripts/kbench_kz.hip, same arrays in one process); tuning switch "fused_kz": 0 = this rule, 8 / 12 force a depth
        const int tkz = h->fused_kz == 8 || h->fused_kz == 12 ? h->fused_kz : (L.nz >= 384 ? 12 : 8);
        const FusedShape T{64, 8, tkz};
        fused_tiles(L, T, nt);
        // The general form fits the shape as well (128 VGPRs, no spills, 70 KB of LDS per block) and gains little from it: nothing at 256^3 (1.064 -> 1.066 ms), 0.7 % at 512^3 with
        // 12-plane chunks (7.447 / 7.469 -> 7.392 / 7.418 ms, two pairs of processes on searched placements) -- it runs this shape from nz = 384 on, or when "fused_tile" = 3 asks for it
        const bool visc = h->viscous_limit && h->visc_ok && dt == INFINITY;
        // round 6 A/B, "fused_tile" = 4: 64 x 16 threads -- fifteen stress rows per tile (y halo 16/15), ONE 16-wave block per CU: fetches 6 % less (16.08 against 17.16 GB per
        // launch) and runs 8 % slower (5.14 - 5.17 against 4.74 - 4.87 ms at 512^3: one block per CU leaves nobody to run while it waits at its two barriers per plane);
        // profiles/r06_y_halo.txt
        if (h->fused_ylds && h->fused_tile == 4) return FusedShape{64, 16, 12};
        if (h->fused_ylds && (h->fused_tile == 3 || ((visc || L.nz >= 384) && h->fused_tile == 2 && !narrow && (long long)nt[0] * nt[1] * nt[2] >= 4096))) return T;
    }
    for (int kz = 8; kz >= 2; kz /= 2) {
        S.kz = kz;
        fused_tiles(L, S, nt);
        if ((long long)nt[0] * nt[1] * nt[2] >= 512) break;
    }
    return S;
}

// ------------------------------------------------------------------------------------------------
// Iteration driver shared by jrx_stokes3d_solve and jrx_stokes3d_iterate_timed.
// One PT iteration m (Stokes3D.jl:78-121) = A_m (stress sweep), B_m (velocity sweep), BCs, halo.
// When nothing observes iteration m's diagnostics and iteration m+1 is certain to run, B_m is fused
// with A_{m+1} (k_fused3d), ping-ponging the state arrays P, τ(6), V(3) between the caller's arrays
// and the handle's scratch set; whatever set is current at the end is copied back.
// ------------------------------------------------------------------------------------------------
struct Iter3D {
    jrx_handle *h;
    const jrx_stokes3d_params *p;
    const double *etatau;
    jrx_stokes3d_fields cur;      // caller's fields with the 10 state pointers of the current set
    Out10 setU, setS;             // caller's arrays / scratch arrays
    // where the current state lives, per group: P and the six stresses / the three velocities.  A fused step moves both groups to the other set; an un-fused stress or
    // velocity sweep can move its own group alone (flip_A / flip_B: it writes out of place -- every sweep reads only the node's own old value of what it writes), which is how a
    // batch with an odd number of fused steps still ends in the caller's arrays without a copy-back and without an extra un-fused iteration (jrx_stokes3d_iterate_timed)
    bool pt_user = true, v_user = true;
    bool flip_A = false, flip_B = false;    // the next un-fused stress sweep / velocity sweep writes its group into the other set (no communicator, no periodic faces)
    bool all_user() const { return pt_user && v_user; }
    bool stress_done = false;     // A of the upcoming iteration already applied (by a fused launch)
    bool fusable = false;
    bool bcs_ordered[2] = {false, false};   // flow_bcs! has run with the reference's pass order on the V of set U / set S
    bool ghosts_stale = false;              // the last fused step left flow_bcs! of its new V to be applied lazily
};

static Out10 out_of(const jrx_stokes3d_fields &f) { return Out10{f.P, f.txx, f.tyy, f.tzz, f.tyz, f.txz, f.txy, f.Vx, f.Vy, f.Vz}; }
static void set_state(jrx_stokes3d_fields &f, const Out10 &o)
{
    f.P = o.P; f.txx = o.txx; f.tyy = o.tyy; f.tzz = o.tzz; f.tyz = o.tyz; f.txz = o.txz; f.txy = o.txy; f.Vx = o.Vx; f.Vy = o.Vy; f.Vz = o.Vz;
}
// P and the stresses of `pt`, the velocities of `v`
static Out10 mix_sets(const Out10 &pt, const Out10 &v) { return Out10{pt.P, pt.txx, pt.tyy, pt.tzz, pt.tyz, pt.txz, pt.txy, v.Vx, v.Vy, v.Vz}; }

static jrx_status ensure_scratch(jrx_handle *h, int nx, int ny, int nz)
{
    const size_t n[10] = {(size_t)nx * ny * nz, (size_t)nx * ny * nz, (size_t)nx * ny * nz, (size_t)nx * ny * nz,
                          (size_t)nx * (ny + 1) * (nz + 1), (size_t)(nx + 1) * ny * (nz + 1), (size_t)(nx + 1) * (ny + 1) * nz,
                          (size_t)(nx + 1) * (ny + 2) * (nz + 2), (size_t)(nx + 2) * (ny + 1) * (nz + 2), (size_t)(nx + 2) * (ny + 2) * (nz + 1)};
    if (h->scratch_dims[0] == nx && h->scratch_dims[1] == ny && h->scratch_dims[2] == nz && h->scratch[0] && h->scratch_stagger_used == h->scratch_stagger + 1000003 * (int)h->scratch_contiguous) return JRX_OK;
    for (int q = 0; q < 10; q++) {
        if (h->scratch_base[q]) JRX_TRY(jrx_dev_free(h, h->scratch_base[q]));
        h->scratch[q] = h->scratch_base[q] = nullptr;
    }
    h->scratch_dims[0] = h->scratch_dims[1] = h->scratch_dims[2] = 0;
    // tuning switch scratch_stagger (bytes, a multiple of 256): array q starts q * stagger bytes into its allocation, so that the ten arrays of the set -- four of them exactly
    // 2^30 bytes at 512^3 -- do not all start at the same offset of the memory system's interleaving pattern
    const size_t stg = (size_t)(h->scratch_stagger > 0 ? h->scratch_stagger : 0) & ~(size_t)255;
    for (int q = 0; q < 10; q++) {
        void *b = nullptr;
        // tuning switch scratch_contiguous: physically contiguous device memory (hipDeviceMallocContiguous), plain hipMalloc when the runtime refuses
        const int keep = h->field_placement;
        if (h->scratch_contiguous) h->field_placement = 2;
        const jrx_status st = jrx_dev_alloc(h, n[q] * sizeof(double) + (size_t)q * stg, &b, 1);
        h->field_placement = keep;
        JRX_TRY(st);
        h->scratch_base[q] = (double *)b;
        h->scratch[q] = (double *)((char *)b + (size_t)q * stg);
        // (test switch "scratch_poison": jrx_dev_alloc fills the allocation with NaNs -- what the set holds before its first use must not matter, tests/test_gpu_stokes3d.py::test_the_second_state_set_may_hold_anything)
    }
    h->scratch_stagger_used = h->scratch_stagger + 1000003 * (int)h->scratch_contiguous;
    h->scratch_dims[0] = nx; h->scratch_dims[1] = ny; h->scratch_dims[2] = nz;
    return JRX_OK;
}

static jrx_status iter_begin(Iter3D &I, jrx_handle *h, const jrx_stokes3d_fields *f, const double *etatau, const jrx_stokes3d_params *p)
{
    I.h = h; I.p = p; I.etatau = etatau;
    JRX_TRY(visc_operands_check(h, f, p));       // may the viscous-limit kernels stand in for the general ones? (dt = Inf only)
    I.cur = *f;
    I.setU = out_of(*f);
    I.pt_user = I.v_user = true; I.flip_A = I.flip_B = false; I.stress_done = false;
    const Lay3 L = make_lay((int)p->nx, (int)p->ny, (int)p->nz);
    // option "fused_comm" = 0 keeps the split sweeps + hidden communication on multi-rank runs (A/B switch; same results);
    // option "scratch_sets" = 0 refuses the library-owned second state set the fused pipeline needs
    // DisplacementBoundaryConditions: flow_bcs! acts on U, the ghosts of V are never refreshed -- the fused kernel's in-kernel BC rules do not apply
    I.fusable = !p->displacement_bcs && (h->kernel_variant == 0 || h->kernel_variant == 3) && h->scratch_sets && (h->fused_comm || !jrx_comm_active(h)) && fits_u32(L) && p->nx >= 48 && p->ny >= 8 && p->nz >= 8;
    if (I.fusable && h->kernel_variant == 0) {
        // auto.  Round 3, with 8-plane chunks and the leaner kernels (scripts/bench_sizes_variants.py, profiles/r03_sizes_variants.txt): the fused pipeline is the
        // faster path at every cube from 48^3 to 512^3 -- by 15 .. 66 % below 160^3, where the round-1 rules (tile fill >= 71 %, >= 1536 tiles) still kept the two
        // sweeps -- except where its 62-column tiles leave a nearly empty last tile while the sweeps' row tiles fill exactly (nx = 125 .. 128: fused -9 %).
        // Lanes per cell row: fused ceil(nx / (TX - 2)) TX-lane tiles with TY - 1 of TY rows updating stresses; sweeps ceil(nx / W) W-lane row tiles (W as in
        // launch_stress); per lane the fused iteration costs 0.62 of the two sweeps (120^3: 9.0 k against 7.4 k it/s at 171 against 128 lanes).  Small grids are
        // bound by the launch count, which favours the fused pipeline whatever the fill.
        const FusedShape S = fused_shape(h, L, p->dt);
        int nt[3];
        fused_tiles(L, S, nt);
        const int W = p->nx > 384 ? 512 : (p->nx > 192 ? 256 : (p->nx > 96 ? 128 : 64));
        const double lanes_f = (double)nt[0] * S.tx * S.ty / (S.ty - 1), lanes_s = (double)((p->nx + W - 1) / W) * W;
        if ((double)p->nx * (double)p->ny * (double)p->nz >= 1e6 && 0.62 * lanes_f > 1.15 * lanes_s) I.fusable = false;
    }
    if (I.fusable) {
        JRX_TRY(ensure_scratch(h, (int)p->nx, (int)p->ny, (int)p->nz));
        double **S = h->scratch;
        I.setS = Out10{S[0], S[1], S[2], S[3], S[4], S[5], S[6], S[7], S[8], S[9]};
        // the scratch V needs the entries no kernel of the fused pipeline writes -- the outer shell of each array (boundary planes with the
        // prescribed normal velocities, ghost planes): copy the shell once
        const int nx = (int)p->nx, ny = (int)p->ny, nz = (int)p->nz;
        const BcArr D[3] = {{I.setS.Vx, {nx + 1, ny + 2, nz + 2}}, {I.setS.Vy, {nx + 2, ny + 1, nz + 2}}, {I.setS.Vz, {nx + 2, ny + 2, nz + 1}}};
        const CBcArr S3[3] = {{f->Vx, {nx + 1, ny + 2, nz + 2}}, {f->Vy, {nx + 2, ny + 1, nz + 2}}, {f->Vz, {nx + 2, ny + 2, nz + 1}}};
        const int m1 = (nx > ny ? nx : ny) + 2, m2 = (ny > nz ? ny : nz) + 2;
        hipLaunchKernelGGL(k_copy_shell3, dim3((unsigned)(((i64)m1 * m2 + 255) / 256), 6, 3), dim3(256), 0, h->stream, D[0], D[1], D[2], S3[0], S3[1], S3[2]);
        JRX_LAUNCH_CHECK(h);
    }
    return JRX_OK;
}

// launch the fused kernel over the box of tiles [b[0], b[1]) x [b[2], b[3]) x [b[4], b[5])
// MW: the second argument of __launch_bounds__, which in HIP is the least number of WAVES PER SIMD the kernel must be able to run with (not CUDA's blocks per multiprocessor): 4, i.e. at most
// 128 VGPRs, for both tile shapes (four 4-wave blocks or two 8-wave blocks per CU); XGV: tile rows per XCD band
static int general_hif_eff(const jrx_handle *h);
// GH: the one-launch forms of the GENERAL kernel (any dt: high-face node layers folded in, in-kernel neighbour faces) are instantiated for this shape (tuning switch "general_hif":
// 4 = built for four waves per SIMD -- 128 VGPRs, 28 dwords of scratch --, 3 = for three -- 155 VGPRs, no scratch --, 0 = the boundary-layer launch behind the kernel as before)
template <int TX, int TY, int KZ, int MW = 4, int XGV = 1, bool GH = false>
static jrx_status launch_fused_t(jrx_handle *h, hipStream_t s, const SweepArgs &a, const FusedBC &bc, const int b[6], bool hiface, bool fold, const FusedShell *shell = nullptr)
{
    // XCD-banded tile order (8 tile rows per XCD): y-halo rows of neighbouring tiles hit in the same L2 (PMC: 45.6 -> 37.3
    // fetched array passes per launch at 512^3)
    const int ntx = b[1] - b[0], nty = b[3] - b[2], ntz = b[5] - b[4];
    if (ntx <= 0 || nty <= 0 || ntz <= 0) return JRX_OK;
    // y-neighbour operands through LDS (measured 8.62 -> 7.94 ms at 512^3); option "fused_ylds" = 0 keeps the lane-shuffle-only form for A/B runs
    // viscous limit (dt = Inf): the operands that 1/(G dt) = 1/(K dt) = 1/dt = 0 multiply are not loaded; option "viscous_limit" = 0 keeps
    // the general kernel.  (Its lower register need -- 111 VGPRs -- leaves room to carry the previous velocity / η planes in registers
    // instead of the third LDS slot: LOWREG off, -1.4 %, scripts/kbench_visc.hip)
    const bool visc = h->viscous_limit && h->visc_ok && a.dt == INFINITY && h->fused_ylds;
    const bool vf = h->visc_fold;      // tuning switch: the folded arithmetic of the viscous limit (k_fused3d, VFOLD; same bits)
    // body-force arrays that hold only +0.0 are not loaded (set by the operand pass): the one-launch viscous-limit form and the general form have those instantiations
    const bool gfold = GH && fold && !visc && h->fused_ylds && general_hif_eff(h) != 0;      // the general form with its high-face node layers inside
    const int nof = ((visc && fold && vf) || (!visc && h->fused_ylds && (!shell || gfold))) ? h->nof : 0;
    if constexpr (GH) {
        if (gfold) {
            const int nblk = shell ? (shell->cls == 0 ? shell->start[1] : shell->start[shell->nbox] - shell->start[1]) : ntx * nty * ntz;
            const FusedShell none = FusedShell{};
            const FusedShell &sh = shell ? *shell : none;
            if (nblk > 0) {
#define GHL(MW_, NBR_, NOF_) hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, MW_, 1, false, XGV, false, true, 3, 1, 0, false, true, false, NBR_, NOF_>), dim3((unsigned)nblk), dim3(TX * TY), 0, s, a, bc, ntx, nty, b[0], b[2], b[4], sh)
                const int key = (general_hif_eff(h) == 3 ? 100 : 0) + (shell ? 10 : 0) + nof;
                switch (key) {
                case 0: GHL(4, false, 0); break; case 1: GHL(4, false, 1); break; case 2: GHL(4, false, 2); break;
                case 10: GHL(4, true, 0); break; case 11: GHL(4, true, 1); break; case 12: GHL(4, true, 2); break;
                case 100: GHL(3, false, 0); break; case 101: GHL(3, false, 1); break; case 102: GHL(3, false, 2); break;
                case 110: GHL(3, true, 0); break; case 111: GHL(3, true, 1); break; default: GHL(3, true, 2); break;
                }
#undef GHL
            }
            JRX_LAUNCH_CHECK(h);
            if (shell && shell->cls == 0) return JRX_OK;       // counted once per iteration, with the second class
            if (shell) h->stat_fused3d_inkernel++;
            if (nof == 1) h->stat_fused3d_nof1++; else if (nof == 2) h->stat_fused3d_nof2++;
            h->stat_fused3d++;
            h->stat_fused3d_general_hif++;
            return JRX_OK;
        }
    }
    if (shell) {        // neighbours, one launch: interior tiles first, the tiles next to a face with a neighbour last (they wait for the exchange's flag)
        if (!(visc && fold)) return jrx_fail(h, JRX_ERR_ARG, "internal: the in-kernel neighbour faces need a one-launch form of the kernel");
        // blk0 = 0: box 0 (tiles that touch no face with a neighbour, beside the exchange); blk0 = start[1]: everything else, behind it
        const int nblk = shell->cls == 0 ? shell->start[1] : shell->start[shell->nbox] - shell->start[1];
        if (nblk > 0) {
            if (nof == 2)
                hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, MW, 1, false, XGV, false, true, 3, 1, 0, true, true, true, true, 2>), dim3((unsigned)nblk), dim3(TX * TY), 0, s, a, bc, ntx, nty, b[0], b[2], b[4], *shell);
            else if (nof == 1)
                hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, MW, 1, false, XGV, false, true, 3, 1, 0, true, true, true, true, 1>), dim3((unsigned)nblk), dim3(TX * TY), 0, s, a, bc, ntx, nty, b[0], b[2], b[4], *shell);
            else
                hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, MW, 1, false, XGV, false, true, 3, 1, 0, true, true, true, true>), dim3((unsigned)nblk), dim3(TX * TY), 0, s, a, bc, ntx, nty, b[0], b[2], b[4], *shell);
        }
        if (shell->cls != 0) { h->stat_fused3d_inkernel++; if (nof == 1) h->stat_fused3d_nof1++; else if (nof == 2) h->stat_fused3d_nof2++; }
        else { JRX_LAUNCH_CHECK(h); return JRX_OK; }       // counted once per iteration, with the second class
    } else if (visc && fold && vf && TX == 64 && TY == 8 && h->fused_ym > 1) {
        // the y march (k_fused3d, YM; tuning switch "fused_ym" = 2 / 4, round 6): a block marches that many tile rows, the y halo row is computed once per march.  Bit-identical,
        // and SLOWER: 4.96 - 5.00 ms (2 rows) / 5.12 - 5.16 ms (4) against 4.74 - 4.79 ms at 512^3 -- under the XCD-banded tile order y neighbours share part of their halo rows in L2; the march gives that up and
        // fetches more through the fabric, not less (17.16 -> 18.15 / 18.60 GB per launch, PMC): profiles/r06_y_halo.txt.  Off.
        const int ym = h->fused_ym >= 3 ? 4 : 2;
        const unsigned nblk = (unsigned)(ntx * ((nty + ym - 1) / ym) * ntz);
#define YML(NOF_, YM_) hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, MW, 1, false, XGV, false, true, 3, 1, 0, true, true, true, false, NOF_, YM_>), dim3(nblk), dim3(TX * TY), 0, s, a, bc, ntx, nty, b[0], b[2], b[4])
        if constexpr (TX == 64 && TY == 8) {
            switch (nof * 10 + (ym >= 3 ? 4 : 2)) {        // built for marches of two and of four tile rows
            case 2: YML(0, 2); break; case 4: YML(0, 4); break;
            case 12: YML(1, 2); break; case 14: YML(1, 4); break;
            case 22: YML(2, 2); break; default: YML(2, 4); break;
            }
        }
#undef YML
        if (nof == 1) h->stat_fused3d_nof1++; else if (nof == 2) h->stat_fused3d_nof2++;
        h->stat_fused3d_ym++;
    } else if (visc && fold && vf) {     // + the high-face node layers inside the kernel: the whole iteration in one launch
        if (nof == 2) {
            hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, MW, 1, false, XGV, false, true, 3, 1, 0, true, true, true, false, 2>), dim3((unsigned)(ntx * nty * ntz)), dim3(TX * TY), 0, s, a, bc, ntx, nty, b[0], b[2], b[4]);
            h->stat_fused3d_nof2++;
        } else if (nof == 1) {
            hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, MW, 1, false, XGV, false, true, 3, 1, 0, true, true, true, false, 1>), dim3((unsigned)(ntx * nty * ntz)), dim3(TX * TY), 0, s, a, bc, ntx, nty, b[0], b[2], b[4]);
            h->stat_fused3d_nof1++;
        } else
            hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, MW, 1, false, XGV, false, true, 3, 1, 0, true, true, true>), dim3((unsigned)(ntx * nty * ntz)), dim3(TX * TY), 0, s, a, bc, ntx, nty, b[0], b[2], b[4]);
    }
    else if (visc && fold)
        hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, MW, 1, false, XGV, false, true, 3, 1, 0, true, true>), dim3((unsigned)(ntx * nty * ntz)), dim3(TX * TY), 0, s, a, bc, ntx, nty, b[0], b[2], b[4]);
    else if (visc && !hiface && vf)
        hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, MW, 1, false, XGV, false, true, 3, 1, 0, true, false, true>), dim3((unsigned)(ntx * nty * ntz)), dim3(TX * TY), 0, s, a, bc, ntx, nty, b[0], b[2], b[4]);
    else if (visc && hiface)
        hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, MW, 1, false, XGV, false, true, 3, 1, 1, true>), dim3((unsigned)(ntx * nty * ntz)), dim3(TX * TY), 0, s, a, bc, ntx, nty, b[0], b[2], b[4]);
    else if (visc)
        hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, MW, 1, false, XGV, false, true, 3, 1, 0, true>), dim3((unsigned)(ntx * nty * ntz)), dim3(TX * TY), 0, s, a, bc, ntx, nty, b[0], b[2], b[4]);
    else if (h->fused_ylds && hiface) {
        if (nof == 2)
            hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, MW, 1, true, XGV, false, true, 3, 1, 1, false, false, false, false, 2>), dim3((unsigned)(ntx * nty * ntz)), dim3(TX * TY), 0, s, a, bc, ntx, nty, b[0], b[2], b[4]);
        else if (nof == 1)
            hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, MW, 1, true, XGV, false, true, 3, 1, 1, false, false, false, false, 1>), dim3((unsigned)(ntx * nty * ntz)), dim3(TX * TY), 0, s, a, bc, ntx, nty, b[0], b[2], b[4]);
        else
            hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, MW, 1, true, XGV, false, true, 3, 1, 1>), dim3((unsigned)(ntx * nty * ntz)), dim3(TX * TY), 0, s, a, bc, ntx, nty, b[0], b[2], b[4]);
        if (nof == 1) h->stat_fused3d_nof1++; else if (nof == 2) h->stat_fused3d_nof2++;
    } else if (h->fused_ylds && nof == 2) {
        hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, MW, 1, true, XGV, false, true, 3, 1, 0, false, false, false, false, 2>), dim3((unsigned)(ntx * nty * ntz)), dim3(TX * TY), 0, s, a, bc, ntx, nty, b[0], b[2], b[4]);
        h->stat_fused3d_nof2++;
    } else if (h->fused_ylds && nof == 1) {
        hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, MW, 1, true, XGV, false, true, 3, 1, 0, false, false, false, false, 1>), dim3((unsigned)(ntx * nty * ntz)), dim3(TX * TY), 0, s, a, bc, ntx, nty, b[0], b[2], b[4]);
        h->stat_fused3d_nof1++;
    } else if (h->fused_ylds)
        // + non-temporal stores: the written set is not read again before the next iteration (PMC: 35.7 -> 34.2 fetched passes, -0.5 .. -1.6 % time)
        // + register diet to 128 VGPRs without spills (4 waves/SIMD): previous velocity plane re-read from a third LDS slot, previous η/G
        //   plane carried as partial sums, the nine stress-only operands requested after the velocity phase (-1 .. -5 %)
        // + 8-plane chunks and tile rows dealt round-robin to the XCDs (XG = 1: the eight XCDs work on eight adjacent tile rows at a time;
        //   kbench 512^3, same box: 8.65 ms with 16 planes / 8-row bands -> 8.04 ms; 256^3: 1.106 -> 1.04 ms)
        hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, MW, 1, true, XGV, false, true, 3, 1>), dim3((unsigned)(ntx * nty * ntz)), dim3(TX * TY), 0, s, a, bc, ntx, nty, b[0], b[2], b[4]);
    else
        hipLaunchKernelGGL((k_fused3d<TX, TY, KZ, 2, 1, false, 8, false, true>), dim3((unsigned)(ntx * nty * ntz)), dim3(TX * TY), 0, s, a, bc, ntx, nty, b[0], b[2], b[4]);
    JRX_LAUNCH_CHECK(h);
    h->stat_fused3d++;
    if (visc) h->stat_fused3d_visc++;
    return JRX_OK;
}
// can this launch also update the high-face node layers (k_fused3d<..., HIF>)?  Only the viscous-limit form over all tiles of a block without neighbours has them
static FusedShape fused_shape(const jrx_handle *h, const Lay3 &L, double dt);
// tuning switch "general_hif": 0 (default) = the pipeline of rounds 1-4 (kernel + boundary-layer launch; early exchange with neighbours).  Measured (gpurun_out/r05t, r05u;
// profiles/r05_general_one_launch.txt): the one-launch instantiation needs 155 VGPRs (three waves per SIMD; built for four it spills 28 dwords: 10.99 ms) and runs at 7.95 ms where kernel +
// boundary-layer launch take 7.47 + 0.11 ms at 512^3 (256^3: 1.178 against 1.146); with neighbours its in-kernel faces do not repay that either (two 512^3 blocks on one device, general
// form: +16.6 % (x split) / +8.6 % (z) over uncoupled blocks against +10.5 % / +11.3 % for the early exchange).  3 / 4 select it for A/B runs and for the parity tests.
static int general_hif_eff(const jrx_handle *h) { return h->general_hif > 0 ? h->general_hif : 0; }
static bool fused_folds_hiface(const jrx_handle *h, const SweepArgs &a)
{
    if (!h->fused_hiface || !h->fused_ylds) return false;
    if (h->viscous_limit && h->visc_ok && a.dt == INFINITY) return true;
    // the general form (round 5): its one-launch instantiations exist for the 64 x 4 tile
    const FusedShape S = fused_shape(h, a.L, a.dt);
    return general_hif_eff(h) != 0 && S.tx == 64 && S.ty == 4;
}
static jrx_status launch_fused(jrx_handle *h, hipStream_t s, const SweepArgs &a, const FusedBC &bc, const int b[6], bool hiface = false, bool fold = false, const FusedShell *shell = nullptr)
{
    const FusedShape S = fused_shape(h, a.L, a.dt);
    const int kz = S.kz;
    if (S.tx == 64 && S.ty == 16) return launch_fused_t<64, 16, 12, 4, 2>(h, s, a, bc, b, hiface, fold, shell);
    if (S.tx == 64 && S.ty == 8) return kz == 12 ? launch_fused_t<64, 8, 12, 4, 4>(h, s, a, bc, b, hiface, fold, shell) : launch_fused_t<64, 8, 8, 4, 4>(h, s, a, bc, b, hiface, fold, shell);
    if (S.tx == 64) {
        if (kz == 8) return launch_fused_t<64, 4, 8, 4, 1, true>(h, s, a, bc, b, hiface, fold, shell);
        if (kz == 4) return launch_fused_t<64, 4, 4, 4, 1, true>(h, s, a, bc, b, hiface, fold, shell);
        return launch_fused_t<64, 4, 2, 4, 1, true>(h, s, a, bc, b, hiface, fold, shell);
    }
    if (kz == 8) return launch_fused_t<32, 8, 8>(h, s, a, bc, b, hiface, fold, shell);
    if (kz == 4) return launch_fused_t<32, 8, 4>(h, s, a, bc, b, hiface, fold, shell);
    return launch_fused_t<32, 8, 2>(h, s, a, bc, b, hiface, fold, shell);
}

// the faces of this rank's block that are physical boundaries (no neighbour): the bits of free_slip / no_slip that flow_bcs! may act on once the planes of the other faces
// hold received values.  3D naming of the reference (App. C.4): free_slip `top` is k = 1 and `bot` k = end, no_slip the other way round
static void physical_faces(const jrx_handle *h, uint32_t *fs_keep, uint32_t *ns_keep)
{
    const uint32_t fs_lo[3] = {JRX_FACE_LEFT, JRX_FACE_FRONT, JRX_FACE_TOP}, fs_hi[3] = {JRX_FACE_RIGHT, JRX_FACE_BACK, JRX_FACE_BOT};
    const uint32_t ns_lo[3] = {JRX_FACE_LEFT, JRX_FACE_FRONT, JRX_FACE_BOT}, ns_hi[3] = {JRX_FACE_RIGHT, JRX_FACE_BACK, JRX_FACE_TOP};
    *fs_keep = *ns_keep = 0;
    for (int d = 0; d < 3; d++) {
        if (!jrx_comm_has_neighbor(h, d, 0)) { *fs_keep |= fs_lo[d]; *ns_keep |= ns_lo[d]; }
        if (!jrx_comm_has_neighbor(h, d, 1)) { *fs_keep |= fs_hi[d]; *ns_keep |= ns_hi[d]; }
    }
}

// tev (optional): events recorded around the sweeps: [0] start, [1] after the stress sweep (if one was launched),
// [2] after the velocity sweep or after the fused launch group (k_fused3d + BCs + boundary planes),
// [3] (fused only) directly after k_fused3d, so that [1] -> [3] is that kernel alone
static inline int imin(int a, int b) { return a < b ? a : b; }
// ncells_timed (optional): the number of cells whose stresses the launch timed by tev[1] -> tev[3] updates
// cev (optional, tuning switch "chain_profile"): six events around the stages of a multi-rank fused step, *chain_mode = 1 (exchange in order) / 2 (early exchange)
static jrx_status iter_step(Iter3D &I, bool diag, bool fuse_next, hipEvent_t *tev, int *was_fused, double *ncells_timed = nullptr, hipEvent_t *cev = nullptr, int *chain_mode = nullptr)
{
    jrx_handle *h = I.h;
    const jrx_stokes3d_params *p = I.p;
    const int nx = (int)p->nx, ny = (int)p->ny, nz = (int)p->nz;
    hipStream_t s = h->stream;
    SweepArgs a = make_args(&I.cur, I.etatau, p);
    if (was_fused) *was_fused = 0;
    if (chain_mode) *chain_mode = 0;
    if (ncells_timed) *ncells_timed = (double)nx * ny * nz;
    if (tev) JRX_HIP(h, hipEventRecord(tev[0], s));
    if (!I.stress_done) {
        if (I.flip_A) {
            // out of place: the new P and stresses go to the other set (the sweep writes every entry of the seven arrays), the velocities stay where they are
            const Out10 other = mix_sets(I.pt_user ? I.setS : I.setU, I.v_user ? I.setU : I.setS);
            a.o = other;
            JRX_TRY(launch_stress(h, s, a, diag, 0, nx + 1, 0, ny + 1, 0, nz + 1));
            set_state(I.cur, other);
            I.pt_user = !I.pt_user;
            I.flip_A = false;
            a = make_args(&I.cur, I.etatau, p);
        } else
            JRX_TRY(launch_stress(h, s, a, diag, 0, nx + 1, 0, ny + 1, 0, nz + 1));
    }
    I.stress_done = false;
    if (tev) JRX_HIP(h, hipEventRecord(tev[1], s));

    if (I.fusable && fuse_next && !diag) {
        // B_m + BCs + A_{m+1}: src = current set, dst = the other set
        const Out10 dst = mix_sets(I.pt_user ? I.setS : I.setU, I.v_user ? I.setS : I.setU);
        a.o = dst;
        FusedBC bc;
        const uint32_t fs = p->free_slip, ns = p->no_slip;
        bc.fsL = !!(fs & JRX_FACE_LEFT); bc.nsL = !!(ns & JRX_FACE_LEFT); bc.fsF = !!(fs & JRX_FACE_FRONT); bc.nsF = !!(ns & JRX_FACE_FRONT);
        bc.fsK0 = !!(fs & JRX_FACE_TOP);  bc.nsK0 = !!(ns & JRX_FACE_BOT);     // k = 1: free_slip `top`, no_slip `bot` (reference naming)
        bc.nsR = !!(ns & JRX_FACE_RIGHT); bc.nsBk = !!(ns & JRX_FACE_BACK); bc.nsK1 = !!(ns & JRX_FACE_TOP);
        bc.fsR = !!(fs & JRX_FACE_RIGHT); bc.fsBk = !!(fs & JRX_FACE_BACK); bc.fsK1 = !!(fs & JRX_FACE_BOT);       // k = end: free_slip `bot`
        bc.nbL = bc.nbR = bc.nbF = bc.nbBk = bc.nbK0 = bc.nbK1 = 0;
        bc.feed = h->nbr_feeder ? 1 : 0;
        // 256 threads per tile, 8 planes per chunk (128 VGPRs -> 4 blocks/CU)
        int nt[3];
        const FusedShape S = fused_shape(h, a.L, a.dt);
        fused_tiles(a.L, S, nt);
        const bool comm = jrx_comm_active(h);
        // periodic_boundary! faces (periodic.jl:56-98) are this block's own neighbour: their planes of V are filled by flow_bcs! after the fused
        // launch, and the stress nodes that read them are redone below exactly like the nodes next to a received halo plane
        const bool per = p->periodic != 0;
        hipStream_t bs = s;            // stream of the boundary work
        // flow_bcs! on the new V: the reference's ordered passes the first time a set is written, one launch for all faces afterwards
        bool &ordered = I.bcs_ordered[I.v_user ? 1 : 0];
        auto fused_bcs = [&](hipStream_t st) -> jrx_status {
            if (ordered && !per) return launch_bcs_faces(h, st, dst.Vx, dst.Vy, dst.Vz, nx, ny, nz, p->free_slip, p->no_slip);
            ordered = true;
            return launch_bcs(h, st, dst.Vx, dst.Vy, dst.Vz, nx, ny, nz, p->free_slip, p->no_slip, p->periodic);
        };
        bool nb[3][2] = {};
        bool folded = false;           // the kernel has updated the high-face node layers itself: nothing is left behind it
        bool rules_with_comm = false;  // neighbours, but flow_bcs! was not applied in memory: the fix-up derives the BC entries of the physical faces by rule
        // option "fused_overlap" = 1: shell of tiles + BCs + exchange on the halo stream, interior tiles concurrently (see below).  Off by
        // default: measured on one device (periodic self neighbour through RCCL, profiles/r01_selfhalo_overlap_*.txt) the RCCL
        // send/recv kernel does not finish before the interior kernel drains, so nothing is hidden and the six small shell launches
        // cost more than they save when an x face is involved (10.7 vs 9.1 ms); to be revisited with real neighbours / DMA copies.
        const bool overlap = h->fused_overlap == 1;
        // option "fused_overlap" = 3 (default): the viscous-limit kernel's own boundary tiles finish the cells next to the received planes (see the branch below); where that
        // form does not run (finite dt, a failed operand check) the early exchange (2) stands in
        const bool visc_form = h->viscous_limit && h->visc_ok && a.dt == INFINITY;
        const bool inkernel = (h->fused_overlap == 3 || h->fused_overlap == 4) && comm && !per && fused_folds_hiface(h, a) && (h->visc_fold || !visc_form);
        const bool early = (h->fused_overlap == 2 || (h->fused_overlap >= 3 && !inkernel)) && comm && !per;
        const bool split = !comm && !per && h->fused_split && nt[0] > 1 && nt[1] > 1 && nt[2] > 1;
        if (split) {
            // Without neighbours the only work behind the fused kernel is the stress update of the high-face node layers (i = nx, j = ny,
            // k = nz): thin, strided (the x face touches one cache line per node and array) and latency-bound -- 0.11 ms at 512^3, 7 % of
            // an iteration at 256^3.  Only the tiles that touch a high face read those nodes in the next iteration, so the iteration is
            // forked: the interior tiles run on the compute stream; the high-face tiles (three disjoint boxes) and, behind them, the
            // boundary layers run on the (high-priority) halo stream; both join before the next iteration.  No tile reads what another
            // tile of the same iteration writes (ping-pong), the boundary layers read new velocities that only high-face tiles produce,
            // and the old stresses they read are never written by a tile: results are those of the single launch, bit for bit.
            // Measured (profiles/r02_ab_fused_split.txt, same box, alternating): 512^3 119.5 / 112.5 it/s forked vs 118.4 / 121.7 single launch,
            // 256^3 822 / 819 vs 847 / 862 -- the interior launch alone takes as long as the launch over all tiles (the high-face column of
            // tiles sweeps every plane of every array beside it and breaks the DRAM page locality of the sweep): option "fused_split", off.
            bs = h->halo_stream;
            JRX_HIP(h, hipEventRecord(h->ev[0], s));
            JRX_HIP(h, hipStreamWaitEvent(bs, h->ev[0], 0));
            const int inner[6] = {0, nt[0] - 1, 0, nt[1] - 1, 0, nt[2] - 1};
            JRX_TRY(launch_fused(h, s, a, bc, inner));
            if (tev) JRX_HIP(h, hipEventRecord(tev[3], s));
            const int hx[6] = {nt[0] - 1, nt[0], 0, nt[1], 0, nt[2]}, hy[6] = {0, nt[0] - 1, nt[1] - 1, nt[1], 0, nt[2]},
                      hz[6] = {0, nt[0] - 1, 0, nt[1] - 1, nt[2] - 1, nt[2]};
            JRX_TRY(launch_fused(h, bs, a, bc, hx, true));
            JRX_TRY(launch_fused(h, bs, a, bc, hy, true));
            JRX_TRY(launch_fused(h, bs, a, bc, hz, true));
            I.ghosts_stale = true;
            if (ncells_timed) *ncells_timed = (double)imin(nx, (nt[0] - 1) * (S.tx - 2)) * (double)imin(ny, (nt[1] - 1) * (S.ty - 1)) * (double)imin(nz, (nt[2] - 1) * S.kz);
        } else if (inkernel) {
            // Neighbour faces inside the kernel (option "fused_overlap" = 3).  As with the early exchange, the velocity phase alone runs first over the boundary slabs of the
            // faces with a neighbour and update_halo!(V) follows on the halo stream, beside k_fused3d on the compute stream.  But the kernel then needs no fix-up: its tiles
            // are launched in two classes -- the ones that touch no face with a neighbour right away, the others behind the event that marks the end of the exchange -- and
            // the second class reads the received planes of the new set where the other tiles apply a flow_bcs! rule: the stress nodes next to a received plane (and the
            // high-face node layers, HIF) come out right the first time.  flow_bcs! is not applied in memory at all (every rule is applied on the fly; the pending application
            // happens before anything reads those entries from memory, `ghosts_stale`).  No BC launch, no fix-up launch, no strided x-face layers.
            // (A first version kept ONE launch and let the boundary tiles, last in the block order, spin on a device-side flag: with two ranks on one device the spinning blocks
            // of one rank fill the chip and starve the kernels they wait for -- the time-out fired at 512^3.  A stream dependency cannot deadlock.)
            bs = h->halo_stream;
            JRX_HIP(h, hipEventRecord(h->ev[0], s));
            JRX_HIP(h, hipStreamWaitEvent(bs, h->ev[0], 0));
            bool nbf[3][2];
            for (int d = 0; d < 3; d++) { nbf[d][0] = jrx_comm_has_neighbor(h, d, 0); nbf[d][1] = jrx_comm_has_neighbor(h, d, 1); }
            {
                const int w = 4;
                const int xa = imin(w, nx / 2), ya = imin(w, ny / 2), za = imin(w, nz / 2);
                const bool L0 = nbf[0][0], H0 = nbf[0][1], L1 = nbf[1][0], H1 = nbf[1][1], L2 = nbf[2][0], H2 = nbf[2][1];
                const int z0 = L2 ? za : 0, z1 = H2 ? nz - za : nz, y0 = L1 ? ya : 0, y1 = H1 ? ny - ya : ny;
                if (L2) JRX_TRY(launch_velocity(h, bs, a, false, 0, nx, 0, ny, 0, za));
                if (H2) JRX_TRY(launch_velocity(h, bs, a, false, 0, nx, 0, ny, nz - za, nz));
                if (L1) JRX_TRY(launch_velocity(h, bs, a, false, 0, nx, 0, ya, z0, z1));
                if (H1) JRX_TRY(launch_velocity(h, bs, a, false, 0, nx, ny - ya, ny, z0, z1));
                if (L0) JRX_TRY(launch_velocity(h, bs, a, false, 0, xa, y0, y1, z0, z1));
                if (H0) JRX_TRY(launch_velocity(h, bs, a, false, nx - xa, nx, y0, y1, z0, z1));
            }
            if (cev) { JRX_HIP(h, hipEventRecord(cev[0], bs)); JRX_HIP(h, hipEventRecord(cev[1], bs)); }
            {
                double *arrs[3] = {dst.Vx, dst.Vy, dst.Vz};
                const int64_t ext[3][3] = {{nx + 1, ny + 2, nz + 2}, {nx + 2, ny + 1, nz + 2}, {nx + 2, ny + 2, nz + 1}};
                const int64_t n[3] = {nx, ny, nz};
                JRX_TRY(jrx_halo_exchange(h, bs, 3, arrs, ext, n));
            }
            JRX_HIP(h, hipEventRecord(h->ev[2], bs));            // update_halo!(V) has delivered
            if (cev) JRX_HIP(h, hipEventRecord(cev[2], bs));
            FusedShell sh;
            memset(&sh, 0, sizeof(sh));
            int lo[3], hi[3];
            for (int d = 0; d < 3; d++) { lo[d] = (nbf[d][0] && nt[d] > 1) ? 1 : 0; hi[d] = (nbf[d][1] && nt[d] > 1) ? nt[d] - 1 : nt[d]; if (hi[d] < lo[d]) hi[d] = lo[d]; }
            for (int d = 0; d < 3; d++) if ((nbf[d][0] || nbf[d][1]) && nt[d] == 1) { lo[d] = 0; hi[d] = 0; }       // a single tile next to a neighbour: no interior tile in that dimension
            // box 0: the first z chunks of the interior tiles -- tuning switch "fused_first_pct" (default 15 % of the interior chunks: ~0.9 ms of work at 512^3 for an exchange of ~0.3 ms);
            // a launch over ALL interior tiles first would leave the tiles of an x face to a launch of their own, which is slow (a column of 64-lane tiles touches 512 B of every 4 KB
            // row of every array: few HBM channels carry it; profiles/r04_inkernel_faces.txt) -- in the second launch they run among their row neighbours
            int zs = lo[2] + ((hi[2] - lo[2]) * h->fused_first_pct + 99) / 100;
            if (zs > hi[2]) zs = hi[2];
            if (hi[0] <= lo[0] || hi[1] <= lo[1]) zs = lo[2];
            {
                const int boxes[7][6] = {{lo[0], hi[0], lo[1], hi[1], lo[2], zs},                                  // class 1
                                         {0, nt[0], 0, nt[1], zs, nt[2]},                                          // class 2: the rest of the block, natural order
                                         {0, nt[0], 0, nt[1], 0, lo[2]},                                           //          the z-lo slab of shell tiles
                                         {0, nt[0], 0, lo[1], lo[2], zs}, {0, nt[0], hi[1], nt[1], lo[2], zs},     //          the shell tiles beside box 0: y slabs,
                                         {0, lo[0], lo[1], hi[1], lo[2], zs}, {hi[0], nt[0], lo[1], hi[1], lo[2], zs}};   //      x slabs
                const int band[7] = {1, 1, 1, 0, 0, 0, 0};
                int tot = 0;
                for (int q = 0; q < 7; q++) {
                    const int *b = boxes[q];
                    const bool empty = b[1] <= b[0] || b[3] <= b[2] || b[5] <= b[4];
                    if (empty && q > 0) continue;
                    for (int c = 0; c < 6; c++) sh.box[sh.nbox][c] = empty ? 0 : b[c];
                    if (empty) { sh.box[sh.nbox][1] = sh.box[sh.nbox][3] = sh.box[sh.nbox][5] = 1; }      // (keeps the divisions of the tile map defined; no block maps to an empty box 0)
                    sh.banded[sh.nbox] = band[q];
                    sh.start[sh.nbox] = tot;
                    tot += empty ? 0 : (b[1] - b[0]) * (b[3] - b[2]) * (b[5] - b[4]);
                    sh.nbox++;
                }
                sh.start[sh.nbox] = tot;
                if (tot != nt[0] * nt[1] * nt[2]) return jrx_fail(h, JRX_ERR_ARG, "internal: the tile classes of the fused kernel do not cover the block (%d of %d)", tot, nt[0] * nt[1] * nt[2]);
            }
            FusedBC bn = bc;
            bn.nbL = nbf[0][0]; bn.nbR = nbf[0][1]; bn.nbF = nbf[1][0]; bn.nbBk = nbf[1][1]; bn.nbK0 = nbf[2][0]; bn.nbK1 = nbf[2][1];
            if (bn.nbL) bn.fsL = bn.nsL = 0;
            if (bn.nbR) bn.fsR = bn.nsR = 0;
            if (bn.nbF) bn.fsF = bn.nsF = 0;
            if (bn.nbBk) bn.fsBk = bn.nsBk = 0;
            if (bn.nbK0) bn.fsK0 = bn.nsK0 = 0;
            if (bn.nbK1) bn.fsK1 = bn.nsK1 = 0;
            const int all[6] = {0, nt[0], 0, nt[1], 0, nt[2]};
            sh.blk0 = 0; sh.cls = 0;
            JRX_TRY(launch_fused(h, s, a, bn, all, false, true, &sh));           // the tiles that read no received plane: beside the exchange
            JRX_HIP(h, hipStreamWaitEvent(s, h->ev[2], 0));
            sh.blk0 = sh.start[1]; sh.cls = 1;
            JRX_TRY(launch_fused(h, s, a, bn, all, false, true, &sh));           // the tiles next to a face with a neighbour: behind it
            if (tev) JRX_HIP(h, hipEventRecord(tev[3], s));
            bs = s;
            if (cev) { JRX_HIP(h, hipEventRecord(cev[3], s)); if (chain_mode) *chain_mode = 2; }
            I.ghosts_stale = true;
            folded = true;
        } else if (early) {
            // Early exchange (option "fused_overlap" = 2).  What the neighbours need of V_m -- the planes 2 / n - 2 (normal component) and the cell layers 2 / n - 3
            // (tangential ones) -- lies within four cells of the face.  So the velocity phase alone runs first over the boundary slabs of the faces WITH a neighbour
            // (the un-fused velocity kernel, into the new set), followed on the halo stream by flow_bcs! (the sent planes carry BC entries of their own rows) and the
            // whole update_halo!(V), x then y then z, while k_fused3d runs over all tiles on the compute stream: it writes the same values to the slab cells again and
            // never touches the planes the exchange receives into.  Behind both: flow_bcs! on the faces WITHOUT a neighbour (the received planes must survive it;
            // where such a ghost row crosses a received plane the copy / negation of the received value is what the neighbour's own flow_bcs! had sent), then the fix-up.
            bs = h->halo_stream;
            JRX_HIP(h, hipEventRecord(h->ev[0], s));
            JRX_HIP(h, hipStreamWaitEvent(bs, h->ev[0], 0));
            {
                const int w = 4;
                const int xa = imin(w, nx / 2), ya = imin(w, ny / 2), za = imin(w, nz / 2);
                const bool L0 = jrx_comm_has_neighbor(h, 0, 0), H0 = jrx_comm_has_neighbor(h, 0, 1), L1 = jrx_comm_has_neighbor(h, 1, 0), H1 = jrx_comm_has_neighbor(h, 1, 1),
                           L2 = jrx_comm_has_neighbor(h, 2, 0), H2 = jrx_comm_has_neighbor(h, 2, 1);
                // disjoint slabs as in jrx3d_velocity_hidden, only on faces with a neighbour
                const int z0 = L2 ? za : 0, z1 = H2 ? nz - za : nz, y0 = L1 ? ya : 0, y1 = H1 ? ny - ya : ny;
                if (L2) JRX_TRY(launch_velocity(h, bs, a, false, 0, nx, 0, ny, 0, za));
                if (H2) JRX_TRY(launch_velocity(h, bs, a, false, 0, nx, 0, ny, nz - za, nz));
                if (L1) JRX_TRY(launch_velocity(h, bs, a, false, 0, nx, 0, ya, z0, z1));
                if (H1) JRX_TRY(launch_velocity(h, bs, a, false, 0, nx, ny - ya, ny, z0, z1));
                if (L0) JRX_TRY(launch_velocity(h, bs, a, false, 0, xa, y0, y1, z0, z1));
                if (H0) JRX_TRY(launch_velocity(h, bs, a, false, nx - xa, nx, y0, y1, z0, z1));
            }
            if (cev) JRX_HIP(h, hipEventRecord(cev[0], bs));
            // flow_bcs! in memory before the exchange (the sent planes carry BC entries of their own rows) -- not needed when nothing reads those entries from memory: the
            // fix-up below derives them by rule, k_fused3d always did, and every path that does read them (un-fused iterations, results handed back) applies the
            // pending flow_bcs! first (tuning switch "comm_bcs_lazy", default OFF: measured 2 % slower than the two BC launches it saves, profiles/r04_comm_bcs_lazy_ab.txt; a non-default A/B path
            // covered by the `fused_early_lazy_bcs` configurations of tests/test_gpu_two_blocks.py only)
            const bool lazy = h->comm_bcs_lazy;
            if (!lazy) JRX_TRY(fused_bcs(bs));
            if (cev) JRX_HIP(h, hipEventRecord(cev[1], bs));
            {
                double *arrs[3] = {dst.Vx, dst.Vy, dst.Vz};
                const int64_t ext[3][3] = {{nx + 1, ny + 2, nz + 2}, {nx + 2, ny + 1, nz + 2}, {nx + 2, ny + 2, nz + 1}};
                const int64_t n[3] = {nx, ny, nz};
                JRX_TRY(jrx_halo_exchange(h, bs, 3, arrs, ext, n));
            }
            if (cev) JRX_HIP(h, hipEventRecord(cev[2], bs));
            const int all[6] = {0, nt[0], 0, nt[1], 0, nt[2]};
            JRX_TRY(launch_fused(h, s, a, bc, all));
            if (tev) JRX_HIP(h, hipEventRecord(tev[3], s));
            JRX_HIP(h, hipEventRecord(h->ev[2], bs));
            JRX_HIP(h, hipStreamWaitEvent(s, h->ev[2], 0));
            bs = s;
            if (lazy) { I.ghosts_stale = true; rules_with_comm = true; }
            else {   // flow_bcs! of the complete new V on the faces that are physical boundaries
                uint32_t fs_keep, ns_keep;
                physical_faces(h, &fs_keep, &ns_keep);
                JRX_TRY(launch_bcs_faces(h, s, dst.Vx, dst.Vy, dst.Vz, nx, ny, nz, p->free_slip & fs_keep, p->no_slip & ns_keep));
            }
            if (cev) { JRX_HIP(h, hipEventRecord(cev[3], s)); if (chain_mode) *chain_mode = 2; }
            for (int d = 0; d < 3; d++) { nb[d][0] = jrx_comm_has_neighbor(h, d, 0); nb[d][1] = jrx_comm_has_neighbor(h, d, 1); }
        } else if (!comm || !overlap) {
            const int all[6] = {0, nt[0], 0, nt[1], 0, nt[2]};
            folded = !comm && !per && fused_folds_hiface(h, a);
            JRX_TRY(launch_fused(h, s, a, bc, all, false, folded));
            if (tev) JRX_HIP(h, hipEventRecord(tev[3], s));
            // Without neighbours no flow_bcs! launch is needed here: the fused kernel and the boundary-layer launch below derive the
            // boundary entries of V by rule, and every path that reads them from memory (un-fused sweeps, results handed back) is
            // preceded by a flow_bcs! launch of its own.  With neighbours the exchange ships those entries, so they must be in memory.
            if (per || (comm && !h->comm_bcs_lazy)) JRX_TRY(fused_bcs(s));
            else { I.ghosts_stale = true; rules_with_comm = comm; }
            if (cev && comm) JRX_HIP(h, hipEventRecord(cev[1], s));
            if (per) {
                const uint32_t lo[3] = {JRX_FACE_LEFT, JRX_FACE_FRONT, JRX_FACE_BOT}, hi[3] = {JRX_FACE_RIGHT, JRX_FACE_BACK, JRX_FACE_TOP};
                for (int d = 0; d < 3; d++) { nb[d][0] = (p->periodic & lo[d]) != 0; nb[d][1] = (p->periodic & hi[d]) != 0; }
            }
            if (comm) {
                // update_halo!(V) after the BCs (Stokes3D.jl:117-120): the neighbours' new velocities land in the boundary planes of dst
                double *arrs[3] = {dst.Vx, dst.Vy, dst.Vz};
                const int64_t ext[3][3] = {{nx + 1, ny + 2, nz + 2}, {nx + 2, ny + 1, nz + 2}, {nx + 2, ny + 2, nz + 1}};
                const int64_t n[3] = {nx, ny, nz};
                JRX_TRY(jrx_halo_exchange(h, s, 3, arrs, ext, n));
                for (int d = 0; d < 3; d++) { nb[d][0] |= jrx_comm_has_neighbor(h, d, 0); nb[d][1] |= jrx_comm_has_neighbor(h, d, 1); }
                if (cev) { JRX_HIP(h, hipEventRecord(cev[2], s)); if (chain_mode) *chain_mode = 1; }
            }
        } else {
            // The role of @hide_communication (Stokes3D.jl:104-121) for the fused kernel: the shell of tiles that touches a face of
            // the block runs first on the halo stream, followed there by the BCs, update_halo!(V) and the stress fix-up below, while
            // the interior tiles run on the compute stream.  No tile reads what another tile of the same launch writes (ping-pong).
            bs = h->halo_stream;
            JRX_HIP(h, hipEventRecord(h->ev[0], s));
            JRX_HIP(h, hipStreamWaitEvent(bs, h->ev[0], 0));
            int lo[3][2], mid[3][2], hi[3][2];
            for (int d = 0; d < 3; d++) {
                lo[d][0] = 0; lo[d][1] = 1;
                hi[d][0] = nt[d] > 1 ? nt[d] - 1 : 1; hi[d][1] = nt[d];
                mid[d][0] = 1; mid[d][1] = hi[d][0];
            }
            const int boxes[6][6] = {{0, nt[0], 0, nt[1], lo[2][0], lo[2][1]},  {0, nt[0], 0, nt[1], hi[2][0], hi[2][1]},
                                     {0, nt[0], lo[1][0], lo[1][1], mid[2][0], mid[2][1]}, {0, nt[0], hi[1][0], hi[1][1], mid[2][0], mid[2][1]},
                                     {lo[0][0], lo[0][1], mid[1][0], mid[1][1], mid[2][0], mid[2][1]}, {hi[0][0], hi[0][1], mid[1][0], mid[1][1], mid[2][0], mid[2][1]}};
            for (int q = 0; q < 6; q++) JRX_TRY(launch_fused(h, bs, a, bc, boxes[q]));
            const int inner[6] = {mid[0][0], mid[0][1], mid[1][0], mid[1][1], mid[2][0], mid[2][1]};
            JRX_TRY(launch_fused(h, s, a, bc, inner));
            JRX_TRY(fused_bcs(bs));
            // update_halo!(V) after the BCs (Stokes3D.jl:117-120): the neighbours' new velocities land in the boundary planes of dst
            double *arrs[3] = {dst.Vx, dst.Vy, dst.Vz};
            const int64_t ext[3][3] = {{nx + 1, ny + 2, nz + 2}, {nx + 2, ny + 1, nz + 2}, {nx + 2, ny + 2, nz + 1}};
            const int64_t n[3] = {nx, ny, nz};
            JRX_TRY(jrx_halo_exchange(h, bs, 3, arrs, ext, n));
            for (int d = 0; d < 3; d++) { nb[d][0] = jrx_comm_has_neighbor(h, d, 0); nb[d][1] = jrx_comm_has_neighbor(h, d, 1); }
            if (per) {
                const uint32_t lo[3] = {JRX_FACE_LEFT, JRX_FACE_FRONT, JRX_FACE_BOT}, hi[3] = {JRX_FACE_RIGHT, JRX_FACE_BACK, JRX_FACE_TOP};
                for (int d = 0; d < 3; d++) { nb[d][0] |= (p->periodic & lo[d]) != 0; nb[d][1] |= (p->periodic & hi[d]) != 0; }
            }
        }
        // Stress nodes whose stencil reads a velocity plane that only now has its final value are redone from the old τ of the
        // current set and the new V of dst: always the planes i = nx, j = ny, k = nz (high-face BC planes); on a face with a
        // neighbour also the first layer (reads the received plane 0) and the last cell layer (reads the received plane n).
        SweepArgs e = a;
        e.f.Vx = dst.Vx; e.f.Vy = dst.Vy; e.f.Vz = dst.Vz;
        const int ilo = nb[0][0] ? 1 : 0, ihi = nb[0][1] ? nx - 1 : nx, jlo = nb[1][0] ? 1 : 0, jhi = nb[1][1] ? ny - 1 : ny,
                  klo = nb[2][0] ? 1 : 0, khi = nb[2][1] ? nz - 1 : nz;
        const int fix[6][6] = {{0, ilo, 0, ny + 1, 0, nz + 1}, {ihi, nx + 1, 0, ny + 1, 0, nz + 1}, {ilo, ihi, 0, jlo, 0, nz + 1},
                               {ilo, ihi, jhi, ny + 1, 0, nz + 1}, {ilo, ihi, jlo, jhi, 0, klo}, {ilo, ihi, jlo, jhi, khi, nz + 1}};
        GhostRule gr;
        {
            auto ty = [&](uint32_t fsbit, uint32_t nsbit) { return (p->no_slip & nsbit) ? 2 : ((p->free_slip & fsbit) ? 1 : 0); };
            const int t6[6] = {ty(JRX_FACE_LEFT, JRX_FACE_LEFT), ty(JRX_FACE_RIGHT, JRX_FACE_RIGHT), ty(JRX_FACE_FRONT, JRX_FACE_FRONT),
                               ty(JRX_FACE_BACK, JRX_FACE_BACK), ty(JRX_FACE_TOP, JRX_FACE_BOT), ty(JRX_FACE_BOT, JRX_FACE_TOP)};
            for (int q = 0; q < 6; q++) gr.t[q] = t6[q];
            if (rules_with_comm)       // a face with a neighbour holds received values: read them
                for (int d = 0; d < 3; d++) { if (nb[d][0]) gr.t[2 * d] = 0; if (nb[d][1]) gr.t[2 * d + 1] = 0; }
        }
        if (!folded) JRX_TRY(launch_stress_boxes(h, bs, e, fix, 6, false, ((comm || per) && !rules_with_comm) ? nullptr : &gr));
        if (bs != s) {
            JRX_HIP(h, hipEventRecord(h->ev[2], bs));
            JRX_HIP(h, hipStreamWaitEvent(s, h->ev[2], 0));
            if (tev && !split) JRX_HIP(h, hipEventRecord(tev[3], s));     // shell/interior overlap with neighbours: only the whole group can be timed
        }
        set_state(I.cur, dst);
        I.pt_user = !I.pt_user; I.v_user = !I.v_user;
        I.stress_done = true;
        if (was_fused) *was_fused = 1;
        if (tev) JRX_HIP(h, hipEventRecord(tev[2], s));
        return JRX_OK;
    }

    const jrx_stokes3d_fields *f = &I.cur;
    if (!jrx_comm_active(h)) {
        if (I.ghosts_stale && diag) {
            // U = V dt below copies the boundary entries of V as flow_bcs! of the previous iteration left them (the reference applies
            // flow_bcs! after velocity2displacement!): apply that pending flow_bcs! now, before V is updated
            JRX_TRY(launch_bcs(h, s, f->Vx, f->Vy, f->Vz, nx, ny, nz, p->free_slip, p->no_slip, p->periodic));
            I.bcs_ordered[I.v_user ? 0 : 1] = true;
        }
        I.ghosts_stale = false;
        if (I.flip_B && !p->displacement_bcs) {
            // out of place: the new velocities go to the other set, whose boundary and ghost entries hold the caller's values (its own arrays, or their copy: k_copy_shell3);
            // flow_bcs! below then acts on that set
            const Out10 other = mix_sets(I.pt_user ? I.setU : I.setS, I.v_user ? I.setS : I.setU);
            a.o = other;
            JRX_TRY(launch_velocity(h, s, a, diag, 0, nx, 0, ny, 0, nz, h->nof));
            set_state(I.cur, other);
            I.v_user = !I.v_user;
        } else
            JRX_TRY(launch_velocity(h, s, a, diag, 0, nx, 0, ny, 0, nz, h->nof));      // h->nof: what iter_begin's pass over this call's body-force arrays has found
        I.flip_B = false;
        if (tev) JRX_HIP(h, hipEventRecord(tev[2], s));
        if (diag) JRX_TRY(launch_scaleU(h, s, f, p));
        if (p->displacement_bcs) {
            // flow_bcs!(stokes, ::DisplacementBoundaryConditions) acts on U = V dt, which the next iteration overwrites: only observable ones matter
            if (diag) JRX_TRY(launch_bcs(h, s, f->Ux, f->Uy, f->Uz, nx, ny, nz, p->free_slip, p->no_slip, p->periodic));
            return JRX_OK;
        }
        // flow_bcs!: the reference's ordered passes on observable iterations and the first time a set is written, otherwise all faces in
        // one launch (same values wherever a stencil reads them)
        bool &ordered = I.bcs_ordered[I.v_user ? 0 : 1];
        if (!diag && ordered && p->periodic == 0) JRX_TRY(launch_bcs_faces(h, s, f->Vx, f->Vy, f->Vz, nx, ny, nz, p->free_slip, p->no_slip));
        else {
            JRX_TRY(launch_bcs(h, s, f->Vx, f->Vy, f->Vz, nx, ny, nz, p->free_slip, p->no_slip, p->periodic));
            ordered = true;
        }
        return JRX_OK;
    }
    if (I.ghosts_stale) {
        // the fused steps before left flow_bcs! of the physical faces to be applied lazily: now, before U = V dt copies those entries and before anything reads them from memory
        // (the planes of the faces with a neighbour hold received values and stay; where a ghost row crosses one, the copy / negation of the received value is what the
        // neighbour's own flow_bcs! would have sent)
        uint32_t fs_keep, ns_keep;
        physical_faces(h, &fs_keep, &ns_keep);
        JRX_TRY(launch_bcs(h, s, f->Vx, f->Vy, f->Vz, nx, ny, nz, p->free_slip & fs_keep, p->no_slip & ns_keep, 0));
        I.ghosts_stale = false;
    }
    return jrx3d_velocity_hidden(h, f, I.etatau, p, diag, p->displacement_bcs ? (diag ? 2 : 3) : 0);
}

// leave the results in the caller's arrays
static jrx_status iter_end(Iter3D &I)
{
    jrx_handle *h = I.h;
    if (I.stress_done) return jrx_fail(h, JRX_ERR_ARG, "internal: iteration pipeline ended with a pending fused stress sweep");
    if (I.all_user()) return JRX_OK;
    const jrx_stokes3d_params *p = I.p;
    const i64 nc = (i64)p->nx * p->ny * p->nz, nyz = (i64)p->nx * (p->ny + 1) * (p->nz + 1), nxz = (i64)(p->nx + 1) * p->ny * (p->nz + 1),
              nxy = (i64)(p->nx + 1) * (p->ny + 1) * p->nz, n0 = (i64)(p->nx + 1) * (p->ny + 2) * (p->nz + 2),
              n1 = (i64)(p->nx + 2) * (p->ny + 1) * (p->nz + 2), n2 = (i64)(p->nx + 2) * (p->ny + 2) * (p->nz + 1);
    const Out10 &S = I.setS, &U = I.setU;
    const i64 zp = I.pt_user ? 0 : 1, zv = I.v_user ? 0 : 1;        // a group that is in the caller's arrays already is not copied (count 0)
    if (zp) {
        hipLaunchKernelGGL(k_copy6, dim3(2048), dim3(256), 0, h->stream, U.P, (const double *)S.P, nc, U.txx, (const double *)S.txx, nc, U.tyy,
                           (const double *)S.tyy, nc, U.tzz, (const double *)S.tzz, nc, U.tyz, (const double *)S.tyz, nyz, U.txz, (const double *)S.txz, nxz);
        JRX_LAUNCH_CHECK(h);
    }
    hipLaunchKernelGGL(k_copy6, dim3(2048), dim3(256), 0, h->stream, U.txy, (const double *)S.txy, nxy * zp, U.Vx, (const double *)S.Vx, n0 * zv, U.Vy,
                       (const double *)S.Vy, n1 * zv, U.Vz, (const double *)S.Vz, n2 * zv, (double *)nullptr, (const double *)nullptr, (i64)0, (double *)nullptr,
                       (const double *)nullptr, (i64)0);
    JRX_LAUNCH_CHECK(h);
    set_state(I.cur, U);
    I.pt_user = I.v_user = true;
    return JRX_OK;
}

extern "C" {

jrx_status jrx_stokes3d_solve(jrx_handle *h, const jrx_stokes3d_fields *f, const jrx_stokes3d_params *p, jrx_solve_result *res)
{
    JRX_TRY(check_params(h, f, p));
    JRX_TRY(check_diag(h, f));
    if (!res) return jrx_fail(h, JRX_ERR_ARG, "null result");
    if (p->nout < 1) return jrx_fail(h, JRX_ERR_ARG, "nout must be >= 1");
    const int nx = (int)p->nx, ny = (int)p->ny, nz = (int)p->nz;
    const size_t n = (size_t)nx * ny * nz;
    hipStream_t s = h->stream;

    // ητ = deepcopy(η); compute_maxloc!(ητ, η); update_halo!(ητ)   (Stokes3D.jl:55-57)
    JRX_TRY(jrx_ensure_etatau(h, n));
    hipLaunchKernelGGL(k_maxloc, dim3((unsigned)(((i64)nx * ny + 255) / 256), (unsigned)nz), dim3(256), 0, s, h->etatau, f->eta, nx, ny, nz);
    JRX_LAUNCH_CHECK(h);
    if (jrx_comm_active(h)) {
        double *arrs[1] = {h->etatau};
        const int64_t ext[1][3] = {{nx, ny, nz}};
        const int64_t nn[3] = {nx, ny, nz};
        JRX_TRY(jrx_halo_exchange(h, s, 1, arrs, ext, nn));
    }

    if (p->displacement_bcs) {    // displacement2velocity!(stokes, dt, flow_bcs) (Stokes3D.jl:72): V = U * inv(dt)
        hipLaunchKernelGGL(k_scale3, dim3(2048), dim3(256), 0, s, f->Vx, (const double *)f->Ux, (i64)(nx + 1) * (ny + 2) * (nz + 2), f->Vy,
                           (const double *)f->Uy, (i64)(nx + 2) * (ny + 1) * (nz + 2), f->Vz, (const double *)f->Uz, (i64)(nx + 2) * (ny + 2) * (nz + 1),
                           1.0 / p->dt);
        JRX_LAUNCH_CHECK(h);
    }
    double err_it1 = 1.0, err = 1.0;
    int64_t iter = 0, cont = 0;
    res->iter = 0; res->nchecks = 0;
    const int rank = jrx_comm_rank(h);
    hipEvent_t t0 = h->ev[6], t1 = h->ev[7];
    Iter3D I;
    JRX_TRY(iter_begin(I, h, f, h->etatau, p));
    JRX_HIP(h, hipEventRecord(t0, s));
    auto keep_going = [&](int64_t it) { return it < 2 || (((err / err_it1) > p->eps_rel && err > p->eps_abs) && it <= p->iterMax); };
    auto is_check = [&](int64_t it1) { return (it1 % p->nout == 0) && it1 > 1; };
    // Small grids (the reference's own 3D tests run at 8^3 .. 32^3, test/test_stokes_solvi3D.jl:25-55) are launch-bound: three to four dependent launches
    // of a few microseconds each per iteration.  Runs of unobserved iterations of the un-fused loop replay as a captured graph of GIT iterations
    // (option "loop_graphs"; same kernels, same order: identical results; profiles/r03_small_grids_graphs.txt).
    constexpr int GIT = 16;
    GraphExecs gexec;
    bool graphs = h->loop_graphs && !I.fusable && !jrx_comm_active(h) && !p->displacement_bcs && p->periodic == 0 && (double)n <= kGraphCells3D;
    while (keep_going(iter)) {
        if (graphs && iter >= 2 && I.bcs_ordered[0] && I.all_user() && !I.stress_done) {
            // observed iterations: the multiples of nout and iteration iterMax + 1; between them err does not change, so keep_going holds
            int64_t nxt = ((iter / p->nout) + 1) * p->nout;
            if (nxt > p->iterMax + 1) nxt = p->iterMax + 1;
            int64_t run = nxt - 1 - iter;                     // unobserved iterations from here
            if (run >= GIT) {
                if (!gexec[0]) {
                    JRX_TRY(jrx_capture_graph(s, &gexec[0], [&]() -> jrx_status {
                        for (int q = 0; q < GIT; q++) JRX_TRY(iter_step(I, false, false, nullptr, nullptr));
                        return JRX_OK;
                    }));
                    if (!gexec[0]) graphs = false;
                }
                if (gexec[0]) {
                    while (run >= GIT) {
                        JRX_HIP(h, hipGraphLaunch(gexec[0], s));
                        iter += GIT; run -= GIT;
                        h->stat_graph_replays++;
                    }
                    continue;
                }
            }
        }
        const int64_t it1 = iter + 1;
        const bool check = is_check(it1);
        const bool diag = check || !keep_going(it1);   // results observable after this iteration
        // iteration it1+1 certainly runs and is not observable either -> its stress sweep can ride on this velocity sweep
        const bool fuse_next = !diag && keep_going(it1) && !(is_check(it1 + 1) || !keep_going(it1 + 1));
        JRX_TRY(iter_step(I, diag, fuse_next, nullptr, nullptr));
        iter = it1;
        if (check) {
            JRX_TRY(launch_sumsq(h, s, &I.cur, p));
            JRX_HIP(h, hipMemcpyAsync(h->h_sums, h->d_sums, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
            JRX_HIP(h, hipStreamSynchronize(s));
            double ss[4] = {h->h_sums[0], h->h_sums[1], h->h_sums[2], h->h_sums[3]};
            JRX_TRY(jrx_allreduce_sum_host(h, ss, 4));       // norm_mpi: sqrt(Allreduce(Σx²)) (Utils.jl:698-701)
            const double nRx = sqrt(ss[0]) / (double)((p->nxg - 2) * (p->nyg - 1) * (p->nzg - 1));
            const double nRy = sqrt(ss[1]) / (double)((p->nxg - 1) * (p->nyg - 2) * (p->nzg - 1));
            const double nRz = sqrt(ss[2]) / (double)((p->nxg - 1) * (p->nyg - 1) * (p->nzg - 2));
            const double nDV = sqrt(ss[3]) / (double)(p->nxg * p->nyg * p->nzg);
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
                printf("iter = %lld, abs_err = %1.3e, rel_err = %1.3e [norm_Rx=%1.3e, norm_Ry=%1.3e, norm_Rz=%1.3e, norm_∇V=%1.3e] \n",
                       (long long)iter, err, err / err_it1, nRx, nRy, nRz, nDV);
            if (std::isnan(err)) {
                res->iter = iter; res->nchecks = cont < res->cap ? cont : res->cap;
                (void)iter_end(I);
                (void)hipStreamSynchronize(s);
                return jrx_fail(h, JRX_ERR_NAN, "NaN(s)");
            }
        }
    }
    JRX_HIP(h, hipEventRecord(t1, s));
    JRX_TRY(iter_end(I));
    // multi_copy! τ -> τ_o, staggered set then centre set (Stokes3D.jl:172-173); τ_o is an operand of the next call: a cached verdict of the operand pass goes
    h->opv.valid = false;
    const i64 nc = (i64)n, nyz = (i64)nx * (ny + 1) * (nz + 1), nxz = (i64)(nx + 1) * ny * (nz + 1), nxy = (i64)(nx + 1) * (ny + 1) * nz;
    hipLaunchKernelGGL(k_copy6, dim3(2048), dim3(256), 0, s, f->toxx, (const double *)f->txx, nc, f->toyy, (const double *)f->tyy, nc, f->tozz,
                       (const double *)f->tzz, nc, f->toyz, (const double *)f->tyz, nyz, f->toxz, (const double *)f->txz, nxz, f->toxy,
                       (const double *)f->txy, nxy);
    JRX_LAUNCH_CHECK(h);
    if (f->tyz_c && f->toyz_c && f->txz_c && f->toxz_c && f->txy_c && f->toxy_c) {
        hipLaunchKernelGGL(k_copy6, dim3(2048), dim3(256), 0, s, f->toyz_c, (const double *)f->tyz_c, nc, f->toxz_c, (const double *)f->txz_c, nc,
                           f->toxy_c, (const double *)f->txy_c, nc, (double *)nullptr, (const double *)nullptr, (i64)0, (double *)nullptr,
                           (const double *)nullptr, (i64)0, (double *)nullptr, (const double *)nullptr, (i64)0);
        JRX_LAUNCH_CHECK(h);
    }
    JRX_HIP(h, hipStreamSynchronize(s));
    float ms = 0.f;
    JRX_HIP(h, hipEventElapsedTime(&ms, t0, t1));
    res->iter = iter;
    res->nchecks = cont < res->cap ? cont : res->cap;
    res->time_s = ms * 1e-3;
    res->av_time_s = iter > 1 ? res->time_s / (double)(iter - 1) : res->time_s;
    return JRX_OK;
}

jrx_status jrx_stokes3d_iterate_timed(jrx_handle *h, const jrx_stokes3d_fields *f, const double *etatau,
                                      const jrx_stokes3d_params *p, int64_t iters, double times_ms[6])
{
    JRX_TRY(check_params(h, f, p));
    if (!etatau) return jrx_fail(h, JRX_ERR_ARG, "etatau is NULL");
    if (iters < 1) return jrx_fail(h, JRX_ERR_ARG, "iters must be >= 1");
    if (!times_ms) return jrx_fail(h, JRX_ERR_ARG, "times_ms is NULL");
    hipStream_t s = h->stream;
    // hipEvents around the launches of sampled iterations *inside* the timed batch (on the stream the
    // kernels run on); at most 64 samples so that event bookkeeping stays negligible (four records per sampled iteration cost 10-20 us: 2 % of a 256^3 iteration)
    // (an odd stride: consecutive fused iterations alternate between the two state sets, whose allocations run 1-2 % apart -- sampling every second one would see one direction only)
    const int64_t stride = iters > 64 ? (((iters + 63) / 64) | 1) : 1;
    const int nsamp = (int)((iters + stride - 1) / stride);
    const bool chain = h->chain_profile && jrx_comm_active(h);
    std::vector<hipEvent_t> evs((size_t)nsamp * (chain ? 8 : 4));
    std::vector<int> fused((size_t)nsamp, 0), cmode((size_t)nsamp, 0);
    std::vector<double> ncell((size_t)nsamp, 0.0);
    // the events are destroyed on every exit path
    struct EvGuard { std::vector<hipEvent_t> &v; ~EvGuard() { for (auto &e : v) if (e) (void)hipEventDestroy(e); } } guard{evs};
    for (auto &e : evs) e = nullptr;
    for (auto &e : evs) JRX_HIP(h, hipEventCreate(&e));
    Iter3D I;
    JRX_TRY(iter_begin(I, h, f, etatau, p));
    JRX_HIP(h, hipStreamSynchronize(s));
    JRX_HIP(h, hipEventRecord(h->ev[6], s));
    // every fused step flips the ping-pong set of both groups (P + stresses, velocities); the batch has iters - 1 of them between the first stress sweep and the last
    // velocity sweep.  With an odd number the batch would end in the scratch set (copy-back of ten arrays: ~4 ms at 512^3).  Instead the two un-fused sweeps at its ends
    // write out of place: the first stress sweep puts P and the stresses into the scratch set, the last velocity sweep brings the velocities back -- every group ends
    // in the caller's arrays, no copy, no extra launch (tuning switch end_flips = 0: the first step stays un-fused instead, one extra sweep pair, ~2 ms at 512^3).
    // With a communicator, periodic faces or an observed batch the first step stays un-fused as before.
    const bool odd = I.fusable && iters >= 2 && ((iters - 1) % 2 == 1);
    const bool flips = odd && h->end_flips && !jrx_comm_active(h) && p->periodic == 0 && !p->displacement_bcs;
    const bool first_unfused = odd && !flips;
    for (int64_t it = 0; it < iters; it++) {
        const bool samp = it % stride == 0;
        const bool fuse_next = it + 1 < iters && !(first_unfused && it == 0);
        if (flips && it == 0) I.flip_A = true;
        if (flips && it == iters - 1) I.flip_B = true;
        const size_t q = (size_t)(it / stride);
        JRX_TRY(iter_step(I, false, fuse_next, samp ? &evs[q * 4] : nullptr, samp ? &fused[q] : nullptr, samp ? &ncell[q] : nullptr,
                          samp && chain ? &evs[(size_t)nsamp * 4 + q * 4] : nullptr, samp && chain ? &cmode[q] : nullptr));
    }
    JRX_HIP(h, hipEventRecord(h->ev[7], s));
    JRX_TRY(iter_end(I));
    JRX_HIP(h, hipStreamSynchronize(s));
    float ms = 0.f;
    JRX_HIP(h, hipEventElapsedTime(&ms, h->ev[6], h->ev[7]));
    times_ms[0] = ms; times_ms[1] = times_ms[2] = times_ms[3] = times_ms[4] = times_ms[5] = 0.0;
    double sa = 0.0, sb = 0.0, sf = 0.0, sk = 0.0, sc = 0.0;
    int na = 0, nb = 0, nf = 0;
    // with a communicator the un-fused iterations run on two streams (no per-sweep events); fused ones stay on `s`
    const bool comm = jrx_comm_active(h);
    for (int q = 0; q < nsamp; q++) {
        if (comm && !fused[q]) continue;
        float m1 = 0.f, m2 = 0.f;
        JRX_HIP(h, hipEventElapsedTime(&m1, evs[(size_t)q * 4], evs[(size_t)q * 4 + 1]));
        JRX_HIP(h, hipEventElapsedTime(&m2, evs[(size_t)q * 4 + 1], evs[(size_t)q * 4 + 2]));
        if (fused[q]) {
            float m3 = 0.f;
            JRX_HIP(h, hipEventElapsedTime(&m3, evs[(size_t)q * 4 + 1], evs[(size_t)q * 4 + 3]));
            sf += m2; sk += m3; sc += ncell[q]; nf++;
        }
        else { sb += m2; nb++; }
        if (m1 > 1e-3f) { sa += m1; na++; }      // a stress sweep ran as its own launch in this iteration
    }
    if (chain) {
        // the rank's chain per fused iteration, microseconds: [0] k_fused3d, [1] boundary-slab velocity launches, [2] flow_bcs! before the exchange, [3] update_halo!(V)
        // (pack, transport, waits for the neighbour, unpack), [4] flow_bcs! behind the join, [5] stress fix-up, [6] whole step, [7] what the step takes beyond the kernel
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int nc = 0;
        auto el = [&](hipEvent_t a, hipEvent_t b) { float m = 0.f; (void)hipEventElapsedTime(&m, a, b); return (double)m * 1e3; };
        for (int q = 0; q < nsamp; q++) {
            if (!fused[q] || !cmode[q]) continue;
            hipEvent_t *t = &evs[(size_t)q * 4], *c = &evs[(size_t)nsamp * 4 + (size_t)q * 4];
            const double step = el(t[1], t[2]), kern = el(t[1], t[3]);
            if (cmode[q] == 2) {
                const double post = fmin(el(t[3], c[3]), el(c[2], c[3]));       // flow_bcs! starts when both the kernel and the exchange are done
                acc[0] += kern; acc[1] += el(t[1], c[0]); acc[2] += el(c[0], c[1]); acc[3] += el(c[1], c[2]); acc[4] += post; acc[5] += el(c[3], t[2]);
            } else {
                acc[0] += kern; acc[2] += el(t[3], c[1]); acc[3] += el(c[1], c[2]); acc[5] += el(c[2], t[2]);
            }
            acc[6] += step; acc[7] += step - kern;
            nc++;
        }
        for (int q = 0; q < 8; q++) h->chain_us[q] = nc ? acc[q] / nc : 0.0;
        h->chain_n = nc;
        (void)hipGetLastError();
    }
    if (na) times_ms[1] = sa / na;
    if (nb) times_ms[2] = sb / nb;
    if (nf) { times_ms[3] = sf / nf; times_ms[4] = sk / nf; times_ms[5] = sc / nf; }
    return JRX_OK;
}

}   // extern "C"
