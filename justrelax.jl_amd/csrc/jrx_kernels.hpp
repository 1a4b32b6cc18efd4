// jrx_kernels.hpp -- small kernels shared by the 2D / 3D / thermal translation units
// (boundary planes, clamped window maximum, sum-of-squares reductions, scaled / plain copies).
#pragma once
#include "jrx_internal.hpp"

namespace {

// U = V*dt over the full arrays (types/displacement.jl:17-28)
__global__ __launch_bounds__(256) void k_scale3(double *__restrict__ U0, const double *__restrict__ V0, i64 n0,
                                                double *__restrict__ U1, const double *__restrict__ V1, i64 n1,
                                                double *__restrict__ U2, const double *__restrict__ V2, i64 n2, double dt)
{
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n0 || t < n1 || t < n2; t += stride) {
        if (t < n0) U0[t] = V0[t] * dt;
        if (t < n1) U1[t] = V1[t] * dt;
        if (U2 && t < n2) U2[t] = V2[t] * dt;
    }
}

// ------------------------------------------------------------------------------------------------
// Boundary conditions.  One launch per (type, dimension): both faces of a dimension touch
// disjoint planes, and the launches are issued in the reference's source order so that edge
// ghosts get the value a sequential execution of the reference's branches would produce.
// type: 0 free_slip (tangential ghost = interior), 1 no_slip (normal plane = 0, tangential ghost =
// -interior), 2 periodic (normal: lo plane = hi plane; tangential ghosts = opposite interior).
// ------------------------------------------------------------------------------------------------
struct BcArr {
    double *p;
    int n[3];
};

__global__ __launch_bounds__(256) void k_bc3d(BcArr A0, BcArr A1, BcArr A2, int type, int dim, int lo_on, int hi_on)
{
    const BcArr arrs[3] = {A0, A1, A2};
    const int d1 = dim == 0 ? 1 : 0, d2 = dim == 2 ? 1 : 2;   // the two in-plane dims, lower first
    const int ta = blockIdx.x * blockDim.x + threadIdx.x;      // index along d1
    const int tb = blockIdx.y;                                 // index along d2
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const BcArr &A = arrs[c];
        if (ta >= A.n[d1] || tb >= A.n[d2]) continue;
        const bool normal = (c == dim);
        if (type == 0 && normal) continue;
        const i64 s[3] = {1, A.n[0], (i64)A.n[0] * A.n[1]};
        const i64 base = ta * s[d1] + tb * s[d2];
        const int e = A.n[dim];
        double *p = A.p;
        if (type == 0) {
            if (lo_on) p[base] = p[base + s[dim]];
            if (hi_on) p[base + (e - 1) * s[dim]] = p[base + (e - 2) * s[dim]];
        } else if (type == 1) {
            if (normal) {
                if (lo_on) p[base] = 0.0;
                if (hi_on) p[base + (e - 1) * s[dim]] = 0.0;
            } else {
                if (lo_on) p[base] = -p[base + s[dim]];
                if (hi_on) p[base + (e - 1) * s[dim]] = -p[base + (e - 2) * s[dim]];
            }
        } else {
            if (normal) {
                if (lo_on) p[base] = p[base + (e - 1) * s[dim]];
            } else {
                if (lo_on) p[base] = p[base + (e - 2) * s[dim]];
                if (hi_on) p[base + (e - 1) * s[dim]] = p[base + s[dim]];
            }
        }
    }
}

// copy of the outer shell (every entry with an index on the first or last plane of a dimension) of three arrays:
// blockIdx.y = face (2 * dim + side), blockIdx.z = array
struct CBcArr {
    const double *p;
    int n[3];
};
__global__ __launch_bounds__(256) void k_copy_shell3(BcArr D0, BcArr D1, BcArr D2, CBcArr S0, CBcArr S1, CBcArr S2)
{
    const BcArr D = blockIdx.z == 0 ? D0 : (blockIdx.z == 1 ? D1 : D2);
    const CBcArr S = blockIdx.z == 0 ? S0 : (blockIdx.z == 1 ? S1 : S2);
    const int dim = blockIdx.y >> 1, side = blockIdx.y & 1;
    const int d1 = dim == 0 ? 1 : 0, d2 = dim == 2 ? 1 : 2;
    const i64 u = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= (i64)D.n[d1] * D.n[d2]) return;
    const int a = (int)(u % D.n[d1]), b = (int)(u / D.n[d1]);
    const i64 s[3] = {1, D.n[0], (i64)D.n[0] * D.n[1]};
    const i64 idx = a * s[d1] + b * s[d2] + (side ? (i64)(D.n[dim] - 1) * s[dim] : 0);
    D.p[idx] = S.p[idx];
}

// free_slip / no_slip on all six faces in ONE launch (blockIdx.z = dimension; tlo/thi: 0 none, 1 free slip, 2 no slip).
// Every ghost entry that lies on exactly one ghost plane gets the value the ordered passes above give it; entries on two or three
// ghost planes (edge / corner ghosts) depend on the pass order there and are left racy here -- no stencil of the Stokes kernels
// reads them.  Used between fused iterations only; the ordered passes run before results are handed back.
__global__ __launch_bounds__(256) void k_bc3d_faces(BcArr A0, BcArr A1, BcArr A2, int tlo0, int thi0, int tlo1, int thi1, int tlo2, int thi2)
{
    const BcArr arrs[3] = {A0, A1, A2};
    const int dim = blockIdx.z;
    const int tlo = dim == 0 ? tlo0 : (dim == 1 ? tlo1 : tlo2), thi = dim == 0 ? thi0 : (dim == 1 ? thi1 : thi2);
    if (!tlo && !thi) return;
    const int d1 = dim == 0 ? 1 : 0, d2 = dim == 2 ? 1 : 2;
    const int ta = blockIdx.x * blockDim.x + threadIdx.x;
    const int tb = blockIdx.y;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const BcArr &A = arrs[c];
        if (ta >= A.n[d1] || tb >= A.n[d2]) continue;
        const bool normal = (c == dim);
        const i64 s[3] = {1, A.n[0], (i64)A.n[0] * A.n[1]};
        const i64 base = ta * s[d1] + tb * s[d2];
        const int e = A.n[dim];
        double *p = A.p;
        if (normal) {
            if (tlo == 2) p[base] = 0.0;
            if (thi == 2) p[base + (e - 1) * s[dim]] = 0.0;
        } else {
            if (tlo == 1) p[base] = p[base + s[dim]];
            else if (tlo == 2) p[base] = -p[base + s[dim]];
            if (thi == 1) p[base + (e - 1) * s[dim]] = p[base + (e - 2) * s[dim]];
            else if (thi == 2) p[base + (e - 1) * s[dim]] = -p[base + (e - 2) * s[dim]];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// compute_maxloc! (Utils.jl:409-461), clamped 3x3x3 (or 3x3 when nz == 1) window maximum
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_maxloc(double *__restrict__ B, const double *__restrict__ A, int nx, int ny, int nz)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = t / nx, i = t - j * nx, k = blockIdx.y;
    if (j >= ny) return;
    double x = -INFINITY;
    for (int kk = k - 1; kk <= k + 1; kk++) {
        const int kc = clampi(kk, 0, nz - 1);
        for (int jj = j - 1; jj <= j + 1; jj++) {
            const int jc = clampi(jj, 0, ny - 1);
            for (int ii = i - 1; ii <= i + 1; ii++) {
                const int ic = clampi(ii, 0, nx - 1);
                const double v = A[ic + (i64)nx * jc + (i64)nx * ny * kc];
                if (v > x) x = v;
            }
        }
    }
    B[i + (i64)nx * j + (i64)nx * ny * k] = x;
}

// ------------------------------------------------------------------------------------------------
// Σx² reductions: wave64 shuffle -> LDS across the 4 waves -> one partial per block -> final pass
// in a single block (bitwise reproducible run to run; no float atomics).
// arrays 0..2: interior slice [1, n-1) in every dim; array 3: everything.
// ------------------------------------------------------------------------------------------------
struct RedArr {
    const double *p;
    int n[3];
    int inner;
};

__global__ __launch_bounds__(256) void k_sumsq_partial(RedArr A0, RedArr A1, RedArr A2, RedArr A3, double *__restrict__ partials)
{
    const RedArr arrs[4] = {A0, A1, A2, A3};
    __shared__ double sm[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i64 stride = (i64)gridDim.x * blockDim.x;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const RedArr &A = arrs[c];
        double s = 0.0;
        if (A.p) {
            const i64 n = (i64)A.n[0] * A.n[1] * A.n[2];
            for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
                bool take = true;
                if (A.inner) {
                    const int i = (int)(t % A.n[0]);
                    const i64 q = t / A.n[0];
                    const int j = (int)(q % A.n[1]);
                    const int k = (int)(q / A.n[1]);
                    take = i > 0 && i < A.n[0] - 1 && j > 0 && j < A.n[1] - 1 && (A.n[2] == 1 || (k > 0 && k < A.n[2] - 1));
                }
                if (take) {
                    const double v = A.p[t];
                    s += v * v;
                }
            }
        }
        s = wave_sum(s);
        if (lane == 0) sm[c][wave] = s;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int c = threadIdx.x;
        partials[(i64)blockIdx.x * 4 + c] = (sm[c][0] + sm[c][1]) + (sm[c][2] + sm[c][3]);
    }
}

__global__ __launch_bounds__(256) void k_sumsq_final(const double *__restrict__ partials, int nblocks, double *__restrict__ out)
{
    __shared__ double sm[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        double s = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += blockDim.x) s += partials[(i64)b * 4 + c];
        s = wave_sum(s);
        if (lane == 0) sm[c][wave] = s;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int c = threadIdx.x;
        out[c] = (sm[c][0] + sm[c][1]) + (sm[c][2] + sm[c][3]);
    }
}

// max |x| of up to three arrays (compute_dt, Utils.jl:512-519): per-block partial maxima [block][4], then one block reduces them
__global__ __launch_bounds__(256) void k_maxabs_partial(const double *__restrict__ A0, i64 n0, const double *__restrict__ A1, i64 n1,
                                                        const double *__restrict__ A2, i64 n2, double *__restrict__ partials)
{
    __shared__ double sm[3][4];
    const double *A[3] = {A0, A1, A2};
    const i64 n[3] = {n0, n1, n2};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i64 stride = (i64)gridDim.x * blockDim.x;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        double m = 0.0;
        if (A[c])
            for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n[c]; t += stride) {
                const double v = fabs(A[c][t]);
                m = (v > m || v != v) ? v : m;          // NaN propagates, as maximum(abs.(x)) does
            }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_down(m, off, 64);
            m = (o > m || o != o) ? o : m;
        }
        if (lane == 0) sm[c][wave] = m;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        double m = sm[c][0];
        for (int w = 1; w < 4; w++) m = (sm[c][w] > m || sm[c][w] != sm[c][w]) ? sm[c][w] : m;
        partials[(i64)blockIdx.x * 4 + c] = m;
    }
}
__global__ __launch_bounds__(256) void k_maxabs_final(const double *__restrict__ partials, int nblocks, double *__restrict__ out)
{
    __shared__ double sm[3][256];
    for (int c = 0; c < 3; c++) {
        double m = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
            const double v = partials[(i64)b * 4 + c];
            m = (v > m || v != v) ? v : m;
        }
        sm[c][threadIdx.x] = m;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        double m = 0.0;
        for (int t = 0; t < 256; t++) m = (sm[c][t] > m || sm[c][t] != sm[c][t]) ? sm[c][t] : m;
        out[c] = m;
    }
}

__global__ __launch_bounds__(256) void k_copy6(double *d0, const double *s0, i64 n0, double *d1, const double *s1, i64 n1,
                                               double *d2, const double *s2, i64 n2, double *d3, const double *s3, i64 n3,
                                               double *d4, const double *s4, i64 n4, double *d5, const double *s5, i64 n5)
{
    const i64 stride = (i64)gridDim.x * blockDim.x;
    i64 nmax = n0;
    nmax = n1 > nmax ? n1 : nmax; nmax = n2 > nmax ? n2 : nmax; nmax = n3 > nmax ? n3 : nmax;
    nmax = n4 > nmax ? n4 : nmax; nmax = n5 > nmax ? n5 : nmax;
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < nmax; t += stride) {
        if (d0 && t < n0) d0[t] = s0[t];
        if (d1 && t < n1) d1[t] = s1[t];
        if (d2 && t < n2) d2[t] = s2[t];
        if (d3 && t < n3) d3[t] = s3[t];
        if (d4 && t < n4) d4[t] = s4[t];
        if (d5 && t < n5) d5[t] = s5[t];
    }
}


}   // namespace
