// halo.hip -- neighbour halo exchange over RCCL (xGMI) and the norm all-reduce.
//
// Replaces ImplicitGlobalGrid's update_halo! (MPI) at the reference's call sites
// src/stokes/Stokes3D.jl:57,120, src/stokes/Stokes2D.jl:209,268,
// src/thermal_diffusion/DiffusionPT_solver.jl:110, and norm_mpi's MPI.Allreduce
// (src/Utils.jl:698-701).  One process per GPU; per dimension (x, then y, then z, so that edges
// propagate) every array's boundary plane is packed into one contiguous buffer per side, exchanged
// with ncclSend/ncclRecv inside one group, and unpacked into the ghost plane.  librccl is
// dlopen'ed on first use so that single-GPU users never load it.
#include "jrx_internal.hpp"
#include <dlfcn.h>
#include <cstdlib>
#include <rccl/rccl.h>

struct jrx_comm_state {
    void *lib = nullptr;
    ncclComm_t comm = nullptr;
    jrx_cart cart;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    double *sbuf[2] = {nullptr, nullptr}, *rbuf[2] = {nullptr, nullptr};
    size_t cap = 0;     // doubles per buffer
    double *d_red = nullptr;
    bool self_through_rccl = false;   // option halo_self_rccl = 1: route self-neighbour planes through ncclSend/ncclRecv (test hook)
};

namespace {

static void *g_rccl = nullptr;

template <class F>
bool sym(void *lib, const char *name, F &out)
{
    out = reinterpret_cast<F>(dlsym(lib, name));
    return out != nullptr;
}

jrx_status load_rccl(jrx_handle *h, jrx_comm_state *c)
{
    if (!g_rccl) {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            g_rccl = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl) break;
        }
    }
    if (!g_rccl) return jrx_fail(h, JRX_ERR_RCCL, "cannot dlopen librccl: %s", dlerror());
    c->lib = g_rccl;
    bool ok = sym(g_rccl, "ncclGetUniqueId", c->GetUniqueId) && sym(g_rccl, "ncclCommInitRank", c->CommInitRank) &&
              sym(g_rccl, "ncclCommDestroy", c->CommDestroy) && sym(g_rccl, "ncclSend", c->Send) && sym(g_rccl, "ncclRecv", c->Recv) &&
              sym(g_rccl, "ncclGroupStart", c->GroupStart) && sym(g_rccl, "ncclGroupEnd", c->GroupEnd) &&
              sym(g_rccl, "ncclAllReduce", c->AllReduce) && sym(g_rccl, "ncclGetErrorString", c->GetErrorString) &&
              sym(g_rccl, "ncclCommCount", c->CommCount);
    if (!ok) return jrx_fail(h, JRX_ERR_RCCL, "librccl is missing a required symbol");
    return JRX_OK;
}

#define JRX_NCCL(h, c, call)                                                                       \
    do {                                                                                           \
        ncclResult_t r_ = (call);                                                                  \
        if (r_ != ncclSuccess)                                                                     \
            return jrx_fail((h), JRX_ERR_RCCL, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, (c)->GetErrorString(r_)); \
    } while (0)

struct PlaneDesc {
    double *p;
    int n[3];
    int send_plane[2], recv_plane[2];   // [left,right]
    i64 off;                            // offset (doubles) into the packed buffer
    i64 count;                          // plane size
};
struct PlaneSet {
    PlaneDesc d[8];
    int narr;
    int dim;
    double *buf[2];                     // [left,right] buffers (send for pack, recv for unpack)
    int on[2];
};

// pack (dir=0) the send planes / unpack (dir=1) into the ghost planes; blockIdx.y = array*2 + side
__global__ __launch_bounds__(256) void k_planes(PlaneSet S, int dir)
{
    const int a = blockIdx.y >> 1, side = blockIdx.y & 1;
    if (!S.on[side]) return;
    const PlaneDesc &D = S.d[a];
    const int dim = S.dim;
    const int d1 = dim == 0 ? 1 : 0, d2 = dim == 2 ? 1 : 2;
    const i64 s[3] = {1, D.n[0], (i64)D.n[0] * D.n[1]};
    const int n1 = D.n[d1];
    double *buf = S.buf[side] + D.off;
    const int plane = dir == 0 ? D.send_plane[side] : D.recv_plane[side];
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < D.count; t += (i64)gridDim.x * blockDim.x) {
        const int v = (int)(t / n1), u = (int)(t - (i64)v * n1);
        const i64 idx = plane * s[dim] + u * s[d1] + v * s[d2];
        if (dir == 0) buf[t] = D.p[idx];
        else D.p[idx] = buf[t];
    }
}

}   // namespace

static bool has_self_neighbor(const jrx_cart &c)
{
    for (int d = 0; d < 3; d++)
        if (c.neighbor[d][0] == c.rank || c.neighbor[d][1] == c.rank) return true;
    return false;
}
// true when update_halo! has something to do: other ranks (RCCL) or a periodic dimension held by this rank alone
bool jrx_comm_active(const jrx_handle *h)
{
    if (!h || !h->comm) return false;
    return (h->comm->comm && h->comm->cart.nprocs > 1) || has_self_neighbor(h->comm->cart);
}
int jrx_comm_rank(const jrx_handle *h) { return (h && h->comm) ? h->comm->cart.rank : 0; }
// does update_halo! overwrite the boundary plane of dimension d on side 0 (low) / 1 (high)?
bool jrx_comm_has_neighbor(const jrx_handle *h, int d, int side) { return jrx_comm_active(h) && h->comm->cart.neighbor[d][side] >= 0; }

jrx_status jrx_allreduce_sum_host(jrx_handle *h, double *vals, int count) { return jrx_allreduce_host(h, vals, count, 0); }

// op: 0 sum (norm_mpi), 1 max (maximum_mpi)
jrx_status jrx_allreduce_host(jrx_handle *h, double *vals, int count, int op)
{
    if (!jrx_comm_active(h) || !h->comm->comm || h->comm->cart.nprocs == 1) return JRX_OK;
    jrx_comm_state *c = h->comm;
    if (count > 8) return jrx_fail(h, JRX_ERR_ARG, "allreduce of more than 8 values");
    JRX_HIP(h, hipMemcpyAsync(c->d_red, vals, count * sizeof(double), hipMemcpyHostToDevice, h->stream));
    JRX_NCCL(h, c, c->AllReduce(c->d_red, c->d_red, (size_t)count, ncclDouble, op == 1 ? ncclMax : ncclSum, c->comm, h->stream));
    JRX_HIP(h, hipMemcpyAsync(h->h_sums, c->d_red, count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < count; i++) vals[i] = h->h_sums[i];
    return JRX_OK;
}

jrx_status jrx_halo_exchange(jrx_handle *h, hipStream_t s, int narrays, double *const *arrays,
                             const int64_t (*ext)[3], const int64_t n[3])
{
    if (!jrx_comm_active(h)) return JRX_OK;
    if (narrays < 1 || narrays > 8) return jrx_fail(h, JRX_ERR_ARG, "update_halo!: 1..8 arrays per call");
    jrx_comm_state *c = h->comm;
    for (int dim = 0; dim < 3; dim++) {
        const int left = c->cart.neighbor[dim][0], right = c->cart.neighbor[dim][1];
        if (left < 0 && right < 0) continue;
        PlaneSet S;
        memset(&S, 0, sizeof(S));
        S.dim = dim;
        i64 total = 0;
        int na = 0;
        for (int a = 0; a < narrays; a++) {
            int64_t sl, sr, rl, rr;
            if (jrx_halo_planes(n[dim], ext[a][dim], &sl, &sr, &rl, &rr) != JRX_OK) continue;   // not exchangeable in this dim
            PlaneDesc &D = S.d[na++];
            D.p = arrays[a];
            for (int q = 0; q < 3; q++) D.n[q] = (int)ext[a][q];
            D.send_plane[0] = (int)sl; D.send_plane[1] = (int)sr;
            D.recv_plane[0] = (int)rl; D.recv_plane[1] = (int)rr;
            D.count = 1;
            for (int q = 0; q < 3; q++)
                if (q != dim) D.count *= ext[a][q];
            D.off = total;
            total += D.count;
        }
        if (na == 0) continue;
        S.narr = na;
        S.on[0] = left >= 0; S.on[1] = right >= 0;
        if ((size_t)total > c->cap) {
            JRX_HIP(h, hipStreamSynchronize(s));
            for (int q = 0; q < 2; q++) {
                if (c->sbuf[q]) JRX_HIP(h, hipFree(c->sbuf[q]));
                if (c->rbuf[q]) JRX_HIP(h, hipFree(c->rbuf[q]));
                JRX_HIP(h, hipMalloc(&c->sbuf[q], (size_t)total * sizeof(double)));
                JRX_HIP(h, hipMalloc(&c->rbuf[q], (size_t)total * sizeof(double)));
            }
            c->cap = (size_t)total;
        }
        i64 maxcount = 0;
        for (int a = 0; a < na; a++) maxcount = S.d[a].count > maxcount ? S.d[a].count : maxcount;
        int gx = (int)((maxcount + 255) / 256);
        gx = gx > 1024 ? 1024 : gx;
        S.buf[0] = c->sbuf[0]; S.buf[1] = c->sbuf[1];
        hipLaunchKernelGGL(k_planes, dim3(gx, na * 2), dim3(256), 0, s, S, 0);
        JRX_LAUNCH_CHECK(h);
        const bool self = left == c->cart.rank && right == c->cart.rank;
        if (self && !(c->comm && c->self_through_rccl)) {
            // periodic dimension owned by this rank alone: the left ghost plane takes the right send plane and
            // vice versa (ImplicitGlobalGrid copies locally in this case)
            S.buf[0] = c->sbuf[1]; S.buf[1] = c->sbuf[0];
            hipLaunchKernelGGL(k_planes, dim3(gx, na * 2), dim3(256), 0, s, S, 1);
            JRX_LAUNCH_CHECK(h);
            continue;
        }
        if (!c->comm) return jrx_fail(h, JRX_ERR_RCCL, "update_halo!: no RCCL communicator (jrx_comm_init not called?)");
        // Sends first (left, right), then receives in the order right, left: when both neighbours are the same
        // rank (periodic, 2 ranks in this dimension, or the rank itself) the k-th send to a peer is matched by that
        // peer's k-th receive, and a left-going plane must land in the peer's right ghost plane.
        JRX_NCCL(h, c, c->GroupStart());
        if (left >= 0) JRX_NCCL(h, c, c->Send(c->sbuf[0], (size_t)total, ncclDouble, left, c->comm, s));
        if (right >= 0) JRX_NCCL(h, c, c->Send(c->sbuf[1], (size_t)total, ncclDouble, right, c->comm, s));
        if (right >= 0) JRX_NCCL(h, c, c->Recv(c->rbuf[1], (size_t)total, ncclDouble, right, c->comm, s));
        if (left >= 0) JRX_NCCL(h, c, c->Recv(c->rbuf[0], (size_t)total, ncclDouble, left, c->comm, s));
        JRX_NCCL(h, c, c->GroupEnd());
        S.buf[0] = c->rbuf[0]; S.buf[1] = c->rbuf[1];
        hipLaunchKernelGGL(k_planes, dim3(gx, na * 2), dim3(256), 0, s, S, 1);
        JRX_LAUNCH_CHECK(h);
    }
    return JRX_OK;
}

extern "C" {

jrx_status jrx_comm_unique_id(uint8_t id[JRX_UNIQUE_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) == JRX_UNIQUE_ID_BYTES, "ncclUniqueId size");
    jrx_comm_state tmp;
    JRX_TRY(load_rccl(nullptr, &tmp));
    ncclUniqueId uid;
    ncclResult_t r = tmp.GetUniqueId(&uid);
    if (r != ncclSuccess) return jrx_fail(nullptr, JRX_ERR_RCCL, "ncclGetUniqueId -> %s", tmp.GetErrorString(r));
    memcpy(id, &uid, JRX_UNIQUE_ID_BYTES);
    return JRX_OK;
}

jrx_status jrx_comm_init(jrx_handle *h, const uint8_t id[JRX_UNIQUE_ID_BYTES], const jrx_cart *cart)
{
    if (!h) return JRX_ERR_ARG;
    if (!cart) return jrx_fail(h, JRX_ERR_ARG, "jrx_comm_init: cart is NULL");
    if (h->comm) JRX_TRY(jrx_comm_destroy(h));
    jrx_comm_state *c = new jrx_comm_state();
    c->cart = *cart;
    h->comm = c;
    c->self_through_rccl = h->halo_self_rccl;      // option "halo_self_rccl" (test hook)
    if (cart->nprocs == 1 && !(c->self_through_rccl && id)) return JRX_OK;      // no other rank; norms are local
    if (!id) return jrx_fail(h, JRX_ERR_ARG, "jrx_comm_init: id is NULL");
    JRX_TRY(load_rccl(h, c));
    JRX_HIP(h, hipSetDevice(h->device));
    ncclUniqueId uid;
    memcpy(&uid, id, JRX_UNIQUE_ID_BYTES);
    JRX_NCCL(h, c, c->CommInitRank(&c->comm, cart->nprocs, uid, cart->rank));
    JRX_HIP(h, hipMalloc(&c->d_red, 8 * sizeof(double)));
    return JRX_OK;
}

jrx_status jrx_comm_count(jrx_handle *h, int32_t *count)
{
    if (!h) return JRX_ERR_ARG;
    if (!count) return jrx_fail(h, JRX_ERR_ARG, "jrx_comm_count: count is NULL");
    *count = 0;
    if (!h->comm || !h->comm->comm) return JRX_OK;
    int n = 0;
    JRX_NCCL(h, h->comm, h->comm->CommCount(h->comm->comm, &n));
    *count = n;
    return JRX_OK;
}

jrx_status jrx_comm_destroy(jrx_handle *h)
{
    if (!h || !h->comm) return JRX_OK;
    jrx_comm_state *c = h->comm;
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->halo_stream) (void)hipStreamSynchronize(h->halo_stream);
    if (c->comm && c->CommDestroy) (void)c->CommDestroy(c->comm);
    for (int q = 0; q < 2; q++) {
        if (c->sbuf[q]) (void)hipFree(c->sbuf[q]);
        if (c->rbuf[q]) (void)hipFree(c->rbuf[q]);
    }
    if (c->d_red) (void)hipFree(c->d_red);
    delete c;
    h->comm = nullptr;
    return JRX_OK;
}

jrx_status jrx_update_halo(jrx_handle *h, int32_t narrays, double *const *arrays, const int64_t (*ext)[3], const int64_t n[3])
{
    if (!h) return JRX_ERR_ARG;
    if (!arrays || !ext || !n) return jrx_fail(h, JRX_ERR_ARG, "update_halo!: null argument");
    JRX_TRY(jrx_halo_exchange(h, h->stream, narrays, arrays, ext, n));
    JRX_HIP(h, hipStreamSynchronize(h->stream));
    return JRX_OK;
}

}   // extern "C"
