// halo.hip -- neighbour halo exchange over RCCL (xGMI) and the norm all-reduce.
//
// Replaces ImplicitGlobalGrid's update_halo! (MPI) at the reference's call sites
// src/stokes/Stokes3D.jl:57,120, src/stokes/Stokes2D.jl:209,268,
// src/thermal_diffusion/DiffusionPT_solver.jl:110, and norm_mpi's MPI.Allreduce
// (src/Utils.jl:698-701).  One process per GPU; per dimension (x, then y, then z, so that edges
// propagate) every array's boundary plane is packed into one contiguous buffer per side, exchanged
// with ncclSend/ncclRecv inside one group, and unpacked into the ghost plane.  librccl is
// dlopen'ed on first use so that single-GPU users never load it.
//
// Second transport, "local" (jrx_comm_init_local): the ranks are handles of ONE process, each driven by its own host thread, on one
// device or on peer-accessible devices.  The packed planes are pushed straight into the neighbour's receive buffer with
// hipMemcpyAsync / hipMemcpyPeerAsync (copy engines: no send/recv kernel holding CUs beside the interior kernel), ordered by events:
// the receiver's stream waits on the sender's "sent" event before unpacking, the sender's stream waits on the receiver's "unpacked"
// event of the previous exchange before overwriting the buffer.  Which records have been issued is agreed through per-link sequence
// numbers under the group's mutex; the host never waits for the device.  Norms are all-reduced on the host in rank order.
//
// Third transport, "ipc" (jrx_comm_init_ipc): the same push protocol between PROCESSES of one node -- one process per rank, the reference's own model
// (mpiexec -n N, test/runtests.jl:73-90; what the Julia extension's MPI ranks would use).  Every rank owns one receive buffer per (dimension, side) in
// uncached device memory and exports it with hipIpcGetMemHandle; the neighbour maps it (hipIpcOpenMemHandle) and pushes its packed planes into it with
// hipMemcpyAsync on the exchange's stream (copy engines / xGMI, no send/recv kernel beside the interior kernel).  Ordering is by sequence flags in a small
// POSIX shared-memory segment that every rank maps and registers with HIP: the sender's stream posts `sent = k` behind its copy, the receiver's stream
// waits for it (a one-wave kernel polling the flag, with a time-out) before it unpacks and then posts `unpacked = k`, which the sender's stream waits for
// before it overwrites the buffer in exchange k + 1.  The host only meets the neighbour's host once per exchange (is its buffer large enough?) and never
// waits for a device.  Norms are all-reduced through the same segment in rank order, so every rank holds the same bits.
#include "jrx_internal.hpp"
#include "ipc_ctl.hpp"
#include "local_group.hpp"
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <time.h>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <vector>
#include <rccl/rccl.h>

struct jrx_comm_state;
static constexpr int kMaxLocalRanks = 64;
// the ranks of an in-process group (local transport): local_group.hpp (free of HIP, shared with the CPU sanitizer harness)
typedef jrx_local::Group jrx_local_group;
static_assert(jrx_local::kMaxRanks == kMaxLocalRanks, "one rank limit");

// ---- ipc transport: the control segment (POSIX shared memory, mapped by every rank and registered with HIP so that device kernels can poll / post the flags).
// Its layout and the whole host-side protocol (attach, failure flag, waits with a time-out, all-reduce) live in ipc_ctl.hpp, free of HIP, so that the CPU sanitizer
// job (tests/host/ctl_harness.cpp) builds exactly this code with -fsanitize=thread / address,undefined.
typedef jrx_ipc::Link IpcLink;
typedef jrx_ipc::Ctl IpcCtl;
static_assert(sizeof(hipIpcMemHandle_t) == sizeof(IpcLink::mem), "IpcLink::mem holds a hipIpcMemHandle_t");
static_assert(jrx_ipc::kMaxRanks == kMaxLocalRanks, "one rank limit for the in-process and the ipc transport");

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
    // ---- local transport (jrx_comm_init_local); sequence numbers are guarded by grp->m
    jrx_local_group *grp = nullptr;
    int device = 0;
    double *lrbuf[3][2] = {};          // receive buffer per (dimension, side): written by the neighbour's copy
    size_t lcap[3][2] = {};
    hipEvent_t ev_sent[3][2] = {}, ev_unpacked[3][2] = {};
    uint64_t ready[3][2] = {}, sent[3][2] = {};   // exchanges entered (buffer large enough, previous unpack recorded) / pushed, per link
    int64_t stat_bytes_pushed = 0;
    // ---- ipc transport (jrx_comm_init_ipc)
    IpcCtl *ctl = nullptr, *ctl_dev = nullptr;     // the control segment as the host / the device of this rank sees it
    bool ctl_registered = false;
    double *irbuf[3][2] = {};          // my receive buffers (uncached device memory, exported)
    size_t icap[3][2] = {};
    double *peer_buf[3][2] = {};       // the neighbour's receive buffer behind my face (dimension, side), mapped into this process
    uint64_t peer_gen[3][2] = {};
    uint64_t ik[3][2] = {};            // exchanges entered per face
    struct Retired { double *p; uint64_t gen; int dim, side; };
    std::vector<Retired> retired;      // receive buffers replaced by larger ones: freed once the neighbour has closed its mapping of that generation (IpcLink::closed_gen)
    double ipc_timeout_s = 120.0;
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
    return ((h->comm->comm || h->comm->grp || h->comm->ctl) && h->comm->cart.nprocs > 1) || has_self_neighbor(h->comm->cart);
}
int jrx_comm_rank(const jrx_handle *h) { return (h && h->comm) ? h->comm->cart.rank : 0; }
void jrx_comm_set_timeout(jrx_handle *h, double seconds)
{
    if (!h || !h->comm) return;
    h->comm->ipc_timeout_s = seconds;
    if (!h->comm->grp) return;
    std::lock_guard<std::mutex> lk(h->comm->grp->m);
    h->comm->grp->timeout_s = seconds;
}
// does update_halo! overwrite the boundary plane of dimension d on side 0 (low) / 1 (high)?
bool jrx_comm_has_neighbor(const jrx_handle *h, int d, int side) { return jrx_comm_active(h) && h->comm->cart.neighbor[d][side] >= 0; }


// ------------------------------------------------------------------------------------------------ local transport
// wait (under the group's mutex) until pred() holds; an absent peer is an error after timeout_s, never a hang
template <class Pred>
static jrx_status local_wait(jrx_handle *h, jrx_local_group *g, std::unique_lock<std::mutex> &lk, Pred pred, const char *what)
{
    const jrx_local::Status st = jrx_local::wait(g, lk, pred);
    if (st == jrx_local::FAILED) return jrx_fail(h, JRX_ERR_RCCL, "local transport: a rank of the group failed or left while this one waited for %s", what);
    if (st == jrx_local::TIMEOUT)
        return jrx_fail(h, JRX_ERR_RCCL, "local transport: timed out after %.0f s waiting for %s (every rank of the group must be driven "
                                          "by its own host thread, all in the same call sequence)", g->timeout_s, what);
    return JRX_OK;
}

// norm_mpi / maximum_mpi over the ranks of an in-process group: deposit, barrier, combine in rank order (every rank gets the same bits) -- jrx_local::allreduce
static jrx_status local_allreduce(jrx_handle *h, jrx_comm_state *c, double *vals, int count, int op)
{
    jrx_local_group *g = c->grp;
    const jrx_local::Status st = jrx_local::allreduce(g, c->cart.rank, vals, count, op);
    if (st == jrx_local::FAILED) return jrx_fail(h, JRX_ERR_RCCL, "local transport: the group has failed, or a rank left while this one waited for the all-reduce of the other ranks");
    if (st == jrx_local::TIMEOUT) return jrx_fail(h, JRX_ERR_RCCL, "local transport: timed out after %.0f s waiting for the all-reduce of the other ranks", g->timeout_s);
    return JRX_OK;
}

// one dimension of update_halo! between handles of this process.  The send planes are already packed into c->sbuf[side] on `s`.
// a rank that leaves an exchange half way (a HIP error, a size mismatch) breaks the group for everybody: the others' waits return at once instead of timing out
struct LocalFailGuard {
    jrx_local_group *g;
    bool ok = false;
    ~LocalFailGuard()
    {
        if (ok) return;
        { std::lock_guard<std::mutex> lk(g->m); g->failed = true; }
        g->cv.notify_all();
    }
};

static jrx_status local_exchange_dim(jrx_handle *h, jrx_comm_state *c, hipStream_t s, int dim, const int nb[2], size_t total, PlaneSet &S, int gx, int na)
{
    jrx_local_group *g = c->grp;
    LocalFailGuard guard{g};
    uint64_t k[2] = {0, 0};
    for (int side = 0; side < 2; side++) {
        if (nb[side] < 0) continue;
        if (c->lcap[dim][side] < total) {
            // nobody is copying into the old buffer: its last copy was waited for by the unpack behind it on this handle's streams
            JRX_HIP(h, hipStreamSynchronize(h->stream));
            JRX_HIP(h, hipStreamSynchronize(h->halo_stream));
            if (c->lrbuf[dim][side]) JRX_HIP(h, hipFree(c->lrbuf[dim][side]));
            c->lrbuf[dim][side] = nullptr; c->lcap[dim][side] = 0;
            JRX_HIP(h, hipMalloc(&c->lrbuf[dim][side], total * sizeof(double)));
            c->lcap[dim][side] = total;
        }
    }
    {
        std::lock_guard<std::mutex> lk(g->m);
        for (int side = 0; side < 2; side++)
            if (nb[side] >= 0) k[side] = ++c->ready[dim][side];
    }
    g->cv.notify_all();
    // push: my send plane of `side` lands in the neighbour's receive buffer of the opposite side
    for (int side = 0; side < 2; side++) {
        if (nb[side] < 0) continue;
        const int opp = 1 - side;
        jrx_comm_state *peer = nullptr;
        {
            std::unique_lock<std::mutex> lk(g->m);
            JRX_TRY(local_wait(h, g, lk, [&] { return g->member[nb[side]] && g->member[nb[side]]->ready[dim][opp] >= k[side]; }, "a neighbour to enter update_halo!"));
            peer = g->member[nb[side]];
        }
        if (peer->lcap[dim][opp] < total) return jrx_fail(h, JRX_ERR_ARG, "local transport: the neighbour exchanges %zu values where this rank sends %zu "
                                                                            "(the ranks of a group must call update_halo! with the same arrays)", peer->lcap[dim][opp], total);
        if (k[side] > 1) JRX_HIP(h, hipStreamWaitEvent(s, peer->ev_unpacked[dim][opp], 0));     // its previous unpack from that buffer
        if (peer->device == c->device) JRX_HIP(h, hipMemcpyAsync(peer->lrbuf[dim][opp], c->sbuf[side], total * sizeof(double), hipMemcpyDeviceToDevice, s));
        else JRX_HIP(h, hipMemcpyPeerAsync(peer->lrbuf[dim][opp], peer->device, c->sbuf[side], c->device, total * sizeof(double), s));
        JRX_HIP(h, hipEventRecord(c->ev_sent[dim][side], s));
        c->stat_bytes_pushed += (int64_t)(total * sizeof(double));
        {
            std::lock_guard<std::mutex> lk(g->m);
            c->sent[dim][side] = k[side];
        }
        g->cv.notify_all();
    }
    // receive: the neighbour's push into my buffer of `side`
    for (int side = 0; side < 2; side++) {
        if (nb[side] < 0) continue;
        const int opp = 1 - side;
        jrx_comm_state *peer = nullptr;
        {
            std::unique_lock<std::mutex> lk(g->m);
            JRX_TRY(local_wait(h, g, lk, [&] { return g->member[nb[side]] && g->member[nb[side]]->sent[dim][opp] >= k[side]; }, "a neighbour's planes"));
            peer = g->member[nb[side]];
        }
        JRX_HIP(h, hipStreamWaitEvent(s, peer->ev_sent[dim][opp], 0));
    }
    S.buf[0] = c->lrbuf[dim][0]; S.buf[1] = c->lrbuf[dim][1];
    hipLaunchKernelGGL(k_planes, dim3(gx, na * 2), dim3(256), 0, s, S, 1);
    JRX_LAUNCH_CHECK(h);
    for (int side = 0; side < 2; side++)
        if (nb[side] >= 0) JRX_HIP(h, hipEventRecord(c->ev_unpacked[dim][side], s));
    guard.ok = true;
    return JRX_OK;
}


// ------------------------------------------------------------------------------------------------ ipc transport
namespace {
using jrx_ipc::now_s;
template <class T> inline T ipc_load(const T *p) { return jrx_ipc::ctl_load(p); }
template <class T> inline void ipc_store(T *p, T v) { jrx_ipc::ctl_store(p, v); }
inline void ipc_relax(int spins) { jrx_ipc::relax(spins); }
jrx_status ipc_status(jrx_handle *h, jrx_comm_state *c, jrx_ipc::Status st, const char *what)
{
    if (st == jrx_ipc::OK) return JRX_OK;
    if (st == jrx_ipc::FAILED) return jrx_fail(h, JRX_ERR_RCCL, "ipc transport: a rank of the group failed or timed out while this one waited for %s", what);
    return jrx_fail(h, JRX_ERR_RCCL, "ipc transport: timed out after %.0f s waiting for %s (every rank must make the same sequence of calls)", c->ipc_timeout_s, what);
}
// host wait on the control segment; an absent / failed peer is an error after the time-out, never a hang
template <class Pred>
jrx_status ipc_wait(jrx_handle *h, jrx_comm_state *c, Pred pred, const char *what)
{
    return ipc_status(h, c, jrx_ipc::wait(c->ctl, c->ipc_timeout_s, pred), what);
}

// one lane per flag: wait until *flag >= want.  The flags live in host memory shared by the processes of the node; a peer that never posts is a time-out
// (the group is marked failed and every host call of every rank then returns an error), never a hung queue.
__global__ void k_ipc_wait(const uint64_t *f0, uint64_t w0, const uint64_t *f1, uint64_t w1, uint32_t *failed, uint64_t timeout_ticks)
{
    const uint64_t *f = threadIdx.x == 0 ? f0 : f1;
    const uint64_t w = threadIdx.x == 0 ? w0 : w1;
    if (!f) return;
    const uint64_t t0 = wall_clock64();
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < w) {
        __builtin_amdgcn_s_sleep(64);
        if (__hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return;
        if (wall_clock64() - t0 > timeout_ticks) {
            __hip_atomic_store(failed, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
    }
}
// post *flag = v behind everything the stream has done so far
__global__ void k_ipc_post(uint64_t *f0, uint64_t v0, uint64_t *f1, uint64_t v1)
{
    uint64_t *f = threadIdx.x == 0 ? f0 : f1;
    const uint64_t v = threadIdx.x == 0 ? v0 : v1;
    if (!f) return;
    __threadfence_system();
    __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}   // namespace

static jrx_status ipc_check(jrx_handle *h, jrx_comm_state *c)
{
    if (c->ctl && ipc_load(&c->ctl->failed))
        return jrx_fail(h, JRX_ERR_RCCL, "ipc transport: the group has failed (a rank left, reported an error, or a device-side wait for a neighbour's planes timed out)");
    return JRX_OK;
}
struct IpcFailGuard {
    jrx_comm_state *c;
    bool ok = false;
    ~IpcFailGuard() { if (!ok && c->ctl) ipc_store(&c->ctl->failed, 1u); }
};

// norm_mpi / maximum_mpi over the ranks of the node: deposit, barrier, combine in rank order (every rank gets the same bits) -- jrx_ipc::allreduce
static jrx_status ipc_allreduce(jrx_handle *h, jrx_comm_state *c, double *vals, int count, int op)
{
    JRX_TRY(ipc_check(h, c));
    return ipc_status(h, c, jrx_ipc::allreduce(c->ctl, c->cart.rank, vals, count, op, c->ipc_timeout_s), "the all-reduce of the other ranks");
}

// one dimension of update_halo! between processes.  The send planes are already packed into c->sbuf[side] on `s`.
static jrx_status ipc_exchange_dim(jrx_handle *h, jrx_comm_state *c, hipStream_t s, int dim, const int nb[2], size_t total, PlaneSet &S, int gx, int na)
{
    IpcCtl *ctl = c->ctl, *dctl = c->ctl_dev;
    const int me = c->cart.rank;
    JRX_TRY(ipc_check(h, c));
    IpcFailGuard guard{c};
    uint64_t k[2] = {0, 0};
    const uint64_t ticks = (uint64_t)(c->ipc_timeout_s * 1e8);           // wall_clock64 counts at 100 MHz
    // 1. my receive buffers
    for (int side = 0; side < 2; side++) {
        if (nb[side] < 0 || c->icap[dim][side] >= total) continue;
        // nobody copies into the old buffer any more: every copy into it was waited for by the unpack behind it on this rank's streams
        JRX_HIP(h, hipStreamSynchronize(h->stream));
        JRX_HIP(h, hipStreamSynchronize(h->halo_stream));
        // the neighbour may still have the old buffer mapped (it closes the mapping when it sees the new generation): the old one is retired and freed once the
        // neighbour has said so (closed_gen), or with the communicator
        if (c->irbuf[dim][side]) c->retired.push_back({c->irbuf[dim][side], ipc_load(&ctl->link[me][dim][side].buf_gen), dim, side});
        c->irbuf[dim][side] = nullptr; c->icap[dim][side] = 0;
        // uncached (fine-grained) device memory: the planes are written by another process's copy and read here behind a flag, not behind a kernel boundary the
        // runtime knows about -- the L2 of this device must not serve them from a stale line
        JRX_HIP(h, hipExtMallocWithFlags((void **)&c->irbuf[dim][side], total * sizeof(double), hipDeviceMallocUncached));
        c->icap[dim][side] = total;
        IpcLink &L = ctl->link[me][dim][side];
        hipIpcMemHandle_t mh;
        JRX_HIP(h, hipIpcGetMemHandle(&mh, c->irbuf[dim][side]));
        jrx_ipc::publish_buffer(L, &mh, (uint64_t)total);
    }
    for (size_t q = 0; q < c->retired.size();) {          // old receive buffers whose mapping the neighbour has closed
        const auto r = c->retired[q];
        if (ipc_load(&ctl->link[me][r.dim][r.side].closed_gen) >= r.gen) { JRX_HIP(h, hipFree(r.p)); c->retired.erase(c->retired.begin() + (long)q); }
        else q++;
    }
    // 2. enter
    for (int side = 0; side < 2; side++)
        if (nb[side] >= 0) { k[side] = ++c->ik[dim][side]; jrx_ipc::enter(ctl->link[me][dim][side], k[side]); }
    // 3. push: my send plane of `side` lands in the neighbour's receive buffer of the opposite side
    uint64_t *wf[2] = {nullptr, nullptr}, wv[2] = {0, 0};
    for (int side = 0; side < 2; side++) {
        if (nb[side] < 0) continue;
        const int opp = 1 - side, peer = nb[side];
        IpcLink &P = ctl->link[peer][dim][opp];
        JRX_TRY(ipc_status(h, c, jrx_ipc::wait_entered(ctl, P, k[side], c->ipc_timeout_s), "a neighbour to enter update_halo!"));
        if (ipc_load(&P.cap) < total)
            return jrx_fail(h, JRX_ERR_ARG, "ipc transport: the neighbour exchanges %llu values where this rank sends %zu (all ranks must call update_halo! with the same arrays)",
                            (unsigned long long)ipc_load(&P.cap), total);
        const uint64_t gen = ipc_load(&P.buf_gen);
        if (gen != c->peer_gen[dim][side]) {
            if (c->peer_buf[dim][side]) {
                JRX_HIP(h, hipStreamSynchronize(h->stream));
                JRX_HIP(h, hipStreamSynchronize(h->halo_stream));
                JRX_HIP(h, hipIpcCloseMemHandle(c->peer_buf[dim][side]));
                c->peer_buf[dim][side] = nullptr;
                ipc_store(&P.closed_gen, c->peer_gen[dim][side]);        // the owner may free that generation's buffer now
            }
            hipIpcMemHandle_t mh;
            memcpy(&mh, (const void *)P.mem, sizeof(mh));
            void *q = nullptr;
            JRX_HIP(h, hipIpcOpenMemHandle(&q, mh, hipIpcMemLazyEnablePeerAccess));
            c->peer_buf[dim][side] = (double *)q;
            c->peer_gen[dim][side] = gen;
        }
        if (k[side] > 1) { wf[side] = &dctl->link[peer][dim][opp].unpacked; wv[side] = k[side] - 1; }     // its previous unpack from that buffer
    }
    if (wf[0] || wf[1]) {
        hipLaunchKernelGGL(k_ipc_wait, dim3(1), dim3(2), 0, s, (const uint64_t *)wf[0], wv[0], (const uint64_t *)wf[1], wv[1], &dctl->failed, ticks);
        JRX_LAUNCH_CHECK(h);
    }
    uint64_t *pf[2] = {nullptr, nullptr};
    for (int side = 0; side < 2; side++) {
        if (nb[side] < 0) continue;
        JRX_HIP(h, hipMemcpyAsync(c->peer_buf[dim][side], c->sbuf[side], total * sizeof(double), hipMemcpyDeviceToDevice, s));
        c->stat_bytes_pushed += (int64_t)(total * sizeof(double));
        pf[side] = &dctl->link[me][dim][side].sent;
    }
    hipLaunchKernelGGL(k_ipc_post, dim3(1), dim3(2), 0, s, pf[0], k[0], pf[1], k[1]);
    JRX_LAUNCH_CHECK(h);
    // 4. receive: the neighbour's push into my buffer of `side`
    for (int side = 0; side < 2; side++) {
        wf[side] = nullptr;
        if (nb[side] >= 0) { wf[side] = &dctl->link[nb[side]][dim][1 - side].sent; wv[side] = k[side]; }
    }
    hipLaunchKernelGGL(k_ipc_wait, dim3(1), dim3(2), 0, s, (const uint64_t *)wf[0], wv[0], (const uint64_t *)wf[1], wv[1], &dctl->failed, ticks);
    JRX_LAUNCH_CHECK(h);
    S.buf[0] = c->irbuf[dim][0]; S.buf[1] = c->irbuf[dim][1];
    hipLaunchKernelGGL(k_planes, dim3(gx, na * 2), dim3(256), 0, s, S, 1);
    JRX_LAUNCH_CHECK(h);
    for (int side = 0; side < 2; side++) pf[side] = nb[side] >= 0 ? &dctl->link[me][dim][side].unpacked : nullptr;
    hipLaunchKernelGGL(k_ipc_post, dim3(1), dim3(2), 0, s, pf[0], k[0], pf[1], k[1]);
    JRX_LAUNCH_CHECK(h);
    guard.ok = true;
    return JRX_OK;
}

static void ipc_teardown(jrx_comm_state *c)
{
    for (int d = 0; d < 3; d++)
        for (int q = 0; q < 2; q++) {
            if (c->peer_buf[d][q]) (void)hipIpcCloseMemHandle(c->peer_buf[d][q]);
            if (c->irbuf[d][q]) (void)hipFree(c->irbuf[d][q]);
            c->peer_buf[d][q] = nullptr; c->irbuf[d][q] = nullptr;
        }
    for (auto &r : c->retired) (void)hipFree(r.p);
    c->retired.clear();
    if (c->ctl) {
        if (c->ctl_registered) (void)hipHostUnregister(c->ctl);
        jrx_ipc::leave(c->ctl, true);
        c->ctl = c->ctl_dev = nullptr;
    }
    (void)hipGetLastError();
}

jrx_status jrx_allreduce_sum_host(jrx_handle *h, double *vals, int count) { return jrx_allreduce_host(h, vals, count, 0); }

// op: 0 sum (norm_mpi), 1 max (maximum_mpi)
jrx_status jrx_allreduce_host(jrx_handle *h, double *vals, int count, int op)
{
    if (!jrx_comm_active(h) || !(h->comm->comm || h->comm->grp || h->comm->ctl) || h->comm->cart.nprocs == 1) return JRX_OK;
    jrx_comm_state *c = h->comm;
    if (count > 8) return jrx_fail(h, JRX_ERR_ARG, "allreduce of more than 8 values");
    if (c->grp) return local_allreduce(h, c, vals, count, op);
    if (c->ctl) return ipc_allreduce(h, c, vals, count, op);
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
            JRX_HIP(h, hipStreamSynchronize(h->stream));        // an earlier exchange may have run on the other stream
            JRX_HIP(h, hipStreamSynchronize(h->halo_stream));
            for (int q = 0; q < 2; q++) {
                if (c->sbuf[q]) JRX_HIP(h, hipFree(c->sbuf[q]));
                if (c->rbuf[q]) JRX_HIP(h, hipFree(c->rbuf[q]));
                c->sbuf[q] = c->rbuf[q] = nullptr;
                JRX_HIP(h, hipMalloc(&c->sbuf[q], (size_t)total * sizeof(double)));
                if (!c->grp && !c->ctl) JRX_HIP(h, hipMalloc(&c->rbuf[q], (size_t)total * sizeof(double)));    // the local / ipc transports receive per (dimension, side)
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
        if (c->grp) {
            const int nb[2] = {left, right};
            JRX_TRY(local_exchange_dim(h, c, s, dim, nb, (size_t)total, S, gx, na));
            continue;
        }
        if (c->ctl) {
            const int nb[2] = {left, right};
            JRX_TRY(ipc_exchange_dim(h, c, s, dim, nb, (size_t)total, S, gx, na));
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

// In-process group: handles[r] becomes rank r of carts[r] (carts[r].rank == r, .nprocs == n).  Afterwards every rank must be driven by
// its own host thread (the entry points rendezvous on the host); devices may be the same one or peer-accessible ones.
jrx_status jrx_comm_init_local(jrx_handle *const *handles, int32_t n, const jrx_cart *carts)
{
    if (!handles || !handles[0]) return JRX_ERR_ARG;
    jrx_handle *h0 = handles[0];
    if (!carts || n < 1 || n > kMaxLocalRanks) return jrx_fail(h0, JRX_ERR_ARG, "jrx_comm_init_local: 1..%d ranks and their carts", kMaxLocalRanks);
    for (int r = 0; r < n; r++) {
        if (!handles[r]) return jrx_fail(h0, JRX_ERR_ARG, "jrx_comm_init_local: handle %d is NULL", r);
        if (carts[r].rank != r || carts[r].nprocs != n) return jrx_fail(h0, JRX_ERR_ARG, "jrx_comm_init_local: carts[%d] is rank %d of %d", r, carts[r].rank, carts[r].nprocs);
        for (int q = 0; q < r; q++)
            if (handles[q] == handles[r]) return jrx_fail(h0, JRX_ERR_ARG, "jrx_comm_init_local: handle %d appears twice", r);
    }
    int prev_device = -1;
    (void)hipGetDevice(&prev_device);
    jrx_local_group *g = new jrx_local_group();
    g->n = n;
    g->timeout_s = handles[0]->comm_timeout_ms * 1e-3;
    jrx_status st = JRX_OK;
    for (int r = 0; r < n && st == JRX_OK; r++) {
        jrx_handle *h = handles[r];
        if (h->comm) st = jrx_comm_destroy(h);
        if (st != JRX_OK) break;
        jrx_comm_state *c = new jrx_comm_state();
        c->cart = carts[r];
        c->grp = g;
        c->device = h->device;
        h->comm = c;
        g->member[r] = c;
        g->refs++;
        auto hip_ok = [&](hipError_t e, const char *what) {
            if (e != hipSuccess && st == JRX_OK) st = jrx_fail(h0, JRX_ERR_HIP, "jrx_comm_init_local: %s -> %s", what, hipGetErrorString(e));
        };
        hip_ok(hipSetDevice(h->device), "hipSetDevice");
        for (int d = 0; d < 3 && st == JRX_OK; d++)
            for (int q = 0; q < 2 && st == JRX_OK; q++) {
                hip_ok(hipEventCreateWithFlags(&c->ev_sent[d][q], hipEventDisableTiming), "hipEventCreate");
                hip_ok(hipEventCreateWithFlags(&c->ev_unpacked[d][q], hipEventDisableTiming), "hipEventCreate");
            }
        // peer access for the pushes into a neighbour on another device
        for (int q = 0; q < r && st == JRX_OK; q++) {
            if (handles[q]->device == h->device) continue;
            int can = 0;
            hip_ok(hipDeviceCanAccessPeer(&can, h->device, handles[q]->device), "hipDeviceCanAccessPeer");
            if (!can && st == JRX_OK) st = jrx_fail(h0, JRX_ERR_UNSUPPORTED, "jrx_comm_init_local: device %d cannot access device %d", h->device, handles[q]->device);
            hipError_t e = hipDeviceEnablePeerAccess(handles[q]->device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) hip_ok(e, "hipDeviceEnablePeerAccess");
            (void)hipGetLastError();
            (void)hipSetDevice(handles[q]->device);
            e = hipDeviceEnablePeerAccess(h->device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) hip_ok(e, "hipDeviceEnablePeerAccess");
            (void)hipGetLastError();
            (void)hipSetDevice(h->device);
        }
    }
    if (st != JRX_OK) {
        bool joined = false;
        for (int r = 0; r < n; r++)
            if (handles[r]->comm && handles[r]->comm->grp == g) { joined = true; (void)jrx_comm_destroy(handles[r]); }      // the last member out deletes the group
        if (!joined) delete g;
    }
    if (prev_device >= 0) (void)hipSetDevice(prev_device);      // init and the destroys above select the handles' devices: the caller's current device is restored
    return st;
}

// id for jrx_comm_init_ipc: 128 bytes that name the group's control segment (rank 0 makes it, the others get it out of band, like the RCCL unique id)
jrx_status jrx_comm_ipc_id(uint8_t id[JRX_UNIQUE_ID_BYTES])
{
    if (!id) return JRX_ERR_ARG;
    memset(id, 0, JRX_UNIQUE_ID_BYTES);
    int fd = open("/dev/urandom", O_RDONLY);
    ssize_t got = fd >= 0 ? read(fd, id, 32) : -1;
    if (fd >= 0) close(fd);
    if (got != 32) {
        uint64_t v[4] = {(uint64_t)getpid(), (uint64_t)time(nullptr), (uint64_t)(now_s() * 1e9), (uint64_t)(uintptr_t)id};
        memcpy(id, v, 32);
    }
    return JRX_OK;
}

jrx_status jrx_comm_init_ipc(jrx_handle *h, const uint8_t id[JRX_UNIQUE_ID_BYTES], const jrx_cart *cart)
{
    if (!h) return JRX_ERR_ARG;
    if (!cart || !id) return jrx_fail(h, JRX_ERR_ARG, "jrx_comm_init_ipc: null argument");
    if (cart->nprocs < 1 || cart->nprocs > kMaxLocalRanks || cart->rank < 0 || cart->rank >= cart->nprocs)
        return jrx_fail(h, JRX_ERR_ARG, "jrx_comm_init_ipc: rank %d of %d (1..%d ranks)", cart->rank, cart->nprocs, kMaxLocalRanks);
    JRX_TRY(jrx_check_device(h));
    if (h->comm) JRX_TRY(jrx_comm_destroy(h));
    jrx_comm_state *c = new jrx_comm_state();
    c->cart = *cart;
    c->device = h->device;
    c->ipc_timeout_s = h->comm_timeout_ms * 1e-3;
    h->comm = c;
    if (cart->nprocs == 1) return JRX_OK;      // no other rank; a periodic dimension is copied locally, norms are local
    char name[64];
    jrx_ipc::segment_name(id, name);
    const size_t bytes = sizeof(IpcCtl);
    {
        const char *what = "";
        if (jrx_ipc::map_segment(name, cart->rank, c->ipc_timeout_s, &c->ctl, &what) != jrx_ipc::OK) {
            const int e = errno;
            delete c;
            h->comm = nullptr;
            return jrx_fail(h, JRX_ERR_RCCL, "jrx_comm_init_ipc: %s (%s): %s", what, name, strerror(e));
        }
    }
    jrx_status st = JRX_OK;
    {
        const jrx_ipc::Status js = jrx_ipc::join(c->ctl, cart->rank, cart->nprocs, h->device, c->ipc_timeout_s);
        if (js == jrx_ipc::MISMATCH) st = jrx_fail(h, JRX_ERR_ARG, "jrx_comm_init_ipc: the segment was made for %u ranks, this cart has %d", c->ctl->nranks, cart->nprocs);
        else st = ipc_status(h, c, js, "every rank to attach to the control segment");
    }
    if (cart->rank == 0) (void)shm_unlink(name);     // every rank has it mapped (or the group failed): the name can go
    if (st == JRX_OK) {
        // the flags are polled and posted by device kernels: pin the segment and map it into the device's address space
        hipError_t e = hipHostRegister(c->ctl, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
        if (e == hipSuccess) { c->ctl_registered = true; e = hipHostGetDevicePointer((void **)&c->ctl_dev, c->ctl, 0); }
        if (e != hipSuccess) st = jrx_fail(h, JRX_ERR_HIP, "jrx_comm_init_ipc: registering the control segment with HIP -> %s", hipGetErrorString(e));
    }
    if (st != JRX_OK) {
        ipc_store(&c->ctl->failed, 1u);
        ipc_teardown(c);
        delete c;
        h->comm = nullptr;
        return st;
    }
    return JRX_OK;
}

jrx_status jrx_comm_count(jrx_handle *h, int32_t *count)
{
    if (!h) return JRX_ERR_ARG;
    if (!count) return jrx_fail(h, JRX_ERR_ARG, "jrx_comm_count: count is NULL");
    *count = 0;
    if (h->comm && h->comm->grp) { *count = h->comm->grp->n; return JRX_OK; }
    if (h->comm && h->comm->ctl) { JRX_TRY(ipc_check(h, h->comm)); *count = (int32_t)ipc_load(&h->comm->ctl->attached); return JRX_OK; }
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
    // the frees below belong to the handle's device; the caller's current device is put back on return
    struct DeviceGuard { int prev = -1; DeviceGuard(int d) { (void)hipGetDevice(&prev); if (prev != d) (void)hipSetDevice(d); else prev = -1; } ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); } } guard(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->halo_stream) (void)hipStreamSynchronize(h->halo_stream);
    if (c->comm && c->CommDestroy) (void)c->CommDestroy(c->comm);
    if (c->grp) {
        // leaving breaks the group for the ranks that remain (their waits return an error); the last one out frees it
        jrx_local_group *g = c->grp;
        bool last;
        {
            std::lock_guard<std::mutex> lk(g->m);
            g->member[c->cart.rank] = nullptr;
            last = --g->refs == 0;
            if (!last) g->failed = true;
        }
        g->cv.notify_all();
        if (last) delete g;
        for (int d = 0; d < 3; d++)
            for (int q = 0; q < 2; q++) {
                if (c->ev_sent[d][q]) (void)hipEventDestroy(c->ev_sent[d][q]);
                if (c->ev_unpacked[d][q]) (void)hipEventDestroy(c->ev_unpacked[d][q]);
                if (c->lrbuf[d][q]) (void)hipFree(c->lrbuf[d][q]);
            }
    }
    if (c->ctl) ipc_teardown(c);     // the streams are idle: every copy into my buffers was unpacked, every copy of mine has completed
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
    if (h->comm && h->comm->ctl) JRX_TRY(ipc_check(h, h->comm));      // a device-side wait that timed out leaves garbage in the ghost planes
    return JRX_OK;
}

}   // extern "C"
