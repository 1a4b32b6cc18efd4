// ipc_ctl.hpp -- the host side of the ipc transport's control segment (halo.hip, jrx_comm_init_ipc): naming, creating / attaching, the
// failure flag, waits with a time-out, and the all-reduce through the segment.  No HIP types and no HIP calls: the same code is compiled by
// hipcc into libjrx_hip.so and by g++ with -fsanitize=thread / address,undefined into tests/host/ctl_harness.cpp (the CPU sanitizer job,
// VERDICT r4 item 8), where threads and forked processes play the ranks.
//
// Memory model: every field another rank reads while this one may write it is accessed through ctl_load / ctl_store (acquire / release) or
// an atomic read-modify-write; plain accesses are confined to fields with a single writer that publishes them behind a release store
// (red_slot behind red_arrived / red_gen, Link::mem and cap behind buf_gen / ready).
#pragma once
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

namespace jrx_ipc {

static constexpr int kMaxRanks = 64;
static constexpr uint64_t kMagic = 0x4a52584950433031ull;      // "JRXIPC01"

struct Link {                    // state of rank r's face (dimension, side): its receive buffer there and its sends through it
    uint64_t ready;              // host of r: exchanges r has entered through this face (its buffer then holds `cap` values)
    uint64_t cap;                // host of r: capacity of the receive buffer (doubles)
    uint64_t buf_gen;            // host of r: bumped whenever the buffer is re-allocated (the neighbour then re-opens `mem`)
    uint8_t mem[64];             // host of r: IPC handle of the receive buffer (hipIpcMemHandle_t; halo.hip asserts the size)
    uint64_t sent;               // stream of r: payloads of r through this face that have landed in the neighbour's buffer
    uint64_t unpacked;           // stream of r: exchanges r has unpacked from its receive buffer of this face
    uint64_t closed_gen;         // host of the NEIGHBOUR behind this face: the last buf_gen of r's buffer whose mapping it has closed (r frees an old buffer only once this has caught up)
    uint64_t pad[1];
};
struct Ctl {
    uint64_t magic;
    uint32_t nranks, attached, failed, left;
    uint64_t red_gen;
    uint32_t red_arrived, pad_;
    double red_slot[2][kMaxRanks][8];
    int32_t device[kMaxRanks];
    Link link[kMaxRanks][3][2];
};

template <class T> inline T ctl_load(const T *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
template <class T> inline void ctl_store(T *p, T v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
inline double now_s()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
inline void relax(int spins)
{
    if (spins < 2000) { sched_yield(); return; }
    timespec ts = {0, 50000};       // 50 us
    nanosleep(&ts, nullptr);
}

enum Status { OK = 0, FAILED = 1, TIMEOUT = 2, SYS = 3, MISMATCH = 4 };

// wait until pred() holds.  FAILED: another rank has marked the group failed; TIMEOUT: this one does so after timeout_s -- an absent peer is an error, never a hang
template <class Pred>
inline Status wait(Ctl *ctl, double timeout_s, Pred pred)
{
    const double t0 = now_s();
    for (int spins = 0; !pred(); spins++) {
        if (ctl_load(&ctl->failed)) return FAILED;
        if (now_s() - t0 > timeout_s) {
            ctl_store(&ctl->failed, 1u);
            return TIMEOUT;
        }
        relax(spins);
    }
    return OK;
}

// norm_mpi / maximum_mpi over the ranks of the node: deposit, barrier, combine in rank order (every rank gets the same bits).  op: 0 sum, 1 max; count <= 8
inline Status allreduce(Ctl *ctl, int me, double *vals, int count, int op, double timeout_s)
{
    if (ctl_load(&ctl->failed)) return FAILED;
    const int n = (int)ctl->nranks;
    const uint64_t gen = ctl_load(&ctl->red_gen);
    double (*slot)[8] = ctl->red_slot[gen & 1];
    for (int i = 0; i < count; i++) slot[me][i] = vals[i];
    if ((int)__atomic_add_fetch(&ctl->red_arrived, 1u, __ATOMIC_ACQ_REL) == n) {
        ctl_store(&ctl->red_arrived, 0u);
        ctl_store(&ctl->red_gen, gen + 1);
    } else {
        const Status st = wait(ctl, timeout_s, [&] { return ctl_load(&ctl->red_gen) != gen; });
        if (st != OK) return st;
    }
    // slot[gen & 1] is written again in generation gen + 2, which starts only after every rank has arrived in gen + 1, i.e. after this read
    for (int i = 0; i < count; i++) {
        double acc = slot[0][i];
        for (int r = 1; r < n; r++) acc = op == 1 ? fmax(acc, slot[r][i]) : acc + slot[r][i];
        vals[i] = acc;
    }
    return OK;
}

// "/jrx_ipc_" + 32 hex digits of the first 16 bytes of the group's id
inline void segment_name(const uint8_t *id, char name[64])
{
    static const char *hx = "0123456789abcdef";
    char *q = name;
    q += snprintf(q, 16, "/jrx_ipc_");
    for (int i = 0; i < 16; i++) { *q++ = hx[id[i] >> 4]; *q++ = hx[id[i] & 15]; }
    *q = 0;
}

// join a control segment that is already mapped at `ctl` (rank 0 initialises it; the memory must be zero or stale from an earlier group -- rank 0 clears it):
// wait for the magic, check the rank count, announce this rank's device, wait for everybody.  MISMATCH: the segment was made for another number of ranks.
inline Status join(Ctl *ctl, int rank, int nprocs, int device, double timeout_s)
{
    if (rank == 0) {
        memset((void *)ctl, 0, sizeof(Ctl));
        ctl->nranks = (uint32_t)nprocs;
        ctl_store(&ctl->magic, kMagic);
    }
    Status st = wait(ctl, timeout_s, [&] { return ctl_load(&ctl->magic) == kMagic; });
    if (st != OK) return st;
    if ((int)ctl->nranks != nprocs) return MISMATCH;
    ctl->device[rank] = device;
    (void)__atomic_add_fetch(&ctl->attached, 1u, __ATOMIC_ACQ_REL);
    return wait(ctl, timeout_s, [&] { return (int)ctl_load(&ctl->attached) >= nprocs; });
}

// map the named POSIX shared-memory segment (rank 0 creates it, the others wait for it to appear with its full size).  On SYS *what names the call that failed (errno is set).
inline Status map_segment(const char *name, int rank, double timeout_s, Ctl **out, const char **what)
{
    const size_t bytes = sizeof(Ctl);
    *out = nullptr;
    int fd = -1;
    const double t0 = now_s();
    if (rank == 0) {
        (void)shm_unlink(name);
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) { *what = "shm_open (create)"; return SYS; }
        if (ftruncate(fd, (off_t)bytes) != 0) { *what = "ftruncate"; close(fd); return SYS; }
    } else {
        for (int spins = 0;; spins++) {
            fd = shm_open(name, O_RDWR, 0600);
            if (fd >= 0) {
                struct stat sb;
                if (fstat(fd, &sb) == 0 && (size_t)sb.st_size >= bytes) break;
                close(fd); fd = -1;
            }
            if (now_s() - t0 > timeout_s) { errno = ETIMEDOUT; *what = "shm_open (rank 0 never created the segment)"; return SYS; }
            relax(spins + 2000);
        }
    }
    void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    const int e = errno;
    close(fd);
    if (m == MAP_FAILED) { errno = e; *what = "mmap"; return SYS; }
    *out = (Ctl *)m;
    return OK;
}

// a rank leaves: the others' waits then fail instead of hanging (the caller has marked `failed` if it leaves a group that is still working)
inline void leave(Ctl *ctl, bool unmap)
{
    (void)__atomic_add_fetch(&ctl->left, 1u, __ATOMIC_ACQ_REL);
    if (unmap) (void)munmap((void *)ctl, sizeof(Ctl));
}

// ---- the per-face handshake of one exchange (host side; the device-side flags `sent` / `unpacked` are posted and polled by kernels on the exchange's stream)
// my receive buffer behind face (dim, side) has been (re)allocated: publish its IPC handle and capacity, then bump the generation the neighbour compares
inline void publish_buffer(Link &L, const void *handle64, uint64_t cap)
{
    memcpy((void *)L.mem, handle64, sizeof(L.mem));
    ctl_store(&L.cap, cap);
    ctl_store(&L.buf_gen, ctl_load(&L.buf_gen) + 1);
}
// enter exchange k through my face
inline void enter(Link &L, uint64_t k) { ctl_store(&L.ready, k); }
// the neighbour's face P: wait until it has entered exchange k; its capacity and buffer generation are then valid for this exchange
inline Status wait_entered(Ctl *ctl, const Link &P, uint64_t k, double timeout_s) { return wait(ctl, timeout_s, [&] { return ctl_load(&P.ready) >= k; }); }

}   // namespace jrx_ipc
