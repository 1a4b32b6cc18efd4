// local_group.hpp -- the host side of the in-process transport (halo.hip, jrx_comm_init_local): the group every rank's host thread meets in, waits under its mutex with a
// time-out, and the all-reduce in rank order.  No HIP: also built by g++ with -fsanitize=thread into tests/host/ctl_harness.cpp.
#pragma once
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <mutex>

struct jrx_comm_state;
namespace jrx_local {
static constexpr int kMaxRanks = 64;
// the ranks of an in-process group
struct Group {
    std::mutex m;
    std::condition_variable cv;
    int n = 0;
    jrx_comm_state *member[kMaxRanks] = {};
    int refs = 0;
    bool failed = false;                     // a member left or timed out: every wait returns an error instead of blocking
    double slot[2][kMaxRanks][8] = {};       // host all-reduce: the values of generation g live in slot[g & 1]
    int arrived = 0;
    uint64_t gen = 0;
    double timeout_s = 120.0;
};
enum Status { OK = 0, FAILED = 1, TIMEOUT = 2 };

// wait (under the group's mutex) until pred() holds; an absent peer is an error after timeout_s, never a hang
template <class Pred>
inline Status wait(Group *g, std::unique_lock<std::mutex> &lk, Pred pred)
{
    // (ThreadSanitizer builds: the system clock -- libstdc++ then waits through pthread_cond_timedwait, which gcc 11's libtsan intercepts; the steady clock's
    // pthread_cond_clockwait it does not, and it then reports the mutex the wait has released as held twice)
#if defined(__SANITIZE_THREAD__)
    typedef std::chrono::system_clock WaitClock;
#else
    typedef std::chrono::steady_clock WaitClock;
#endif
    const auto deadline = WaitClock::now() + std::chrono::duration_cast<WaitClock::duration>(std::chrono::duration<double>(g->timeout_s));
    while (!pred()) {
        if (g->failed) return FAILED;
        if (g->cv.wait_until(lk, deadline) == std::cv_status::timeout && !pred()) {
            g->failed = true;
            g->cv.notify_all();
            return TIMEOUT;
        }
    }
    return OK;
}

// norm_mpi / maximum_mpi over the ranks of an in-process group: deposit, barrier, combine in rank order (every rank gets the same bits).  op: 0 sum, 1 max
inline Status allreduce(Group *g, int rank, double *vals, int count, int op)
{
    std::unique_lock<std::mutex> lk(g->m);
    if (g->failed) return FAILED;
    const uint64_t gen = g->gen;
    double (*slot)[8] = g->slot[gen & 1];
    for (int i = 0; i < count; i++) slot[rank][i] = vals[i];
    if (++g->arrived == g->n) {
        g->arrived = 0;
        g->gen++;
        g->cv.notify_all();
    } else {
        const Status st = wait(g, lk, [&] { return g->gen != gen; });
        if (st != OK) return st;
    }
    // slot[gen & 1] is overwritten at generation gen + 2 at the earliest, which every rank enters only after this read (it holds the mutex)
    for (int i = 0; i < count; i++) {
        double acc = slot[0][i];
        for (int r = 1; r < g->n; r++) acc = op == 1 ? fmax(acc, slot[r][i]) : acc + slot[r][i];
        vals[i] = acc;
    }
    return OK;
}
}   // namespace jrx_local
