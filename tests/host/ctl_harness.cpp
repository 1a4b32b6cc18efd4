// ctl_harness.cpp -- CPU sanitizer job for the host side of the two copy-engine transports.  Builds exactly the headers libjrx_hip.so is built from
// (justrelax.jl_amd/csrc/ipc_ctl.hpp, local_group.hpp; no HIP) and lets host threads / forked processes play the ranks:
//   threads N K   N threads share one control segment (heap memory, so that ThreadSanitizer sees every access): join, K all-reduces (sum and max, checked), K exchanges along a
//                 ring of faces with a buffer re-allocation on the way -- the `sent` / `unpacked` flags a device kernel posts in the library are posted by the threads here, and
//                 the payload they guard is plain memory, so a missing release / acquire is a reported race --, then the failure paths: a rank that leaves, a rank that never comes
//   procs N K     N forked processes through the real shm_open / mmap path: join, K all-reduces, leave (AddressSanitizer / UBSan build)
//   local N K     N threads in an in-process group (mutex / condition variable): K all-reduces, a rank that leaves, a time-out
// Exit code 0 and "ok" on stdout, or a message and 1.
//   g++ -std=c++17 -O1 -g -fsanitize=thread            -I justrelax.jl_amd/csrc tests/host/ctl_harness.cpp -o ctl_tsan -lpthread -lrt
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -I justrelax.jl_amd/csrc tests/host/ctl_harness.cpp -o ctl_asan -lpthread -lrt
#include "ipc_ctl.hpp"
#include "local_group.hpp"
#include <atomic>
#include <cstdlib>
#include <sys/wait.h>
#include <thread>
#include <vector>

using namespace jrx_ipc;
static std::atomic<int> g_bad{0};
#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); g_bad++; } } while (0)

static double val(int r, int k, int i) { return (double)((r + 1) * 1000 + k) + 0.125 * i; }

static void reduce_loop(Ctl *ctl, int me, int n, int K, double timeout)
{
    for (int k = 0; k < K; k++) {
        for (int op = 0; op < 2; op++) {
            double v[8], want[8];
            for (int i = 0; i < 8; i++) {
                v[i] = val(me, k, i);
                double acc = val(0, k, i);
                for (int r = 1; r < n; r++) acc = op ? fmax(acc, val(r, k, i)) : acc + val(r, k, i);
                want[i] = acc;
            }
            const Status st = allreduce(ctl, me, v, 8, op, timeout);
            CHECK(st == OK, "rank %d all-reduce %d/%d -> %d", me, k, op, (int)st);
            for (int i = 0; i < 8; i++) CHECK(v[i] == want[i], "rank %d all-reduce %d/%d entry %d: %.17g != %.17g", me, k, op, i, v[i], want[i]);
        }
    }
}

// one rank of the ring 0 - 1 - ... - (n-1) along dimension 0: the protocol of ipc_exchange_dim with host threads in the place of the streams
struct Payload { std::vector<double> buf[2]; };          // my receive buffers (side 0: from the left neighbour, side 1: from the right one); plain memory on purpose
static void exchange_loop(Ctl *ctl, std::vector<Payload> &pay, int me, int n, int K, double timeout)
{
    const int nb[2] = {me > 0 ? me - 1 : -1, me + 1 < n ? me + 1 : -1};
    uint64_t peer_gen[2] = {0, 0};
    size_t cap[2] = {0, 0};
    for (uint64_t k = 1; k <= (uint64_t)K; k++) {
        const size_t total = 16 + 8 * (size_t)(k / 7);            // grows now and then: re-allocation + re-publication
        for (int side = 0; side < 2; side++) {
            if (nb[side] < 0 || cap[side] >= total) continue;
            // nobody writes into my old buffer any more: every push into it was unpacked (k - 1) before the neighbour could start pushing k -- it waits for ready >= k first
            pay[me].buf[side].assign(total, -1.0);
            cap[side] = total;
            uint8_t handle[64];
            for (int b = 0; b < 64; b++) handle[b] = (uint8_t)(me * 7 + side * 3 + b + (int)total);
            publish_buffer(ctl->link[me][0][side], handle, (uint64_t)total);
        }
        for (int side = 0; side < 2; side++)
            if (nb[side] >= 0) enter(ctl->link[me][0][side], k);
        for (int side = 0; side < 2; side++) {
            if (nb[side] < 0) continue;
            const int opp = 1 - side, peer = nb[side];
            Link &P = ctl->link[peer][0][opp];
            const Status st = wait_entered(ctl, P, k, timeout);
            CHECK(st == OK, "rank %d exchange %llu: wait_entered -> %d", me, (unsigned long long)k, (int)st);
            if (st != OK) return;
            CHECK(ctl_load(&P.cap) >= total, "rank %d: the neighbour's capacity %llu < %zu", me, (unsigned long long)ctl_load(&P.cap), total);
            const uint64_t gen = ctl_load(&P.buf_gen);
            if (gen != peer_gen[side]) {
                if (peer_gen[side]) ctl_store(&P.closed_gen, peer_gen[side]);
                uint8_t handle[64];
                memcpy(handle, (const void *)P.mem, 64);
                const size_t pcap = (size_t)ctl_load(&P.cap);
                for (int b = 0; b < 64; b++) CHECK(handle[b] == (uint8_t)(peer * 7 + opp * 3 + b + (int)pcap), "rank %d: torn IPC handle of rank %d", me, peer);
                peer_gen[side] = gen;
            }
            // the neighbour has unpacked my previous payload before I overwrite its buffer (k_ipc_wait on `unpacked` in the library)
            if (k > 1) {
                const Status s2 = wait(ctl, timeout, [&] { return ctl_load(&P.unpacked) >= k - 1; });
                CHECK(s2 == OK, "rank %d exchange %llu: waiting for unpacked -> %d", me, (unsigned long long)k, (int)s2);
                if (s2 != OK) return;
            }
            for (size_t q = 0; q < total; q++) pay[peer].buf[opp][q] = (double)(me * 100000 + (int)k * 10) + 1e-3 * (double)q;          // the push (hipMemcpyAsync there)
            ctl_store(&ctl->link[me][0][side].sent, k);                                                                              // k_ipc_post behind it
        }
        for (int side = 0; side < 2; side++) {
            if (nb[side] < 0) continue;
            const Link &Q = ctl->link[nb[side]][0][1 - side];
            const Status st = wait(ctl, timeout, [&] { return ctl_load(&Q.sent) >= k; });
            CHECK(st == OK, "rank %d exchange %llu: waiting for the neighbour's planes -> %d", me, (unsigned long long)k, (int)st);
            if (st != OK) return;
            for (size_t q = 0; q < total; q++)                                                                                       // the unpack
                CHECK(pay[me].buf[side][q] == (double)(nb[side] * 100000 + (int)k * 10) + 1e-3 * (double)q, "rank %d exchange %llu side %d entry %zu: %.17g", me,
                      (unsigned long long)k, side, q, pay[me].buf[side][q]);
            ctl_store(&ctl->link[me][0][side].unpacked, k);
        }
    }
}

static int run_threads(int n, int K)
{
    Ctl *ctl = (Ctl *)aligned_alloc(64, (sizeof(Ctl) + 63) / 64 * 64);
    memset((void *)ctl, 0xA5, sizeof(Ctl));          // stale contents of an earlier group: rank 0 clears them
    ctl->magic = 0;
    std::vector<Payload> pay((size_t)n);
    std::vector<std::thread> th;
    for (int r = 0; r < n; r++)
        th.emplace_back([&, r] {
            const Status st = join(ctl, r, n, r % 8, 20.0);
            CHECK(st == OK, "rank %d join -> %d", r, (int)st);
            if (st != OK) return;
            reduce_loop(ctl, r, n, K, 20.0);
            exchange_loop(ctl, pay, r, n, K, 20.0);
            reduce_loop(ctl, r, n, 3, 20.0);
        });
    for (auto &t : th) t.join();
    CHECK(ctl_load(&ctl->attached) == (uint32_t)n && !ctl_load(&ctl->failed), "attached %u failed %u", ctl->attached, ctl->failed);
    // a rank leaves while the others wait in an all-reduce: they fail at once instead of timing out
    {
        th.clear();
        const double t0 = now_s();
        for (int r = 0; r < n; r++)
            th.emplace_back([&, r] {
                if (r == n - 1) { ctl_store(&ctl->failed, 1u); leave(ctl, false); return; }
                double v[2] = {1.0, 2.0};
                const Status st = allreduce(ctl, r, v, 2, 0, 30.0);
                CHECK(st == FAILED, "rank %d: all-reduce beside a rank that left -> %d", r, (int)st);
            });
        for (auto &t : th) t.join();
        CHECK(now_s() - t0 < 10.0, "the ranks waited %.1f s for a rank that had left", now_s() - t0);
        CHECK(ctl_load(&ctl->left) == 1u, "left = %u", ctl->left);
    }
    // a rank that never comes: the join of the others times out (and marks the group failed)
    if (n > 1) {
        th.clear();
        memset((void *)ctl, 0, sizeof(Ctl));
        for (int r = 0; r < n - 1; r++)
            th.emplace_back([&, r] {
                const Status st = join(ctl, r, n, 0, 0.3);
                CHECK(st == TIMEOUT || st == FAILED, "rank %d: join without rank %d -> %d", r, n - 1, (int)st);
            });
        for (auto &t : th) t.join();
        CHECK(ctl_load(&ctl->failed) == 1u, "the group is not marked failed");
    }
    free(ctl);
    return g_bad.load();
}

static int run_procs(int n, int K)
{
    uint8_t id[16];
    for (int i = 0; i < 16; i++) id[i] = (uint8_t)(getpid() >> (i % 4 * 8)) ^ (uint8_t)(i * 37);
    char name[64];
    segment_name(id, name);
    std::vector<pid_t> kids;
    for (int r = 0; r < n; r++) {
        const pid_t pid = fork();
        if (pid == 0) {
            Ctl *ctl = nullptr;
            const char *what = "";
            if (map_segment(name, r, 20.0, &ctl, &what) != OK) { fprintf(stderr, "rank %d: %s: %s\n", r, what, strerror(errno)); _exit(2); }
            Status st = join(ctl, r, n, r, 20.0);
            if (r == 0) (void)shm_unlink(name);
            if (st != OK) { fprintf(stderr, "rank %d: join -> %d\n", r, (int)st); _exit(3); }
            reduce_loop(ctl, r, n, K, 20.0);
            leave(ctl, true);
            _exit(g_bad.load() ? 1 : 0);
        }
        kids.push_back(pid);
    }
    int bad = 0;
    for (pid_t p : kids) {
        int stt = 0;
        waitpid(p, &stt, 0);
        if (!WIFEXITED(stt) || WEXITSTATUS(stt) != 0) { fprintf(stderr, "child %d ended with status %d\n", (int)p, stt); bad++; }
    }
    // a second group under the same name with another rank count: the late rank reports the mismatch
    return bad;
}

static int run_local(int n, int K)
{
    jrx_local::Group *g = new jrx_local::Group();
    g->n = n; g->refs = n; g->timeout_s = 20.0;
    std::vector<std::thread> th;
    for (int r = 0; r < n; r++)
        th.emplace_back([&, r] {
            for (int k = 0; k < K; k++)
                for (int op = 0; op < 2; op++) {
                    double v[8], want[8];
                    for (int i = 0; i < 8; i++) {
                        v[i] = val(r, k, i);
                        double acc = val(0, k, i);
                        for (int q = 1; q < n; q++) acc = op ? fmax(acc, val(q, k, i)) : acc + val(q, k, i);
                        want[i] = acc;
                    }
                    const jrx_local::Status st = jrx_local::allreduce(g, r, v, 8, op);
                    CHECK(st == jrx_local::OK, "local rank %d all-reduce -> %d", r, (int)st);
                    for (int i = 0; i < 8; i++) CHECK(v[i] == want[i], "local rank %d all-reduce %d/%d entry %d", r, k, op, i);
                }
        });
    for (auto &t : th) t.join();
    th.clear();
    // a rank that leaves (jrx_comm_destroy of one member): the others fail at once
    for (int r = 0; r < n; r++)
        th.emplace_back([&, r] {
            if (r == 0) {
                { std::lock_guard<std::mutex> lk(g->m); g->failed = true; }
                g->cv.notify_all();
                return;
            }
            double v[1] = {1.0};
            const jrx_local::Status st = jrx_local::allreduce(g, r, v, 1, 0);
            CHECK(st == jrx_local::FAILED, "local rank %d beside a rank that left -> %d", r, (int)st);
        });
    for (auto &t : th) t.join();
    th.clear();
    // a rank that never comes: time-out
    if (n > 1) {
        jrx_local::Group *g2 = new jrx_local::Group();
        g2->n = n; g2->timeout_s = 0.3;
        for (int r = 0; r < n - 1; r++)
            th.emplace_back([&, r] {
                double v[1] = {1.0};
                const jrx_local::Status st = jrx_local::allreduce(g2, r, v, 1, 0);
                CHECK(st == jrx_local::TIMEOUT || st == jrx_local::FAILED, "local rank %d without rank %d -> %d", r, n - 1, (int)st);
            });
        for (auto &t : th) t.join();
        delete g2;
    }
    delete g;
    return g_bad.load();
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s threads|procs|local N K\n", argv[0]); return 2; }
    const int n = atoi(argv[2]), K = atoi(argv[3]);
    if (n < 1 || n > kMaxRanks) { fprintf(stderr, "1 .. %d ranks\n", kMaxRanks); return 2; }
    int bad = 0;
    if (!strcmp(argv[1], "threads")) bad = run_threads(n, K);
    else if (!strcmp(argv[1], "procs")) bad = run_procs(n, K);
    else if (!strcmp(argv[1], "local")) bad = run_local(n, K);
    else return 2;
    if (bad) { fprintf(stderr, "%d check(s) failed\n", bad); return 1; }
    printf("ok\n");
    return 0;
}
