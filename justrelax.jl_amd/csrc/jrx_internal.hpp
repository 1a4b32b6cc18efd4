// jrx_internal.hpp -- shared internals of libjrx_hip (handle, error plumbing, index helpers).
// gfx950 / CDNA4 only.  Not part of the public ABI (include/jrx.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cmath>
#include "jrx.h"

struct jrx_comm_state;   // halo.hip
struct jrx_field_pool;   // fieldpool.hip

struct jrx_handle {
    int device = 0;
    hipStream_t stream = nullptr;        // compute stream
    hipStream_t halo_stream = nullptr;   // boundary slabs + pack/unpack + RCCL
    hipEvent_t ev[8] = {};
    double *d_partials = nullptr;        // reduction scratch [kMaxRedBlocks][4]
    double *d_sums = nullptr;            // [16] final sums and flags (device)
    double *h_sums = nullptr;            // [16] pinned host mirror
    double *etatau = nullptr;            // library-owned ητ (capacity etatau_cap doubles)
    size_t etatau_cap = 0;
    jrx_comm_state *comm = nullptr;
    jrx_field_pool *pool = nullptr;      // jrx_field_alloc / jrx_field_free: the state arrays the library hands out, and its own large arrays
    int field_placement = 0;             // option: 0 hipMalloc, 1 physical chunks mapped in shuffled order (virtual memory management), 2 physically contiguous (A/B: the slow rate)
    int field_chunk_mib = 64;            // tuning: size of a physical chunk (0: every array ONE chunk of its own size, no pool); a run that wants the pool sets it to its largest array
    bool field_shuffle = true;           // tuning: 0 = chunks in creation order (A/B of the random dealing itself)
    int scratch_poison = 0;                          // test switch, bit mask: arrays of jrx_dev_alloc are filled with NaNs when they are allocated -- 1 the second state sets, 2 ητ, 4 jrx_field_alloc
    int fused_kz = 0;                                // tuning: chunk depth of the 64 x 8 tile of k_fused3d (0: 12 planes from nz = 384 on, else 8; 8 / 12 force)
    int field_pool_pct = 70;                         // tuning: "field_placement" = 1, chunks >= 128 MiB: the first allocation of a chunk size fills a pool of chunks for that share of the free memory, every array takes random chunks of it; 0 = off
    double *scratch_base[10] = {};       // what hipMalloc returned for scratch[q] (scratch[q] may start scratch_stagger * q bytes into it)
    bool scratch_contiguous = false;     // tuning switch: the second 3D state set in physically contiguous device memory (hipDeviceMallocContiguous)
    int scratch_stagger = 0, scratch_stagger_used = 0;   // tuning switch (bytes; see ensure_scratch) and the value the current allocation was made with
    double *scratch[10] = {};            // ping-pong set for the fused iteration kernel (P, τ(6), V(3))
    int scratch_dims[3] = {0, 0, 0};
    double *tscratch[4] = {};            // ping-pong set of the fused 3D heat-diffusion kernel (T, qTx, qTy, qTz)
    int tscratch_dims[3] = {0, 0, 0};
    double *scratch2d[6] = {};           // ping-pong set of the fused 2D Stokes kernel (P, τxx, τyy, τxy, Vx, Vy)
    int scratch2d_dims[2] = {0, 0};
    double *tscratch2[3] = {};           // the same for the 2D loop (T, qTx, qTy)
    int tscratch2_dims[2] = {0, 0};
    // ---- options (jrx_set_option; nothing in the library reads the process environment)
    bool loop_graphs = true;             // launch-bound 2D loops: runs of unobserved iterations replay as captured hipGraphs (option "loop_graphs")
    bool thermal_fused = true;           // heat diffusion: one fused launch per unobserved iteration (option "thermal_fused")
    int fused_overlap = 3;               // multi-rank fused pipeline: 3 (= 4; viscous-limit form) the kernel's own boundary tiles read the received planes -- a second launch of the kernel over those
                                         // tiles behind update_halo!(V) -- no BC launch, no fix-up, on every rank; 0 exchange behind the kernel, in order; 1 shell tiles + exchange on the halo stream, interior tiles
                                         // concurrently; 2 boundary slabs of the velocity phase + BCs + the whole exchange on the halo stream beside the kernel (early exchange; what 3 falls back to for finite dt)
    int kernel_variant = 0;              // 0 auto (fused PT pipeline where it pays), 1 per-node v1 kernels, 2 z-marching sweeps only, 3 fused wherever legal
    bool fused_split = false;            // no neighbours: high-face tiles + boundary stress layers on the halo stream, interior tiles concurrently
                                         // (measured slower, profiles/r02_ab_fused_split.txt: off)
    int fused_tile = 2;                  // fused kernel tile: 0 = 64 x 4 threads, 1 = 32 x 8, 2 = by nx (32 x 8 for nx = 63 .. 90)
    bool fused_comm = true;              // multi-rank runs use the fused pipeline (0: split sweeps + hidden communication)
    bool vep_store_all = false;          // VEP loops: every iteration stores the output-only arrays (A/B of the skipped stores)
    bool viscous_limit = true;           // dt = Inf: the fused 3D kernel skips the operands multiplied by 1/(G dt) = 1/(K dt) = 1/dt = 0
    bool visc_ok = false;                // set per driver call by the operand check: every τ_o, P0, Q finite and K, G neither NaN nor 0, so the viscous-limit kernels give the general ones' result
    int fused_first_pct = 15;            // in-kernel neighbour faces: share of the interior z chunks launched beside the exchange (the rest of the block follows behind it)
    bool comm_bcs_lazy = false;          // multi-rank fused pipeline: 1 = flow_bcs! of the physical faces is not applied in memory every iteration (the fix-up derives those entries by rule) but
                                         // lazily, before anything reads them.  Measured (two 512^3 blocks, profiles/r04_comm_bcs_lazy_ab.txt): the rule form of the fix-up costs more than the two
                                         // BC launches it saves (-2 %): off
    bool end_flips = true;               // jrx_stokes3d_iterate_timed: a batch with an odd number of fused steps ends in the caller's arrays through out-of-place end sweeps (0: first step un-fused)
    bool zero_forces = true;             // viscous-limit one-launch kernel: body-force arrays whose every entry is +0.0 (all bits zero; the operand pass looks) are not loaded (k_fused3d, NOF; same bits)
    // option "operand_cache": the verdict of the operand pass (visc_ok, nof) is kept per set of operand pointers, extents and dt and reused by the next driver call until the caller
    // declares the operand arrays changed (jrx_fields_dirty) or the library writes one of them itself
    bool operand_cache = false;
    struct { bool valid = false; const void *ptr[14] = {}; int64_t n[3] = {}; double dt = 0.0; int flags = 0; bool visc_ok = false; int nof = 0; } opv;
    int64_t stat_operand_cache_hits = 0;
    int nof = 0;                         // set per driver call by the operand pass: 0 = every ρg array is loaded, 1 = ρg_x and ρg_y hold only +0.0, 2 = all three do
    bool visc_fold = true;               // viscous-limit fused kernel: the arithmetic with the exact zeros folded away (one division per thread for dτ_r, no division by 1 in compute_P!; same bits; A/B)
    bool fused_hiface = true;            // viscous-limit fused kernel without neighbours: the high-face node layers inside the kernel (0: the boundary-layer launch behind it, A/B)
    bool fused_ylds = true;              // fused kernel: y-neighbour operands through LDS (0: lane-shuffle-only form, A/B)
    int general_hif = 0;                 // general (any dt) fused kernel, 64 x 4 tile: the high-face node layers inside the kernel and, with neighbours, the in-kernel faces: 4 / 3 = built for that many
                                         // waves per SIMD, 0 = the boundary-layer launch behind the kernel and the early exchange (the pipeline of rounds 1-4, default: measured faster)
    int64_t stat_fused3d_general_hif = 0;
    bool nbr_feeder = true;              // tuning (round 6): in-kernel neighbour faces, low x face: the idle feeder lane holds the received plane (k_fused3d, feedL); 0 = column 0 loads it itself behind the barrier
    int fused_ym = 0;                    // tuning (round 6): one-launch viscous-limit kernel, 64 x 8 tile: a block marches this many tile rows in y (0 / 1: one tile per block)
    int64_t stat_fused3d_ym = 0;         // launches of the y-marching form
    int b_width_opt[3] = {0, 0, 0};      // > 0 overrides jrx_stokes3d_params.b_width (tuning; the split does not change results)
    bool halo_self_rccl = false;         // test hook: a rank that is its own periodic neighbour routes the planes through ncclSend/ncclRecv
    int thermal_cfg = 0, thermal_xg = 8; // fused 3D heat-diffusion tile shape / XCD band override (tuning)
    bool thermal_nt = false;             // tuning (round 6): fused 3D heat-diffusion kernels store the new (T, qT) set with non-temporal stores
    int thermal_tile = 0;                // tuning (round 6): fused 3D heat diffusion, array / rheology form: 4 / 8 = 64 x TY tiles with the rows j +- 1 through LDS (k_thermal3d_fused_t)
    bool fused2d = true;                 // 2D visco-elastic loop: one-launch iterations on launch-bound grids
    bool vep3_peel = true;                   // z-marching edge kernel: a nearly empty last lane segment goes to the node kernel (A/B)
    bool vep3_peel_fork = false;             // ... that thin launch on the halo stream beside the main edge kernel (measured 1 % slower at 256^3: off; A/B)
    bool vep3_nt = false;                    // 3D VEP kernels: non-temporal stores of the outputs (measured neutral at 256^3: off; A/B)
    int vep3_prekz = 0;                      // planes per thread of k_vep3_pre: 0 = by the number of blocks (default), 1 .. 32 force a depth (A/B)
    int vep3_cfg = 0;                        // z-marching edge kernel: KZ * 10 + min blocks per CU, 0 = default
    int vep3_edges = 4;                      // 3D VEP edge pass: 4 z-marching kernel, the three family waves of a row share the centre and shear operands through LDS;
                                             // 3 centre operands only, 1 no LDS (one family per block), 2 one launch per family, 0 one node per thread (A/B)
    int vep3_hide_comm = 0;                  // multi-rank 3D VEP driver: 2 = the three exchanges of an iteration on the halo stream beside independent kernels; 1 = ητ and the edge stresses only,
                                             // update_halo!(V) behind the whole velocity sweep; 0 (default) = everything in order on the compute stream.  Two 256^3 blocks on one device, with the fused
                                             // pre / centre kernel (no centre pass left to hide the edge-stress exchange behind): +20 % (2), +17 % (1), +10 % (0) over two uncoupled blocks
                                             // (gpurun_out/r04c; before the fusion +9.7 / +7.9 / +8.8 %) -- until a real link shows otherwise the measured-best form is the default
    bool fused2d_batch = true;               // 2D one-launch iteration: the form with every operand requested up front (k_fused2d_b), and its viscous-limit instantiation for dt = Inf
    int fused2d_max_nodes = 1200000;         // ... runs on grids of up to this many nodes (SolCx: faster than the two-kernel iteration up to 1024^2, slower at 1280^2 ... 1536^2; the control-flow form: 200,000)
    bool thermal_fused_ph = true;            // 3D heat diffusion, phase-ratio form with a constant phase count: unobserved iterations as one launch (k_thermal3d_fused_ph)
    bool thermal_np_const = true;            // phase-ratio form of the heat-diffusion kernels (2D and 3D): instantiations with the phase count as a constant (1..4)
    int vep3_prec_tile = 2;                  // fused pre / centre kernel: 1 = 64 x 4 tiles of node columns per block, 0 = 256 consecutive nodes of the flattened plane, 2 (default) = tiles from 16,384 node columns per plane
    bool vep3_np_const = true;               // 3D VEP centre pass (and the fused pre / centre kernel): instantiations with the phase count as a constant (1..4): ratios loaded in one batch, phase loops unrolled
    bool vep3_fuse_pc = true;                // 3D VEP driver without neighbours, linear laws: k_vep3_pre + k_vep3_visc + k_vep3_centre as one kernel ahead of the edge pass (k_vep3_prec; second sets of η and τxx, τyy, τzz)
    bool vep3_fork = false;                  // 3D VEP driver without neighbours: the centre pass of the stress update on the halo stream beside the edge pass (second set of τxx, τyy, τzz).
                                             // Measured 256^3 311.6 / 311.4 / 312.2 it/s forked vs 312.7 / 313.2 / 312.2 in order (profiles/r04_vep3d_fork.txt): nothing to gain, off
    bool vep3_map = true, vep3_xcd = true;   // 3D VEP edge kernel thread mapping / XCD slab order (A/B)
    bool scratch_sets = true;            // the fused pipelines may allocate their library-owned second state set (0: never -- un-fused paths)
    // ---- read-only counters (jrx_get_option "stat_*"): launches of the fused kernels since jrx_create, so that tests and the bench can
    //      prove which kernel path ran
    int64_t stat_fused3d_inkernel = 0;   // launches of k_fused3d that finished the neighbour faces themselves (fused_overlap = 3)
    int64_t stat_fused3d_visc = 0, stat_visc_checks = 0, stat_visc_fallbacks = 0, stat_fused3d_nof1 = 0, stat_fused3d_nof2 = 0;     // launches of the viscous-limit form of k_fused3d; operand checks run / failed
    int64_t stat_fused3d = 0, stat_fused2d = 0, stat_thermal_fused = 0, stat_vep3_fused = 0, stat_graph_replays = 0;
    bool chain_profile = false;          // tuning switch: jrx_stokes3d_iterate_timed also times the stages of a multi-rank fused step (jrx_tuning_chain_profile)
    double chain_us[8] = {};
    int64_t chain_n = 0;
    int comm_timeout_ms = 120000;        // in-process transport: how long a rank waits on the host for a neighbour before it reports an error
    char err[512] = {0};
};

static constexpr int kMaxRedBlocks = 2048;

extern char g_jrx_create_err[512];

static inline jrx_status jrx_fail(jrx_handle *h, jrx_status st, const char *fmt, ...)
{
    char *dst = h ? h->err : g_jrx_create_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return st;
}

#define JRX_HIP(h, call)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return jrx_fail((h), JRX_ERR_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #call,      \
                            hipGetErrorString(e_));                                              \
    } while (0)

#define JRX_TRY(call)                          \
    do {                                       \
        jrx_status s_ = (call);                \
        if (s_ != JRX_OK) return s_;           \
    } while (0)

#define JRX_LAUNCH_CHECK(h) JRX_HIP(h, hipGetLastError())

// the instantiated hipGraphs of one driver call: released on every exit path (error returns included)
struct GraphExecs {
    hipGraphExec_t g[2] = {nullptr, nullptr};
    GraphExecs() = default;
    GraphExecs(const GraphExecs &) = delete;
    GraphExecs &operator=(const GraphExecs &) = delete;
    ~GraphExecs() { reset(); }
    void reset() { for (auto &x : g) if (x) { (void)hipGraphExecDestroy(x); x = nullptr; } }
    hipGraphExec_t &operator[](int i) { return g[i]; }
};

// Capture what `body` enqueues on stream s into an instantiated graph.  *out stays nullptr when the runtime refuses the capture (the caller then
// keeps plain launches); an error of the body itself is returned.  Thread-local capture mode: other host threads (other handles) are not affected.
template <class F>
static inline jrx_status jrx_capture_graph(hipStream_t s, hipGraphExec_t *out, F &&body)
{
    *out = nullptr;
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return JRX_OK; }
    const jrx_status st = body();
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(s, &g);
    if (st == JRX_OK && e == hipSuccess && g && hipGraphInstantiate(out, g, nullptr, nullptr, 0) != hipSuccess) *out = nullptr;
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    return st;
}
// grids up to this many cells run their 3D loops launch-bound (a dependent launch costs ~5 us): runs of unobserved iterations replay as graphs.
// Measured (profiles/r03_small_grids_graphs.txt): +3 .. +8 % at 16^3 and 32^3, nothing at 64^3, -7 % for the heat loop at 96^3 -> up to 48^3 cells.
static constexpr double kGraphCells3D = 48.0 * 48.0 * 48.0;

// A handle is bound to one device (jrx_create); entry points that launch or allocate require that device to be the calling thread's
// current one -- checked, never changed behind the caller's back
jrx_status jrx_check_device(jrx_handle *h);

// fieldpool.hip: where every large library-owned array comes from (the handle's placement option applies); freed with the handle at the latest
jrx_status jrx_dev_alloc(jrx_handle *h, size_t bytes, void **out, int tag = 4);      // tag: which bit of the test switch "scratch_poison" fills it with NaNs (1 second state sets, 2 ητ, 4 jrx_field_alloc)
jrx_status jrx_dev_free(jrx_handle *h, void *p);
void jrx_pool_destroy(jrx_handle *h);

// ensure the library-owned ητ scratch holds n doubles
jrx_status jrx_ensure_etatau(jrx_handle *h, size_t n);

// reduction helper: Σ over a wave64 via DPP-free shuffles
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

typedef long long i64;

__device__ __forceinline__ i64 clampll(i64 v, i64 lo, i64 hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// src/rheology/StressUpdate.jl:70
__device__ __forceinline__ double dev_dtau_r(double theta_dtau, double eta, double _Gdt)
{
    return 1.0 / (theta_dtau + fma(eta, _Gdt, 1.0));
}
// src/stokes/StressKernels.jl:2-5
__device__ __forceinline__ double dev_stress_inc(double t, double to, double eta, double e, double _Gdt, double dtr)
{
    return dtr * fma(2.0 * eta, e, fma(-(t - to) * eta, _Gdt, -t));
}

// halo.hip: exchange of the velocity (or any) fields; no-op without a communicator
jrx_status jrx_halo_exchange(jrx_handle *h, hipStream_t s, int narrays, double *const *arrays,
                             const int64_t (*ext)[3], const int64_t n[3]);
bool jrx_comm_active(const jrx_handle *h);
// all-reduce (sum) of `count` doubles in place on the host values (uses RCCL when active)
jrx_status jrx_allreduce_sum_host(jrx_handle *h, double *vals, int count);
jrx_status jrx_allreduce_host(jrx_handle *h, double *vals, int count, int op);   // op: 0 sum, 1 max
int jrx_comm_rank(const jrx_handle *h);
void jrx_comm_set_timeout(jrx_handle *h, double seconds);          // in-process group of the handle (no-op without one)
bool jrx_comm_has_neighbor(const jrx_handle *h, int d, int side);   // the halo exchange receives into that boundary plane

// stokes3d.hip: pieces of the 3D visco-elastic path that the 3D VEP driver (stokes3d_vep.hip) reuses.  Asynchronous on `s`.
// velocity sweep = compute_V! 3D (+ residuals when diag); sumsq leaves Σx² of Rx, Ry, Rz (interior slices) and RP in h->d_sums
// nof: body-force arrays that the caller has just seen to hold only +0.0 (jrx3d_forces_zero) and that the unobserved sweep therefore does not load (1: fx, fy; 2: all three)
jrx_status jrx3d_velocity_sweep(jrx_handle *h, hipStream_t s, const jrx_stokes3d_fields *f, const double *etatau, const jrx_stokes3d_params *p, bool diag, int nof = 0);
jrx_status jrx3d_forces_zero(jrx_handle *h, hipStream_t s, const double *fx, const double *fy, const double *fz, int64_t nc, int *nof);
// the same sweep as @hide_communication runs it (boundary slabs, BCs and update_halo!(V) on the halo stream, interior on the compute stream); see stokes3d.hip
jrx_status jrx3d_velocity_hidden(jrx_handle *h, const jrx_stokes3d_fields *f, const double *etatau, const jrx_stokes3d_params *p, bool diag, int bc_kind);
jrx_status jrx3d_scaleU(jrx_handle *h, hipStream_t s, const jrx_stokes3d_fields *f, const jrx_stokes3d_params *p);
jrx_status jrx3d_bcs(jrx_handle *h, hipStream_t s, double *Vx, double *Vy, double *Vz, int nx, int ny, int nz, uint32_t fs, uint32_t ns, uint32_t pe);
// all free-slip / no-slip faces in one launch: equal to jrx3d_bcs on every entry a Stokes stencil reads once jrx3d_bcs has run on the same arrays
jrx_status jrx3d_bcs_faces(jrx_handle *h, hipStream_t s, double *Vx, double *Vy, double *Vz, int nx, int ny, int nz, uint32_t fs, uint32_t ns);
jrx_status jrx3d_sumsq(jrx_handle *h, hipStream_t s, const jrx_stokes3d_fields *f, const jrx_stokes3d_params *p);
