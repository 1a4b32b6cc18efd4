// handle.hip -- handle lifetime, error strings, and the host-only block-decomposition helpers.
#include "jrx_internal.hpp"
#include "jrx_tuning.h"

char g_jrx_create_err[512] = {0};

jrx_status jrx_ensure_etatau(jrx_handle *h, size_t n)
{
    if (h->etatau_cap >= n) return JRX_OK;
    if (h->etatau) JRX_TRY(jrx_dev_free(h, h->etatau));
    h->etatau = nullptr;
    h->etatau_cap = 0;
    JRX_TRY(jrx_dev_alloc(h, n * sizeof(double), (void **)&h->etatau, 2));
    h->etatau_cap = n;
    return JRX_OK;
}

jrx_status jrx_check_device(jrx_handle *h)
{
    if (!h) return JRX_ERR_ARG;
    int cur = -1;
    JRX_HIP(h, hipGetDevice(&cur));
    if (cur != h->device)
        return jrx_fail(h, JRX_ERR_ARG, "this handle is bound to device %d but the calling thread's current device is %d (hipSetDevice / torch.cuda.set_device "
                                        "first: the library does not change it)", h->device, cur);
    return JRX_OK;
}

extern "C" {

int32_t jrx_version(void) { return JRX_VERSION; }

// sha256 over csrc/* and include/jrx.h at build time (justrelax.jl_amd/build.py passes -DJRX_BUILD_ID); the marker string lets the
// builder read the id of an existing .so without loading it
#ifndef JRX_BUILD_ID
#define JRX_BUILD_ID "unknown"
#endif
static const char g_jrx_build_marker[] = "JRX_BUILD_ID=" JRX_BUILD_ID;
const char *jrx_build_id(void) { return g_jrx_build_marker + 13; }

const char *jrx_last_error(const jrx_handle *h) { return h ? h->err : g_jrx_create_err; }

jrx_status jrx_create(int32_t device, jrx_handle **out)
{
    if (!out) return jrx_fail(nullptr, JRX_ERR_ARG, "jrx_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return jrx_fail(nullptr, JRX_ERR_HIP, "jrx_create: no HIP device visible");
    if (device < 0 || device >= ndev) return jrx_fail(nullptr, JRX_ERR_ARG, "jrx_create: device %d out of range [0,%d)", device, ndev);
    jrx_handle *h = new jrx_handle();
    h->device = device;
    int prev_device = -1;
    (void)hipGetDevice(&prev_device);       // restored below: creating a handle does not change the caller's current device
#define CK(call)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            jrx_status st_ = jrx_fail(nullptr, JRX_ERR_HIP, "jrx_create: %s -> %s", #call, hipGetErrorString(e_)); \
            (void)jrx_destroy(h);              /* streams, events and buffers created so far */    \
            if (prev_device >= 0) (void)hipSetDevice(prev_device);                                 \
            return st_;                                                                            \
        }                                                                                          \
    } while (0)
    CK(hipSetDevice(device));
    CK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    {
        // the boundary work (slabs / shell tiles, BCs, pack, send/recv, unpack) should win the dispatch race against the interior
        int lo = 0, hi = 0;
        CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        CK(hipStreamCreateWithPriority(&h->halo_stream, hipStreamNonBlocking, hi));
    }
    for (int i = 0; i < 3; i++) CK(hipEventCreateWithFlags(&h->ev[i], hipEventDisableTiming));   // stream joins
    for (int i = 3; i < 8; i++) CK(hipEventCreate(&h->ev[i]));                                    // timing
    CK(hipMalloc(&h->d_partials, sizeof(double) * 4 * kMaxRedBlocks));
    CK(hipMalloc(&h->d_sums, sizeof(double) * 16));         // [0..7] reductions / flags of the drivers, [8] the halo flag of the in-kernel neighbour faces, [9] its time-out mark
    CK(hipMemset(h->d_sums, 0, sizeof(double) * 16));
    CK(hipHostMalloc(&h->h_sums, sizeof(double) * 16, hipHostMallocDefault));
#undef CK
    if (prev_device >= 0 && prev_device != device) (void)hipSetDevice(prev_device);
    *out = h;
    return JRX_OK;
}

jrx_status jrx_destroy(jrx_handle *h)
{
    if (!h) return JRX_OK;
    (void)hipSetDevice(h->device);
    if (h->comm) (void)jrx_comm_destroy(h);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->halo_stream) (void)hipStreamSynchronize(h->halo_stream);
    for (int i = 0; i < 8; i++)
        if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
    if (h->d_partials) (void)hipFree(h->d_partials);
    if (h->d_sums) (void)hipFree(h->d_sums);
    if (h->h_sums) (void)hipHostFree(h->h_sums);
    jrx_pool_destroy(h);                 // ητ, the second 3D state sets and every array jrx_field_alloc handed out and the caller did not free
    for (int q = 0; q < 3; q++)
        if (h->tscratch2[q]) (void)hipFree(h->tscratch2[q]);
    for (int q = 0; q < 6; q++)
        if (h->scratch2d[q]) (void)hipFree(h->scratch2d[q]);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->halo_stream) (void)hipStreamDestroy(h->halo_stream);
    delete h;
    return JRX_OK;
}

// ---------------------------------------------------------------- options
// Two tables: the options of the drop-in ABI (include/jrx.h: what a caller of solve! may want to choose -- kernel path, memory, overlap --
// and the read-only launch counters) and the tuning / test switches of the measurements in profiles/ (include/jrx_tuning.h).
namespace {
struct OptRef { const char *key; int kind; void *p; };     // kind 0: bool, 1: int, 2: read-only int64 counter
int find_opt(jrx_handle *h, const char *key, bool tuning, OptRef *out)
{
    const OptRef pub[] = {
        {"kernel_variant", 1, &h->kernel_variant}, {"fused_overlap", 1, &h->fused_overlap}, {"thermal_fused", 0, &h->thermal_fused},
        {"fused_comm", 0, &h->fused_comm}, {"loop_graphs", 0, &h->loop_graphs}, {"scratch_sets", 0, &h->scratch_sets},
        {"viscous_limit", 0, &h->viscous_limit}, {"field_placement", 1, &h->field_placement}, {"field_chunk_mib", 1, &h->field_chunk_mib}, {"operand_cache", 0, &h->operand_cache}, {"stat_operand_cache_hits", 2, &h->stat_operand_cache_hits}, {"stat_fused3d_general_hif", 2, &h->stat_fused3d_general_hif},
        {"stat_fused3d", 2, &h->stat_fused3d}, {"stat_fused2d", 2, &h->stat_fused2d}, {"stat_thermal_fused", 2, &h->stat_thermal_fused},
        {"stat_vep3_fused", 2, &h->stat_vep3_fused}, {"stat_graph_replays", 2, &h->stat_graph_replays},
        {"stat_fused3d_visc", 2, &h->stat_fused3d_visc}, {"stat_fused3d_inkernel", 2, &h->stat_fused3d_inkernel}, {"stat_visc_checks", 2, &h->stat_visc_checks}, {"stat_visc_fallbacks", 2, &h->stat_visc_fallbacks}, {"stat_fused3d_nof1", 2, &h->stat_fused3d_nof1}, {"stat_fused3d_nof2", 2, &h->stat_fused3d_nof2},
    };
    const OptRef tun[] = {
        {"fused_split", 0, &h->fused_split}, {"fused_tile", 1, &h->fused_tile}, {"fused_ylds", 0, &h->fused_ylds}, {"fused_hiface", 0, &h->fused_hiface}, {"visc_fold", 0, &h->visc_fold}, {"zero_forces", 0, &h->zero_forces}, {"scratch_stagger", 1, &h->scratch_stagger}, {"scratch_contiguous", 0, &h->scratch_contiguous}, {"end_flips", 0, &h->end_flips}, {"comm_bcs_lazy", 0, &h->comm_bcs_lazy}, {"fused_first_pct", 1, &h->fused_first_pct},
        {"b_width_x", 1, &h->b_width_opt[0]}, {"b_width_y", 1, &h->b_width_opt[1]}, {"b_width_z", 1, &h->b_width_opt[2]},
        {"halo_self_rccl", 0, &h->halo_self_rccl}, {"thermal_cfg", 1, &h->thermal_cfg}, {"thermal_xg", 1, &h->thermal_xg}, {"thermal_tile", 1, &h->thermal_tile}, {"thermal_nt", 0, &h->thermal_nt},
        {"fused2d", 0, &h->fused2d}, {"vep3_edges", 1, &h->vep3_edges}, {"vep3_cfg", 1, &h->vep3_cfg}, {"vep3_peel", 0, &h->vep3_peel}, {"vep3_peel_fork", 0, &h->vep3_peel_fork}, {"vep3_nt", 0, &h->vep3_nt}, {"vep3_prekz", 1, &h->vep3_prekz},
        {"vep3_hide_comm", 1, &h->vep3_hide_comm}, {"vep3_fork", 0, &h->vep3_fork}, {"vep3_fuse_pc", 0, &h->vep3_fuse_pc}, {"vep3_np_const", 0, &h->vep3_np_const}, {"vep3_prec_tile", 1, &h->vep3_prec_tile}, {"thermal_np_const", 0, &h->thermal_np_const}, {"thermal_fused_ph", 0, &h->thermal_fused_ph}, {"fused2d_batch", 0, &h->fused2d_batch}, {"fused2d_max_nodes", 1, &h->fused2d_max_nodes}, {"vep3_map", 0, &h->vep3_map}, {"vep3_xcd", 0, &h->vep3_xcd}, {"comm_timeout_ms", 1, &h->comm_timeout_ms}, {"vep_store_all", 0, &h->vep_store_all}, {"chain_profile", 0, &h->chain_profile},
        {"general_hif", 1, &h->general_hif}, {"field_shuffle", 0, &h->field_shuffle}, {"field_pool_pct", 1, &h->field_pool_pct}, {"fused_kz", 1, &h->fused_kz}, {"fused_ym", 1, &h->fused_ym}, {"nbr_feeder", 0, &h->nbr_feeder}, {"stat_fused3d_ym", 2, &h->stat_fused3d_ym}, {"scratch_poison", 1, &h->scratch_poison},
    };
    if (tuning) {
        for (const OptRef &o : tun)
            if (strcmp(o.key, key) == 0) { *out = o; return 1; }
    } else {
        for (const OptRef &o : pub)
            if (strcmp(o.key, key) == 0) { *out = o; return 1; }
    }
    return 0;
}
jrx_status opt_set(jrx_handle *h, const char *key, int64_t value, bool tuning, const char *fn)
{
    if (!h) return JRX_ERR_ARG;
    if (!key) return jrx_fail(h, JRX_ERR_ARG, "%s: key is NULL", fn);
    OptRef o;
    if (!find_opt(h, key, tuning, &o)) {
        if (find_opt(h, key, !tuning, &o))
            return jrx_fail(h, JRX_ERR_ARG, "%s: '%s' is %s", fn, key, tuning ? "an option of the public ABI (jrx_set_option)" : "a tuning switch (jrx_tuning_set, include/jrx_tuning.h)");
        return jrx_fail(h, JRX_ERR_ARG, "%s: unknown key '%s'", fn, key);
    }
    if (o.kind == 2) return jrx_fail(h, JRX_ERR_ARG, "%s: '%s' is a read-only counter", fn, key);
    if (o.kind == 0) *(bool *)o.p = value != 0;
    else *(int *)o.p = (int)value;
    if (strcmp(key, "comm_timeout_ms") == 0) jrx_comm_set_timeout(h, (double)value * 1e-3);
    return JRX_OK;
}
jrx_status opt_get(jrx_handle *h, const char *key, int64_t *value, bool tuning, const char *fn)
{
    if (!h) return JRX_ERR_ARG;
    if (!key || !value) return jrx_fail(h, JRX_ERR_ARG, "%s: null argument", fn);
    OptRef o;
    if (!find_opt(h, key, tuning, &o)) return jrx_fail(h, JRX_ERR_ARG, "%s: unknown key '%s'", fn, key);
    *value = o.kind == 0 ? (int64_t)*(bool *)o.p : (o.kind == 1 ? (int64_t)*(int *)o.p : *(int64_t *)o.p);
    return JRX_OK;
}
}   // namespace

jrx_status jrx_fields_dirty(jrx_handle *h)
{
    if (!h) return JRX_ERR_ARG;
    h->opv.valid = false;
    return JRX_OK;
}
jrx_status jrx_set_option(jrx_handle *h, const char *key, int64_t value) { return opt_set(h, key, value, false, "jrx_set_option"); }
jrx_status jrx_get_option(jrx_handle *h, const char *key, int64_t *value) { return opt_get(h, key, value, false, "jrx_get_option"); }
jrx_status jrx_tuning_set(jrx_handle *h, const char *key, int64_t value) { return opt_set(h, key, value, true, "jrx_tuning_set"); }
jrx_status jrx_tuning_get(jrx_handle *h, const char *key, int64_t *value) { return opt_get(h, key, value, true, "jrx_tuning_get"); }
jrx_status jrx_tuning_chain_profile(jrx_handle *h, double out_us[8], int64_t *samples)
{
    if (!h) return JRX_ERR_ARG;
    if (!out_us || !samples) return jrx_fail(h, JRX_ERR_ARG, "jrx_tuning_chain_profile: null argument");
    for (int q = 0; q < 8; q++) out_us[q] = h->chain_us[q];
    *samples = h->chain_n;
    return JRX_OK;
}

// ---------------------------------------------------------------- block decomposition (host only)
int64_t jrx_n_global(int64_t n, int32_t dims, int32_t periodic)
{
    if (n == 1) return 1;
    return (int64_t)dims * (n - 2) + (periodic ? 0 : 2);
}

jrx_status jrx_halo_planes(int64_t n, int64_t nA, int64_t *send_left, int64_t *send_right, int64_t *recv_left, int64_t *recv_right)
{
    const int64_t ol = 2 + (nA - n);          // overlap of this array along the dimension
    if (ol < 2 || nA < ol) return JRX_ERR_ARG;   // e.g. Rx (extent n-1): not exchangeable
    if (send_left) *send_left = ol - 1;       // 1-based plane ol_A
    if (send_right) *send_right = nA - ol;    // 1-based plane nA - ol_A + 1
    if (recv_left) *recv_left = 0;
    if (recv_right) *recv_right = nA - 1;
    return JRX_OK;
}

jrx_status jrx_cart_create(int32_t rank, int32_t nprocs, const int64_t n[3], const int32_t dims_in[3],
                           const int32_t periods[3], jrx_cart *out)
{
    if (!out || !n || nprocs < 1 || rank < 0 || rank >= nprocs) return JRX_ERR_ARG;
    int32_t dims[3] = {1, 1, 1};
    const bool fixed = dims_in && (dims_in[0] || dims_in[1] || dims_in[2]);
    if (fixed) {
        for (int d = 0; d < 3; d++) dims[d] = dims_in[d] ? dims_in[d] : 1;
    } else {
        // balanced factorisation (MPI_Dims_create): largest prime factors first onto the smallest dim
        int idx[3], na = 0;
        for (int d = 0; d < 3; d++)
            if (n[d] > 1) idx[na++] = d;
        if (na > 0) {
            int fac[32], nf = 0, m = nprocs;
            for (int p = 2; m > 1; p++)
                while (m % p == 0) { fac[nf++] = p; m /= p; }
            int v[3] = {1, 1, 1};
            for (int q = nf - 1; q >= 0; q--) {
                int j = 0;
                for (int a = 1; a < na; a++)
                    if (v[a] < v[j]) j = a;
                v[j] *= fac[q];
            }
            // non-increasing order over the active dims
            for (int a = 0; a < na; a++)
                for (int b = a + 1; b < na; b++)
                    if (v[b] > v[a]) { int t = v[a]; v[a] = v[b]; v[b] = t; }
            for (int a = 0; a < na; a++) dims[idx[a]] = v[a];
        }
    }
    if ((int64_t)dims[0] * dims[1] * dims[2] != nprocs) return JRX_ERR_ARG;
    out->rank = rank; out->nprocs = nprocs;
    int r = rank;
    for (int d = 2; d >= 0; d--) { out->coords[d] = r % dims[d]; r /= dims[d]; }
    for (int d = 0; d < 3; d++) { out->dims[d] = dims[d]; out->periods[d] = periods ? periods[d] : 0; }
    for (int d = 0; d < 3; d++)
        for (int sgn = 0; sgn < 2; sgn++) {
            int c[3] = {out->coords[0], out->coords[1], out->coords[2]};
            c[d] += sgn ? 1 : -1;
            int nb = -1;
            if (c[d] >= 0 && c[d] < dims[d]) nb = (c[0] * dims[1] + c[1]) * dims[2] + c[2];
            else if (out->periods[d]) {            // dims[d] == 1: the rank is its own neighbour (IGG local copy)
                c[d] = (c[d] + dims[d]) % dims[d];
                nb = (c[0] * dims[1] + c[1]) * dims[2] + c[2];
            }
            out->neighbor[d][sgn] = nb;
        }
    return JRX_OK;
}

}   // extern "C"
