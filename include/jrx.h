/*
 * jrx.h -- C ABI of the MI355X-native pseudo-transient (PT) Stokes / heat-diffusion hot path.
 *
 * This is the drop-in boundary for JustRelax.jl's backend method table: a Julia extension
 * defines solve!(::Trait, stokes, ...), heatdiffusion_PT!(::Trait, thermal, ...), flow_bcs!,
 * thermal_bcs!, velocity2displacement!, compute_maxloc! for its trait and forwards each to one
 * entry point below with `ccall`, passing `pointer(A)` of its ROCArray{Float64} fields
 * (INTEGRATION.md shows the shim).  Reference seam being replaced:
 *   src/ext/AMDGPU/3D.jl:397-407, src/ext/AMDGPU/2D.jl (forwarding methods onto _solve!),
 *   src/stokes/Stokes3D.jl:18-23, src/stokes/Stokes2D.jl:12-17,
 *   src/thermal_diffusion/DiffusionPT_solver.jl:11-17.
 * All file:line citations are relative to the reference checkout (PTsolvers/JustRelax.jl v0.7.1).
 *
 * Conventions
 *  - every array is device memory, fp64, dense, column-major (x fastest) -- Julia's layout; the
 *    library never allocates, frees or re-lays-out a caller array (the caller keeps them alive
 *    for the duration of the call: GC.@preserve);
 *  - extents follow src/types/constructors/stokes.jl and .../heat_diffusion.jl, e.g. in 3D with
 *    ni = (nx,ny,nz): Vx (nx+1,ny+2,nz+2), txy (nx+1,ny+1,nz), Rx (nx-1,ny,nz), T (nx+2,ny+2);
 *  - calls are synchronous at the ABI (they return after the handle's stream has drained) except
 *    the *_async kernels-only entry points, which only enqueue;
 *  - every function returns a jrx_status; jrx_last_error() gives the message. Never aborts.
 *  - a handle is bound to one GPU and is not thread-safe (one handle per GPU / per rank).
 */
#ifndef JRX_H
#define JRX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JRX_VERSION 230

typedef enum jrx_status {
    JRX_OK = 0,
    JRX_ERR_NAN = 1,        /* error("NaN(s)") -- src/stokes/Stokes3D.jl:162 */
    JRX_ERR_HIP = 2,
    JRX_ERR_RCCL = 3,
    JRX_ERR_ARG = 4,
    JRX_ERR_UNSUPPORTED = 5
} jrx_status;

/* Boundary faces: one bit per face, names as in the reference's NamedTuples
 * (src/boundaryconditions/types.jl:108-157).  NOTE the reference's own 3D conventions, kept
 * verbatim: free_slip! maps `top` to k = 1 and `bot` to k = end (free_slip.jl:35-50); no_slip!
 * maps `bot` to k = 1 and `top` to k = end (no_slip.jl:44-53); periodic: bot = k = 1.
 * In 2D: bot <-> j = 1, top <-> j = end for all three. */
enum {
    JRX_FACE_LEFT = 1, JRX_FACE_RIGHT = 2, JRX_FACE_FRONT = 4, JRX_FACE_BACK = 8,
    JRX_FACE_TOP = 16, JRX_FACE_BOT = 32
};

typedef struct jrx_handle jrx_handle;

/* ------------------------------------------------------------------ lifetime */
/* Creates streams, events and reduction scratch on `device`. */
jrx_status jrx_create(int32_t device, jrx_handle **out);
jrx_status jrx_destroy(jrx_handle *h);
const char *jrx_last_error(const jrx_handle *h);   /* h may be NULL: last creation error */
int32_t jrx_version(void);
/* sha256 of the sources (csrc/ + include/jrx.h) this binary was built from; the Python binding refuses a library whose id differs from
 * the sources beside it */
const char *jrx_build_id(void);

/* ------------------------------------------------------------------ state arrays
 * Device memory for the fields, handed out by the library.  Replaces the array constructor the backend owns in the reference:
 * StokesArrays(::Type{AMDGPUBackend}, ni) / ThermalArrays(...) -> @zeros(ni...) -> ROCArray (src/ext/AMDGPU/3D.jl:46-48,
 * src/types/constructors/stokes.jl:279-303) -- a binding wraps the pointer (Julia: unsafe_wrap(ROCArray, ptr, dims; lock = false) plus a
 * finalizer that calls jrx_field_free).  Contents are NOT initialised (the constructor fills with zeros as @zeros does).  Using it is
 * optional: every entry point takes any device pointer.  What it buys: the option "field_placement" decides how the arrays are backed
 * physically, and the large 3D kernels are sensitive to that (the same launch takes 4.7 .. 6.9 ms at 512^3: DESIGN.md section 3, profiles/r05_placement_search.txt).  With
 * "field_placement" = 1 every large array is mapped ONCE, at a virtual range never used before, onto physical chunks picked at random from a pool that spans most of the free
 * memory (tuning keys "field_chunk_mib", "field_pool_pct": include/jrx_tuning.h); the library's own large arrays (second state sets, ητ) follow the same option.  An array is
 * never moved afterwards.  jrx_destroy releases whatever the caller has not freed.  jrx_field_trim returns the pool's unused chunks to the driver (call it once the arrays of a
 * run exist -- the library's second state set is made by the first driver call).  The pool is filled when a large array is requested while no array of that chunk size is live
 * (the first array of a run); arrays made later get chunks created on the spot.
 * jrx_field_stats: [0] live arrays (the library's own included), [1] their bytes, [2] physical chunks created, [3] spare chunks, [4] us in hipMemCreate, [5] us mapping. */
jrx_status jrx_field_alloc(jrx_handle *h, int64_t count, double **out);     /* count doubles */
jrx_status jrx_field_free(jrx_handle *h, double *p);
jrx_status jrx_field_trim(jrx_handle *h);
/* jrx_field_list: the live arrays of the handle (bytes[i] < 0: not chunk-backed), count = how many there are (may exceed cap). */
jrx_status jrx_field_list(jrx_handle *h, int64_t cap, double **ptrs, int64_t *bytes, int64_t *count);
jrx_status jrx_field_stats(jrx_handle *h, int64_t out[6]);

/* Options of the handle: what a caller of solve! may want to choose.  (The A/B switches of the measurements in profiles/ and the test
 * hooks are NOT part of this ABI: include/jrx_tuning.h.)  The library never reads the process environment.  Keys:
 * "kernel_variant" (3D Stokes):
 *   0 = default: fused PT pipeline where it applies (nx >= 48, ny, nz >= 8, and not one of the few nx -- 125 .. 128 -- whose last
 *       62-column tile stays nearly empty while the sweeps' row tiles fill exactly): one kernel runs velocity sweep m + BCs + stress sweep m+1 with ping-pong
 *       state arrays (the handle then owns a second set of the 10 state arrays); with a communicator the exchange
 *       of V follows and the stress nodes next to a received plane are redone; otherwise, and on iterations whose
 *       results are observed, the two z-marching sweeps;
 *   1 = simple one-thread-per-node kernels;  2 = the two sweeps only (z-marching; per-node on blocks up to ~88^3; no ping-pong set);
 *   3 = fused pipeline wherever it is legal (ignores that rule).  All variants produce bit-identical results.
 * "fused_comm" (0/1, default 1): multi-rank runs use the fused pipeline; 0 = split sweeps + hidden communication (same results).
 * "fused_overlap" (0/1/2/3/4, default 3): how the multi-rank fused pipeline places update_halo!(V) (same results in every mode):
 *   3 = neighbour faces inside the kernel, for every rank (4 is accepted as a synonym; viscous-limit form, dt = Inf; other runs use 2): boundary slabs + the exchange on a second stream beside the fused kernel as in 2, but the
 *       kernel's tiles next to a face with a neighbour are launched behind the exchange (the others beside it) and read the received planes themselves: no flow_bcs! launch,
 *       no fix-up launch;
 *   2 = early exchange: the velocity phase alone over the boundary slabs of the faces with a neighbour, flow_bcs! and the whole exchange on a second
 *       stream BESIDE the fused kernel (which recomputes those cells with the same values), then flow_bcs! on the physical faces and the fix-up;
 *   1 = the shell of tiles, BCs and exchange on the second stream while the interior tiles run;  0 = everything behind the kernel, in order.
 * "thermal_fused" (0/1, default 1): jrx_heatdiffusion_PT2d / _PT3d run unobserved iterations as one fused launch with a
 *   library-owned second (T, qT) set; 0 = always compute_flux! and update_T! as two launches (same results).
 * "scratch_sets" (0/1, default 1): 0 forbids every library-owned second state set (3D fused pipeline: + 10 arrays, i.e.
 *   + 10.8 GB at 512^3; 2D fused loop; fused heat diffusion) -- the un-fused kernels then run; same results, less memory.
 * "loop_graphs" (0/1, default 1): runs of unobserved iterations of the launch-bound loops (the 2D loops; the 3D loops on small grids) replay as
 *   captured hipGraphs; same results, shorter gaps between launches.
 * "viscous_limit" (0/1, default 1): with dt = Inf (the reference's purely viscous runs: SolVi3D, Burstedde, TaylorGreen) the 3D visco-elastic
 *   stress kernels (fused iteration, z-marching sweep, boundary layers) do not load the operands that 1/(G dt) = 1/(K dt) = 1/dt = 0 multiply
 *   (old stresses, P0, K, G, Q).  Every driver call first checks those ten arrays in one streaming pass (all of tau_o, P0, Q finite; K, G neither NaN
 *   nor 0): only then do the results equal the general kernels', and only then does this form run -- otherwise the general kernels run and a NaN
 *   there ends the solve with JRX_ERR_NAN exactly as error("NaN(s)") of Stokes3D.jl:162 would.  0 = always the general kernels.
 * "operand_cache" (0/1, default 0): the drivers of the 3D visco-elastic path look at their operand arrays once per call (dt = Inf: may the viscous-limit kernels stand in --
 *   tau_o, P0, Q, eta finite, K, G neither NaN nor 0; any dt: are body-force arrays +0.0 throughout): one streaming pass and a host synchronisation, 3.6 ms at 512^3 -- nothing
 *   inside a solve! of thousands of iterations, 3 % of a 20-iteration batch.  1 = the verdict is kept per (operand pointers, extents, dt) and reused until the caller states that
 *   it has written to one of those arrays: jrx_fields_dirty(h).  The library's own writes (the tau -> tau_o copy at the end of solve!) invalidate it themselves.  Default 0: every call looks.
 * "field_placement" (0/1/2, default 0): backing of the arrays of jrx_field_alloc and of the library's own large arrays: 0 = hipMalloc; 1 = physical chunks
 *   (hipMemCreate) picked at random from a pool and mapped once onto a fresh virtual range per array; 2 = physically contiguous (hipDeviceMallocContiguous; the slowest placement there
 *   is, for A/B runs).  Results never depend on it.
 * "field_chunk_mib" (default 64): size of a physical chunk of "field_placement" = 1 in MiB (0: every array ONE chunk of its own size).  The pool -- chunks for most of the free
 *   memory, created by the first large allocation, dealt at random -- exists for chunks of >= 128 MiB; a caller sets the size of its largest array (for an (nx, ny, nz) block:
 *   (nx + 2)(ny + 2)(nz + 2) x 8 B rounded up to 2 MiB) so that every array is ONE chunk of one common size, which is the arrangement that was measured
 *   (profiles/r05_placement_search.txt section 13, profiles/r06_ten_processes.txt).
 * Read-only counters (jrx_get_option): "stat_fused3d", "stat_fused2d", "stat_thermal_fused", "stat_vep3_fused" = launches of the fused
 *   kernels since jrx_create, "stat_fused3d_visc" = those of "stat_fused3d" that ran the viscous-limit form, "stat_fused3d_inkernel" = those that finished the faces with a neighbour themselves ("fused_overlap" = 3), "stat_visc_checks" /
 *   "stat_visc_fallbacks" = operand checks run / failed (general kernels used), "stat_operand_cache_hits" = driver calls that reused the operand verdict, "stat_graph_replays" = hipGraphLaunch calls -- so that a caller
 *   (and the tests, and bench.py for the kernel it prices) can prove which path ran. */
jrx_status jrx_set_option(jrx_handle *h, const char *key, int64_t value);
/* the caller has written to an operand array (tau_o, P0, Q, K, G, eta, rho g) since the last driver call: a cached verdict of the operand pass ("operand_cache") is dropped */
jrx_status jrx_fields_dirty(jrx_handle *h);
jrx_status jrx_get_option(jrx_handle *h, const char *key, int64_t *value);

/* ------------------------------------------------------------------ block decomposition (host logic; no GPU needed)
 * ImplicitGlobalGrid semantics used by the reference: local arrays of n cells overlap
 * the neighbour by 2 cells; an array of extent nA along a split dimension has overlap
 * ol_A = 2 + (nA - n); update_halo! sends plane ol_A (1-based) to the left neighbour and plane
 * nA - ol_A + 1 to the right one, and receives into planes 1 and nA. */
typedef struct jrx_cart {
    int32_t rank, nprocs;
    int32_t dims[3], coords[3];
    int32_t periods[3];
    int32_t neighbor[3][2];     /* [dim][0=left,1=right], -1 = physical boundary; == rank for a periodic dim with dims[d]==1 */
} jrx_cart;
/* dims = all zeros -> balanced factorisation over the dimensions with n[d] > 1 */
jrx_status jrx_cart_create(int32_t rank, int32_t nprocs, const int64_t n[3], const int32_t dims_in[3],
                           const int32_t periods[3], jrx_cart *out);
/* 0-based plane indices for an array of extent nA in a dimension of n local cells */
jrx_status jrx_halo_planes(int64_t n, int64_t nA, int64_t *send_left, int64_t *send_right,
                           int64_t *recv_left, int64_t *recv_right);
int64_t jrx_n_global(int64_t n, int32_t dims, int32_t periodic);   /* nx_g() */

/* RCCL communicator for the halo exchange + norm all-reduce (one rank per GPU).
 * Rank 0 calls jrx_comm_unique_id and ships the 128 bytes to the others out of band
 * (MPI.Bcast in Julia, torch.distributed in the Python host), then every rank calls init. */
#define JRX_UNIQUE_ID_BYTES 128
jrx_status jrx_comm_unique_id(uint8_t id[JRX_UNIQUE_ID_BYTES]);
jrx_status jrx_comm_init(jrx_handle *h, const uint8_t id[JRX_UNIQUE_ID_BYTES], const jrx_cart *cart);
jrx_status jrx_comm_destroy(jrx_handle *h);
/* The same communicator between PROCESSES of one node with copy engines as transport ("ipc"; one process per rank as the reference runs --
 * mpiexec -n N, test/runtests.jl:73-90 -- and as the Julia extension's MPI ranks would).  Every rank exports one receive buffer per (dimension, side)
 * (hipIpcGetMemHandle), the neighbour maps it and pushes its packed planes into it with hipMemcpyAsync on the exchange's stream; sequence flags in a
 * POSIX shared-memory segment named by `id` order the copies against the unpack kernels (device-side waits with a time-out: a neighbour that never
 * arrives is JRX_ERR_RCCL, not a hang); norms are all-reduced through the segment in rank order.  Rank 0 calls jrx_comm_ipc_id and ships the 128
 * bytes out of band exactly like the RCCL id; every rank then calls jrx_comm_init_ipc.  Needs HSA_ENABLE_IPC_MODE_LEGACY=0 where the driver only
 * supports dmabuf IPC.  Replaces the same call sites as jrx_comm_init. */
jrx_status jrx_comm_ipc_id(uint8_t id[JRX_UNIQUE_ID_BYTES]);
jrx_status jrx_comm_init_ipc(jrx_handle *h, const uint8_t id[JRX_UNIQUE_ID_BYTES], const jrx_cart *cart);
/* The same communicator for ranks that are handles of ONE process (in-process transport): handles[r] becomes rank r of carts[r]
 * (carts[r].rank == r, carts[r].nprocs == n); the handles may sit on one device or on peer-accessible devices.  update_halo! then
 * pushes the packed planes into the neighbour's receive buffer with hipMemcpyAsync / hipMemcpyPeerAsync on the exchange's stream
 * (copy engines instead of send/recv kernels), ordered by events; norms are all-reduced on the host in rank order.  Afterwards every
 * rank must be driven by its own host thread: the entry points that exchange (the solves, jrx_update_halo, jrx_compute_dt) meet on
 * the host and return JRX_ERR_RCCL after 120 s if a neighbour never arrives.  Replaces the same call sites as jrx_comm_init
 * (update_halo! / norm_mpi of src/stokes/Stokes3D.jl:57,120,127-147); jrx_comm_destroy leaves the group. */
jrx_status jrx_comm_init_local(jrx_handle *const *handles, int32_t n, const jrx_cart *carts);
/* the number of ranks of the handle's communicator: ncclCommCount as RCCL itself reports it, or the size of the in-process group
 * (0 = neither) */
jrx_status jrx_comm_count(jrx_handle *h, int32_t *count);
/* update_halo!(A...) for up to 8 arrays, each of extents ext[a][0..2] on a local grid of n cells
 * (call sites: src/stokes/Stokes3D.jl:57,120; src/stokes/Stokes2D.jl:209,268;
 * src/thermal_diffusion/DiffusionPT_solver.jl:110).  Dimension by dimension (x, y, z). */
jrx_status jrx_update_halo(jrx_handle *h, int32_t narrays, double *const *arrays, const int64_t (*ext)[3],
                           const int64_t n[3]);

/* ------------------------------------------------------------------ 3D Stokes, isoviscous visco-elastic variant */
typedef struct jrx_stokes3d_fields {
    double *P, *P0, *divV, *Q;                       /* ni                         (stokes.P, P0, ∇V, Q) */
    double *Vx, *Vy, *Vz;                            /* staggered                  (stokes.V)            */
    double *Ux, *Uy, *Uz;                            /* as V                       (stokes.U)            */
    double *txx, *tyy, *tzz, *tyz, *txz, *txy;       /* @stress(stokes)  -- Voigt order of src/Utils.jl:230-239 */
    double *toxx, *toyy, *tozz, *toyz, *toxz, *toxy; /* @tensor(stokes.τ_o)                              */
    double *exx, *eyy, *ezz, *eyz, *exz, *exy;       /* @strain(stokes)                                  */
    double *eta;                                     /* stokes.viscosity.η, ni                           */
    double *K, *G;                                   /* bulk / shear modulus arrays, ni (may hold Inf)   */
    double *fx, *fy, *fz;                            /* ρg, ni                                           */
    double *RP, *Rx, *Ry, *Rz;                       /* stokes.R                                         */
    double *tyz_c, *txz_c, *txy_c, *toyz_c, *toxz_c, *toxy_c;  /* centre shear copies for multi_copy!, may be NULL */
} jrx_stokes3d_fields;

typedef struct jrx_stokes3d_params {
    int64_t nx, ny, nz;            /* size(stokes.P) on this rank */
    int64_t nxg, nyg, nzg;         /* nx_g(), ny_g(), nz_g() */
    double _dx, _dy, _dz;          /* grid._di.center */
    double dt;                     /* may be Inf */
    double r, theta_dtau, eta_dtau;/* PTStokesCoeffs: r, θ_dτ, ηdτ (src/types/stokes.jl:203-229) */
    double eps_rel, eps_abs;       /* ϵ_rel, ϵ_abs */
    int64_t iterMax, nout;         /* kwargs of solve! (Stokes3D.jl:35-41) */
    uint32_t free_slip, no_slip, periodic;   /* JRX_FACE_* masks of flow_bcs */
    int32_t b_width[3];            /* boundary-slab width for comm/compute overlap (default 4,4,4) */
    int32_t verbose;               /* print the reference's per-check line on rank 0 */
    int32_t displacement_bcs;      /* flow_bcs is a DisplacementBoundaryConditions: displacement2velocity! first (Stokes3D.jl:72), flow_bcs! on U */
} jrx_stokes3d_params;

typedef struct jrx_solve_result {
    int64_t iter;                  /* iterations executed */
    int64_t nchecks;               /* entries filled in the histories below */
    int64_t cap;                   /* capacity of each history buffer (>= iterMax/nout + 1) */
    double *err_evo1; int64_t *err_evo2;
    double *norm_Rx, *norm_Ry, *norm_Rz, *norm_divV;   /* norm_Rz unused in 2D */
    double time_s, av_time_s;      /* wtime0 and wtime0/(iter-1) as in Stokes3D.jl:170,183-184 */
} jrx_solve_result;

/* solve!(stokes, pt_stokes, grid, flow_bcs, ρg, K, G, dt, igg; kwargs) -- src/stokes/Stokes3D.jl:25-186.
 * Runs the whole PT loop on the device: compute_maxloc!(ητ, η) (+halo), then per iteration
 * compute_∇V!, compute_P!, compute_strain_rate!, compute_τ!, compute_V!, velocity2displacement!,
 * flow_bcs!, update_halo!(V), L2 residual norms every nout iterations, and the final τ -> τ_o copy.
 * All outputs the reference leaves in StokesArrays after the call (∇V, ε, R, U included) are
 * left identical. */
jrx_status jrx_stokes3d_solve(jrx_handle *h, const jrx_stokes3d_fields *f, const jrx_stokes3d_params *p,
                              jrx_solve_result *res);

/* Finer-grained entry points (each mirrors reference kernels; used by the parity tests, and by
 * callers that drive the loop themselves).  `etatau` is the ητ array of compute_maxloc!.
 * flags for the fused sweeps: */
enum {
    JRX_OUT_STATE_ONLY = 0,     /* write only the state the next sweep needs (P, τ / V) */
    JRX_OUT_DIAG = 1            /* also write ∇V, ε, RP (stress sweep) / R, U (velocity sweep) */
};
/* compute_∇V! + compute_P! + compute_strain_rate! + compute_τ!  (VelocityKernels.jl:3-6,59-104;
 * PressureKernels.jl:10-15,186-195; StressKernels.jl:149-230) fused into one sweep over ni.+1 */
jrx_status jrx_stokes3d_sweep_stress(jrx_handle *h, const jrx_stokes3d_fields *f, const jrx_stokes3d_params *p,
                                     int32_t flags);
/* compute_V! (VelocityKernels.jl:182-242) + velocity2displacement! (types/displacement.jl:17-28) */
jrx_status jrx_stokes3d_sweep_velocity(jrx_handle *h, const jrx_stokes3d_fields *f, const double *etatau,
                                       const jrx_stokes3d_params *p, int32_t flags);
/* flow_bcs! -- BoundaryConditions.jl:86-100 (no_slip, free_slip, periodic, in that order) */
jrx_status jrx_flow_bcs3d(jrx_handle *h, double *Vx, double *Vy, double *Vz, int64_t nx, int64_t ny, int64_t nz,
                          uint32_t free_slip, uint32_t no_slip, uint32_t periodic);
/* Σx² of Rx,Ry,Rz[2:end-1,2:end-1,2:end-1] and of RP (local part of norm_mpi, Stokes3D.jl:127-142) */
jrx_status jrx_stokes3d_residual_sumsq(jrx_handle *h, const jrx_stokes3d_fields *f, const jrx_stokes3d_params *p,
                                       double out[4]);
/* compute_maxloc!(B, A; window=(1,1,1)) -- src/Utils.jl:409-461 ; nz = 1 selects the 2D form */
jrx_status jrx_compute_maxloc(jrx_handle *h, double *B, const double *A, int64_t nx, int64_t ny, int64_t nz);

/* ------------------------------------------------------------------ 2D Stokes, visco-elastic variant */
typedef struct jrx_stokes2d_fields {
    double *P, *P0, *divV, *Q;
    double *Vx, *Vy, *Ux, *Uy;
    double *txx, *tyy, *txy, *toxx, *toyy, *toxy;
    double *exx, *eyy, *exy;
    double *eta, *K, *G;
    double *fx, *fy;
    double *RP, *Rx, *Ry;
    double *txy_c, *toxy_c;        /* may be NULL */
} jrx_stokes2d_fields;

typedef struct jrx_stokes2d_params {
    int64_t nx, ny, nxg, nyg;
    double _dx, _dy;
    double dt, r, theta_dtau, eta_dtau, eps_rel, eps_abs;
    int64_t iterMax, nout;
    uint32_t free_slip, no_slip, periodic;
    int32_t verbose;
    int32_t displacement_bcs;      /* as in jrx_stokes3d_params (Stokes2D.jl:223) */
    /* Non-uniform Geometry (Geometry(xvi...), src/grid/Cartesian.jl:77-100): device arrays of INVERSE spacings, all six NULL on a uniform grid (then _dx, _dy
     * apply).  [0] _di.vertex[1] (nx), [1] _di.vertex[2] (ny), [2] _di.center[1] (nx-1), [3] _di.center[2] (ny-1), [4] _di.velocity[1][2] (y spacing of
     * the Vx grid, ny+1), [5] _di.velocity[2][1] (x spacing of the Vy grid, nx+1).  Each stencil takes the array the reference's kernel takes
     * (VelocityKernels.jl:3-44,108-180,246-307).  2D drivers only; the one-launch iteration (k_fused2d) is not used on such a grid. */
    const double *inv_spacing[6];
} jrx_stokes2d_params;

/* solve!(stokes, pt_stokes, grid, flow_bcs, ρg, G, K, dt, igg; kwargs) -- src/stokes/Stokes2D.jl:181-325 */
jrx_status jrx_stokes2d_solve(jrx_handle *h, const jrx_stokes2d_fields *f, const jrx_stokes2d_params *p,
                              jrx_solve_result *res);
jrx_status jrx_stokes2d_sweep_stress(jrx_handle *h, const jrx_stokes2d_fields *f, const double *etatau,
                                     const jrx_stokes2d_params *p, int32_t flags);
jrx_status jrx_stokes2d_sweep_velocity(jrx_handle *h, const jrx_stokes2d_fields *f, const double *etatau,
                                       const jrx_stokes2d_params *p, int32_t flags);
/* compute_Res! -- VelocityKernels.jl:246-269 */
jrx_status jrx_stokes2d_compute_res(jrx_handle *h, const jrx_stokes2d_fields *f, const jrx_stokes2d_params *p);
jrx_status jrx_flow_bcs2d(jrx_handle *h, double *Vx, double *Vy, int64_t nx, int64_t ny,
                          uint32_t free_slip, uint32_t no_slip, uint32_t periodic);
jrx_status jrx_stokes2d_residual_sumsq(jrx_handle *h, const jrx_stokes2d_fields *f, const jrx_stokes2d_params *p,
                                       double out[3]);

/* ------------------------------------------------------------------ 2D multiphase visco-elasto-plastic Stokes (config 5)
 * solve!(stokes, pt_stokes, grid, flow_bcs, ρg, phase_ratios, rheology, args, dt, igg; kwargs)
 * -- src/stokes/Stokes2D.jl:577-866 with update_stresses_center_vertex_ps! (src/stokes/StressKernels.jl:992-1144),
 * compute_P! phase-ratio form (PressureKernels.jl:47-106), update_viscosity_τII! (rheology/Viscosity.jl:67-106,382-418)
 * and the post-loop epilogue (vorticity, shear2center!, accumulate_tensor!, accumulate_vol!, multi_copy!).
 * The rheology is passed as a table instead of GeoParams objects: per phase a LinearViscous viscosity, ConstantElasticity
 * (G, Kb) and an optional DruckerPrager_regularised(C, ϕ, ψ, η_vp) -- what test/test_shearband2D.jl uses.  Phase ratios
 * are JustPIC's CellArray layout: phase index fastest, [nphase][nx][ny] at centres and [nphase][nx+1][ny+1] at vertices. */
#define JRX_MAXPHASE 8
typedef struct jrx_rheology {
    int32_t nphase;
    double eta[JRX_MAXPHASE], G[JRX_MAXPHASE], Kb[JRX_MAXPHASE];
    int32_t is_pl[JRX_MAXPHASE];
    double C[JRX_MAXPHASE], sinphi[JRX_MAXPHASE], cosphi[JRX_MAXPHASE], sinpsi[JRX_MAXPHASE], eta_vp[JRX_MAXPHASE];
    /* ---- appended in round 2; a zero-initialised tail gives the round-1 behaviour (LinearViscous, NoSoftening, ρg owned by the caller) ----
     * Density and gravity -- compute_ρg! / update_ρg! (rheology/BuoyancyForces.jl:37-60,153-167).  has_density = 0: the ρg arrays
     * are the caller's and never recomputed.  rho_kind: 0 ConstantDensity(rho0), 1 PT_Density rho0 (1 - alpha (T - T0) + beta (P - P0)),
     * 2 T_Density rho0 (1 - alpha (T - T0)), 3 Compressible_Density rho0 exp(beta (P - P0)) [GeoParams forms, assumed].
     * gravity = compute_gravity(first(rheology)): a scalar, it fills the last component of ρg (BuoyancyForces.jl:69-70). */
    int32_t has_density;
    int32_t rho_kind[JRX_MAXPHASE];
    double rho0[JRX_MAXPHASE], alpha[JRX_MAXPHASE], beta[JRX_MAXPHASE], T0[JRX_MAXPHASE], P0[JRX_MAXPHASE];
    double gravity;
    /* Strain softening of the cohesion and of the friction angle, evaluated at the accumulated plastic strain EII_pl (the EII keyword
     * of compute_yieldfunction_phase, StressKernels.jl:1053-1105; GeoParams softening_C / softening_ϕ, forms assumed):
     * kind 0 NoSoftening, 1 LinearSoftening((a = min, b = max), (c = lo, d = hi)): b for EII <= lo, a for EII >= hi, linear between;
     * 2 NonLinearSoftening(a = ξ₀, b = Δ, c = μ, d = σ) = ξ₀ - Δ/2 erfc(-(EII - μ)/σ).  phi_deg is the unsoftened friction angle. */
    int32_t softC_kind[JRX_MAXPHASE], softphi_kind[JRX_MAXPHASE];
    double softC_a[JRX_MAXPHASE], softC_b[JRX_MAXPHASE], softC_c[JRX_MAXPHASE], softC_d[JRX_MAXPHASE];
    double softphi_a[JRX_MAXPHASE], softphi_b[JRX_MAXPHASE], softphi_c[JRX_MAXPHASE], softphi_d[JRX_MAXPHASE], phi_deg[JRX_MAXPHASE];
    /* Creep law of the viscous element for compute_viscosity! / compute_viscosity_τII! with dt = Inf (rheology/Viscosity.jl:142-167):
     * visc_kind 0 LinearViscous(eta); 1 Arrhenius: eta exp((Ea + P Va)/(Rgas T) - Ea/(Rgas Tref)), clamped to [visc_lo, visc_hi]
     * (the CustomRheology of test/test_WENO5.jl:37-42 with depth = 0);
     * 2 power-law creep (GeoParams DislocationCreep with r = 0; DiffusionCreep is n = 1 with the grain-size factor folded into A):
     *   compute_εII: ε = creep_A (τII creep_FT)^creep_n exp(-(Ea + P Va)/(Rgas T)) / creep_FE
     *   compute_τII: τ = creep_A^(-1/n) (εII creep_FE)^(1/n) exp((Ea + P Va)/(n Rgas T)) / creep_FT
     *   compute_viscosity_τII = τII / (2 ε(τII)),  compute_viscosity_εII = τ(εII) / (2 εII)      [forms ASSUMED, parity unpinned]
     * (the 2D single-material driver hands its in-loop compute_viscosity_τII! the strain-rate invariant, as the reference's _compute_viscosity! does.)
     * FT, FE: GeoParams' apparatus corrections (AxialCompression √3, 2/√3; SimpleShear 2, 2; Invariant 1, 1).  One creep element per phase.
     * The invariant a law of kind 2 is evaluated at is the one the reference's kernels form (Viscosity.jl:382-418,455-503): compute_viscosity! from the strain
     * rate, update_viscosity_τII! from the stress, eps() on the normal components of an all-zero tensor. */
    int32_t visc_kind[JRX_MAXPHASE];
    double Ea[JRX_MAXPHASE], Va[JRX_MAXPHASE], Tref[JRX_MAXPHASE], Rgas[JRX_MAXPHASE], visc_lo[JRX_MAXPHASE], visc_hi[JRX_MAXPHASE];
    double creep_A[JRX_MAXPHASE], creep_n[JRX_MAXPHASE], creep_FT[JRX_MAXPHASE], creep_FE[JRX_MAXPHASE];      /* visc_kind 2 */
} jrx_rheology;

typedef struct jrx_vep2d_fields {
    double *P, *P0, *divV, *Q;                     /* stokes.P, P0, ∇V, Q */
    double *Vx, *Vy, *Ux, *Uy;
    double *exx, *eyy, *exy, *exy_c;               /* stokes.ε: xx, yy (centres), xy (vertices), xy_c */
    double *eplxx, *eplyy, *eplxy, *eplxy_c;       /* stokes.ε_pl */
    double *dexy_c, *dexy;                         /* stokes.Δε.xy_c, .xy (shear2center! only; may be NULL) */
    double *txx, *tyy, *txy, *txy_c, *tII;         /* stokes.τ: xx, yy, xy (vertices), xy_c, II */
    double *toxx, *toyy, *toxy, *toxy_c;           /* stokes.τ_o */
    double *eta, *eta_v, *eta_vep;                 /* stokes.viscosity.η, ηv (may be NULL), η_vep */
    double *EII_pl, *evol_pl, *EVol_pl;            /* stokes.EII_pl, ε_vol_pl, EVol_pl */
    double *fx, *fy;                               /* ρg */
    double *RP, *Rx, *Ry;
    double *omega_xy;                              /* stokes.ω.xy (may be NULL) */
    double *phase_c, *phase_v;                     /* phase_ratios.center / .vertex */
    const double *T;                               /* args.T at the cell centres (ni) for the density laws; may be NULL (T = 0) */
    double *dexx, *deyy, *divU;                    /* strain_increment variant: Δε.xx, Δε.yy, stokes.∇U (ni); may be NULL otherwise */
} jrx_vep2d_fields;

typedef struct jrx_vep2d_params {
    int64_t nx, ny, nxg, nyg;
    double _dx, _dy;
    double dt, r, theta_dtau, eta_dtau, eps_rel, eps_abs;
    int64_t iterMax, iterMin, nout;                /* kwargs (Stokes2D.jl:588-599): iterMax=50e3, iterMin=100, nout=500 */
    uint32_t free_slip, no_slip, periodic;
    double lambda_relaxation, viscosity_relaxation, cutoff_lo, cutoff_hi;
    int32_t verbose;
    int32_t free_surface;                          /* kwarg free_surface: compute_V! / compute_Res! get dt * free_surface (Stokes2D.jl:773,797) */
    int32_t displacement_bcs;                      /* flow_bcs is a DisplacementBoundaryConditions: V = U / dt first, flow_bcs! acts on U */
    int32_t T_ghosted;                             /* args.T is thermal.T (nx+2, ny+2), indexed as the reference does (densities at the cell's own [i, j]) */
    int32_t strain_increment;                      /* kwarg strain_increment (jrx_stokes2d_vep_solve only; Stokes2D.jl:588,659-734, StressKernels.jl:1147-1302): strains
                                                    * from the displacement increments U = V dt, Δε form of the stress update; U and its BCs are refreshed every iteration */
    const double *inv_spacing[6];                  /* non-uniform Geometry: as in jrx_stokes2d_params (all NULL: uniform); not with strain_increment */
} jrx_vep2d_params;

jrx_status jrx_stokes2d_vep_solve(jrx_handle *h, const jrx_vep2d_fields *f, const jrx_rheology *rh, const jrx_vep2d_params *p,
                                  jrx_solve_result *res);
/* solve!(stokes, pt_stokes, grid, flow_bcs, ρg, rheology::MaterialParams, args, dt, igg; kwargs) -- src/stokes/Stokes2D.jl:345-557: the 2D
 * single-phase visco-elasto-plastic driver built on compute_τ_nonlinear! and center2vertex! (test/test_WENO5.jl:226-291).  The rheology is
 * phase 0 of the table (creep law, ConstantElasticity, optional DruckerPrager_regularised, density); phase_c / phase_v are not read.
 * f->T = args.T: cell-centred (ni), or thermal.T (ni.+2) with p->T_ghosted = 1 -- then read at [i, j] by the density and at [i+1, j+1]
 * by the viscosity, as the reference's two argument helpers do. */
jrx_status jrx_stokes2d_nonlinear_solve(jrx_handle *h, const jrx_vep2d_fields *f, const jrx_rheology *rh, const jrx_vep2d_params *p,
                                        jrx_solve_result *res);
/* update_stresses_center_vertex_ps! alone (θ, λ, λv are caller arrays of extents ni, ni, ni.+1) -- for parity tests */
jrx_status jrx_vep2d_update_stresses(jrx_handle *h, const jrx_vep2d_fields *f, const double *theta, double *lambda, double *lambda_v,
                                     const jrx_rheology *rh, const jrx_vep2d_params *p);
/* compute_τ_nonlinear! 2D alone: single phase (multiphase = 0; StressKernels.jl:266-307, uses rheology phase 0) or phases at
 * the cell centres (multiphase = 1; :310-351) with _compute_τ_nonlinear! (rheology/StressUpdate.jl:2-57).  Centre-only
 * visco-elasto-plastic update: writes τ.xx, τ.yy, τ.xy_c, τ.II, η_vep, ε_pl.xx, ε_pl.yy, ε_pl.xy[i,j], λ and θ (caller
 * arrays of extent ni).  τ_o.xy and ε_pl.xy are the vertex arrays addressed with the centre index, as the reference's
 * caller passes them (Stokes2D.jl:442-458).  Softening laws are not modelled (NoSoftening). */
jrx_status jrx_compute_tau_nonlinear2d(jrx_handle *h, const jrx_vep2d_fields *f, double *theta, double *lambda, const jrx_rheology *rh,
                                       const jrx_vep2d_params *p, int32_t multiphase);
/* center2vertex!(vertex, center) 2D (Interpolations.jl:101-114): vertex is (nx+1, ny+1), center (nx, ny) */
jrx_status jrx_center2vertex2d(jrx_handle *h, double *vertex, const double *center, int64_t nx, int64_t ny);
/* tensor_invariant!(A): II = second_invariant_staggered(xx, yy, gather(xy)) -- StressKernels.jl:443-470 */
jrx_status jrx_tensor_invariant2d(jrx_handle *h, double *II, const double *xx, const double *yy, const double *xy, int64_t nx, int64_t ny);
/* compute_viscosity! (fn_viscosity = compute_viscosity_εII) for the table rheology: η <- ν·η_phase + (1-ν)·η, clamped to the cutoff; creep laws that read
 * fields take T (f->T with p->T_ghosted as in the solve), P = f->P and the invariant of @strain_center as the reference's kernel does (Viscosity.jl:382-418) */
jrx_status jrx_vep2d_compute_viscosity(jrx_handle *h, const jrx_vep2d_fields *f, const jrx_rheology *rh, const jrx_vep2d_params *p, double nu);
/* the same with fn_viscosity = compute_viscosity_τII (update_viscosity_τII!, Viscosity.jl:67-106): a power-law creep is evaluated at the invariant of
 * @stress_center (vertices: τ.xy alone, the PT solvers leave τ.xx_v, τ.yy_v zero) instead of the strain rate's; identical for the other creep laws */
jrx_status jrx_vep2d_compute_viscosity_tauII(jrx_handle *h, const jrx_vep2d_fields *f, const jrx_rheology *rh, const jrx_vep2d_params *p, double nu);

/* ------------------------------------------------------------------ 3D multiphase visco-elasto-plastic Stokes
 * solve!(stokes, pt_stokes, grid, flow_bcs, ρg, phase_ratios, rheology, args, dt, igg; kwargs) for 3D grids --
 * src/stokes/Stokes3D.jl:447-668 with update_stresses_center_vertex_ps! 3D (src/stokes/StressKernels.jl:604-989), as
 * test/test_shearband3D_MPI.jl drives it.  Same rheology table as the 2D driver.  Extents: centres ni; edge arrays
 * yz (nx, ny+1, nz+1), xz (nx+1, ny, nz+1), xy (nx+1, ny+1, nz); phase arrays [nphase][extent] with the phase index
 * fastest (JustPIC CellArray).  Pointers marked optional may be NULL. */
typedef struct jrx_vep3d_fields {
    double *P, *P0, *divV, *Q;
    double *Vx, *Vy, *Vz, *Ux, *Uy, *Uz;
    double *exx, *eyy, *ezz, *eyz, *exz, *exy;            /* ε: normals at centres, shear on edges */
    double *eyz_c, *exz_c, *exy_c;                        /* optional: shear2center!(stokes.ε) targets */
    double *eplxx, *eplyy, *eplzz, *eplyz, *eplxz, *eplxy;/* ε_pl */
    double *eplyz_c, *eplxz_c, *eplxy_c;                  /* optional */
    double *deyz, *dexz, *dexy, *deyz_c, *dexz_c, *dexy_c;/* optional: Δε shear and its centre copies */
    double *txx, *tyy, *tzz, *tyz, *txz, *txy;            /* τ: normals at centres, shear on edges */
    double *tyz_c, *txz_c, *txy_c, *tII;                  /* τ shear at centres, second invariant */
    double *toxx, *toyy, *tozz, *toyz, *toxz, *toxy, *toyz_c, *toxz_c, *toxy_c;
    double *eta, *eta_vep;
    double *EII_pl, *evol_pl, *EVol_pl;
    double *fx, *fy, *fz;                                 /* ρg */
    double *RP, *Rx, *Ry, *Rz;
    double *omega_yz, *omega_xz, *omega_xy;               /* optional: vorticity on the edges */
    const double *phase_c, *phase_yz, *phase_xz, *phase_xy;
    const double *T;                                      /* args.T at the cell centres (ni); may be NULL (T = 0) */
} jrx_vep3d_fields;

typedef struct jrx_vep3d_params {
    int64_t nx, ny, nz, nxg, nyg, nzg;
    double _dx, _dy, _dz;
    double dt, r, theta_dtau, eta_dtau, eps_rel, eps_abs;
    int64_t iterMax, nout;
    uint32_t free_slip, no_slip, periodic;
    double lambda_relaxation, viscosity_relaxation, cutoff_lo, cutoff_hi;
    int32_t verbose;
    int32_t displacement_bcs;
    int32_t T_ghosted;             /* f->T = args.T is thermal.T (ni .+ 2), as the miniapps pass it (RisingBlob3D/Blob3D.jl:355): update_ρg! reads it at the cell's own
                                    * [i, j, k], unshifted (getindex_NamedTuple(args, I...), BuoyancyForces.jl:52); 0: cell-centred (ni) */
    int32_t b_width[3];            /* kwargs.b_width (Stokes3D.jl:460): boundary-slab widths of the hidden update_halo!(V) (multi-rank runs; <= 0: 4) */
} jrx_vep3d_params;

jrx_status jrx_stokes3d_vep_solve(jrx_handle *h, const jrx_vep3d_fields *f, const jrx_rheology *rh, const jrx_vep3d_params *p,
                                  jrx_solve_result *res);
/* update_stresses_center_vertex_ps! 3D alone (θ, λ of extent ni; λv = {yz, xz, xy} edge arrays) -- for parity tests.
 * Every update reads the stresses of the previous call (the reference's single launch races on neighbouring values). */
jrx_status jrx_vep3d_update_stresses(jrx_handle *h, const jrx_vep3d_fields *f, const double *theta, double *lambda, double *const lambda_v[3],
                                     const jrx_rheology *rh, const jrx_vep3d_params *p);
/* compute_viscosity! 3D (εII form; ..._tauII: update_viscosity_τII!) for the table rheology: η <- ν·η_phase + (1-ν)·η, clamped to the cutoff; creep laws that
 * read fields: T, P at the cell, invariant of @strain / @stress with the edge components gathered (Viscosity.jl:455-503) */
jrx_status jrx_vep3d_compute_viscosity(jrx_handle *h, const jrx_vep3d_fields *f, const jrx_rheology *rh, const jrx_vep3d_params *p, double nu);
jrx_status jrx_vep3d_compute_viscosity_tauII(jrx_handle *h, const jrx_vep3d_fields *f, const jrx_rheology *rh, const jrx_vep3d_params *p, double nu);
/* tensor_invariant!(A) 3D -- StressKernels.jl:472-487 */
jrx_status jrx_tensor_invariant3d(jrx_handle *h, double *II, const double *xx, const double *yy, const double *zz, const double *yz,
                                  const double *xz, const double *xy, int64_t nx, int64_t ny, int64_t nz);

/* ------------------------------------------------------------------ 2D PT heat diffusion */
typedef struct jrx_thermal2d_fields {
    double *T, *Told, *dT;                 /* (nx+2, ny+2): thermal.T, Told, ΔT */
    double *qTx, *qTx2;                    /* (nx+1, ny)   */
    double *qTy, *qTy2;                    /* (nx, ny+1)   */
    double *H, *shear_heating, *ResT;      /* (nx, ny)     */
    double *K, *rhoCp;                     /* (nx, ny); unused (may be NULL) in the rheology form */
    double *thetar_dtau, *dtau_rho;        /* (nx, ny): pt_thermal.θr_dτ, dτ_ρ */
    /* optional (NULL = absent; the one-launch iterations are not used when any is given) */
    const double *adiabatic;               /* (nx, ny): thermal.adiabatic, the term + adiabatic * T of the rheology forms (DiffusionPT_kernels.jl:553-601,631-668) */
    const double *dirichlet_mask;          /* (nx+2, ny+2): thermal_bc.dirichlet.mask -- cells with mask != 0 take T = (1 - m) T + m value instead of the update
                                            * and have ResT = 0 (Dirichlet.jl:72-105, mask/mask.jl:47-50) */
    const double *dirichlet_value;         /* (nx+2, ny+2) values; NULL with a ConstantDirichletBoundaryCondition (params.dirichlet_const) */
} jrx_thermal2d_fields;

typedef struct jrx_thermal2d_params {
    int64_t nx, ny;
    double _dx, _dy;
    double dt, eps;                        /* pt_thermal.ϵ */
    int64_t iterMax, nout;
    /* faces in the order left, right, top, bot (2D: bot <-> j = 1) */
    int32_t no_flux[4];
    int32_t constant_value_on[4]; double constant_value[4];
    int32_t constant_flux_on[4];  double constant_flux[4];
    int32_t periodic[4];
    /* 0: array-coefficient form (DiffusionPT_solver.jl:34-149);
     * 1: rheology form restricted to constant k, constant Cp, rho = rho0*(1 - alpha*(T - T0))
     *    (DiffusionPT_solver.jl:181-305 as evaluated by test/test_diffusion2D.jl) */
    int32_t rheology_form;
    double k_const, Cp, rho0, alpha, T0;
    int32_t verbose;
    double dirichlet_const;                /* value of a ConstantDirichletBoundaryCondition (read where dirichlet_mask != 0 and dirichlet_value is NULL) */
    /* Non-uniform Geometry (device arrays of inverse spacings, all four NULL on a uniform grid): [0], [1] = _di.center x (nx-1), y (ny-1): compute_flux! reads them
     * at clamp(i, 1, nx-1) (DiffusionPT_kernels.jl:338,354,405,433,482,510); [2], [3] = _di.vertex x (nx), y (ny): update_T! of the rheology / phase forms and
     * check_res! (:579-580,616,647-648).  Not with the array form (K, ρCp arrays): its update_T! indexes _di.center beyond its extent in the reference (:532). */
    const double *inv_spacing[4];
} jrx_thermal2d_params;

/* heatdiffusion_PT!(thermal, pt_thermal, thermal_bc, K, ρCp | rheology, args, dt, grid; kwargs) */
jrx_status jrx_heatdiffusion_PT2d(jrx_handle *h, const jrx_thermal2d_fields *t, const jrx_thermal2d_params *p,
                                  int64_t *iter_count, double *norm_ResT, int64_t cap, int64_t *nnorms);
/* thermal_bcs! -- BoundaryConditions.jl:39-53 */
jrx_status jrx_thermal_bcs2d(jrx_handle *h, double *T, const jrx_thermal2d_params *p);
/* one PT iteration: compute_flux! + update_T! + thermal_bcs! (DiffusionPT_solver.jl:236-261) */
jrx_status jrx_thermal2d_iteration(jrx_handle *h, const jrx_thermal2d_fields *t, const jrx_thermal2d_params *p);
/* check_res! (DiffusionPT_kernels.jl:603-668) */
jrx_status jrx_thermal2d_check_res(jrx_handle *h, const jrx_thermal2d_fields *t, const jrx_thermal2d_params *p);

/* ------------------------------------------------------------------ small backend generics of the method table
 * (src/ext/AMDGPU/3D.jl:205-239): velocity2displacement!, displacement2velocity! (types/displacement.jl:2-60), compute_dt (Utils.jl:492-519).
 * Arrays in the order x, y, z with n[d] elements each; the third pointer may be NULL in 2D. */
jrx_status jrx_velocity2displacement(jrx_handle *h, double *const U[3], const double *const V[3], const int64_t n[3], double dt);
jrx_status jrx_displacement2velocity(jrx_handle *h, double *const V[3], const double *const U[3], const int64_t n[3], double dt);
/* dt = min(dt_diff, 0.9 * min_d(di[d] / max|V_d|)); max over all ranks when a communicator is active; dt_diff = INFINITY to ignore */
jrx_status jrx_compute_dt(jrx_handle *h, const double *const V[3], const int64_t n[3], const double di[3], int32_t ndim, double dt_diff, double *dt_out);

/* ------------------------------------------------------------------ post-loop epilogue operators, stand-alone
 * The VEP drivers run these themselves (Stokes2D.jl:831-846, Stokes3D.jl:640-658); time-stepping scripts also call them directly. */
/* shear2center!(A): shear components averaged to the cell centres -- Interpolations.jl:291-323 */
jrx_status jrx_shear2center2d(jrx_handle *h, double *xy_c, const double *xy, int64_t nx, int64_t ny);
jrx_status jrx_shear2center3d(jrx_handle *h, double *yz_c, double *xz_c, double *xy_c, const double *yz, const double *xz, const double *xy,
                              int64_t nx, int64_t ny, int64_t nz);
/* accumulate_tensor!(II, A, dt): II += second_invariant_staggered(A) * dt -- StressKernels.jl:364-408 */
jrx_status jrx_accumulate_tensor2d(jrx_handle *h, double *II, const double *xx, const double *yy, const double *xy, double dt, int64_t nx, int64_t ny);
jrx_status jrx_accumulate_tensor3d(jrx_handle *h, double *II, const double *xx, const double *yy, const double *zz, const double *yz, const double *xz,
                                   const double *xy, double dt, int64_t nx, int64_t ny, int64_t nz);
/* accumulate_vol!(EVol_pl, ε_vol_pl, dt): EVol_pl += dt * ε_vol_pl over n cells -- StressKernels.jl:410-431 */
jrx_status jrx_accumulate_vol(jrx_handle *h, double *EVol, const double *evol, double dt, int64_t n);
/* compute_vorticity! as the drivers call it -- stress_rotation_particles.jl:17-50 (2D at the vertices with the velocity-node
 * spacings; 3D on the edges, forward differences at the un-shifted velocity index exactly as the reference's `_di` method) */
jrx_status jrx_compute_vorticity2d(jrx_handle *h, double *wxy, const double *Vx, const double *Vy, int64_t nx, int64_t ny, double _dx, double _dy);
jrx_status jrx_compute_vorticity3d(jrx_handle *h, double *wyz, double *wxz, double *wxy, const double *Vx, const double *Vy, const double *Vz,
                                   int64_t nx, int64_t ny, int64_t nz, double _dx, double _dy, double _dz);

/* ------------------------------------------------------------------ 3D PT heat diffusion
 * heatdiffusion_PT!(thermal, pt_thermal, thermal_bc, K, ρCp | rheology, args, dt, grid; kwargs) for 3D grids --
 * src/thermal_diffusion/DiffusionPT_solver.jl:34-149,181-305 with the 3D kernels DiffusionPT_kernels.jl:6-61,160-199,250-282.
 * Faces in the order left, right, front, back, top, bot; 3D thermal naming: bot <-> k = 1, top <-> k = end
 * (constant_value.jl:15-33) -- the opposite of the 3D velocity free-slip naming. */
typedef struct jrx_thermal3d_fields {
    double *T, *Told, *dT;                 /* (nx+2, ny+2, nz+2): thermal.T, Told, ΔT */
    double *qTx, *qTx2;                    /* (nx+1, ny, nz) */
    double *qTy, *qTy2;                    /* (nx, ny+1, nz) */
    double *qTz, *qTz2;                    /* (nx, ny, nz+1) */
    double *H, *shear_heating, *ResT;      /* ni */
    const double *K, *rhoCp;               /* ni: array-coefficient form; NULL in the rheology form */
    const double *thetar_dtau, *dtau_rho;  /* ni: PTThermalCoeffs (rewritten every iteration by jrx_heatdiffusion_PT3d_phases) */
    const double *adiabatic, *dirichlet_mask, *dirichlet_value;   /* optional, as in jrx_thermal2d_fields; mask / value (nx+2, ny+2, nz+2) */
} jrx_thermal3d_fields;

typedef struct jrx_thermal3d_params {
    int64_t nx, ny, nz;
    double _dx, _dy, _dz;
    double dt, eps;
    int64_t iterMax, nout;
    int32_t no_flux[6];
    int32_t constant_value_on[6]; double constant_value[6];
    int32_t constant_flux_on[6];  double constant_flux[6];
    int32_t periodic[6];
    int32_t rheology_form;                 /* 1: k_const, Cp, PT_Density(rho0, alpha, T0) as in test/test_diffusion3D.jl */
    double k_const, Cp, rho0, alpha, T0;
    int32_t verbose;
    double dirichlet_const;
} jrx_thermal3d_params;

jrx_status jrx_heatdiffusion_PT3d(jrx_handle *h, const jrx_thermal3d_fields *t, const jrx_thermal3d_params *p, int64_t *iter_count, double *norm_ResT,
                                  int64_t cap, int64_t *nnorms);
jrx_status jrx_thermal_bcs3d(jrx_handle *h, double *T, const jrx_thermal3d_params *p);
jrx_status jrx_thermal3d_iteration(jrx_handle *h, const jrx_thermal3d_fields *t, const jrx_thermal3d_params *p);
jrx_status jrx_thermal3d_check_res(jrx_handle *h, const jrx_thermal3d_fields *t, const jrx_thermal3d_params *p);

/* ------------------------------------------------------------------ phase-ratio form of the heat-diffusion path
 * heatdiffusion_PT!(thermal, pt_thermal, thermal_bc, rheology, args, dt, grid; kwargs = (phase = phase_ratios, ...)) --
 * src/thermal_diffusion/DiffusionPT_solver.jl:181-305 with phase !== nothing: update_pt_thermal_arrays! every iteration
 * (DiffusionPT_coefficients.jl:105-136), conductivity from the face phase ratios (DiffusionPT_kernels.jl:366-440 / :62-158), ρCp and
 * radioactive heat from the centre ratios (:553-601, :631-668 / :200-249, :283-325).  Per phase: ConstantConductivity k,
 * ConstantHeatCapacity Cp, ConstantRadioactiveHeat H_r and a density law (rho_kind as in jrx_rheology: 0 constant, 1 PT_Density,
 * 2 T_Density, 3 Compressible_Density).  p->rheology_form is ignored (set to 2 internally); t->K / t->rhoCp are not read. */
typedef struct jrx_thermal_phases {
    int32_t nphase;
    double k[JRX_MAXPHASE], Cp[JRX_MAXPHASE], Hr[JRX_MAXPHASE];
    int32_t rho_kind[JRX_MAXPHASE];
    double rho0[JRX_MAXPHASE], alpha[JRX_MAXPHASE], beta[JRX_MAXPHASE], T0[JRX_MAXPHASE], P0[JRX_MAXPHASE];
    double max_lxyz, Vpdtau;               /* pt_thermal.max_lxyz, pt_thermal.Vpdτ */
} jrx_thermal_phases;

typedef struct jrx_thermal_phase_fields {
    const double *P;                       /* args.P (ni) */
    const double *phase_c;                 /* phase_ratios.center [nphase][ni], phase index fastest */
    const double *phase_qx, *phase_qy, *phase_qz;   /* phase_ratios.Vx (nx+1, ny[, nz]), .Vy (nx, ny+1[, nz]), .Vz (nx, ny, nz+1); qz NULL in 2D */
} jrx_thermal_phase_fields;

jrx_status jrx_heatdiffusion_PT2d_phases(jrx_handle *h, const jrx_thermal2d_fields *t, const jrx_thermal2d_params *p, const jrx_thermal_phases *ph,
                                         const jrx_thermal_phase_fields *pf, int64_t *iter_count, double *norm_ResT, int64_t cap, int64_t *nnorms);
jrx_status jrx_heatdiffusion_PT3d_phases(jrx_handle *h, const jrx_thermal3d_fields *t, const jrx_thermal3d_params *p, const jrx_thermal_phases *ph,
                                         const jrx_thermal_phase_fields *pf, int64_t *iter_count, double *norm_ResT, int64_t cap, int64_t *nnorms);
/* update_pt_thermal_arrays!(pt_thermal, phase_ratios, rheology, args, _dt) -- DiffusionPT_coefficients.jl:105-136: writes thetar_dtau, dtau_rho (ni)
 * from T (ni.+2, read at Idx.+1), P and the centre ratios; n = {nx, ny, nz} (nz = 1 in 2D) */
jrx_status jrx_update_pt_thermal_arrays(jrx_handle *h, double *thetar_dtau, double *dtau_rho, const double *T, const int64_t n[3], int32_t ndim, double dt,
                                        const jrx_thermal_phases *ph, const jrx_thermal_phase_fields *pf);
/* adiabatic_heating!(thermal, stokes, rheology, phases, _dt) -- DiffusionPT_kernels.jl:720-746: A = (P - P0) * α * _dt over ncells cells, α the phase-weighted thermal
 * expansivity of the density laws (PT_Density, T_Density: α; ConstantDensity: 0 -- the values test/test_rheology.jl:57-64 asserts of get_α; Compressible_Density: 0,
 * assumed); phase_c NULL: phase 0 alone */
jrx_status jrx_adiabatic_heating(jrx_handle *h, double *adiabatic, const double *P, const double *P0, int64_t ncells, double dt, const jrx_thermal_phases *ph,
                                 const double *phase_c);

/* ------------------------------------------------------------------ grid operators of a time step, either side of solve! / heatdiffusion_PT! */
/* The methods the reference's AMDGPU extension forwards to the generic kernels (src/ext/AMDGPU/2D.jl:301-352, 3D.jl:311-362).  Shapes as in
 * StokesArrays: Vx (nx+1, ny+2[, nz+2]), Vy (nx+2, ny+1[, nz+2]), Vz (nx+2, ny+2, nz+1); shear yz (nx, ny+1, nz+1), xz (nx+1, ny, nz+1), xy (nx+1, ny+1[, nz]).
 * velocity2vertex!(Vx_v, Vy_v[, Vz_v], Vx, Vy[, Vz]) -- Interpolations.jl:212-249: the kernel runs over size(Vx_v) = (mx, my[, mz]) <= ni .+ 1 (the caller's
 *   choice, as in the reference: ni .+ 1 in the miniapps, ni in test/test_Interpolations.jl:150-164);
 * velocity2center! -- :257-289, outputs ni;
 * vertex2center!(center, vertex; ghost_x, ghost_y, ghost_z) -- :72-96: over size(vertex) .- 1, written at I .+ ghost of a centre array of extents cdim;
 * center2vertex_harm! -- :116-137 (2D, clamped harmonic mean; vertex (nx+1, ny+1));
 * center2vertex!(vertex_yz, vertex_xz, vertex_xy, center_yz, center_xz, center_xy) -- :139-178 (3D; boundary edges are not written).
 * The 2D center2vertex!(vertex, center) is jrx_center2vertex2d above. */
jrx_status jrx_velocity2vertex2d(jrx_handle *h, double *Vx_v, double *Vy_v, const double *Vx, const double *Vy, int64_t nx, int64_t ny, int64_t mx, int64_t my);
jrx_status jrx_velocity2vertex3d(jrx_handle *h, double *Vx_v, double *Vy_v, double *Vz_v, const double *Vx, const double *Vy, const double *Vz, int64_t nx,
                                 int64_t ny, int64_t nz, int64_t mx, int64_t my, int64_t mz);
jrx_status jrx_velocity2center2d(jrx_handle *h, double *Vx_c, double *Vy_c, const double *Vx, const double *Vy, int64_t nx, int64_t ny);
jrx_status jrx_velocity2center3d(jrx_handle *h, double *Vx_c, double *Vy_c, double *Vz_c, const double *Vx, const double *Vy, const double *Vz, int64_t nx,
                                 int64_t ny, int64_t nz);
jrx_status jrx_vertex2center(jrx_handle *h, double *center, const double *vertex, const int64_t vdim[3], const int64_t cdim[3], int32_t ndim, int32_t ghost_x,
                             int32_t ghost_y, int32_t ghost_z);
jrx_status jrx_center2vertex_harm2d(jrx_handle *h, double *vertex, const double *center, int64_t nx, int64_t ny);
jrx_status jrx_center2vertex3d(jrx_handle *h, double *vertex_yz, double *vertex_xz, double *vertex_xy, const double *center_yz, const double *center_xz,
                               const double *center_xy, int64_t nx, int64_t ny, int64_t nz);
/* compute_ρg!(ρg[end], [phase_ratios,] rheology, (; T, P)) -- rheology/BuoyancyForces.jl:6-60, the scalar-gravity form: rhog = density * gravity of the first
 * phase over the n = {nx, ny[, nz]} cells (the density laws of jrx_rheology; has_density must be set).  phase_c NULL: single-phase form (phase 0); T, P may be
 * NULL (= 0); P has the extents n.  tdim: the extents of args.T (NULL: n) -- T is read at the cell's own [i, j, k] of that array, so a ghosted thermal.T passed
 * as args.T is read without the shift to the centres, as the reference's getindex_NamedTuple(args, I...) does (test/test_WENO5.jl:208-214). */
jrx_status jrx_compute_rhog(jrx_handle *h, double *rhog, const jrx_rheology *rh, const double *phase_c, const double *T, const double *P, const int64_t n[3],
                            const int64_t tdim[3], int32_t ndim);
/* compute_lithostatic_pressure!(P, ρg, dz) -- src/Utils.jl:521-573: P[j] = Σ_{k>j} ρg[k] dz[k] + ρg[j] dz[j] / 2 down the columns of the last dimension (the
 * vertical, pointing up); dz_cells: one height per cell of that dimension, or NULL for the constant height dz.  n = {nx, ny[, nz]}.  The four-argument IGG
 * form (weight of the ranks stacked above) is not built: refused when the handle's communicator splits the vertical direction. */
jrx_status jrx_compute_lithostatic_pressure(jrx_handle *h, double *P, const double *rhog, double dz, const double *dz_cells, const int64_t n[3], int32_t ndim);
/* compute_viscosity!(stokes, args, rheology::MaterialParams, cutoff; relaxation = ν) -- rheology/Viscosity.jl:118-167, for the creep laws of the rheology table
 * (phase 0: LinearViscous or the Arrhenius table; they do not depend on the strain rate): eta <- clamp(ν η_creep(T, P) + (1 - ν) eta, cutoff).  args.T has the
 * extents tdim: ni .+ 2 (the ghosted thermal.T, read at I .+ 1 as local_viscosity_args does, Viscosity.jl:513-523) or ni / NULL (cell centres); P: ni or NULL.
 * The phase-ratio form is jrx_vep{2d,3d}_compute_viscosity. */
jrx_status jrx_compute_viscosity_single(jrx_handle *h, double *eta, const jrx_rheology *rh, const double *T, const double *P, const int64_t n[3],
                                        const int64_t tdim[3], int32_t ndim, double nu, double cutoff_lo, double cutoff_hi, const double *AII, int32_t tau_form);
/* AII (ni, or NULL): the invariant array of compute_viscosity_εII! / compute_viscosity_τII!(η, ν, AII, args, rheology, cutoff) (Viscosity.jl:169-196);
 * tau_form: AII is a stress invariant.  A power-law creep (visc_kind 2) needs AII here; the 2D single-material driver forms it from @strain(stokes) itself. */
/* compute_shear_heating!(thermal, stokes, [phase_ratios,] rheology, dt) -- thermal_diffusion/ShearHeating.jl:14-71:
 * shear_heating = max(0, Χ τ : (ε - ε_el)), ε_el = (τ - τ_o) / (2 G dt), at the cell centres.  tau, tau_o: @tensor_center(stokes.τ / τ_o) in Voigt order
 * (2D: xx, yy, xy_c; 3D: xx, yy, zz, yz_c, xz_c, xy_c), eps: @strain(stokes) (shear components on their edges, averaged to the centre as cache_tensors does).
 * G from rh (fn_ratio(get_shear_modulus, ...) with phase_c, phase 0 without); chi[q] = Χ of phase q's ConstantShearheating (0: no law).
 * [compute_shearheating is GeoParams': Χ Σ τ_ij (ε_ij - ε_el_ij) with the shear terms of the Voigt tuple counted twice -- form ASSUMED, parity unpinned.]
 * n = {nx, ny, nz}; ndim 2 or 3. */
jrx_status jrx_compute_shear_heating(jrx_handle *h, double *shear_heating, const double *const *tau, const double *const *tau_o, const double *const *eps,
                                     const double *phase_c, const jrx_rheology *rh, const double *chi, double dt, const int64_t n[3], int32_t ndim);

/* ------------------------------------------------------------------ timing hooks for bench.py */
/* Runs `iters` PT iterations of the 3D loop body back to back (no norm checks) and reports device times
 * measured with hipEvents on the handle's stream inside that batch:
 *   times_ms[0] whole batch; [1] mean stand-alone stress sweep; [2] mean stand-alone velocity sweep;
 *   [3] mean fused launch group (k_fused3d = velocity sweep m + BCs + stress sweep m+1, then the ghost-plane and
 *   boundary-plane launches), 0 when nothing was fused; [4] mean k_fused3d launch alone (without neighbours: the launch over the
 *   interior tiles, while the high-face tiles and the boundary layers run beside it on the halo stream); [5] the number of cells
 *   whose stresses the launch timed in [4] updates (its units; 0 when nothing was fused). */
jrx_status jrx_stokes3d_iterate_timed(jrx_handle *h, const jrx_stokes3d_fields *f, const double *etatau,
                                      const jrx_stokes3d_params *p, int64_t iters, double times_ms[6]);

#ifdef __cplusplus
}
#endif
#endif /* JRX_H */
