/* jrx_tuning.h -- tuning / A-B / test switches of libjrx_hip.  NOT part of the drop-in ABI (include/jrx.h): nothing a caller of
 * solve! needs is here, results never depend on any of these, and keys may change between builds.  They exist so that the
 * measurements under profiles/ and the parity tests can select a kernel form at run time instead of rebuilding.
 *
 * Keys (int64 values):
 *   "fused_ylds" (default 1)      3D fused kernel: y-neighbour operands through LDS (0: lane-shuffle-only form)
 *   "visc_fold" (1)               3D fused kernel, viscous-limit form: the arithmetic with its exact zeros folded away for finite η (dτ_r = 1 / (θ_dτ + 1) once per thread, Δτ = dτ_r fma(2η, ε, -τ),
 *                                 no division by 1 + 0 ψ); same bits (0: the general expressions with zero operands)
 *   "zero_forces" (1)             3D fused kernel (one-launch viscous-limit form and general form) and z-marching velocity sweep (3D Stokes and 3D VEP drivers): ρg arrays whose every entry
 *                                 is +0.0 (all 64 bits zero; a pass of the driver call looks at them) are not loaded -- ρg_x and ρg_y (gravity along z), or all three (SolVi3D, ShearBand3D);
 *                                 x - (+0.0) = x for every x, so the bits are the same (0: always loaded)
 *   "end_flips" (1)               jrx_stokes3d_iterate_timed, no neighbours: a batch whose number of fused steps is odd ends in the caller's arrays because its first stress sweep and its last
 *                                 velocity sweep write out of place into the other state set (0: the first iteration stays un-fused instead -- one more sweep pair; same results)
 *   "scratch_stagger" (0)         bytes (multiple of 256): array q of the library's second 3D state set starts q * stagger bytes into its allocation (scripts/bench_alloc_stagger.py: the spread
 *                                 of the 512^3 kernel between allocations does not depend on it -- it comes with the physical placement, not with the low address bits)
 *   "scratch_contiguous" (0)      the library's second 3D state set in physically contiguous device memory (hipExtMallocWithFlags, hipDeviceMallocContiguous): the probe of
 *                                 profiles/r04_alloc_stagger.txt -- a process whose arrays are ALL physically contiguous runs the 512^3 kernel at the slow rate, every time
 *   "fused_hiface" (1)            3D fused kernel, viscous-limit form, no neighbours: the stress nodes on the high faces i = nx, j = ny, k = nz are updated inside the kernel
 *                                 (0: by the boundary-layer launch behind it)
 *   "fused_first_pct" (15)        multi-rank fused pipeline with the neighbour faces inside the kernel (fused_overlap = 3): share (%) of the interior z chunks whose tiles are launched
 *                                 beside update_halo!(V); the rest of the block -- shell tiles among their row neighbours -- follows behind it
 *   "comm_bcs_lazy" (0)           multi-rank fused pipeline: 1 = flow_bcs! of the physical faces applied lazily (before anything reads those entries from memory) instead of twice per
 *                                 iteration; the fix-up next to the received planes derives them by rule (measured 2 % slower than the two launches: off)
 *   "fused_tile" (2)              3D fused kernel tile: 0 = 64 x 4 threads, 1 = 32 x 8, 3 = 64 x 8 (two 8-wave blocks per CU, XCD bands of four tile rows), 2 = chosen by the grid (32 x 8 for nx = 63 .. 90,
 *                                 where three 32-lane tiles replace two 64-lane ones; 64 x 8 where the launch keeps >= 4,096 blocks, i.e. from ~230^3 on)
 *   "fused_split" (0)             no neighbours: high-face tiles + boundary stress layers forked onto the halo stream
 *   "b_width_x/y/z" (0)           > 0 overrides jrx_stokes3d_params.b_width of the split sweeps
 *   "fused2d" (1)                 2D visco-elastic loop: one-launch iterations on launch-bound grids
 *   "vep3_edges" (4)              3D VEP edge pass: 4 = z-marching kernel, the three family waves of a row share the centre and shear operands through LDS;
 *                                 6 = the same with a fourth wave per workgroup that loads and publishes every operand (phase ratios and λv included) one plane step ahead of the
 *                                 family waves, which then have no load of their own (bit-identical; measured equal to 4, profiles/r04_vep3d_fused_pre_centre.txt);
 *                                 3 = centre operands only; 1 = no LDS, one family per block; 2 = one launch per family; 0 = one node per thread
 *   "vep3_cfg", "vep3_peel", "vep3_peel_fork", "vep3_map", "vep3_xcd"     z-marching edge kernel: chunk depth / occupancy, peeling of a thin last segment, thread map, XCD slabs
 *   "vep3_nt" (0), "vep3_prekz" (0)   3D VEP: non-temporal stores of the edge pass; planes per thread of the z-marching pre kernel (0 = chosen by the grid size; 1, 2, 4, 8, 16, 32)
 *   "vep3_hide_comm" (0)          multi-rank 3D VEP driver: 2 = ητ, edge-stress and V exchanges on the halo stream beside independent kernels; 1 = the first two only, update_halo!(V) behind the
 *                                 whole velocity sweep; 0 (default: the fastest on one device since the pre / centre kernels are fused) = everything on the compute stream, in order
 *   "vep3_fuse_pc" (1)            3D VEP driver (with and without neighbours), viscosity laws that read no field, no softening law: compute_∇V! / compute_P! / compute_strain_rate!, update_viscosity_τII! and the centre half of
 *                                 update_stresses_center_vertex_ps! run as ONE kernel ahead of the edge half (0 = the three kernels, centre half behind the edge half); bit-identical
 *   "vep3_np_const" (1)           VEP kernels, 2D and 3D (3D: centre pass, fused pre / centre kernel, per-node edge kernel; 2D: the merged stress kernel): 1 = instantiations with the number of
 *                                 phases as a compile-time constant (1..4: ratios loaded in one batch, phase loops unrolled); 0 = the run-time loops (A/B)
 *   "thermal_np_const" (1)        phase-ratio form of the heat-diffusion kernels (2D, 3D): 1 = instantiations with the number of phases as a compile-time constant (1..4; in 3D also the flux kernel
 *                                 with batched loads and, with "thermal_fused_ph", the one-launch iteration); 0 = run-time loops, two kernels per iteration (A/B)
 *   "fused2d_batch" (1)           2D visco-elastic loop, one-launch iteration: 1 = the form that requests every operand up front (k_fused2d_b; dt = Inf: its viscous-limit instantiation,
 *                                 which does not load τ_o, P0, K, G, Q, behind the operand check of "viscous_limit"); 0 = the control-flow form (A/B)
 *   "fused2d_max_nodes" (1200000)  ... on grids of up to this many nodes (larger: the two-kernel iteration)
 *   "vep3_prec_tile" (2)          thread map of the fused 3D VEP pre / centre kernel: 1 = 64 x 4 tiles of node columns, 0 = 256 consecutive nodes of the flattened plane, 2 = tiles from 16,384 node columns per plane
 *   "thermal_fused_ph" (1)        3D heat diffusion, phase-ratio form (phase count 1..4): 1 = unobserved iterations as one launch (flux + update + BCs + next PT coefficients; θr_dτ ping-pongs), 0 = two kernels
 *   "vep3_fork" (0)               3D VEP driver without neighbours: 1 = the centre pass of update_stresses_center_vertex_ps! runs on a second stream beside the edge pass (it writes a
 *                                 second set of τxx, τyy, τzz, adopted by pointer swap); measured equal to one pass after the other: off
 *   "vep_store_all" (0)           VEP loops (2D and 3D): 1 = every iteration stores ∇V, RP, ε_pl, ε_vol_pl, τII, η_vep (default: only iterations whose results can be observed)
 *   "thermal_cfg", "thermal_xg"   fused 3D heat-diffusion tile shape / XCD band
 *   "thermal_tile" (0), "thermal_nt" (0)   round 6 A/Bs of the fused 3D heat-diffusion iteration: 4 / 8 = 64 x TY tiles whose rows exchange (T, K, θ) and the y flux through LDS (bit-identical, 9 - 16 %
 *                                 slower); 1 = non-temporal stores of the new (T, qT) set (neutral) -- profiles/r06_thermal3d_tile.txt
 *   "halo_self_rccl" (0)          test hook: a rank that is its own periodic neighbour routes its planes through ncclSend/ncclRecv
 *   "comm_timeout_ms" (120000)    in-process and ipc transports (jrx_comm_init_local / _ipc): how long a rank waits for a neighbour (host waits and the device-side flag waits)
 *   "chain_profile" (0)           jrx_stokes3d_iterate_timed on a multi-rank handle also records events around the stages of every sampled fused step; read with
 *                                 jrx_tuning_chain_profile
 *   "field_shuffle" (1)          "field_placement" = 1: 0 = chunks dealt in creation order (A/B of the random dealing)
 *   "scratch_poison" (0)         test switch, bit mask: arrays the library allocates are filled with NaNs first (what they hold before their first use must not matter): 1 = the second state sets,
 *                                 2 = the library-owned ητ, 4 = the arrays of jrx_field_alloc
 *   "nbr_feeder" (1)             in-kernel neighbour faces ("fused_overlap" = 3), low x face: the idle feeder lane of the face's tiles holds the received plane as "column -1" (round 6: -2 % on
 *                                 the coupled kernel, profiles/r06_low_face_feeder.txt); 0 = column 0 loads those entries itself behind the barrier (A/B)
 *   "fused_ym" (0)               one-launch viscous-limit kernel, 64 x 8 tile: 2 / 4 = a block marches that many tile rows in y and hands the halo row on in LDS (round 6; bit-identical, measured
 *                                 4 - 8 % slower at 512^3 and fetching more, not less: the march loses the L2 sharing between concurrent y neighbours -- profiles/r06_y_halo.txt); 0 = one tile per block.
 *                                 "stat_fused3d_ym" (read-only) counts its launches.  "fused_tile" = 4: a 64 x 16 tile, one 16-wave block per CU (6 % fewer bytes fetched, 8 % slower)
 *   "fused_kz" (0)               chunk depth of the 64 x 8 tile of k_fused3d: 0 = 12 planes from nz = 384 on, 8 below (scripts/kbench_kz.hip); 8 / 12 force a depth
 *   "field_pool_pct" (70)         "field_placement" = 1 with chunks of >= 128 MiB: the first allocation of a chunk size creates chunks for this share of the free memory (less 6 GiB), and every
 *                                 array takes random chunks of that pool -- chunks from all over the device's memory are what makes a placement good (profiles/r05_placement_search.txt,
 *                                 section 13); jrx_field_trim releases what nobody took; 0 = no pool (chunks are created as needed)
 *   "general_hif" (0)            3D fused kernel, general form (any dt), 64 x 4 tile: the stress nodes on the high faces i = nx, j = ny, k = nz are updated inside the kernel (one launch per
 *                                 unobserved iteration) and, with neighbours, the kernel's boundary tiles read the received planes ("fused_overlap" = 3: no flow_bcs! launch, no fix-up):
 *                                 4 / 3 = the instantiation built for four (128 VGPRs + 28 dwords of scratch: 10.99 ms at 512^3) / three (155 VGPRs: 7.95 ms) waves per SIMD; 0 = boundary-layer launch (7.47 + 0.11 ms)
 *                                 and early exchange (default: the faster pipeline, with and without neighbours -- profiles/r05_general_one_launch.txt)
 */
#ifndef JRX_TUNING_H
#define JRX_TUNING_H
#include "jrx.h"
#ifdef __cplusplus
extern "C" {
#endif
jrx_status jrx_tuning_set(jrx_handle *h, const char *key, int64_t value);
jrx_status jrx_tuning_get(jrx_handle *h, const char *key, int64_t *value);
/* what a rank's fused iteration spent where in the last jrx_stokes3d_iterate_timed call with "chain_profile" = 1, averages in microseconds over `samples` steps:
 * [0] k_fused3d, [1] boundary-slab velocity launches of the early exchange, [2] flow_bcs! before the exchange, [3] update_halo!(V) (pack, transport, waiting for
 * the neighbour, unpack), [4] flow_bcs! behind the join, [5] stress fix-up next to the received planes, [6] the whole step, [7] the step beyond k_fused3d.  With the
 * early exchange [1]..[3] run on the halo stream beside [0]. */
jrx_status jrx_tuning_chain_profile(jrx_handle *h, double out_us[8], int64_t *samples);
#ifdef __cplusplus
}
#endif
#endif
