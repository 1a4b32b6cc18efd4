/*
 * jrx_oracle.h -- CPU restatement ("oracle") of the JustRelax.jl pseudo-transient hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (justrelax.jl_amd/, include/) may
 * include, link, import or call this code.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and there only as the checker / CPU baseline.
 *
 * Parity status: the reference is pure Julia and cannot be run in the build container
 * (no julia binary, no network).  The restatement is pinned by the reference's own
 * known-answer tests (test/test_mini_kernels.jl, test/test_Utils.jl, test/test_types.jl,
 * analytic benchmark thresholds of test_stokes_taylor_green.jl / test_stokes_solvi3D.jl /
 * test_stokes_solcx.jl / test_stokes_elastic_buildup.jl / test_diffusion2D.jl) -- see
 * tests/test_oracle_*.py.  It has never been diffed against outputs of the Julia code.
 *
 * Every function cites the reference file:line (relative to /root/reference) it follows.
 * One loop nest per reference kernel, no fusion, outer-loop OpenMP threading (this mirrors
 * ParallelStencil's Threads backend and is what bench.py times as cpu_baseline kind="port").
 * Arrays are dense, column-major (x fastest), fp64 -- Julia's layout.
 * Build with -ffp-contract=off: fma() appears exactly where the reference has fma/muladd.
 */
#ifndef JRX_ORACLE_H
#define JRX_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- MiniKernels.jl scalar helpers (1-based i,j,k exactly as in the reference tests) ---- */
double orc_mini2(const char *name, const double *A, int n1, int n2, double d, int i, int j);
double orc_mini3(const char *name, const double *A, int n1, int n2, int n3, double d, int i, int j, int k);
double orc_div2(const double *Ax, const double *Ay, int n1, int n2, double _dx, double _dy, int i, int j);
double orc_div3(const double *Ax, const double *Ay, const double *Az, int n1, int n2, int n3,
                double _dx, double _dy, double _dz, int i, int j, int k);
double orc_mysum(int use_inv, const double *A, int n1, int n2, int n3,
                 int i0, int i1, int j0, int j1, int k0, int k1);
double orc_compute_dtau_r(double theta_dtau, double eta, double _Gdt);

/* ---- 3D Stokes fields (all pointers host memory) ---- */
typedef struct orc_fields3d {
    double *P, *P0, *divV, *Q;
    double *Vx, *Vy, *Vz;
    double *Ux, *Uy, *Uz;
    double *txx, *tyy, *tzz, *tyz, *txz, *txy;
    double *toxx, *toyy, *tozz, *toyz, *toxz, *toxy;
    double *exx, *eyy, *ezz, *eyz, *exz, *exy;
    double *eta, *K, *G;
    double *fx, *fy, *fz;
    double *RP, *Rx, *Ry, *Rz;
    /* centre copies of the shear stresses (xx/yy/zz alias the staggered ones); may be NULL */
    double *tyz_c, *txz_c, *txy_c, *toyz_c, *toxz_c, *toxy_c;
} orc_fields3d;

typedef struct orc_params3d {
    int64_t nx, ny, nz;          /* local cell counts = size(stokes.P) */
    int64_t nxg, nyg, nzg;       /* nx_g(), ny_g(), nz_g() (== nx,ny,nz on one rank) */
    double _dx, _dy, _dz;        /* inverse spacings */
    double dt, r, theta_dtau, eta_dtau;
    double eps_rel, eps_abs;
    int64_t iterMax, nout;
    uint32_t free_slip, no_slip, periodic; /* bit per face, see JRX_FACE_* in include/jrx.h */
    int32_t displacement_bcs;    /* DisplacementBoundaryConditions: V = U / dt first (Stokes3D.jl:72), flow_bcs! acts on U (BoundaryConditions.jl:71-78) */
} orc_params3d;

typedef struct orc_result {
    int64_t iter;
    int64_t nchecks;
    int32_t status;              /* 0 ok, 1 NaN residual */
    double *err_evo1; int64_t *err_evo2;   /* capacity cap */
    double *norm_Rx, *norm_Ry, *norm_Rz, *norm_divV;
    int64_t cap;
    double time_s;
} orc_result;

/* single kernels, one per reference kernel */
void orc_compute_divV3d(double *divV, const double *Vx, const double *Vy, const double *Vz,
                        int64_t nx, int64_t ny, int64_t nz, double _dx, double _dy, double _dz);
void orc_compute_P3d(double *P, const double *P0, double *RP, const double *divV, const double *Q,
                     const double *eta, const double *K, const double *G, int64_t n,
                     double dt, double r, double theta_dtau);
void orc_compute_strain_rate3d(const orc_fields3d *f, const orc_params3d *p);
void orc_compute_tau3d(const orc_fields3d *f, const orc_params3d *p);
void orc_compute_V3d(const orc_fields3d *f, const double *etatau, const orc_params3d *p);
void orc_velocity2displacement3d(const orc_fields3d *f, const orc_params3d *p);
void orc_flow_bcs3d(double *Vx, double *Vy, double *Vz, int64_t nx, int64_t ny, int64_t nz,
                    uint32_t free_slip, uint32_t no_slip, uint32_t periodic);
void orc_compute_maxloc3d(double *B, const double *A, int64_t nx, int64_t ny, int64_t nz);
void orc_compute_maxloc2d(double *B, const double *A, int64_t nx, int64_t ny);
/* sums of squares: Rx/Ry/Rz interior slice [2:end-1]^3, RP whole -- Stokes3D.jl:127-142 */
void orc_residual_sumsq3d(const orc_fields3d *f, const orc_params3d *p, double out[4]);
/* one PT iteration = Stokes3D.jl:78-121 (no halo exchange: single rank) */
void orc_stokes3d_iteration(const orc_fields3d *f, const double *etatau, const orc_params3d *p);
/* full driver = Stokes3D.jl:25-186 */
int32_t orc_stokes3d_solve(const orc_fields3d *f, const orc_params3d *p, orc_result *res);

/* ---- 2D Stokes (visco-elastic variant, Stokes2D.jl:181-325) ---- */
typedef struct orc_fields2d {
    double *P, *P0, *divV, *Q;
    double *Vx, *Vy, *Ux, *Uy;
    double *txx, *tyy, *txy, *toxx, *toyy, *toxy;
    double *exx, *eyy, *exy;
    double *eta, *K, *G;
    double *fx, *fy;
    double *RP, *Rx, *Ry;
    double *txy_c, *toxy_c;      /* may be NULL */
} orc_fields2d;

typedef struct orc_params2d {
    int64_t nx, ny, nxg, nyg;
    double _dx, _dy;
    double dt, r, theta_dtau, eta_dtau, eps_rel, eps_abs;
    int64_t iterMax, nout;
    uint32_t free_slip, no_slip, periodic;
    int32_t displacement_bcs;    /* as in orc_params3d (Stokes2D.jl:223) */
    /* non-uniform Geometry (src/grid/Cartesian.jl:77-100): inverse spacings, all NULL on a uniform grid: [0] _di.vertex[1] (nx), [1] _di.vertex[2] (ny),
     * [2] _di.center[1] (nx-1), [3] _di.center[2] (ny-1), [4] _di.velocity[1][2] (ny+1), [5] _di.velocity[2][1] (nx+1) */
    const double *inv_spacing[6];
} orc_params2d;

void orc_compute_divV2d_sp(double *divV, const double *Vx, const double *Vy, int64_t nx, int64_t ny, double _dx, double _dy, const double *const *sp);
void orc_compute_divV2d(double *divV, const double *Vx, const double *Vy, int64_t nx, int64_t ny,
                        double _dx, double _dy);
void orc_compute_strain_rate2d(const orc_fields2d *f, const orc_params2d *p);
void orc_compute_tau2d(const orc_fields2d *f, const orc_params2d *p);
void orc_compute_V2d(const orc_fields2d *f, const double *etatau, const orc_params2d *p);
void orc_compute_Res2d(const orc_fields2d *f, const orc_params2d *p);
void orc_compute_V2d_fs(const orc_fields2d *f, const double *etatau, const orc_params2d *p, double fs_dt);
void orc_compute_Res2d_fs(const orc_fields2d *f, const orc_params2d *p, double fs_dt);
void orc_velocity2displacement2d(const orc_fields2d *f, const orc_params2d *p);
void orc_flow_bcs2d(double *Vx, double *Vy, int64_t nx, int64_t ny,
                    uint32_t free_slip, uint32_t no_slip, uint32_t periodic);
void orc_residual_sumsq2d(const orc_fields2d *f, const orc_params2d *p, double out[3]);
void orc_stokes2d_iteration(const orc_fields2d *f, const double *etatau, const orc_params2d *p);
int32_t orc_stokes2d_solve(const orc_fields2d *f, const orc_params2d *p, orc_result *res);

#define ORC_MAXPHASE8 8
/* ---- 2D PT heat diffusion (DiffusionPT_solver.jl / DiffusionPT_kernels.jl) ---- */
typedef struct orc_thermal2d {
    double *T, *Told, *dT;        /* (nx+2, ny+2) ; dT = ΔT */
    double *qTx, *qTx2;           /* (nx+1, ny)   */
    double *qTy, *qTy2;           /* (nx, ny+1)   */
    double *H, *shear_heating, *ResT;   /* (nx, ny) */
    double *K, *rhoCp;            /* (nx, ny) array-coefficient form; may be NULL in rheology form */
    double *thetar_dtau, *dtau_rho;     /* (nx, ny) PTThermalCoeffs */
    const double *adiabatic;            /* (nx, ny) thermal.adiabatic of the rheology forms (adiabatic_heating!, DiffusionPT_kernels.jl:720-746); NULL = absent */
    const double *dirichlet_mask;       /* (nx+2, ny+2) thermal_bc.dirichlet.mask (Dirichlet.jl, mask/mask.jl); NULL = none */
    const double *dirichlet_value;      /* (nx+2, ny+2) values, or NULL with dirichlet_const (ConstantDirichletBoundaryCondition) */
} orc_thermal2d;

typedef struct orc_thermal_params2d {
    int64_t nx, ny;
    double _dx, _dy;
    double dt, eps;
    int64_t iterMax, nout;
    /* thermal BCs, faces in order left,right,top,bot (2D naming of the reference) */
    int32_t no_flux[4];
    int32_t constant_value_on[4]; double constant_value[4];
    int32_t constant_flux_on[4];  double constant_flux[4];
    int32_t periodic[4];
    /* rheology form (test_diffusion2D.jl): constant conductivity, Cp, PT_Density(rho0, alpha), H */
    int32_t rheology_form;
    double k_const, Cp, rho0, alpha, T0, H_const;
    double dirichlet_const;             /* value of a ConstantDirichletBoundaryCondition (used when dirichlet_value is NULL) */
    const double *inv_spacing[4];   /* non-uniform Geometry: _di.center x (nx-1), y (ny-1), _di.vertex x (nx), y (ny); all NULL: uniform */
} orc_thermal_params2d;

/* phase-ratio form (rheology_form = 2): per-phase thermal properties + the arrays heatdiffusion_PT!(...; phase = phase_ratios) reads */
typedef struct orc_thermal_phases {
    int32_t nphase;
    double k[ORC_MAXPHASE8], Cp[ORC_MAXPHASE8], Hr[ORC_MAXPHASE8];
    int32_t rho_kind[ORC_MAXPHASE8];
    double rho0[ORC_MAXPHASE8], alpha[ORC_MAXPHASE8], beta[ORC_MAXPHASE8], T0[ORC_MAXPHASE8], P0[ORC_MAXPHASE8];
    double max_lxyz, Vpdtau;      /* pt_thermal.max_lxyz, pt_thermal.Vpdτ */
} orc_thermal_phases;
typedef struct orc_thermal_phase_fields {
    const double *P;              /* args.P, ni */
    const double *phase_c;        /* phase_ratios.center [nphase][ni] */
    const double *phase_qx, *phase_qy, *phase_qz;   /* phase_ratios.Vx (nx+1, ny[, nz]), .Vy, .Vz; phase index fastest */
} orc_thermal_phase_fields;
void orc_thermal_set_phases(const orc_thermal_phases *ph, const orc_thermal_phase_fields *pf);   /* NULL, NULL to clear */
/* adiabatic_heating!(thermal, stokes, rheology, phases, _dt) -- DiffusionPT_kernels.jl:720-746: A = (P - P0) * α * _dt with α the phase-weighted thermal
 * expansivity of the density laws (PT_Density, T_Density: α; otherwise 0 -- ASSUMED compute_α of GeoParams); phase_c NULL: phase 0 alone */
void orc_adiabatic_heating(double *A, const double *P, const double *P0, int64_t n, const orc_thermal_phases *ph, const double *phase_c, double _dt);

void orc_thermal_bcs2d(double *T, const orc_thermal_params2d *p);
void orc_thermal2d_iteration(const orc_thermal2d *t, const orc_thermal_params2d *p);
void orc_thermal2d_check_res(const orc_thermal2d *t, const orc_thermal_params2d *p);
int32_t orc_heatdiffusion_PT2d(const orc_thermal2d *t, const orc_thermal_params2d *p,
                               int64_t *iter_out, double *norm_ResT, int64_t cap, int64_t *nnorms);

/* ---- 3D PT heat diffusion (DiffusionPT_kernels.jl:6-61,160-199,250-282; test/test_diffusion3D.jl) ---- */
typedef struct orc_thermal3d {
    double *T, *Told, *dT;              /* (nx+2, ny+2, nz+2) */
    double *qTx, *qTx2;                 /* (nx+1, ny, nz) */
    double *qTy, *qTy2;                 /* (nx, ny+1, nz) */
    double *qTz, *qTz2;                 /* (nx, ny, nz+1) */
    double *H, *shear_heating, *ResT;   /* ni */
    double *K, *rhoCp;                  /* ni; may be NULL in the rheology form */
    double *thetar_dtau, *dtau_rho;     /* ni */
    const double *adiabatic, *dirichlet_mask, *dirichlet_value;      /* as in orc_thermal2d; mask / value (nx+2, ny+2, nz+2) */
} orc_thermal3d;

typedef struct orc_thermal_params3d {
    int64_t nx, ny, nz;
    double _dx, _dy, _dz;
    double dt, eps;
    int64_t iterMax, nout;
    /* faces in order left,right,front,back,top,bot ; 3D thermal naming: bot <-> k = 1, top <-> k = end */
    int32_t no_flux[6];
    int32_t constant_value_on[6]; double constant_value[6];
    int32_t constant_flux_on[6];  double constant_flux[6];
    int32_t periodic[6];
    int32_t rheology_form;
    double k_const, Cp, rho0, alpha, T0;
    double dirichlet_const;
} orc_thermal_params3d;

void orc_thermal_bcs3d(double *T, const orc_thermal_params3d *p);
void orc_thermal3d_iteration(const orc_thermal3d *t, const orc_thermal_params3d *p);
void orc_thermal3d_check_res(const orc_thermal3d *t, const orc_thermal_params3d *p);
int32_t orc_heatdiffusion_PT3d(const orc_thermal3d *t, const orc_thermal_params3d *p, int64_t *iter_out, double *norm_ResT, int64_t cap,
                               int64_t *nnorms);

/* ---- 2D multiphase visco-elasto-plastic Stokes (Stokes2D.jl:577-866; BASELINE config 5: shear band) ----
 * Restricted to what test/test_shearband2D.jl evaluates: per-phase LinearViscous eta, ConstantElasticity (G, Kb),
 * DruckerPrager_regularised (C, phi, psi, eta_vp), constant density (rho*g given as arrays).  GeoParams is not
 * vendored in the reference; the forms used are stated at each function (ASSUMED where marked). */
#define ORC_MAXPHASE 8
typedef struct orc_rheology {
    int32_t nphase;
    double eta[ORC_MAXPHASE], G[ORC_MAXPHASE], Kb[ORC_MAXPHASE];
    int32_t is_pl[ORC_MAXPHASE];
    double C[ORC_MAXPHASE], sinphi[ORC_MAXPHASE], cosphi[ORC_MAXPHASE], sinpsi[ORC_MAXPHASE], eta_vp[ORC_MAXPHASE];
    /* ---- appended in round 2; a zero-initialised tail gives the round-1 behaviour (LinearViscous, NoSoftening, ρg owned by the caller) ----
     * Density and gravity -- compute_ρg! / update_ρg! (rheology/BuoyancyForces.jl:37-60,153-167).  has_density = 0: the ρg arrays
     * are the caller's and never recomputed.  rho_kind: 0 ConstantDensity(rho0), 1 PT_Density rho0 (1 - alpha (T - T0) + beta (P - P0)),
     * 2 T_Density rho0 (1 - alpha (T - T0)), 3 Compressible_Density rho0 exp(beta (P - P0)) [GeoParams forms, assumed].
     * gravity = compute_gravity(first(rheology)): a scalar, it fills the last component of ρg (BuoyancyForces.jl:69-70). */
    int32_t has_density;
    int32_t rho_kind[ORC_MAXPHASE];
    double rho0[ORC_MAXPHASE], alpha[ORC_MAXPHASE], beta[ORC_MAXPHASE], T0[ORC_MAXPHASE], P0[ORC_MAXPHASE];
    double gravity;
    /* Strain softening of the cohesion and of the friction angle, evaluated at the accumulated plastic strain EII_pl (the EII keyword
     * of compute_yieldfunction_phase, StressKernels.jl:1053-1105; GeoParams softening_C / softening_ϕ, forms assumed):
     * kind 0 NoSoftening, 1 LinearSoftening((a = min, b = max), (c = lo, d = hi)): b for EII <= lo, a for EII >= hi, linear between;
     * 2 NonLinearSoftening(a = ξ₀, b = Δ, c = μ, d = σ) = ξ₀ - Δ/2 erfc(-(EII - μ)/σ).  phi_deg is the unsoftened friction angle. */
    int32_t softC_kind[ORC_MAXPHASE], softphi_kind[ORC_MAXPHASE];
    double softC_a[ORC_MAXPHASE], softC_b[ORC_MAXPHASE], softC_c[ORC_MAXPHASE], softC_d[ORC_MAXPHASE];
    double softphi_a[ORC_MAXPHASE], softphi_b[ORC_MAXPHASE], softphi_c[ORC_MAXPHASE], softphi_d[ORC_MAXPHASE], phi_deg[ORC_MAXPHASE];
    /* Creep law of the viscous element for compute_viscosity! / compute_viscosity_τII! with dt = Inf (rheology/Viscosity.jl:142-167):
     * visc_kind 0 LinearViscous(eta); 1 Arrhenius: eta exp((Ea + P Va)/(Rgas T) - Ea/(Rgas Tref)), clamped to [visc_lo, visc_hi]
     * (the CustomRheology of test/test_WENO5.jl:37-42 with depth = 0);
     * 2 power-law creep (GeoParams DislocationCreep, r = 0; forms ASSUMED, parity unpinned):
     *   compute_εII = creep_A (τII FT)^n exp(-(Ea + P Va)/(Rgas T)) / FE;  compute_τII = creep_A^(-1/n) (εII FE)^(1/n) exp((Ea + P Va)/(n Rgas T)) / FT;
     *   compute_viscosity_τII = τII / (2 ε(τII)), compute_viscosity_εII = τ(εII) / (2 εII) */
    int32_t visc_kind[ORC_MAXPHASE];
    double Ea[ORC_MAXPHASE], Va[ORC_MAXPHASE], Tref[ORC_MAXPHASE], Rgas[ORC_MAXPHASE], visc_lo[ORC_MAXPHASE], visc_hi[ORC_MAXPHASE];
    double creep_A[ORC_MAXPHASE], creep_n[ORC_MAXPHASE], creep_FT[ORC_MAXPHASE], creep_FE[ORC_MAXPHASE];
} orc_rheology;

typedef struct orc_vep2d {
    double *P, *P0, *divV, *Q;              /* ni */
    double *Vx, *Vy, *Ux, *Uy;
    double *exx, *eyy, *exy, *exy_c;        /* strain rate: centre, centre, vertex, centre copy */
    double *eplxx, *eplyy, *eplxy, *eplxy_c;/* plastic strain rate */
    double *dexy_c;                         /* Δε.xy_c (shear2center! of a zero tensor here) and its vertex source */
    double *dexy;
    double *txx, *tyy, *txy, *txy_c, *tII;  /* τ: centre, centre, vertex, centre shear, invariant */
    double *toxx, *toyy, *toxy, *toxy_c;
    double *eta, *eta_v, *eta_vep;          /* ni, ni.+1, ni */
    double *EII_pl, *evol_pl, *EVol_pl;     /* ni */
    double *fx, *fy;                        /* ρg */
    double *RP, *Rx, *Ry;
    double *omega_xy;                       /* ni.+1 */
    double *phase_c, *phase_v;              /* [nphase][nx][ny] and [nphase][nx+1][ny+1], phase index fastest (CellArray) */
    const double *T;                        /* args.T at the cell centres (ni) for the density laws; may be NULL (T = 0) */
    double *dexx, *deyy, *divU;             /* strain_increment variant: Δε.xx, Δε.yy (ni), stokes.∇U (ni); may be NULL otherwise */
} orc_vep2d;

typedef struct orc_vep_params2d {
    int64_t nx, ny, nxg, nyg;
    double _dx, _dy;
    double dt, r, theta_dtau, eta_dtau, eps_rel, eps_abs;
    int64_t iterMax, iterMin, nout;
    uint32_t free_slip, no_slip, periodic;
    double lambda_relaxation, viscosity_relaxation, cutoff_lo, cutoff_hi;
    int32_t staggered_invariant_mean_of_squares;  /* 0: (mean xy)^2 ; 1: mean(xy^2)  -- GeoParams second_invariant_staggered */
    int32_t free_surface;                         /* kwarg free_surface: compute_V! / compute_Res! get dt * free_surface (Stokes2D.jl:773,797) */
    int32_t displacement_bcs;                     /* flow_bcs is a DisplacementBoundaryConditions: V = U / dt first, flow_bcs! acts on U (BoundaryConditions.jl:71-78) */
    int32_t T_ghosted;                            /* single-phase driver: args.T is thermal.T (nx+2, ny+2), indexed as the reference does */
    int32_t strain_increment;                     /* kwarg strain_increment (Stokes2D.jl:588,659-734): strains from the displacement increments, Δε form of the stress update */
    const double *inv_spacing[6];                 /* as in orc_params2d */
} orc_vep_params2d;

int32_t orc_stokes2d_vep_solve(const orc_vep2d *f, const orc_rheology *rh, const orc_vep_params2d *p, orc_result *res);
int32_t orc_stokes2d_nonlinear_solve(const orc_vep2d *f, const orc_rheology *rh, const orc_vep_params2d *p, orc_result *res);
void orc_vep2d_stress(const orc_vep2d *f, const double *theta, double *lam, double *lamv, const orc_rheology *rh,
                      const orc_vep_params2d *p);
void orc_compute_tau_nonlinear2d(const orc_vep2d *f, double *theta, double *lam, const orc_rheology *rh, const orc_vep_params2d *p,
                                 int32_t multiphase);
void orc_center2vertex2d(double *v, const double *c, int64_t nx, int64_t ny);
double orc_yieldfunction_phase(const orc_rheology *rh, const double *ratio, double P, double tII);
void orc_plastic_gradients_phase2d(const orc_rheology *rh, const double *ratio, const double t[3], double dQdt[3], double *dQdP, double *dFdP);
int32_t orc_isyielding(int32_t is_pl, double tII_trial, double ty);
double orc_compute_dtau_pl(const double tij[3], const double dtij[3], double ty, double tII_trial, double eta, double lam0, double eta_reg,
                           double dtr, double volume, double dtau_pl[3], double ldq[3]);
void orc_tensor_invariant2d(double *II, const double *xx, const double *yy, const double *xy, int64_t nx, int64_t ny, int32_t mode);
void orc_compute_viscosity2d(const orc_vep2d *f, const orc_rheology *rh, const orc_vep_params2d *p, double nu);      /* compute_viscosity!: εII form */
void orc_compute_viscosity2d_form(const orc_vep2d *f, const orc_rheology *rh, const orc_vep_params2d *p, double nu, int32_t tau);

/* ---- 3D multiphase visco-elasto-plastic Stokes (Stokes3D.jl:447-668; test/test_shearband3D_MPI.jl) ---- */
typedef struct orc_vep3d {
    double *P, *P0, *divV, *Q;                          /* ni */
    double *Vx, *Vy, *Vz, *Ux, *Uy, *Uz;
    double *exx, *eyy, *ezz, *eyz, *exz, *exy;          /* strain rate: centres, then edges yz (nx,ny+1,nz+1), xz (nx+1,ny,nz+1), xy (nx+1,ny+1,nz) */
    double *eyz_c, *exz_c, *exy_c;                      /* shear2center! targets (ni) */
    double *eplxx, *eplyy, *eplzz, *eplyz, *eplxz, *eplxy, *eplyz_c, *eplxz_c, *eplxy_c;
    double *deyz, *dexz, *dexy, *deyz_c, *dexz_c, *dexy_c;    /* Δε shear (may be NULL) */
    double *txx, *tyy, *tzz, *tyz, *txz, *txy, *tyz_c, *txz_c, *txy_c, *tII;
    double *toxx, *toyy, *tozz, *toyz, *toxz, *toxy, *toyz_c, *toxz_c, *toxy_c;
    double *eta, *eta_vep;                              /* ni */
    double *EII_pl, *evol_pl, *EVol_pl;                 /* ni */
    double *fx, *fy, *fz;                               /* ρg */
    double *RP, *Rx, *Ry, *Rz;
    double *omega_yz, *omega_xz, *omega_xy;             /* edge extents */
    double *phase_c, *phase_yz, *phase_xz, *phase_xy;   /* [nphase][extent], phase index fastest (CellArray) */
    const double *T;                                    /* args.T at the cell centres (ni); may be NULL (T = 0) */
} orc_vep3d;

typedef struct orc_vep_params3d {
    int64_t nx, ny, nz, nxg, nyg, nzg;
    double _dx, _dy, _dz;
    double dt, r, theta_dtau, eta_dtau, eps_rel, eps_abs;
    int64_t iterMax, nout;
    uint32_t free_slip, no_slip, periodic;
    double lambda_relaxation, viscosity_relaxation, cutoff_lo, cutoff_hi;
    int32_t displacement_bcs;
    int32_t T_ghosted;           /* args.T is thermal.T (ni .+ 2), read by the density at the cell's own [i, j, k] (BuoyancyForces.jl:52) */
} orc_vep_params3d;

int32_t orc_stokes3d_vep_solve(const orc_vep3d *f, const orc_rheology *rh, const orc_vep_params3d *p, orc_result *res);
void orc_vep3d_stress(const orc_vep3d *f, const double *theta, double *lam, double *const lamv[3], const orc_rheology *rh,
                      const orc_vep_params3d *p);
void orc_compute_viscosity3d(const orc_vep3d *f, const orc_rheology *rh, const orc_vep_params3d *p, double nu);
void orc_compute_viscosity3d_form(const orc_vep3d *f, const orc_rheology *rh, const orc_vep_params3d *p, double nu, int32_t tau);
void orc_tensor_invariant3d(double *II, const double *xx, const double *yy, const double *zz, const double *yz, const double *xz, const double *xy,
                            int64_t nx, int64_t ny, int64_t nz);
void orc_shear2center3d(double *yz_c, double *xz_c, double *xy_c, const double *yz, const double *xz, const double *xy, int64_t nx, int64_t ny, int64_t nz);
void orc_compute_vorticity3d(double *wyz, double *wxz, double *wxy, const double *Vx, const double *Vy, const double *Vz,
                             int64_t nx, int64_t ny, int64_t nz, double _dx, double _dy, double _dz);

/* test hook: periodic self-neighbour update_halo! inside the VEP drivers (see stokes3d_vep.c) */
void orc_set_self_halo(int px, int py, int pz);
void orc_self_halo(double *A, const int64_t ext[3], const int64_t n[3]);

int orc_num_threads(void);
void orc_set_num_threads(int n);
void orc_first_touch_copy(double *dst, const double *src, int64_t slab, int64_t nslab);

#ifdef __cplusplus
}
#endif
/* gridops.c -- grid operators either side of solve! / heatdiffusion_PT! (Interpolations.jl, BuoyancyForces.jl:6-60, ShearHeating.jl) */
void orc_velocity2vertex2d(double *Vxv, double *Vyv, const double *Vx, const double *Vy, int64_t nx, int64_t ny, int64_t mx, int64_t my);
void orc_velocity2vertex3d(double *Vxv, double *Vyv, double *Vzv, const double *Vx, const double *Vy, const double *Vz, int64_t nx, int64_t ny, int64_t nz,
                           int64_t mx, int64_t my, int64_t mz);
void orc_velocity2center2d(double *Vxc, double *Vyc, const double *Vx, const double *Vy, int64_t nx, int64_t ny);
void orc_velocity2center3d(double *Vxc, double *Vyc, double *Vzc, const double *Vx, const double *Vy, const double *Vz, int64_t nx, int64_t ny, int64_t nz);
void orc_vertex2center(double *cen, const double *ver, const int64_t vdim[3], const int64_t cdim[3], int32_t ndim, int32_t gx, int32_t gy, int32_t gz);
void orc_center2vertex_harm2d(double *ver, const double *cen, int64_t nx, int64_t ny);
void orc_center2vertex3d(double *vyz, double *vxz, double *vxy, const double *cyz, const double *cxz, const double *cxy, int64_t nx, int64_t ny, int64_t nz);
void orc_compute_rhog(double *rhog, const orc_rheology *rh, const double *phase_c, const double *T, const double *P, const int64_t n[3], const int64_t tdim[3],
                      int32_t ndim);
void orc_compute_lithostatic_pressure(double *P, const double *rhog, double dz, const double *dz_cells, const int64_t n[3], int32_t ndim);
void orc_compute_viscosity_single(double *eta, const orc_rheology *rh, const double *T, const double *P, const int64_t n[3], const int64_t tdim[3], int32_t ndim,
                                  double nu, double lo, double hi, const double *AII, int32_t tau);
void orc_compute_shear_heating(double *sh, const double *const *tau, const double *const *tau_o, const double *const *eps, const double *phase_c,
                               const orc_rheology *rh, const double *chi, double dt, const int64_t n[3], int32_t ndim);

#endif
