"""ctypes binding of the CPU oracle (oracle/libjrx_oracle.so).

TEST INFRASTRUCTURE ONLY -- see oracle/jrx_oracle.h.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product path never does.

All arrays are numpy float64, Fortran order (Julia's dense column-major layout).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
# JRX_ORACLE_LIB: another build of the same sources (the sanitizer build of oracle/Makefile, tests/test_host_sanitizers.py); it is loaded as it is, never rebuilt here
_LIB_PATH = Path(os.environ["JRX_ORACLE_LIB"]) if os.environ.get("JRX_ORACLE_LIB") else _HERE / "libjrx_oracle.so"

FACE = dict(left=1, right=2, front=4, back=8, top=16, bot=32)


def build(force: bool = False) -> Path:
    """Compile the C restatement with gcc (see oracle/Makefile)."""
    if os.environ.get("JRX_ORACLE_LIB"):
        return _LIB_PATH
    if force or not _LIB_PATH.exists():
        subprocess.check_call(["make", "-C", str(_HERE)] + (["-B"] if force else []))
    return _LIB_PATH


_dp = C.POINTER(C.c_double)

F3_NAMES = ["P", "P0", "divV", "Q", "Vx", "Vy", "Vz", "Ux", "Uy", "Uz",
            "txx", "tyy", "tzz", "tyz", "txz", "txy",
            "toxx", "toyy", "tozz", "toyz", "toxz", "toxy",
            "exx", "eyy", "ezz", "eyz", "exz", "exy",
            "eta", "K", "G", "fx", "fy", "fz", "RP", "Rx", "Ry", "Rz",
            "tyz_c", "txz_c", "txy_c", "toyz_c", "toxz_c", "toxy_c"]

F2_NAMES = ["P", "P0", "divV", "Q", "Vx", "Vy", "Ux", "Uy",
            "txx", "tyy", "txy", "toxx", "toyy", "toxy", "exx", "eyy", "exy",
            "eta", "K", "G", "fx", "fy", "RP", "Rx", "Ry", "txy_c", "toxy_c"]

T2_NAMES = ["T", "Told", "dT", "qTx", "qTx2", "qTy", "qTy2", "H", "shear_heating", "ResT",
            "K", "rhoCp", "thetar_dtau", "dtau_rho"]


def shapes3d(nx, ny, nz):
    """Array extents of StokesArrays in 3D (src/types/constructors/stokes.jl:27-34,196-212,241-247)."""
    c = (nx, ny, nz)
    s = {n: c for n in F3_NAMES}
    s.update(Vx=(nx + 1, ny + 2, nz + 2), Vy=(nx + 2, ny + 1, nz + 2), Vz=(nx + 2, ny + 2, nz + 1))
    s.update(Ux=s["Vx"], Uy=s["Vy"], Uz=s["Vz"])
    for p in ("t", "to", "e"):
        s[p + "xy"] = (nx + 1, ny + 1, nz)
        s[p + "yz"] = (nx, ny + 1, nz + 1)
        s[p + "xz"] = (nx + 1, ny, nz + 1)
    s.update(Rx=(nx - 1, ny, nz), Ry=(nx, ny - 1, nz), Rz=(nx, ny, nz - 1))
    return s


def shapes2d(nx, ny):
    c = (nx, ny)
    s = {n: c for n in F2_NAMES}
    s.update(Vx=(nx + 1, ny + 2), Vy=(nx + 2, ny + 1))
    s.update(Ux=s["Vx"], Uy=s["Vy"])
    for p in ("t", "to", "e"):
        s[p + "xy"] = (nx + 1, ny + 1)
    s.update(Rx=(nx - 1, ny), Ry=(nx, ny - 1))
    return s


def shapes_thermal2d(nx, ny):
    c = (nx, ny)
    s = {n: c for n in T2_NAMES}
    s.update(T=(nx + 2, ny + 2), Told=(nx + 2, ny + 2), dT=(nx + 2, ny + 2),
             qTx=(nx + 1, ny), qTx2=(nx + 1, ny), qTy=(nx, ny + 1), qTy2=(nx, ny + 1))
    return s


def alloc(shapes: dict) -> dict:
    return {k: np.zeros(v, dtype=np.float64, order="F") for k, v in shapes.items()}


def _mkstruct(name, names):
    return type(name, (C.Structure,), {"_fields_": [(n, _dp) for n in names]})


Fields3D = _mkstruct("Fields3D", F3_NAMES)
Fields2D = _mkstruct("Fields2D", F2_NAMES)
T_OPT = ["adiabatic", "dirichlet_mask", "dirichlet_value"]     # optional: NULL when the dict has no such entry
Thermal2D = _mkstruct("Thermal2D", T2_NAMES + T_OPT)


class Params3D(C.Structure):
    _fields_ = [("nx", C.c_int64), ("ny", C.c_int64), ("nz", C.c_int64),
                ("nxg", C.c_int64), ("nyg", C.c_int64), ("nzg", C.c_int64),
                ("_dx", C.c_double), ("_dy", C.c_double), ("_dz", C.c_double),
                ("dt", C.c_double), ("r", C.c_double), ("theta_dtau", C.c_double), ("eta_dtau", C.c_double),
                ("eps_rel", C.c_double), ("eps_abs", C.c_double),
                ("iterMax", C.c_int64), ("nout", C.c_int64),
                ("free_slip", C.c_uint32), ("no_slip", C.c_uint32), ("periodic", C.c_uint32), ("displacement_bcs", C.c_int32)]


class Params2D(C.Structure):
    _fields_ = [("nx", C.c_int64), ("ny", C.c_int64), ("nxg", C.c_int64), ("nyg", C.c_int64),
                ("_dx", C.c_double), ("_dy", C.c_double),
                ("dt", C.c_double), ("r", C.c_double), ("theta_dtau", C.c_double), ("eta_dtau", C.c_double),
                ("eps_rel", C.c_double), ("eps_abs", C.c_double),
                ("iterMax", C.c_int64), ("nout", C.c_int64),
                ("free_slip", C.c_uint32), ("no_slip", C.c_uint32), ("periodic", C.c_uint32), ("displacement_bcs", C.c_int32),
                ("inv_spacing", C.POINTER(C.c_double) * 6)]


def set_spacing2d(p, inv_spacing):
    """non-uniform Geometry: the six inverse-spacing arrays (_di.vertex x / y, _di.center x / y, _di.velocity[1][2], _di.velocity[2][1]) of Params2D / VEPParams2D;
    the arrays are kept alive on the params object"""
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in inv_spacing]
    assert len(arrs) == 6
    p._spacing_keepalive = arrs
    for q, a in enumerate(arrs):
        p.inv_spacing[q] = a.ctypes.data_as(C.POINTER(C.c_double))
    return p


class ThermalParams2D(C.Structure):
    _fields_ = [("nx", C.c_int64), ("ny", C.c_int64), ("_dx", C.c_double), ("_dy", C.c_double),
                ("dt", C.c_double), ("eps", C.c_double), ("iterMax", C.c_int64), ("nout", C.c_int64),
                ("no_flux", C.c_int32 * 4),
                ("constant_value_on", C.c_int32 * 4), ("constant_value", C.c_double * 4),
                ("constant_flux_on", C.c_int32 * 4), ("constant_flux", C.c_double * 4),
                ("periodic", C.c_int32 * 4),
                ("rheology_form", C.c_int32),
                ("k_const", C.c_double), ("Cp", C.c_double), ("rho0", C.c_double), ("alpha", C.c_double),
                ("T0", C.c_double), ("H_const", C.c_double), ("dirichlet_const", C.c_double), ("inv_spacing", C.POINTER(C.c_double) * 4)]


def set_spacing_thermal2d(p, grid_inv):
    """non-uniform Geometry for the 2D heat solver: grid_inv = grid._di (dict with 'center', 'vertex'); arrays kept alive on the params object"""
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (grid_inv["center"][0], grid_inv["center"][1], grid_inv["vertex"][0], grid_inv["vertex"][1])]
    p._spacing_keepalive = arrs
    for q, a in enumerate(arrs):
        p.inv_spacing[q] = a.ctypes.data_as(C.POINTER(C.c_double))
    return p


class Result(C.Structure):
    _fields_ = [("iter", C.c_int64), ("nchecks", C.c_int64), ("status", C.c_int32),
                ("err_evo1", _dp), ("err_evo2", C.POINTER(C.c_int64)),
                ("norm_Rx", _dp), ("norm_Ry", _dp), ("norm_Rz", _dp), ("norm_divV", _dp),
                ("cap", C.c_int64), ("time_s", C.c_double)]


_lib = None


def usable_cpus() -> int:
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (cpu.max = "quota period").  The GPU boxes give a 16-CPU quota
    on a 256-thread host, where OpenMP's default of one thread per logical CPU time-slices and runs ~8x slower than 16 threads."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, n)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(_LIB_PATH))
        L.orc_mini2.restype = C.c_double
        L.orc_mini2.argtypes = [C.c_char_p, _dp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int]
        L.orc_mini3.restype = C.c_double
        L.orc_mini3.argtypes = [C.c_char_p, _dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int]
        L.orc_div2.restype = C.c_double
        L.orc_div2.argtypes = [_dp, _dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]
        L.orc_div3.restype = C.c_double
        L.orc_div3.argtypes = [_dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                               C.c_int, C.c_int, C.c_int]
        L.orc_mysum.restype = C.c_double
        L.orc_mysum.argtypes = [C.c_int, _dp] + [C.c_int] * 9
        L.orc_compute_dtau_r.restype = C.c_double
        L.orc_compute_dtau_r.argtypes = [C.c_double] * 3
        L.orc_num_threads.restype = C.c_int
        L.orc_stokes3d_solve.restype = C.c_int32
        L.orc_stokes2d_solve.restype = C.c_int32
        L.orc_heatdiffusion_PT2d.restype = C.c_int32
        import os
        if "OMP_NUM_THREADS" not in os.environ:          # default: as many threads as the job has CPUs
            L.orc_set_num_threads(C.c_int(usable_cpus()))
        _lib = L
    return _lib


def _p(a):
    if a is None:
        return _dp()
    assert a.dtype == np.float64 and a.flags.f_contiguous, "oracle arrays must be float64, Fortran order"
    return a.ctypes.data_as(_dp)


def fields3d(arr: dict) -> Fields3D:
    f = Fields3D()
    for n in F3_NAMES:
        setattr(f, n, _p(arr.get(n)))
    return f


def fields2d(arr: dict) -> Fields2D:
    f = Fields2D()
    for n in F2_NAMES:
        setattr(f, n, _p(arr.get(n)))
    return f


def thermal2d(arr: dict) -> Thermal2D:
    f = Thermal2D()
    for n in T2_NAMES + T_OPT:
        setattr(f, n, _p(arr.get(n)))
    return f


def bcmask(d) -> int:
    m = 0
    for k, v in (d or {}).items():
        if v:
            m |= FACE[k]
    return m


def params3d(ni, _di, dt, pt, *, iterMax=10_000, nout=500, free_slip=None, no_slip=None, periodic=None,
             ni_g=None, displacement_bcs=False) -> Params3D:
    """pt = dict(r, theta_dtau, eta_dtau, eps_rel, eps_abs)"""
    ni_g = ni_g or ni
    return Params3D(ni[0], ni[1], ni[2], ni_g[0], ni_g[1], ni_g[2], _di[0], _di[1], _di[2],
                    dt, pt["r"], pt["theta_dtau"], pt["eta_dtau"], pt["eps_rel"], pt["eps_abs"],
                    int(iterMax), int(nout), bcmask(free_slip), bcmask(no_slip), bcmask(periodic), int(bool(displacement_bcs)))


def params2d(ni, _di, dt, pt, *, iterMax=10_000, nout=500, free_slip=None, no_slip=None, periodic=None,
             ni_g=None, displacement_bcs=False) -> Params2D:
    ni_g = ni_g or ni
    return Params2D(ni[0], ni[1], ni_g[0], ni_g[1], _di[0], _di[1],
                    dt, pt["r"], pt["theta_dtau"], pt["eta_dtau"], pt["eps_rel"], pt["eps_abs"],
                    int(iterMax), int(nout), bcmask(free_slip), bcmask(no_slip), bcmask(periodic), int(bool(displacement_bcs)))


class _Res:
    def __init__(self, cap):
        self.cap = cap
        self.e1 = np.zeros(cap)
        self.e2 = np.zeros(cap, dtype=np.int64)
        self.n = [np.zeros(cap) for _ in range(4)]
        self.c = Result(0, 0, 0, _p(self.e1), self.e2.ctypes.data_as(C.POINTER(C.c_int64)),
                        _p(self.n[0]), _p(self.n[1]), _p(self.n[2]), _p(self.n[3]), cap, 0.0)

    def asdict(self, dim):
        k = self.c.nchecks
        d = dict(iter=self.c.iter, status=self.c.status, err_evo1=self.e1[:k].copy(), err_evo2=self.e2[:k].copy(),
                 norm_Rx=self.n[0][:k].copy(), norm_Ry=self.n[1][:k].copy(), norm_divV=self.n[3][:k].copy(),
                 time=self.c.time_s)
        if dim == 3:
            d["norm_Rz"] = self.n[2][:k].copy()
        return d


def stokes3d_solve(arr: dict, p: Params3D) -> dict:
    res = _Res(int(p.iterMax // p.nout + 2))
    f = fields3d(arr)
    lib().orc_stokes3d_solve(C.byref(f), C.byref(p), C.byref(res.c))
    return res.asdict(3)


def stokes2d_solve(arr: dict, p: Params2D) -> dict:
    res = _Res(int(p.iterMax // p.nout + 2))
    f = fields2d(arr)
    lib().orc_stokes2d_solve(C.byref(f), C.byref(p), C.byref(res.c))
    return res.asdict(2)


def stokes3d_iteration(arr: dict, etatau, p: Params3D):
    f = fields3d(arr)
    lib().orc_stokes3d_iteration(C.byref(f), _p(etatau), C.byref(p))


def stokes2d_iteration(arr: dict, etatau, p: Params2D):
    f = fields2d(arr)
    lib().orc_stokes2d_iteration(C.byref(f), _p(etatau), C.byref(p))


def call3d(fn: str, arr: dict, p: Params3D, *extra):
    """Call one of the single-kernel entry points taking (fields*, [extra...], params*)."""
    f = fields3d(arr)
    getattr(lib(), fn)(C.byref(f), *extra, C.byref(p))


def call2d(fn: str, arr: dict, p: Params2D, *extra):
    f = fields2d(arr)
    getattr(lib(), fn)(C.byref(f), *extra, C.byref(p))


def compute_maxloc(A):
    B = np.zeros_like(A, order="F")
    if A.ndim == 3:
        lib().orc_compute_maxloc3d(_p(B), _p(A), *[C.c_int64(n) for n in A.shape])
    else:
        lib().orc_compute_maxloc2d(_p(B), _p(A), *[C.c_int64(n) for n in A.shape])
    return B


def flow_bcs3d(Vx, Vy, Vz, ni, free_slip=None, no_slip=None, periodic=None):
    lib().orc_flow_bcs3d(_p(Vx), _p(Vy), _p(Vz), *[C.c_int64(n) for n in ni],
                         C.c_uint32(bcmask(free_slip)), C.c_uint32(bcmask(no_slip)), C.c_uint32(bcmask(periodic)))


def flow_bcs2d(Vx, Vy, ni, free_slip=None, no_slip=None, periodic=None):
    lib().orc_flow_bcs2d(_p(Vx), _p(Vy), *[C.c_int64(n) for n in ni],
                         C.c_uint32(bcmask(free_slip)), C.c_uint32(bcmask(no_slip)), C.c_uint32(bcmask(periodic)))


def residual_sumsq3d(arr, p):
    out = (C.c_double * 4)()
    f = fields3d(arr)
    lib().orc_residual_sumsq3d(C.byref(f), C.byref(p), out)
    return np.array(out[:])


def residual_sumsq2d(arr, p):
    out = (C.c_double * 3)()
    f = fields2d(arr)
    lib().orc_residual_sumsq2d(C.byref(f), C.byref(p), out)
    return np.array(out[:])


def thermal_params2d(ni, _di, dt, eps, *, iterMax=50_000, nout=1000, no_flux=None, constant_value=None,
                     constant_flux=None, periodic=None, rheology=None) -> ThermalParams2D:
    """BC dicts use the reference's 2D face names left/right/top/bot; a value of False/None disables."""
    order = ("left", "right", "top", "bot")
    p = ThermalParams2D()
    p.nx, p.ny, p._dx, p._dy, p.dt, p.eps, p.iterMax, p.nout = ni[0], ni[1], _di[0], _di[1], dt, eps, int(iterMax), int(nout)
    for i, k in enumerate(order):
        p.no_flux[i] = int(bool((no_flux or {}).get(k, False)))
        p.periodic[i] = int(bool((periodic or {}).get(k, False)))
        v = (constant_value or {}).get(k, False)
        p.constant_value_on[i] = int(v is not False and v is not None)
        p.constant_value[i] = float(v) if p.constant_value_on[i] else 0.0
        v = (constant_flux or {}).get(k, False)
        on = not isinstance(v, bool) and v is not None      # `!isa(bc_flux.left, Bool)` in the reference
        p.constant_flux_on[i] = int(on)
        p.constant_flux[i] = float(v) if on else 0.0
    if rheology:
        p.rheology_form = 1
        p.k_const, p.Cp, p.rho0, p.alpha, p.T0 = (rheology["k"], rheology["Cp"], rheology["rho0"],
                                                   rheology["alpha"], rheology.get("T0", 0.0))
    return p


def thermal_bcs2d(T, p: ThermalParams2D):
    lib().orc_thermal_bcs2d(_p(T), C.byref(p))


def heatdiffusion_PT2d(arr: dict, p: ThermalParams2D) -> dict:
    cap = int(p.iterMax // p.nout + 2)
    it = np.zeros(cap, dtype=np.int64)
    nr = np.zeros(cap)
    nn = C.c_int64(0)
    t = thermal2d(arr)
    lib().orc_heatdiffusion_PT2d(C.byref(t), C.byref(p), it.ctypes.data_as(C.POINTER(C.c_int64)), _p(nr),
                                 C.c_int64(cap), C.byref(nn))
    return dict(iter_count=it[:nn.value].copy(), norm_ResT=nr[:nn.value].copy())


def thermal2d_iteration(arr: dict, p: ThermalParams2D):
    t = thermal2d(arr)
    lib().orc_thermal2d_iteration(C.byref(t), C.byref(p))


def thermal2d_check_res(arr: dict, p: ThermalParams2D):
    t = thermal2d(arr)
    lib().orc_thermal2d_check_res(C.byref(t), C.byref(p))


def num_threads() -> int:
    return lib().orc_num_threads()


def set_num_threads(n: int):
    lib().orc_set_num_threads(C.c_int(int(n)))


# ---------------------------------------------------------------- 2D multiphase VEP (shear band)
VEP_NAMES = ["P", "P0", "divV", "Q", "Vx", "Vy", "Ux", "Uy", "exx", "eyy", "exy", "exy_c", "eplxx", "eplyy", "eplxy", "eplxy_c",
             "dexy_c", "dexy", "txx", "tyy", "txy", "txy_c", "tII", "toxx", "toyy", "toxy", "toxy_c", "eta", "eta_v", "eta_vep",
             "EII_pl", "evol_pl", "EVol_pl", "fx", "fy", "RP", "Rx", "Ry", "omega_xy", "phase_c", "phase_v", "T", "dexx", "deyy", "divU"]
VEP2D = _mkstruct("VEP2D", VEP_NAMES)
MAXPHASE = 8


class Rheology(C.Structure):
    _fields_ = [("nphase", C.c_int32)] + [(k, C.c_double * MAXPHASE) for k in ("eta", "G", "Kb")] + [("is_pl", C.c_int32 * MAXPHASE)] + \
               [(k, C.c_double * MAXPHASE) for k in ("C", "sinphi", "cosphi", "sinpsi", "eta_vp")] + \
               [("has_density", C.c_int32), ("rho_kind", C.c_int32 * MAXPHASE)] + \
               [(k, C.c_double * MAXPHASE) for k in ("rho0", "alpha", "beta", "T0", "P0")] + [("gravity", C.c_double)] + \
               [("softC_kind", C.c_int32 * MAXPHASE), ("softphi_kind", C.c_int32 * MAXPHASE)] + \
               [(k, C.c_double * MAXPHASE) for k in ("softC_a", "softC_b", "softC_c", "softC_d", "softphi_a", "softphi_b", "softphi_c",
                                                     "softphi_d", "phi_deg")] + \
               [("visc_kind", C.c_int32 * MAXPHASE)] + [(k, C.c_double * MAXPHASE) for k in ("Ea", "Va", "Tref", "Rgas", "visc_lo", "visc_hi", "creep_A", "creep_n",
                                                                                        "creep_FT", "creep_FE")]


class VEPParams2D(C.Structure):
    _fields_ = [("nx", C.c_int64), ("ny", C.c_int64), ("nxg", C.c_int64), ("nyg", C.c_int64), ("_dx", C.c_double), ("_dy", C.c_double),
                ("dt", C.c_double), ("r", C.c_double), ("theta_dtau", C.c_double), ("eta_dtau", C.c_double),
                ("eps_rel", C.c_double), ("eps_abs", C.c_double), ("iterMax", C.c_int64), ("iterMin", C.c_int64), ("nout", C.c_int64),
                ("free_slip", C.c_uint32), ("no_slip", C.c_uint32), ("periodic", C.c_uint32),
                ("lambda_relaxation", C.c_double), ("viscosity_relaxation", C.c_double), ("cutoff_lo", C.c_double), ("cutoff_hi", C.c_double),
                ("staggered_invariant_mean_of_squares", C.c_int32), ("free_surface", C.c_int32), ("displacement_bcs", C.c_int32), ("T_ghosted", C.c_int32),
                ("strain_increment", C.c_int32), ("inv_spacing", C.POINTER(C.c_double) * 6)]


def vep_shapes2d(nx, ny, nphase):
    c, v = (nx, ny), (nx + 1, ny + 1)
    s = {n: c for n in VEP_NAMES}
    s.update(Vx=(nx + 1, ny + 2), Vy=(nx + 2, ny + 1), Ux=(nx + 1, ny + 2), Uy=(nx + 2, ny + 1), exy=v, eplxy=v, dexy=v, txy=v, toxy=v,
             eta_v=v, omega_xy=v, Rx=(nx - 1, ny), Ry=(nx, ny - 1), phase_c=(nphase, nx, ny), phase_v=(nphase, nx + 1, ny + 1))
    return s


def rheology_struct(phases: list) -> Rheology:
    """phases: list of dict(eta, G, Kb, C, phi_deg, psi_deg, eta_vp) -- C is the GeoParams cohesion parameter"""
    import math
    r = Rheology()
    r.nphase = len(phases)
    for q, ph in enumerate(phases):
        r.eta[q], r.G[q], r.Kb[q] = ph.get("eta", 0.0), ph["G"], ph["Kb"]
        _cr = ph.get("creep")
        if _cr is not None and _cr.get("kind") in ("dislocation", "powerlaw"):
            # GeoParams sums the strain rates of the elements of a CompositeRheology; the table holds ONE viscous element per phase, so a LinearViscous
            # element (eta) beside a DislocationCreep would be dropped silently: refuse it
            if r.eta[q] != 0.0:
                raise ValueError(f"phase {q}: `eta` (a LinearViscous element) in series with a dislocation creep is not built: one viscous element per phase")
        elif "eta" not in ph:
            raise KeyError("eta")
        pl = "C" in ph and ph["C"] is not None
        r.is_pl[q] = int(pl)
        if pl:
            r.C[q], r.eta_vp[q] = ph["C"], ph.get("eta_vp", 0.0)
            r.sinphi[q], r.cosphi[q] = math.sin(math.radians(ph["phi_deg"])), math.cos(math.radians(ph["phi_deg"]))
            r.sinpsi[q] = math.sin(math.radians(ph.get("psi_deg", 0.0)))
        # density / gravity (compute_ρg!): ph["density"] = dict(kind="constant"|"PT"|"T"|"compressible", rho0[, alpha, beta, T0, P0]); ph["g"]
        d = ph.get("density")
        if d is not None:
            r.has_density = 1
            r.rho_kind[q] = {"constant": 0, "PT": 1, "T": 2, "compressible": 3}[d.get("kind", "constant")]
            r.rho0[q], r.alpha[q], r.beta[q] = d["rho0"], d.get("alpha", 0.0), d.get("beta", 0.0)
            r.T0[q], r.P0[q] = d.get("T0", 0.0), d.get("P0", 0.0)
        if q == 0:
            r.gravity = float(ph.get("g", 0.0))
        # strain softening of C and ϕ: dict(kind="linear", min, max, lo, hi) | dict(kind="nonlinear", xi0, Delta[, mu=1, sigma=0.5])
        r.phi_deg[q] = float(ph.get("phi_deg", 0.0))
        for key, pre in (("softening_C", "softC_"), ("softening_phi", "softphi_")):
            sft = ph.get(key)
            if sft is None:
                continue
            if sft["kind"] == "linear":
                vals = (1, sft["min"], sft["max"], sft["lo"], sft["hi"])
            elif sft["kind"] == "nonlinear":
                vals = (2, sft["xi0"], sft["Delta"], sft.get("mu", 1.0), sft.get("sigma", 0.5))
            else:
                raise ValueError(f"unknown softening law {sft['kind']!r}")
            getattr(r, pre + "kind")[q] = vals[0]
            for name, v in zip("abcd", vals[1:]):
                getattr(r, pre + name)[q] = float(v)
        # creep law: dict(kind="arrhenius", Ea, Va, T0, R, cutoff=(lo, hi)) on top of eta (= η0)
        cr = ph.get("creep")
        if cr is not None and cr.get("kind") in ("dislocation", "powerlaw"):
            # DislocationCreep(A, n, E, V, R) with r = 0; apparatus = "AxialCompression" (FT = √3, FE = 2/√3) | "SimpleShear" (2, 2) | "Invariant" (1, 1)
            FT, FE = {"AxialCompression": (3.0 ** 0.5, 2.0 / 3.0 ** 0.5), "SimpleShear": (2.0, 2.0), "Invariant": (1.0, 1.0)}[cr.get("apparatus", "AxialCompression")]
            r.visc_kind[q] = 2
            r.creep_A[q], r.creep_n[q], r.creep_FT[q], r.creep_FE[q] = cr["A"], cr["n"], cr.get("FT", FT), cr.get("FE", FE)
            r.Ea[q], r.Va[q], r.Rgas[q] = cr.get("E", 0.0), cr.get("V", 0.0), cr.get("R", 8.3145)
        elif cr is not None:
            if cr.get("kind") != "arrhenius":
                raise ValueError(f"unknown creep law {cr.get('kind')!r}")
            r.visc_kind[q] = 1
            r.Ea[q], r.Va[q], r.Tref[q], r.Rgas[q] = cr["Ea"], cr["Va"], cr["T0"], cr.get("R", 8.3145)
            lo, hi = cr.get("cutoff", (0.0, float("inf")))
            r.visc_lo[q], r.visc_hi[q] = lo, hi
    return r


def vep_params2d(ni, _di, dt, pt, *, iterMax=50_000, iterMin=100, nout=500, free_slip=None, no_slip=None, periodic=None,
                 lambda_relaxation=0.2, viscosity_relaxation=1e-2, cutoff=(-np.inf, np.inf), stag_mode=0, ni_g=None, free_surface=False,
                 displacement_bcs=False, T_ghosted=False, strain_increment=False) -> VEPParams2D:
    ni_g = ni_g or ni
    return VEPParams2D(ni[0], ni[1], ni_g[0], ni_g[1], _di[0], _di[1], dt, pt["r"], pt["theta_dtau"], pt["eta_dtau"], pt["eps_rel"],
                       pt["eps_abs"], int(iterMax), int(iterMin), int(nout), bcmask(free_slip), bcmask(no_slip), bcmask(periodic),
                       lambda_relaxation, viscosity_relaxation, cutoff[0], cutoff[1], stag_mode, int(bool(free_surface)), int(bool(displacement_bcs)), int(bool(T_ghosted)),
                       int(bool(strain_increment)))


def vep2d(arr: dict) -> VEP2D:
    f = VEP2D()
    for n in VEP_NAMES:
        setattr(f, n, _p(arr.get(n)))
    return f


def stokes2d_vep_solve(arr: dict, rh: Rheology, p: VEPParams2D) -> dict:
    res = _Res(int(p.iterMax // p.nout + 2))
    f = vep2d(arr)
    L = lib()
    L.orc_stokes2d_vep_solve.restype = C.c_int32
    L.orc_stokes2d_vep_solve(C.byref(f), C.byref(rh), C.byref(p), C.byref(res.c))
    return res.asdict(2)


def stokes2d_nonlinear_solve(arr: dict, rh: Rheology, p: VEPParams2D) -> dict:
    """solve!(stokes, pt_stokes, grid, flow_bcs, ρg, rheology::MaterialParams, args, dt, igg) -- Stokes2D.jl:345-557"""
    res = _Res(int(p.iterMax // p.nout + 2))
    f = vep2d(arr)
    L = lib()
    L.orc_stokes2d_nonlinear_solve.restype = C.c_int32
    L.orc_stokes2d_nonlinear_solve(C.byref(f), C.byref(rh), C.byref(p), C.byref(res.c))
    return res.asdict(2)


def tensor_invariant2d(xx, yy, xy, mode=0):
    II = np.zeros_like(xx, order="F")
    lib().orc_tensor_invariant2d(_p(II), _p(xx), _p(yy), _p(xy), C.c_int64(xx.shape[0]), C.c_int64(xx.shape[1]), C.c_int32(mode))
    return II


def compute_viscosity2d(arr, rh, p, nu=1.0, tau=False):
    f = vep2d(arr)
    lib().orc_compute_viscosity2d_form(C.byref(f), C.byref(rh), C.byref(p), C.c_double(nu), C.c_int32(int(tau)))


def compute_viscosity3d(arr, rh, p, nu=1.0, tau=False):
    f = vep3d(arr)
    lib().orc_compute_viscosity3d_form(C.byref(f), C.byref(rh), C.byref(p), C.c_double(nu), C.c_int32(int(tau)))


# ---- 3D multiphase VEP (oracle/stokes3d_vep.c) ----
VEP3_NAMES = ["P", "P0", "divV", "Q", "Vx", "Vy", "Vz", "Ux", "Uy", "Uz",
              "exx", "eyy", "ezz", "eyz", "exz", "exy", "eyz_c", "exz_c", "exy_c",
              "eplxx", "eplyy", "eplzz", "eplyz", "eplxz", "eplxy", "eplyz_c", "eplxz_c", "eplxy_c",
              "deyz", "dexz", "dexy", "deyz_c", "dexz_c", "dexy_c",
              "txx", "tyy", "tzz", "tyz", "txz", "txy", "tyz_c", "txz_c", "txy_c", "tII",
              "toxx", "toyy", "tozz", "toyz", "toxz", "toxy", "toyz_c", "toxz_c", "toxy_c",
              "eta", "eta_vep", "EII_pl", "evol_pl", "EVol_pl", "fx", "fy", "fz", "RP", "Rx", "Ry", "Rz",
              "omega_yz", "omega_xz", "omega_xy", "phase_c", "phase_yz", "phase_xz", "phase_xy", "T"]
VEP3D = _mkstruct("VEP3D", VEP3_NAMES)


class VEPParams3D(C.Structure):
    _fields_ = [("nx", C.c_int64), ("ny", C.c_int64), ("nz", C.c_int64), ("nxg", C.c_int64), ("nyg", C.c_int64), ("nzg", C.c_int64),
                ("_dx", C.c_double), ("_dy", C.c_double), ("_dz", C.c_double),
                ("dt", C.c_double), ("r", C.c_double), ("theta_dtau", C.c_double), ("eta_dtau", C.c_double),
                ("eps_rel", C.c_double), ("eps_abs", C.c_double), ("iterMax", C.c_int64), ("nout", C.c_int64),
                ("free_slip", C.c_uint32), ("no_slip", C.c_uint32), ("periodic", C.c_uint32),
                ("lambda_relaxation", C.c_double), ("viscosity_relaxation", C.c_double), ("cutoff_lo", C.c_double), ("cutoff_hi", C.c_double),
                ("displacement_bcs", C.c_int32), ("T_ghosted", C.c_int32)]


def vep_shapes3d(nx, ny, nz, nphase):
    c = (nx, ny, nz)
    yz, xz, xy = (nx, ny + 1, nz + 1), (nx + 1, ny, nz + 1), (nx + 1, ny + 1, nz)
    s = {n: c for n in VEP3_NAMES}
    for pre in ("e", "epl", "de", "t", "to"):
        s[pre + "yz"], s[pre + "xz"], s[pre + "xy"] = yz, xz, xy
    s.update(Vx=(nx + 1, ny + 2, nz + 2), Vy=(nx + 2, ny + 1, nz + 2), Vz=(nx + 2, ny + 2, nz + 1),
             Ux=(nx + 1, ny + 2, nz + 2), Uy=(nx + 2, ny + 1, nz + 2), Uz=(nx + 2, ny + 2, nz + 1),
             Rx=(nx - 1, ny, nz), Ry=(nx, ny - 1, nz), Rz=(nx, ny, nz - 1), omega_yz=yz, omega_xz=xz, omega_xy=xy,
             phase_c=(nphase,) + c, phase_yz=(nphase,) + yz, phase_xz=(nphase,) + xz, phase_xy=(nphase,) + xy)
    return s


def vep_params3d(ni, _di, dt, pt, *, iterMax=10_000, nout=500, free_slip=None, no_slip=None, periodic=None,
                 lambda_relaxation=0.2, viscosity_relaxation=1e-2, cutoff=(-np.inf, np.inf), ni_g=None, displacement_bcs=False, T_ghosted=False) -> VEPParams3D:
    ni_g = ni_g or ni
    return VEPParams3D(ni[0], ni[1], ni[2], ni_g[0], ni_g[1], ni_g[2], _di[0], _di[1], _di[2], dt, pt["r"], pt["theta_dtau"], pt["eta_dtau"],
                       pt["eps_rel"], pt["eps_abs"], int(iterMax), int(nout), bcmask(free_slip), bcmask(no_slip), bcmask(periodic),
                       lambda_relaxation, viscosity_relaxation, cutoff[0], cutoff[1], int(bool(displacement_bcs)), int(bool(T_ghosted)))


def vep3d(arr: dict) -> VEP3D:
    f = VEP3D()
    for n in VEP3_NAMES:
        setattr(f, n, _p(arr.get(n)))
    return f


def stokes3d_vep_solve(arr: dict, rh: Rheology, p: VEPParams3D) -> dict:
    res = _Res(int(p.iterMax // p.nout + 2))
    f = vep3d(arr)
    L = lib()
    L.orc_stokes3d_vep_solve.restype = C.c_int32
    L.orc_stokes3d_vep_solve(C.byref(f), C.byref(rh), C.byref(p), C.byref(res.c))
    return res.asdict(3)


def vep3d_stress(arr: dict, theta, lam, lamv, rh: Rheology, p: VEPParams3D):
    """update_stresses_center_vertex_ps! 3D; lamv = (λv_yz, λv_xz, λv_xy)"""
    f = vep3d(arr)
    lv = (_dp * 3)(*[_p(a) for a in lamv])
    lib().orc_vep3d_stress(C.byref(f), _p(theta), _p(lam), lv, C.byref(rh), C.byref(p))


# ---- 3D PT heat diffusion (oracle/thermal3d.c) ----
T3_NAMES = ["T", "Told", "dT", "qTx", "qTx2", "qTy", "qTy2", "qTz", "qTz2", "H", "shear_heating", "ResT", "K", "rhoCp", "thetar_dtau", "dtau_rho"]
Thermal3D = _mkstruct("Thermal3D", T3_NAMES + T_OPT)
THERMAL_FACES3 = ("left", "right", "front", "back", "top", "bot")


class ThermalParams3D(C.Structure):
    _fields_ = [("nx", C.c_int64), ("ny", C.c_int64), ("nz", C.c_int64), ("_dx", C.c_double), ("_dy", C.c_double), ("_dz", C.c_double),
                ("dt", C.c_double), ("eps", C.c_double), ("iterMax", C.c_int64), ("nout", C.c_int64),
                ("no_flux", C.c_int32 * 6),
                ("constant_value_on", C.c_int32 * 6), ("constant_value", C.c_double * 6),
                ("constant_flux_on", C.c_int32 * 6), ("constant_flux", C.c_double * 6),
                ("periodic", C.c_int32 * 6),
                ("rheology_form", C.c_int32),
                ("k_const", C.c_double), ("Cp", C.c_double), ("rho0", C.c_double), ("alpha", C.c_double), ("T0", C.c_double),
                ("dirichlet_const", C.c_double)]


def thermal_shapes3d(nx, ny, nz):
    c, g = (nx, ny, nz), (nx + 2, ny + 2, nz + 2)
    s = {k: c for k in ("H", "shear_heating", "ResT", "K", "rhoCp", "thetar_dtau", "dtau_rho")}
    s.update(T=g, Told=g, dT=g, qTx=(nx + 1, ny, nz), qTx2=(nx + 1, ny, nz), qTy=(nx, ny + 1, nz), qTy2=(nx, ny + 1, nz),
             qTz=(nx, ny, nz + 1), qTz2=(nx, ny, nz + 1))
    return s


def thermal_params3d(ni, _di, dt, eps, *, iterMax=50_000, nout=1000, no_flux=None, constant_value=None, constant_flux=None, periodic=None,
                     rheology=None) -> ThermalParams3D:
    """BC dicts use the reference's 3D face names left/right/front/back/top/bot; False/None disables; `True` in constant_value
    counts as the number 1 exactly as `2 * bc.left - T` does in the reference."""
    p = ThermalParams3D()
    p.nx, p.ny, p.nz, p._dx, p._dy, p._dz = ni[0], ni[1], ni[2], _di[0], _di[1], _di[2]
    p.dt, p.eps, p.iterMax, p.nout = dt, eps, int(iterMax), int(nout)
    for i, k in enumerate(THERMAL_FACES3):
        p.no_flux[i] = int(bool((no_flux or {}).get(k, False)))
        p.periodic[i] = int(bool((periodic or {}).get(k, False)))
        v = (constant_value or {}).get(k, False)
        p.constant_value_on[i] = int(v is not False and v is not None)
        p.constant_value[i] = float(v) if p.constant_value_on[i] else 0.0
        v = (constant_flux or {}).get(k, False)
        on = not isinstance(v, bool) and v is not None
        p.constant_flux_on[i] = int(on)
        p.constant_flux[i] = float(v) if on else 0.0
    if rheology:
        p.rheology_form = 1
        p.k_const, p.Cp, p.rho0, p.alpha, p.T0 = (rheology["k"], rheology["Cp"], rheology["rho0"], rheology["alpha"], rheology.get("T0", 0.0))
    return p


def thermal3d(arr: dict) -> Thermal3D:
    f = Thermal3D()
    for n in T3_NAMES + T_OPT:
        setattr(f, n, _p(arr.get(n)))
    return f


def heatdiffusion_PT3d(arr: dict, p: ThermalParams3D) -> dict:
    cap = int(p.iterMax // p.nout + 2)
    it, nr, nn = np.zeros(cap, dtype=np.int64), np.zeros(cap), C.c_int64(0)
    t = thermal3d(arr)
    lib().orc_heatdiffusion_PT3d(C.byref(t), C.byref(p), it.ctypes.data_as(C.POINTER(C.c_int64)), _p(nr), C.c_int64(cap), C.byref(nn))
    return dict(iter_count=it[:nn.value].copy(), norm_ResT=nr[:nn.value].copy())


def thermal3d_iteration(arr: dict, p: ThermalParams3D):
    t = thermal3d(arr)
    lib().orc_thermal3d_iteration(C.byref(t), C.byref(p))


def thermal3d_check_res(arr: dict, p: ThermalParams3D):
    t = thermal3d(arr)
    lib().orc_thermal3d_check_res(C.byref(t), C.byref(p))


# ---- phase-ratio form of the heat-diffusion path (oracle/thermal_phases.h) ----
class ThermalPhases(C.Structure):
    _fields_ = [("nphase", C.c_int32)] + [(k, C.c_double * 8) for k in ("k", "Cp", "Hr")] + [("rho_kind", C.c_int32 * 8)] + \
               [(k, C.c_double * 8) for k in ("rho0", "alpha", "beta", "T0", "P0")] + [("max_lxyz", C.c_double), ("Vpdtau", C.c_double)]


class ThermalPhaseFields(C.Structure):
    _fields_ = [(k, _dp) for k in ("P", "phase_c", "phase_qx", "phase_qy", "phase_qz")]


RHO_KINDS = {"constant": 0, "PT": 1, "T": 2, "compressible": 3}


def thermal_phases(phases: list, max_lxyz: float, Vpdtau: float) -> ThermalPhases:
    """phases: list of dict(k, Cp, Hr=0, density=dict(kind, rho0, alpha, beta, T0, P0))"""
    m = ThermalPhases()
    m.nphase = len(phases)
    for q, ph in enumerate(phases):
        m.k[q], m.Cp[q], m.Hr[q] = ph["k"], ph["Cp"], ph.get("Hr", 0.0)
        d = ph.get("density", {})
        m.rho_kind[q] = RHO_KINDS[d.get("kind", "constant")]
        m.rho0[q], m.alpha[q], m.beta[q], m.T0[q], m.P0[q] = d.get("rho0", 0.0), d.get("alpha", 0.0), d.get("beta", 0.0), d.get("T0", 0.0), d.get("P0", 0.0)
    m.max_lxyz, m.Vpdtau = float(max_lxyz), float(Vpdtau)
    return m


def _phase_fields(ph: dict) -> ThermalPhaseFields:
    f = ThermalPhaseFields()
    for k in ("P", "phase_c", "phase_qx", "phase_qy", "phase_qz"):
        setattr(f, k, _p(ph.get(k)))
    return f


def heatdiffusion_PT_phases(arr: dict, p, m: ThermalPhases, ph: dict) -> dict:
    """heatdiffusion_PT!(...; kwargs = (phase = phase_ratios, ...)), 2D or 3D by the type of p; ph: dict(P, phase_c, phase_qx, phase_qy[, phase_qz])"""
    three = isinstance(p, ThermalParams3D)
    f = _phase_fields(ph)
    p.rheology_form = 2
    lib().orc_thermal_set_phases(C.byref(m), C.byref(f))
    try:
        return heatdiffusion_PT3d(arr, p) if three else heatdiffusion_PT2d(arr, p)
    finally:
        lib().orc_thermal_set_phases(None, None)


def adiabatic_heating(A, P, P0, m: "ThermalPhases", phase_c, _dt):
    """adiabatic_heating!(thermal, stokes, rheology, phases, _dt) -- DiffusionPT_kernels.jl:720-746; phase_c None: phase 0 alone"""
    lib().orc_adiabatic_heating(_p(A), _p(P), _p(P0), C.c_int64(A.size), C.byref(m), _p(phase_c), C.c_double(_dt))


def first_touch(a: np.ndarray) -> np.ndarray:
    """a copy of the Fortran-ordered array `a` whose pages are first touched by the OpenMP threads that own the corresponding slabs of the slowest index
    (NUMA placement for the timed CPU baseline of bench.py)"""
    assert a.flags.f_contiguous and a.dtype == np.float64
    out = np.empty(a.shape, dtype=np.float64, order="F")
    nslab = a.shape[-1]
    lib().orc_first_touch_copy(_p(out), _p(a), C.c_int64(a.size // nslab), C.c_int64(nslab))
    return out


# ----------------------------------------------------------------------------- grid operators either side of the solves (gridops.c)
def _i64s(*ns):
    return [C.c_int64(int(n)) for n in ns]


def velocity2vertex(Vx, Vy, Vz=None, out_shape=None):
    """velocity2vertex!(Vx_v, Vy_v[, Vz_v], Vx, Vy[, Vz]) -- Interpolations.jl:212-249; returns the outputs (shape out_shape, default ni .+ 1)"""
    if Vz is None:
        nx, ny = Vx.shape[0] - 1, Vx.shape[1] - 2
        m = tuple(out_shape or (nx + 1, ny + 1))
        o = [np.zeros(m, order="F") for _ in range(2)]
        lib().orc_velocity2vertex2d(_p(o[0]), _p(o[1]), _p(Vx), _p(Vy), *_i64s(nx, ny, *m))
    else:
        nx, ny, nz = Vx.shape[0] - 1, Vx.shape[1] - 2, Vx.shape[2] - 2
        m = tuple(out_shape or (nx + 1, ny + 1, nz + 1))
        o = [np.zeros(m, order="F") for _ in range(3)]
        lib().orc_velocity2vertex3d(_p(o[0]), _p(o[1]), _p(o[2]), _p(Vx), _p(Vy), _p(Vz), *_i64s(nx, ny, nz, *m))
    return o


def velocity2center(Vx, Vy, Vz=None):
    """velocity2center! -- Interpolations.jl:257-289"""
    if Vz is None:
        nx, ny = Vx.shape[0] - 1, Vx.shape[1] - 2
        o = [np.zeros((nx, ny), order="F") for _ in range(2)]
        lib().orc_velocity2center2d(_p(o[0]), _p(o[1]), _p(Vx), _p(Vy), *_i64s(nx, ny))
    else:
        nx, ny, nz = Vx.shape[0] - 1, Vx.shape[1] - 2, Vx.shape[2] - 2
        o = [np.zeros((nx, ny, nz), order="F") for _ in range(3)]
        lib().orc_velocity2center3d(_p(o[0]), _p(o[1]), _p(o[2]), _p(Vx), _p(Vy), _p(Vz), *_i64s(nx, ny, nz))
    return o


def vertex2center(center, vertex, ghost=(False, False, False)):
    """vertex2center!(center, vertex; ghost_x, ghost_y, ghost_z) in place -- Interpolations.jl:72-96"""
    nd = vertex.ndim
    vd = (C.c_int64 * 3)(*vertex.shape, *([1] * (3 - nd)))
    cd = (C.c_int64 * 3)(*center.shape, *([1] * (3 - nd)))
    g = list(ghost) + [False] * (3 - len(ghost))
    lib().orc_vertex2center(_p(center), _p(vertex), vd, cd, C.c_int32(nd), *[C.c_int32(bool(x)) for x in g])


def center2vertex_harm(center):
    """center2vertex_harm! -- Interpolations.jl:116-137"""
    nx, ny = center.shape
    v = np.zeros((nx + 1, ny + 1), order="F")
    lib().orc_center2vertex_harm2d(_p(v), _p(center), *_i64s(nx, ny))
    return v


def center2vertex3d(vyz, vxz, vxy, cyz, cxz, cxy):
    """center2vertex!(vertex_yz, vertex_xz, vertex_xy, center_yz, center_xz, center_xy) in place -- Interpolations.jl:139-178"""
    lib().orc_center2vertex3d(_p(vyz), _p(vxz), _p(vxy), _p(cyz), _p(cxz), _p(cxy), *_i64s(*cyz.shape))


def compute_rhog(rh: "Rheology", T, P, phase_c=None, shape=None):
    """compute_ρg!(ρg[end], [phase_ratios,] rheology, (; T, P)) -- BuoyancyForces.jl:6-60; T may be larger than ρg (read at [i, j, k] without a shift)"""
    shape = tuple(shape or (P if P is not None else T).shape)
    nd = len(shape)
    out = np.zeros(shape, order="F")
    n = (C.c_int64 * 3)(*shape, *([1] * (3 - nd)))
    td = (C.c_int64 * 3)(*(T.shape if T is not None else shape), *([1] * (3 - nd)))
    lib().orc_compute_rhog(_p(out), C.byref(rh), _p(phase_c), _p(T), _p(P), n, td, C.c_int32(nd))
    return out


def compute_lithostatic_pressure(rhog, dz):
    """compute_lithostatic_pressure!(P, ρg, dz) -- src/Utils.jl:541-573; dz: a number or one height per cell of the last dimension"""
    nd = rhog.ndim
    P = np.zeros(rhog.shape, order="F")
    n = (C.c_int64 * 3)(*rhog.shape, *([1] * (3 - nd)))
    vec = None if np.isscalar(dz) else np.ascontiguousarray(dz, dtype=np.float64)
    lib().orc_compute_lithostatic_pressure(_p(P), _p(rhog), C.c_double(float(dz) if vec is None else 0.0), vec.ctypes.data_as(_dp) if vec is not None else _dp(), n, C.c_int32(nd))
    return P


def compute_viscosity_single(eta, rh: "Rheology", T, P, cutoff=(-np.inf, np.inf), nu=1.0, AII=None, tau=False):
    """compute_viscosity!(stokes, args, rheology::MaterialParams, cutoff; relaxation) in place on eta -- rheology/Viscosity.jl:118-167 (T: ni or ni .+ 2)"""
    nd = eta.ndim
    n = (C.c_int64 * 3)(*eta.shape, *([1] * (3 - nd)))
    td = (C.c_int64 * 3)(*(T.shape if T is not None else eta.shape), *([1] * (3 - nd)))
    lib().orc_compute_viscosity_single(_p(eta), C.byref(rh), _p(T), _p(P), n, td, C.c_int32(nd), C.c_double(nu), C.c_double(cutoff[0]), C.c_double(cutoff[1]),
                                       _p(AII), C.c_int32(int(tau)))


def compute_shear_heating(tau, tau_o, eps, rh: "Rheology", chi, dt, phase_c=None):
    """compute_shear_heating! -- ShearHeating.jl:14-71; tau, tau_o: centre arrays in Voigt order, eps: staggered strain rates; chi: Χ per phase"""
    ni = tau[0].shape
    nd = len(ni)
    out = np.zeros(ni, order="F")
    arr = lambda ts: (_dp * 6)(*[_p(t) for t in ts])
    x = (C.c_double * MAXPHASE)(*[float(c) for c in chi])
    n = (C.c_int64 * 3)(*ni, *([1] * (3 - nd)))
    lib().orc_compute_shear_heating(_p(out), arr(tau), arr(tau_o), arr(eps), _p(phase_c), C.byref(rh), x, C.c_double(dt), n, C.c_int32(nd))
    return out
