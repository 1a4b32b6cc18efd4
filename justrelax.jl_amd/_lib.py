"""ctypes binding of libjrx_hip.so (the C ABI of include/jrx.h).

There is no fallback: if the shared library is missing or a symbol is absent, loading raises.
"""
from __future__ import annotations

import ctypes as C
import re
from pathlib import Path

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "lib" / "libjrx_hip.so"
HEADER = HERE.parent / "include" / "jrx.h"

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)

STATUS = {0: "JRX_OK", 1: "JRX_ERR_NAN", 2: "JRX_ERR_HIP", 3: "JRX_ERR_RCCL", 4: "JRX_ERR_ARG", 5: "JRX_ERR_UNSUPPORTED"}
FACE = dict(left=1, right=2, front=4, back=8, top=16, bot=32)
OUT_STATE_ONLY, OUT_DIAG = 0, 1
UNIQUE_ID_BYTES = 128

F3_NAMES = ["P", "P0", "divV", "Q", "Vx", "Vy", "Vz", "Ux", "Uy", "Uz",
            "txx", "tyy", "tzz", "tyz", "txz", "txy", "toxx", "toyy", "tozz", "toyz", "toxz", "toxy",
            "exx", "eyy", "ezz", "eyz", "exz", "exy", "eta", "K", "G", "fx", "fy", "fz", "RP", "Rx", "Ry", "Rz",
            "tyz_c", "txz_c", "txy_c", "toyz_c", "toxz_c", "toxy_c"]
F2_NAMES = ["P", "P0", "divV", "Q", "Vx", "Vy", "Ux", "Uy", "txx", "tyy", "txy", "toxx", "toyy", "toxy",
            "exx", "eyy", "exy", "eta", "K", "G", "fx", "fy", "RP", "Rx", "Ry", "txy_c", "toxy_c"]
T_OPT = ["adiabatic", "dirichlet_mask", "dirichlet_value"]          # optional members of the thermal field structs (NULL = absent)
T2_NAMES = ["T", "Told", "dT", "qTx", "qTx2", "qTy", "qTy2", "H", "shear_heating", "ResT", "K", "rhoCp",
            "thetar_dtau", "dtau_rho"]


def _ptr_struct(name, names):
    return type(name, (C.Structure,), {"_fields_": [(n, C.c_void_p) for n in names]})


Stokes3DFields = _ptr_struct("Stokes3DFields", F3_NAMES)
Stokes2DFields = _ptr_struct("Stokes2DFields", F2_NAMES)
Thermal2DFields = _ptr_struct("Thermal2DFields", T2_NAMES + T_OPT)


class Stokes3DParams(C.Structure):
    _fields_ = [("nx", C.c_int64), ("ny", C.c_int64), ("nz", C.c_int64),
                ("nxg", C.c_int64), ("nyg", C.c_int64), ("nzg", C.c_int64),
                ("_dx", C.c_double), ("_dy", C.c_double), ("_dz", C.c_double),
                ("dt", C.c_double), ("r", C.c_double), ("theta_dtau", C.c_double), ("eta_dtau", C.c_double),
                ("eps_rel", C.c_double), ("eps_abs", C.c_double), ("iterMax", C.c_int64), ("nout", C.c_int64),
                ("free_slip", C.c_uint32), ("no_slip", C.c_uint32), ("periodic", C.c_uint32),
                ("b_width", C.c_int32 * 3), ("verbose", C.c_int32), ("displacement_bcs", C.c_int32)]


class Stokes2DParams(C.Structure):
    _fields_ = [("nx", C.c_int64), ("ny", C.c_int64), ("nxg", C.c_int64), ("nyg", C.c_int64),
                ("_dx", C.c_double), ("_dy", C.c_double),
                ("dt", C.c_double), ("r", C.c_double), ("theta_dtau", C.c_double), ("eta_dtau", C.c_double),
                ("eps_rel", C.c_double), ("eps_abs", C.c_double), ("iterMax", C.c_int64), ("nout", C.c_int64),
                ("free_slip", C.c_uint32), ("no_slip", C.c_uint32), ("periodic", C.c_uint32), ("verbose", C.c_int32),
                ("displacement_bcs", C.c_int32), ("inv_spacing", C.c_void_p * 6)]


class Thermal2DParams(C.Structure):
    _fields_ = [("nx", C.c_int64), ("ny", C.c_int64), ("_dx", C.c_double), ("_dy", C.c_double),
                ("dt", C.c_double), ("eps", C.c_double), ("iterMax", C.c_int64), ("nout", C.c_int64),
                ("no_flux", C.c_int32 * 4),
                ("constant_value_on", C.c_int32 * 4), ("constant_value", C.c_double * 4),
                ("constant_flux_on", C.c_int32 * 4), ("constant_flux", C.c_double * 4),
                ("periodic", C.c_int32 * 4), ("rheology_form", C.c_int32),
                ("k_const", C.c_double), ("Cp", C.c_double), ("rho0", C.c_double), ("alpha", C.c_double),
                ("T0", C.c_double), ("verbose", C.c_int32), ("dirichlet_const", C.c_double), ("inv_spacing", C.c_void_p * 4)]


T3_NAMES = ["T", "Told", "dT", "qTx", "qTx2", "qTy", "qTy2", "qTz", "qTz2", "H", "shear_heating", "ResT", "K", "rhoCp", "thetar_dtau", "dtau_rho"]
Thermal3DFields = _ptr_struct("Thermal3DFields", T3_NAMES + T_OPT)


class Thermal3DParams(C.Structure):
    _fields_ = [("nx", C.c_int64), ("ny", C.c_int64), ("nz", C.c_int64), ("_dx", C.c_double), ("_dy", C.c_double), ("_dz", C.c_double),
                ("dt", C.c_double), ("eps", C.c_double), ("iterMax", C.c_int64), ("nout", C.c_int64),
                ("no_flux", C.c_int32 * 6),
                ("constant_value_on", C.c_int32 * 6), ("constant_value", C.c_double * 6),
                ("constant_flux_on", C.c_int32 * 6), ("constant_flux", C.c_double * 6),
                ("periodic", C.c_int32 * 6), ("rheology_form", C.c_int32),
                ("k_const", C.c_double), ("Cp", C.c_double), ("rho0", C.c_double), ("alpha", C.c_double), ("T0", C.c_double),
                ("verbose", C.c_int32), ("dirichlet_const", C.c_double)]


class ThermalPhases(C.Structure):
    _fields_ = [("nphase", C.c_int32)] + [(k, C.c_double * 8) for k in ("k", "Cp", "Hr")] + [("rho_kind", C.c_int32 * 8)] + \
               [(k, C.c_double * 8) for k in ("rho0", "alpha", "beta", "T0", "P0")] + [("max_lxyz", C.c_double), ("Vpdtau", C.c_double)]


TPH_NAMES = ["P", "phase_c", "phase_qx", "phase_qy", "phase_qz"]
ThermalPhaseFields = _ptr_struct("ThermalPhaseFields", TPH_NAMES)


VEP_NAMES = ["P", "P0", "divV", "Q", "Vx", "Vy", "Ux", "Uy", "exx", "eyy", "exy", "exy_c", "eplxx", "eplyy", "eplxy", "eplxy_c",
             "dexy_c", "dexy", "txx", "tyy", "txy", "txy_c", "tII", "toxx", "toyy", "toxy", "toxy_c", "eta", "eta_v", "eta_vep",
             "EII_pl", "evol_pl", "EVol_pl", "fx", "fy", "RP", "Rx", "Ry", "omega_xy", "phase_c", "phase_v", "T", "dexx", "deyy", "divU"]
VEP2DFields = _ptr_struct("VEP2DFields", VEP_NAMES)
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


class VEP2DParams(C.Structure):
    _fields_ = [("nx", C.c_int64), ("ny", C.c_int64), ("nxg", C.c_int64), ("nyg", C.c_int64), ("_dx", C.c_double), ("_dy", C.c_double),
                ("dt", C.c_double), ("r", C.c_double), ("theta_dtau", C.c_double), ("eta_dtau", C.c_double),
                ("eps_rel", C.c_double), ("eps_abs", C.c_double), ("iterMax", C.c_int64), ("iterMin", C.c_int64), ("nout", C.c_int64),
                ("free_slip", C.c_uint32), ("no_slip", C.c_uint32), ("periodic", C.c_uint32),
                ("lambda_relaxation", C.c_double), ("viscosity_relaxation", C.c_double), ("cutoff_lo", C.c_double), ("cutoff_hi", C.c_double),
                ("verbose", C.c_int32), ("free_surface", C.c_int32), ("displacement_bcs", C.c_int32), ("T_ghosted", C.c_int32), ("strain_increment", C.c_int32),
                ("inv_spacing", C.c_void_p * 6)]


VEP3_NAMES = ["P", "P0", "divV", "Q", "Vx", "Vy", "Vz", "Ux", "Uy", "Uz",
              "exx", "eyy", "ezz", "eyz", "exz", "exy", "eyz_c", "exz_c", "exy_c",
              "eplxx", "eplyy", "eplzz", "eplyz", "eplxz", "eplxy", "eplyz_c", "eplxz_c", "eplxy_c",
              "deyz", "dexz", "dexy", "deyz_c", "dexz_c", "dexy_c",
              "txx", "tyy", "tzz", "tyz", "txz", "txy", "tyz_c", "txz_c", "txy_c", "tII",
              "toxx", "toyy", "tozz", "toyz", "toxz", "toxy", "toyz_c", "toxz_c", "toxy_c",
              "eta", "eta_vep", "EII_pl", "evol_pl", "EVol_pl", "fx", "fy", "fz", "RP", "Rx", "Ry", "Rz",
              "omega_yz", "omega_xz", "omega_xy", "phase_c", "phase_yz", "phase_xz", "phase_xy", "T"]
VEP3DFields = _ptr_struct("VEP3DFields", VEP3_NAMES)


class VEP3DParams(C.Structure):
    _fields_ = [("nx", C.c_int64), ("ny", C.c_int64), ("nz", C.c_int64), ("nxg", C.c_int64), ("nyg", C.c_int64), ("nzg", C.c_int64),
                ("_dx", C.c_double), ("_dy", C.c_double), ("_dz", C.c_double),
                ("dt", C.c_double), ("r", C.c_double), ("theta_dtau", C.c_double), ("eta_dtau", C.c_double),
                ("eps_rel", C.c_double), ("eps_abs", C.c_double), ("iterMax", C.c_int64), ("nout", C.c_int64),
                ("free_slip", C.c_uint32), ("no_slip", C.c_uint32), ("periodic", C.c_uint32),
                ("lambda_relaxation", C.c_double), ("viscosity_relaxation", C.c_double), ("cutoff_lo", C.c_double), ("cutoff_hi", C.c_double),
                ("verbose", C.c_int32), ("displacement_bcs", C.c_int32), ("T_ghosted", C.c_int32), ("b_width", C.c_int32 * 3)]


class SolveResult(C.Structure):
    _fields_ = [("iter", C.c_int64), ("nchecks", C.c_int64), ("cap", C.c_int64),
                ("err_evo1", _dp), ("err_evo2", _ip),
                ("norm_Rx", _dp), ("norm_Ry", _dp), ("norm_Rz", _dp), ("norm_divV", _dp),
                ("time_s", C.c_double), ("av_time_s", C.c_double)]


class Cart(C.Structure):
    _fields_ = [("rank", C.c_int32), ("nprocs", C.c_int32), ("dims", C.c_int32 * 3), ("coords", C.c_int32 * 3),
                ("periods", C.c_int32 * 3), ("neighbor", (C.c_int32 * 2) * 3)]


TUNING_HEADER = HERE.parent / "include" / "jrx_tuning.h"
ABI_VERSION = 230        # 230: allocation-time pool placement; jrx_field_tune, jrx_stokes3d_tune_placement and the in-place re-mapping are gone; 220: jrx_field_alloc / _free / _trim / _stats, option field_placement; jrx_fields_dirty, option operand_cache; JRX_VERSION this binding's structs and option keys follow (210: jrx_comm_init_ipc, viscous-limit counters; 200 -> 210 also covers round 3's b_width[3] of jrx_vep3d_params)


def header_version() -> int:
    m = re.search(r"#define\s+JRX_VERSION\s+(\d+)", HEADER.read_text())
    return int(m.group(1)) if m else -1


def declared_symbols(headers=None) -> list:
    """Every function include/jrx.h (the drop-in ABI) and include/jrx_tuning.h (tuning / test switches) declare."""
    out = set()
    for hp in (headers or (HEADER, TUNING_HEADER)):
        txt = re.sub(r"/\*.*?\*/", "", hp.read_text(), flags=re.S)
        out |= set(re.findall(r"\b(jrx_[A-Za-z0-9_]+)\s*\(", txt))
    return sorted(out)


_lib = None


def load(check_symbols: bool = False):
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(f"{LIB_PATH} not found: run `python justrelax.jl_amd/build.py` (hipcc, gfx950). "
                               "There is no CPU fallback.")
        import torch  # noqa: F401  -- loads the process-wide HIP runtime (libamdhip64.so.7) first
        L = C.CDLL(str(LIB_PATH))
        # the binary must have been built from the sources beside it (csrc/ + include/jrx.h): a stale .so would silently run old kernels
        L.jrx_build_id.restype = C.c_char_p
        from .build import source_id
        have, want = L.jrx_build_id().decode(), source_id()
        if have != want:
            raise RuntimeError(f"{LIB_PATH} was built from other sources (build id {have[:12]}…, sources {want[:12]}…): "
                               "run `python justrelax.jl_amd/build.py`")
        # ... and speak the ABI revision this binding was written against (struct layouts, option keys): JRX_VERSION of include/jrx.h
        L.jrx_version.restype = C.c_int32
        if L.jrx_version() != ABI_VERSION or header_version() != ABI_VERSION:
            raise RuntimeError(f"ABI revision mismatch: binding {ABI_VERSION}, include/jrx.h {header_version()}, {LIB_PATH.name} {L.jrx_version()}")
        L.jrx_last_error.restype = C.c_char_p
        L.jrx_last_error.argtypes = [C.c_void_p]
        L.jrx_create.argtypes = [C.c_int32, C.POINTER(C.c_void_p)]
        L.jrx_n_global.restype = C.c_int64
        L.jrx_n_global.argtypes = [C.c_int64, C.c_int32, C.c_int32]
        for name in declared_symbols():
            fn = getattr(L, name)          # raises AttributeError if the symbol is not exported
            if name not in ("jrx_last_error", "jrx_n_global", "jrx_version", "jrx_build_id"):
                fn.restype = C.c_int32
        _lib = L
    if check_symbols:
        missing = [s for s in declared_symbols() if not hasattr(_lib, s)]
        if missing:
            raise RuntimeError(f"libjrx_hip.so lacks symbols declared in jrx.h: {missing}")
    return _lib


class JrxError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"{STATUS.get(status, status)}: {msg}")
        self.status = status


class Handle:
    """jrx_handle: streams, events, reduction scratch, ητ, RCCL communicator.  One per GPU."""

    def __init__(self, device: int = 0):
        self.lib = load()
        self._h = C.c_void_p()
        st = self.lib.jrx_create(C.c_int32(device), C.byref(self._h))
        if st != 0:
            raise JrxError(st, self.lib.jrx_last_error(None).decode())
        self.device = device

    def check(self, st):
        if st != 0:
            raise JrxError(st, self.lib.jrx_last_error(self._h).decode())

    def call(self, name, *args):
        self.check(getattr(self.lib, name)(self._h, *args))

    def set_option(self, key: str, value: int):
        """jrx_set_option for the keys of the public ABI, jrx_tuning_set for the tuning / test switches of include/jrx_tuning.h"""
        k = C.c_char_p(key.encode())
        st = self.lib.jrx_set_option(self._h, k, C.c_int64(int(value)))
        if st != 0:
            # only a key that lives in the other table is retried there; any other failure (a read-only counter, a NULL key) is reported as it is
            if "is a tuning switch" not in self.lib.jrx_last_error(self._h).decode():
                self.check(st)
            self.check(self.lib.jrx_tuning_set(self._h, k, C.c_int64(int(value))))

    def get_option(self, key: str) -> int:
        k, v = C.c_char_p(key.encode()), C.c_int64()
        st = self.lib.jrx_get_option(self._h, k, C.byref(v))
        if st != 0:
            self.check(self.lib.jrx_tuning_get(self._h, k, C.byref(v)))
        return v.value

    def fields_dirty(self):
        """the caller has written to an operand array (τ_o, P0, Q, K, G, η, ρg) since the last driver call: a cached operand verdict (option operand_cache) is dropped"""
        self.call("jrx_fields_dirty")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.jrx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_handles = {}


def default_handle(device=None) -> Handle:
    import torch
    if device is None:
        device = torch.cuda.current_device()
    if device not in _handles:
        _handles[device] = Handle(device)
    return _handles[device]


def bcmask(d) -> int:
    m = 0
    for k, v in (d or {}).items():
        if v:
            m |= FACE[k]
    return m
