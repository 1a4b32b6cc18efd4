"""The reference's boundary-condition known-answer tests (test/test_boundary_conditions2D.jl, test_boundary_conditions3D.jl) applied to
the CPU oracle's flow_bcs! / thermal_bcs! restatements."""
import ctypes as C

import numpy as np
import pytest

from _bc_known_answers import check_flow2d, check_flow3d, check_thermal


def test_flow_bcs_2d(oracle):
    def apply(V, fs, ns, pe):
        n = V[0].shape[0] - 1
        oracle.flow_bcs2d(V[0], V[1], (n, n), free_slip=fs, no_slip=ns, periodic=pe)
        return V
    check_flow2d(apply)


def test_flow_bcs_3d(oracle):
    def apply(V, fs, ns, pe):
        n = V[0].shape[0] - 1
        oracle.flow_bcs3d(V[0], V[1], V[2], (n, n, n), free_slip=fs, no_slip=ns, periodic=pe)
        return V
    check_flow3d(apply)


@pytest.mark.parametrize("nD", [2, 3])
def test_thermal_bcs(oracle, nD):
    def apply(T, nf, cv, pe):
        ni = tuple(m - 2 for m in T.shape)
        if nD == 2:
            oracle.thermal_bcs2d(T, oracle.thermal_params2d(ni, (1.0, 1.0), 1.0, 0.0, no_flux=nf, constant_value=cv, periodic=pe))
        else:
            p = oracle.thermal_params3d(ni, (1.0, 1.0, 1.0), 1.0, 0.0, no_flux=nf, constant_value=cv, periodic=pe)
            oracle.lib().orc_thermal_bcs3d(T.ctypes.data_as(C.POINTER(C.c_double)), C.byref(p))
        return T
    check_thermal(apply, nD)


def test_bc_struct_checks(jr):
    """constructor errors of test_boundary_conditions2D.jl:51-82,113-128 / 3D.jl:62-72,117-122"""
    off4 = dict(left=False, right=False, top=False, bot=False)
    with pytest.raises(ValueError):
        jr.VelocityBoundaryConditions(no_slip=dict(off4, left=True), free_slip=dict(left=True, right=True, top=True, bot=True))
    assert isinstance(jr.VelocityBoundaryConditions(no_slip=off4, free_slip=dict(off4)), jr.VelocityBoundaryConditions)
    with pytest.raises(ValueError, match="Periodic boundary conditions must be paired"):
        jr.VelocityBoundaryConditions(no_slip=off4, free_slip=dict(off4), periodic=dict(off4, left=True))
    with pytest.raises(ValueError):
        jr.VelocityBoundaryConditions(no_slip=off4, free_slip=dict(off4, left=True), periodic=dict(off4, left=True, right=True))
    with pytest.raises(ValueError, match="Periodic boundary conditions must be paired"):
        jr.TemperatureBoundaryConditions(no_flux=off4, periodic=dict(off4, left=True))
    with pytest.raises(ValueError):
        jr.TemperatureBoundaryConditions(no_flux=dict(off4, left=True), periodic=dict(off4, left=True, right=True))
