"""The reference's boundary-condition known-answer tests (test/test_boundary_conditions2D.jl, test_boundary_conditions3D.jl) on the device:
flow_bcs! (velocity and displacement arrays share the kernels) and thermal_bcs! through the C ABI."""
import numpy as np
import pytest

from _bc_known_answers import check_flow2d, check_flow3d, check_thermal

pytestmark = pytest.mark.gpu


def _dev():
    import torch
    return torch.device("cuda", torch.cuda.current_device())


@pytest.mark.parametrize("kind", ["velocity", "displacement"])
@pytest.mark.parametrize("nD", [2, 3])
def test_flow_bcs(jr, nD, kind):
    from types import SimpleNamespace
    from justrelax_jl_amd.arrays import from_numpy
    cls = jr.VelocityBoundaryConditions if kind == "velocity" else jr.DisplacementBoundaryConditions

    def apply(V, fs, ns, pe):
        d = [from_numpy(v, _dev()) for v in V]
        names = ("Vx", "Vy", "Vz")[:nD]
        jr.flow_bcs_(SimpleNamespace(**dict(zip(names, d))), cls(free_slip=fs, no_slip=ns, periodic=pe))
        return [jr.to_numpy(t) for t in d]
    (check_flow2d if nD == 2 else check_flow3d)(apply)


@pytest.mark.parametrize("nD", [2, 3])
def test_thermal_bcs(jr, nD):
    from justrelax_jl_amd import thermal as th
    from justrelax_jl_amd.arrays import from_numpy

    def apply(T, nf, cv, pe):
        d = from_numpy(T, _dev())
        th.thermal_bcs_(d, jr.TemperatureBoundaryConditions(no_flux=nf, constant_value=cv, periodic=pe))
        return jr.to_numpy(d)
    check_thermal(apply, nD)
