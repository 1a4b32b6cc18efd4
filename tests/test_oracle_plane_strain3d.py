"""Numeric anchor for the elastic and compressible terms of the 3D visco-elastic kernels.

Every 3D test of the reference runs with dt = Inf and/or G = K = Inf, so the τ_o / 1/(G dt) terms of compute_τ! 3D
(src/stokes/StressKernels.jl:149-230) and the 1/(K dt) term of compute_P! (src/stokes/PressureKernels.jl:186-195) had no known answer.
The 2D kernels have one: the elastic build-up (miniapps/benchmarks/stokes2D/elastic_buildup/Elastic_BuildUp.jl:4,55-56,75-86;
test/test_stokes_elastic_buildup.jl:47-54, mean error vs 2εη(1 − exp(−Gt/η)) <= 5e-3), which tests/test_oracle_golden.py meets.  A 3D
problem whose fields are uniform along z is algebraically the 2D problem (miniapps.plane_strain3d), so the 3D oracle must (i) meet the same
reference bound and (ii) reproduce the pinned 2D oracle run to round-off; the same construction on random fields with finite K, G, dt
exercises every term of the iteration, the compressible one included."""
import math

import numpy as np
import pytest


def _cmp3(s3, a2, k2):
    """every plane (normal to the uniform axis, ghost planes included) of the 3D array that carries the 2D array `k2`, against it"""
    a3 = s3.arrays[s3.extra["names"][k2]]
    scale = max(np.abs(a2).max(), 1e-300)
    return float(np.abs(a3 - np.expand_dims(a2, s3.extra["axis"])).max() / scale)


def _out_of_plane_is_zero(s3):
    oop = s3.extra["out_of_plane"]
    return all(np.abs(s3.arrays[k]).max() == 0.0 for k in [oop["V"]] + ["t" + c for c in oop["shear"]])


@pytest.mark.parametrize("axis,steps", [(2, 200), (1, 60), (0, 60)])
def test_3d_elastic_buildup_meets_the_reference_bound_and_equals_the_2d_run(oracle, jr, axis, steps):
    """axis = 2: the reference's whole test (200 solves, mean error <= 5e-3) on fields uniform along z -- anchors τxx, τyy, τxy, P.  axis = 1 / 0: the first 60
    solves on fields uniform along y / x, where the 2D τyy lives on τzz and the 2D τxy on τxz / τyz -- the components whose τ_o terms had no numeric anchor
    (VERDICT r3 P1): the same numbers as the pinned 2D run at every compared step."""
    from justrelax_jl_amd import checks
    orc = oracle
    s2 = jr.miniapps.elastic_buildup2d(32)
    s3 = jr.miniapps.plane_strain3d(s2, nz=3, axis=axis)
    kyr, η0, εbg, G = (s2.extra[k] for k in ("kyr", "η0", "εbg", "G"))
    t, errs, worst = 0.0, [], 0.0
    nm = s3.extra["names"]
    for step in range(steps):
        dt = 0.05 * kyr
        s2.dt = s3.dt = dt
        r2 = orc.stokes2d_solve(s2.arrays, checks.oracle_params2d(orc, s2))
        r3 = orc.stokes3d_solve(s3.arrays, checks.oracle_params3d(orc, s3))
        assert r3["iter"] == r2["iter"] == 1000, step
        t += dt
        sol = 2 * εbg * η0 * (1 - math.exp(-G * t / η0))
        errs.append(abs(np.abs(s3.arrays[nm["tyy"]]).max() - sol) / sol)
        if step % 20 == 19 or step < 3:
            for k in ("tyy", "txx", "txy", "toyy", "toxx", "toxy", "P", "Vx", "Vy"):
                worst = max(worst, _cmp3(s3, s2.arrays[k], k))
    if steps == 200:
        assert sum(errs) / len(errs) <= 5.0e-3                                # test_stokes_elastic_buildup.jl:47-54
    else:
        assert max(errs) <= 2.0e-2 and errs[-1] <= 6.0e-3                     # the first steps of the same curve (the 2D run's own errors, checked below to round-off)
    assert worst <= 1e-10, worst                                              # the pinned 2D run, to round-off
    assert _out_of_plane_is_zero(s3)
    assert np.abs(s3.arrays[nm["toyy"]]).max() > (0.5 if steps == 200 else 0.2) * 2 * εbg * η0      # the elastic memory term really is in play


@pytest.mark.parametrize("axis", [2, 1, 0])
@pytest.mark.parametrize("bcs", ["free_slip", "no_slip"])
def test_3d_compressible_viscoelastic_iterations_equal_the_2d_ones_on_uniform_fields(oracle, jr, bcs, axis):
    """random V, P, τ, τ_o, Q, G, K (all finite), uniform η, finite dt: 30 PT iterations of the 3D oracle == the 2D oracle (pinned by SolCx / SolKz /
    the elastic build-up) on every plane: pins compute_P!'s 1/(K dt) term and the τ_o terms of the four stress components the orientation puts in plane --
    the three orientations together cover all six"""
    from justrelax_jl_amd import checks
    orc = oracle
    s2 = jr.miniapps.random_fields2d((19, 14), seed=77, iterMax=29, nout=10, bcs=bcs)
    s2.pt.ϵ_rel = s2.pt.ϵ_abs = 1e-30
    # a uniform viscosity: the 2D driver hands compute_P! ητ (Stokes2D.jl:231-233) where the 3D one hands it η (Stokes3D.jl:79-91) -- a quirk of the
    # reference that makes the two iterations differ for a variable η; with a uniform η they are the same number.  G and K stay random
    s2.arrays["eta"][...] = 0.37
    orc.flow_bcs2d(s2.arrays["Vx"], s2.arrays["Vy"], s2.ni, **{k: getattr(s2.flow_bcs, k) for k in ("free_slip", "no_slip", "periodic")})
    s3 = jr.miniapps.plane_strain3d(s2, nz=4, axis=axis)
    p2, p3 = checks.oracle_params2d(orc, s2), checks.oracle_params3d(orc, s3)
    P_before = s2.arrays["P"].copy()
    r2 = orc.stokes2d_solve(s2.arrays, p2)
    r3 = orc.stokes3d_solve(s3.arrays, p3)
    assert r2["iter"] == r3["iter"] == 30
    assert np.isfinite(s2.arrays["K"]).all() and np.abs(s2.arrays["P"] - P_before).max() > 1e-3
    for k in ("P", "Vx", "Vy", "txx", "tyy", "txy", "toxx", "toyy", "toxy", "exx", "eyy", "exy", "RP", "Rx", "Ry", "divV"):
        assert _cmp3(s3, s2.arrays[k], k) <= 1e-11, k
    oop = s3.extra["out_of_plane"]
    assert np.abs(s3.arrays["t" + oop["tn"]]).max() > 0.0          # ε_uu = −∇V/3 ≠ 0 here: the out-of-plane normal stress evolves without feeding back
    assert _out_of_plane_is_zero(s3)
