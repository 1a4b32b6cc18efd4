"""The boundary-condition known-answer checks of the reference's test/test_boundary_conditions2D.jl and
test/test_boundary_conditions3D.jl, written once and applied to both the CPU oracle (tests/test_oracle_bcs.py) and the HIP library
(tests/test_gpu_bcs.py).  `apply_flow(V, free_slip, no_slip, periodic)` and `apply_thermal(T, no_flux, constant_value, periodic)`
take and return host arrays (lists / arrays in the reference's layout)."""
import numpy as np

F4 = ("left", "right", "top", "bot")
F6 = ("left", "right", "front", "back", "top", "bot")


def faces(names, on=()):
    on = names if on == "all" else on
    return {f: (f in on) for f in names}


def check_thermal(apply_thermal, nD):
    """test_boundary_conditions2D.jl:20-49 / 3D.jl:19-52: T = 1:N reshaped; constant_value = true on every face (the value `true` is
    the number 1: ghost = 2 - interior); periodic on every face"""
    names = F4 if nD == 2 else F6
    shape = (6, 7) if nD == 2 else (6, 7, 8)
    T0 = np.asfortranarray(np.arange(1.0, np.prod(shape) + 1).reshape(shape, order="F"))
    inner = slice(1, -1)

    def faces_of(T):
        out = {}
        for d in range(nD):
            for name, (g, i_) in (("lo", (0, 1)), ("hi", (-1, -2))):
                idx = [inner] * nD
                idx[d] = g
                jdx = [inner] * nD
                jdx[d] = i_
                out[(d, name)] = (T[tuple(idx)], tuple(jdx))
        return out
    T = apply_thermal(T0.copy(order="F"), faces(names), {f: True for f in names}, faces(names))
    for (d, name), (ghost, jdx) in faces_of(T).items():
        assert np.array_equal(ghost, 2 - T0[jdx]), ("constant_value", d, name)
    T = apply_thermal(T0.copy(order="F"), faces(names), faces(names), faces(names, "all"))
    for d in range(nD):
        idx = [inner] * nD
        lo, hi, lo_src, hi_src = list(idx), list(idx), list(idx), list(idx)
        lo[d], hi[d], lo_src[d], hi_src[d] = 0, -1, -2, 1
        assert np.array_equal(T[tuple(lo)], T0[tuple(lo_src)]) and np.array_equal(T[tuple(hi)], T0[tuple(hi_src)]), ("periodic", d)


def check_flow2d(apply_flow, n=5, seed=0):
    """test_boundary_conditions2D.jl:84-177 (velocities) and :199-274 (displacements use the same kernels)"""
    rng = np.random.default_rng(seed)
    new = lambda: [np.asfortranarray(rng.random((n + 1, n + 2))), np.asfortranarray(rng.random((n + 2, n + 1)))]
    Vx, Vy = apply_flow(new(), faces(F4, "all"), faces(F4), faces(F4))                       # free slip
    assert np.array_equal(Vx[:, 0], Vx[:, 1]) and np.array_equal(Vx[:, -1], Vx[:, -2])
    assert np.array_equal(Vy[0, :], Vy[1, :]) and np.array_equal(Vy[-1, :], Vy[-2, :])
    V0 = new()
    Vx, Vy = apply_flow([v.copy(order="F") for v in V0], faces(F4), faces(F4), faces(F4, ("left", "right")))   # periodic in x
    assert np.array_equal(Vx[0, :], V0[0][-1, :]) and np.array_equal(Vx[-1, :], V0[0][-1, :])
    assert np.array_equal(Vy[0, :], V0[1][-2, :]) and np.array_equal(Vy[-1, :], V0[1][1, :])
    Vx, Vy = apply_flow(new(), faces(F4), faces(F4, "all"), faces(F4))                       # no slip
    assert not Vx[0, :].any() and not Vx[-1, :].any() and not Vy[:, 0].any() and not Vy[:, -1].any()
    assert np.array_equal(Vy[0, :], -Vy[1, :]) and np.array_equal(Vy[-1, :], -Vy[-2, :])
    assert np.array_equal(Vx[:, 0], -Vx[:, 1]) and np.array_equal(Vx[:, -1], -Vx[:, -2])


def check_flow3d(apply_flow, n=5, seed=0):
    """test_boundary_conditions3D.jl:74-149"""
    rng = np.random.default_rng(seed)
    shp = [(n + 1, n + 2, n + 2), (n + 2, n + 1, n + 2), (n + 2, n + 2, n + 1)]
    new = lambda: [np.asfortranarray(rng.random(s)) for s in shp]
    V = apply_flow(new(), faces(F6, "all"), faces(F6), faces(F6))
    V = apply_flow(V, faces(F6, "all"), faces(F6), faces(F6))                                # the reference applies it twice too (:86-87)
    Vx, Vy, Vz = V
    assert np.array_equal(Vx[:, :, 0], Vx[:, :, 1]) and np.array_equal(Vx[:, :, -1], Vx[:, :, -2])
    assert np.array_equal(Vx[:, 0, :], Vx[:, 1, :]) and np.array_equal(Vx[:, -1, :], Vx[:, -2, :])
    assert np.array_equal(Vy[:, :, 0], Vy[:, :, 1]) and np.array_equal(Vy[:, :, -1], Vy[:, :, -2])
    assert np.array_equal(Vy[0, :, :], Vy[1, :, :]) and np.array_equal(Vy[-1, :, :], Vy[-2, :, :])
    assert np.array_equal(Vz[0, :, :], Vz[1, :, :]) and np.array_equal(Vz[-1, :, :], Vz[-2, :, :])
    assert np.array_equal(Vz[:, 0, :], Vz[:, 1, :]) and np.array_equal(Vz[:, -1, :], Vz[:, -2, :])
    V0 = [v.copy(order="F") for v in V]
    Vx, Vy, Vz = apply_flow([v.copy(order="F") for v in V0], faces(F6), faces(F6), faces(F6, "all"))     # periodic everywhere (:101-116)
    i = slice(1, -1)
    assert np.array_equal(Vx[0, i, i], V0[0][-1, i, i])
    assert np.array_equal(Vy[0, i, i], V0[1][-2, i, i]) and np.array_equal(Vy[-1, i, i], V0[1][1, i, i])
    assert np.array_equal(Vy[i, 0, i], V0[1][i, -1, i])
    assert np.array_equal(Vx[i, 0, i], V0[0][i, -2, i]) and np.array_equal(Vx[i, -1, i], V0[0][i, 1, i])
    assert np.array_equal(Vz[i, i, 0], V0[2][i, i, -1])
    assert np.array_equal(Vx[i, i, 0], V0[0][i, i, -2]) and np.array_equal(Vx[i, i, -1], V0[0][i, i, 1])
    Vx, Vy, Vz = apply_flow(new(), faces(F6), faces(F6, "all"), faces(F6))                   # no slip (:124-149)
    assert not Vx[0].any() and not Vx[-1].any() and not Vy[:, 0].any() and not Vy[:, -1].any() and not Vz[:, :, 0].any() and not Vz[:, :, -1].any()
    assert np.array_equal(Vx[:, 0, :], -Vx[:, 1, :]) and np.array_equal(Vx[:, -1, :], -Vx[:, -2, :])
    assert np.array_equal(Vx[:, :, 0], -Vx[:, :, 1]) and np.array_equal(Vx[:, :, -1], -Vx[:, :, -2])
    assert np.array_equal(Vy[0, :, :], -Vy[1, :, :]) and np.array_equal(Vy[-1, :, :], -Vy[-2, :, :])
    assert np.array_equal(Vy[:, :, 0], -Vy[:, :, 1]) and np.array_equal(Vy[:, :, -1], -Vy[:, :, -2])
    assert np.array_equal(Vz[:, 0, :], -Vz[:, 1, :]) and np.array_equal(Vz[:, -1, :], -Vz[:, -2, :])
    assert np.array_equal(Vz[0, :, :], -Vz[1, :, :]) and np.array_equal(Vz[-1, :, :], -Vz[-2, :, :])
