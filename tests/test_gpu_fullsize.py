"""GPU tests at BASELINE.json's full sizes (256^3 and 512^3 per GPU), where the CPU oracle is too slow to be the checker:
size-independent properties of the 3D PT iteration instead.

  1. the three kernel paths (fused PT pipeline, two z-marching sweeps, one-thread-per-node kernels) must leave bit-identical
     states after the same number of iterations from the same random state (the one-thread-per-node kernels are the ones
     the oracle checks directly at small sizes);
  2. homogeneity: the visco-elastic iteration is linear in (V, τ, τ_o, P, P0, Q, ρg); scaling that state by a power of two is
     exact in fp64, so the result must scale by exactly the same factor, bit for bit.
Everything stays in device memory (48 GB of fields at 512^3)."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


def _build(jr, n, seed=1234):
    import math
    import torch
    import justrelax_jl_amd.grid as g
    from justrelax_jl_amd.arrays import PTStokesCoeffs, VelocityBoundaryConditions
    dev = torch.device("cuda", torch.cuda.current_device())
    ni = (n, n, n)
    g.init_global_grid(*ni)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    li = (1.0, 1.3, 0.9)
    di = tuple(l / m for l, m in zip(li, ni))
    grid = g.Geometry(ni, li, origin=(0.0, 0.0, 0.0))
    pt = PTStokesCoeffs(li, di)
    st = jr.StokesArrays(jr.AMDGPUBackend, ni)

    def fill(t, lo=-1.0, hi=1.0):
        # tensors are Fortran-ordered views; fill the underlying storage
        flat = torch.empty(t.numel(), device=dev, dtype=torch.float64)
        flat.uniform_(lo, hi, generator=gen)
        t.copy_(flat.view(*reversed(t.shape)).permute(*range(t.dim() - 1, -1, -1)))
        del flat
    for t in (st.P, st.P0, st.V.Vx, st.V.Vy, st.V.Vz):
        fill(t)
    fill(st.Q, -0.1, 0.1)
    for T in (st.τ, st.τ_o):
        for c in ("xx", "yy", "zz", "yz", "xz", "xy"):
            fill(getattr(T, c))
    fill(st.viscosity.η, -3.0, 0.0)
    st.viscosity.η.copy_(10.0 ** st.viscosity.η)
    K, G = jr.fzeros(ni, dev), jr.fzeros(ni, dev)
    fill(K, 2.0, 3.0); fill(G, 1.0, 1.5)
    ρg = tuple(jr.fzeros(ni, dev) for _ in range(3))
    for t in ρg:
        fill(t)
    faces = ("left", "right", "front", "back", "top", "bot")
    bcs = VelocityBoundaryConditions(free_slip={f: True for f in faces}, no_slip={f: False for f in faces})
    ητ = jr.fzeros(ni, dev)
    jr.compute_maxloc_(ητ, st.viscosity.η)
    jr.flow_bcs_(st, bcs)
    return st, ρg, K, G, pt, grid, bcs, ητ


STATE = ("P", "τ.xx", "τ.yy", "τ.zz", "τ.yz", "τ.xz", "τ.xy", "V.Vx", "V.Vy", "V.Vz")


def _get(o, path):
    for p in path.split("."):
        o = getattr(o, p)
    return o


def _observable(name, t):
    """views that every kernel path must agree on: all of P and τ; for V the cells that are ghost in at most one direction"""
    if not name.startswith("V."):
        return [t]
    tang = {"V.Vx": (1, 2), "V.Vy": (0, 2), "V.Vz": (0, 1)}[name]
    inner = [slice(None)] * 3
    for d in tang:
        inner[d] = slice(1, -1)
    out = [t[tuple(inner)]]
    for d in tang:
        for e in (0, -1):
            idx = list(inner)
            idx[d] = e
            out.append(t[tuple(idx)])
    return out


@pytest.mark.parametrize("n", [256, 512])
def test_full_size_kernel_paths_agree_and_iteration_is_homogeneous(jr, n):
    import torch
    from justrelax_jl_amd import _lib, stokes
    st, ρg, K, G, pt, grid, bcs, ητ = _build(jr, n)
    dt, iters = 0.25, 4
    h = _lib.default_handle()
    init = {k: _get(st, k).clone() for k in STATE}

    def run(variant, scale=1.0):
        for k in STATE:
            _get(st, k).copy_(init[k])
        if scale != 1.0:
            for t in [_get(st, k) for k in STATE] + [st.P0, st.Q] + [getattr(st.τ_o, c) for c in ("xx", "yy", "zz", "yz", "xz", "xy")] + list(ρg):
                t.mul_(scale)
        h.call("jrx_set_option", C.c_char_p(b"kernel_variant"), C.c_int64(variant))
        times = stokes.iterate_timed_(st, pt, grid, bcs, ρg, K, G, ητ, dt, iters)
        return {k: _get(st, k).clone() for k in STATE}, times

    try:
        fused, t3 = run(3)
        assert t3[3] > 0, "the fused PT pipeline did not run"
        for variant in (2, 1):
            other, _ = run(variant)
            for k in STATE:
                for a, b in zip(_observable(k, fused[k]), _observable(k, other[k])):
                    assert torch.equal(a, b), (n, variant, k)
            del other
        # homogeneity under an exact (power-of-two) scaling of the linear state
        scaled, _ = run(0, scale=4.0)
        for k in STATE:
            for a, b in zip(_observable(k, fused[k]), _observable(k, scaled[k])):
                assert torch.equal(a * 4.0, b), (n, "homogeneity", k)
    finally:
        h.call("jrx_set_option", C.c_char_p(b"kernel_variant"), C.c_int64(0))
    # and the state moved: the iteration did something
    assert not torch.equal(fused["V.Vx"], init["V.Vx"])
