"""CPU-only tests: the C-ABI library loads and exports every symbol include/jrx.h declares, the
host-side block-decomposition logic, and the mirror of the reference's traits / BC structs."""
import ctypes as C

import numpy as np
import pytest


def test_library_exports_every_declared_symbol(jr):
    from justrelax_jl_amd import _lib
    syms = _lib.declared_symbols()
    assert len(syms) >= 28 and "jrx_stokes3d_solve" in syms and "jrx_heatdiffusion_PT2d" in syms
    L = _lib.load(check_symbols=True)
    for s in syms:
        assert hasattr(L, s), s
    L.jrx_version.restype = C.c_int32
    assert L.jrx_version() == 230


def test_library_carries_the_build_id_of_its_sources(jr, tmp_path):
    """jrx_build_id() = sha256 over csrc/ + include/jrx.h; the binding refuses a binary built from other sources, and the builder reads the
    id of an existing .so from the file (no load) to decide whether to rebuild."""
    from justrelax_jl_amd import _lib, build
    L = _lib.load()
    L.jrx_build_id.restype = C.c_char_p
    sid = build.source_id()
    assert len(sid) == 64 and L.jrx_build_id().decode() == sid == build.binary_id()
    assert not build.needs_build()
    stale = tmp_path / "libstale.so"
    data = _lib.LIB_PATH.read_bytes()
    i = data.find(build.MARKER) + len(build.MARKER)
    stale.write_bytes(data[:i] + b"0" * 64 + data[i + 64:])           # same binary, another id
    assert build.binary_id(stale) == "0" * 64 != sid
    old_lib, old_path = _lib._lib, _lib.LIB_PATH
    try:
        _lib._lib, _lib.LIB_PATH = None, stale
        with pytest.raises(RuntimeError, match="built from other sources"):
            _lib.load()
    finally:
        _lib._lib, _lib.LIB_PATH = old_lib, old_path


def test_no_gpu_means_loud_failure(jr):
    """Without a HIP device the product path raises; it never computes on the CPU."""
    import torch
    from justrelax_jl_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.JrxError):
        _lib.Handle(0)
    with pytest.raises(RuntimeError):
        jr.StokesArrays(jr.AMDGPUBackend, (4, 4, 4))


def test_halo_planes_follow_igg_semantics(jr):
    """ol_A = 2 + (nA - n): send plane ol_A / nA-ol_A+1 (1-based), receive into 1 / nA."""
    from justrelax_jl_amd import _lib
    L = _lib.load()
    n = 10
    want = {n: (1, 8, 0, 9), n + 1: (2, 8, 0, 10), n + 2: (3, 8, 0, 11)}     # 0-based (send_l, send_r, recv_l, recv_r)
    for nA, w in want.items():
        v = [C.c_int64() for _ in range(4)]
        assert L.jrx_halo_planes(C.c_int64(n), C.c_int64(nA), *[C.byref(x) for x in v]) == 0
        assert tuple(x.value for x in v) == w
    assert L.jrx_halo_planes(C.c_int64(n), C.c_int64(n - 1), None, None, None, None) != 0      # Rx-like arrays are not exchangeable
    assert L.jrx_n_global(512, 2, 0) == 1022 and L.jrx_n_global(512, 1, 0) == 512 and L.jrx_n_global(1, 4, 0) == 1


@pytest.mark.parametrize("nprocs,dims", [(1, (1, 1, 1)), (2, (2, 1, 1)), (4, (2, 2, 1)), (8, (2, 2, 2)), (6, (3, 2, 1)), (12, (3, 2, 2))])
def test_cart_create_matches_python_grid(jr, nprocs, dims):
    from justrelax_jl_amd import _lib, grid, halo
    for rank in range(nprocs):
        grid.init_global_grid(16, 16, 16, rank=rank, nprocs=nprocs)
        gg = grid.global_grid()
        assert tuple(gg.dims) == dims
        cart = _lib.Cart()
        n = (C.c_int64 * 3)(16, 16, 16)
        assert _lib.load().jrx_cart_create(rank, nprocs, n, None, None, C.byref(cart)) == 0      # library's own factorisation
        assert tuple(cart.dims) == dims and tuple(cart.coords) == tuple(gg.coords)
        nb = grid.neighbors(gg.coords, gg.dims)
        assert [tuple(x) for x in cart.neighbor] == nb
        for d in range(3):                                   # neighbour relation is symmetric
            for side in (0, 1):
                o = cart.neighbor[d][side]
                if o >= 0:
                    oc = grid.cart_coords(o, dims)
                    assert grid.neighbors(oc, dims)[d][1 - side] == rank
    grid.init_global_grid(16, 16, 1, rank=0, nprocs=4)
    assert tuple(grid.global_grid().dims) == (2, 2, 1)       # 2D problems never split z
    grid.finalize_global_grid()


def test_global_grid_coordinates(jr):
    """nx_g = dims*(nx-2)+2 and x_g offsets of src/grid/Utils.jl:24-40; Geometry of src/grid/Grid.jl:56-143"""
    from justrelax_jl_amd import grid
    grid.init_global_grid(10, 9, 8, rank=1, nprocs=2)
    assert (grid.nx_g(), grid.ny_g(), grid.nz_g()) == (18, 9, 8)
    g = jr.Geometry((10, 9, 8), (18.0, 9.0, 8.0))
    assert g.di["center"] == (1.0, 1.0, 1.0)
    assert g.xvi[0][0] == 8.0 and g.xvi[0][-1] == 18.0 and g.xci[0][0] == 8.5      # rank 1 starts at cell 8 = 1*(10-2)
    grid.finalize_global_grid()
    g = jr.Geometry((4, 4), (2.0, 1.0), origin=(0.0, -1.0))
    assert np.allclose(g.xci[1], [-0.875, -0.625, -0.375, -0.125]) and g._di["center"] == (2.0, 4.0)


def test_traits_and_backends(jr):
    """test/test_traits.jl:49-95"""
    import torch
    st = jr.StokesArrays(jr.CPUBackend, (4, 4))
    assert isinstance(jr.backend(st), jr.CPUBackendTrait) and isinstance(jr.backend(st.P), jr.CPUBackendTrait)
    assert isinstance(jr.backend(jr.ThermalArrays(jr.CPUBackend, (4, 4))), jr.CPUBackendTrait)
    assert issubclass(jr.AMDGPUBackendTrait, jr.GPUBackendTrait) and issubclass(jr.GPUBackendTrait, jr.BackendTrait)
    with pytest.raises(ValueError):
        jr.backend(3.0)
    with pytest.raises(ValueError):
        jr.PTArray(int)
    a = jr.PTArray(jr.CPUBackend)(np.ones((2, 3)))
    assert isinstance(a, torch.Tensor) and a.dtype == torch.float64


def test_boundary_condition_structs(jr):
    """src/boundaryconditions/types.jl:108-195 ; test_boundary_conditions2D/3D.jl error cases"""
    f6 = ("left", "right", "front", "back", "top", "bot")
    on, off = {k: True for k in f6}, {k: False for k in f6}
    with pytest.raises(ValueError, match="Incompatible"):
        jr.VelocityBoundaryConditions(no_slip=on, free_slip=dict(off, right=True))
    b = jr.VelocityBoundaryConditions(no_slip=off, free_slip=off)
    assert b.nD == 3 and not any(b.free_slip.values())
    with pytest.raises(ValueError, match="Periodic boundary conditions must be paired"):
        jr.VelocityBoundaryConditions(no_slip=off, free_slip=off, periodic=dict(off, front=True))
    b2 = jr.VelocityBoundaryConditions()
    assert b2.nD == 2 and all(b2.free_slip.values())
    with pytest.raises(ValueError):
        jr.VelocityBoundaryConditions(free_slip=dict(left=False, right=False, top=False, bot=False),
                                      periodic=dict(left=False, right=False, top=True, bot=True), free_surface=True)
    t = jr.TemperatureBoundaryConditions(no_flux=dict(left=True, right=True, top=False, bot=False),
                                         constant_value=dict(left=True, right=True, top=300.0, bot=3500.0))
    assert t.nD == 2 and t.constant_value["top"] == 300.0 and t.constant_flux["left"] is False
    with pytest.raises(ValueError):
        jr.TemperatureBoundaryConditions(no_flux=dict(left=False, right=False, top=False, bot=False),
                                         periodic=dict(left=False, right=False, front=True, back=False, top=False, bot=False))
    with pytest.raises(ValueError):
        jr.TemperatureBoundaryConditions(no_flux=dict(left=True, right=True, top=False, bot=False),
                                         periodic=dict(left=True, right=True, top=False, bot=False))


def test_column_major_layout(jr):
    import torch
    from justrelax_jl_amd.arrays import is_fortran
    t = jr.fzeros((3, 4, 5), torch.device("cpu"))
    assert tuple(t.shape) == (3, 4, 5) and t.stride() == (1, 3, 12) and is_fortran(t)
    a = np.arange(60, dtype=np.float64).reshape(3, 4, 5, order="F")
    t = jr.from_numpy(a, torch.device("cpu"))
    assert is_fortran(t) and float(t[1, 2, 3]) == a[1, 2, 3]
    assert np.array_equal(jr.to_numpy(t), a) and jr.to_numpy(t).flags.f_contiguous


def test_device_builder_matches_host_builder_on_cpu(jr):
    """solvi3d_device (torch ops, used by bench.py at 512^3) == solvi3d (numpy) on the same grid."""
    from justrelax_jl_amd.miniapps.stokes3d import solvi3d_device
    import justrelax_jl_amd.grid as g
    g.finalize_global_grid()
    st, ρg, K, G, pt, grid, bcs, dt = solvi3d_device(12, jr.CPUBackend)
    g.finalize_global_grid()
    s = jr.miniapps.solvi3d(12)
    assert np.allclose(jr.to_numpy(st.viscosity.η), s.arrays["eta"], rtol=1e-14, atol=0)
    Vx = jr.to_numpy(st.V.Vx)
    assert np.array_equal(Vx[:, 1:-1, 1:-1], s.arrays["Vx"][:, 1:-1, 1:-1])
    assert pt.θ_dτ == s.pt.θ_dτ and dt == s.dt


def test_header_is_c99_and_callable_from_c(tmp_path):
    """The boundary is a C ABI: include/jrx.h compiles as strict C99, and a plain-C program linked against libjrx_hip.so calls the
    host-only entry points (block decomposition, halo planes: ImplicitGlobalGrid's arithmetic) without a GPU."""
    import shutil
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    gcc = shutil.which("gcc")
    assert gcc, "gcc not found"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", str(root / "include" / "jrx.h")], check=True)
    libdir = root / "justrelax.jl_amd" / "lib"
    assert (libdir / "libjrx_hip.so").exists(), "build the library first (__graft_entry__.build())"
    exe = tmp_path / "c_abi_smoke"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-I", str(root / "include"), str(root / "tests" / "c_abi_smoke.c"), "-L", str(libdir),
                    "-ljrx_hip", f"-Wl,-rpath,{libdir}", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines()
    assert out[0] == "dims 2 2 2 coords 1 0 1"            # rank 5 of 2 x 2 x 2 = (1*2 + 0)*2 + 1: MPI_Cart_create's row-major order, as IGG uses
    assert out[1] == "neighbors 1 -1 -1 7 4 -1"           # x-left (0,0,1) = 1, y-right (1,1,1) = 7, z-left (1,0,0) = 4; -1 = physical boundary
    assert out[2] == "halo planes 2 510 0 512"
    assert out[3] == "nx_g 1022"


def test_every_driver_kwarg_reaches_its_parameter_struct(jr):
    """the kwargs the reference's drivers accept (Stokes3D.jl:25-41,447-466; Stokes2D.jl:181-196,345-362,577-599) are forwarded into the
    parameter structs of the C ABI -- by value on the Python host, by name in the Julia extension (INTEGRATION.md, table "kwargs of the reference's drivers")"""
    import re
    from pathlib import Path
    from types import SimpleNamespace
    from justrelax_jl_amd import stokes as st
    pt = jr.PTStokesCoeffs((1.0, 1.0, 1.0), (0.1, 0.1, 0.1))
    grid3 = SimpleNamespace(_di=dict(center=(10.0, 10.0, 10.0)), nonuniform=False)
    grid2 = SimpleNamespace(_di=dict(center=(10.0, 10.0)), nonuniform=False)
    bcs3 = jr.VelocityBoundaryConditions(free_slip={f: True for f in ("left", "right", "front", "back", "top", "bot")})
    bcs2 = jr.VelocityBoundaryConditions(free_slip={f: True for f in ("left", "right", "top", "bot")})
    s3, s2 = SimpleNamespace(_ni=(8, 9, 10)), SimpleNamespace(_ni=(8, 9))
    p = st.params3d(s3, pt, grid3, bcs3, 0.5, iterMax=77, nout=11, b_width=(3, 2, 5), verbose=False, viscosity_relaxation=0.3)
    assert (p.iterMax, p.nout, tuple(p.b_width), p.verbose) == (77, 11, (3, 2, 5), 0)
    p = st.params2d(s2, pt, grid2, bcs2, 0.5, iterMax=78, nout=12, b_width=(4, 4, 1), verbose=True)
    assert (p.iterMax, p.nout, p.verbose) == (78, 12, 1)
    p = st.vep_params2d(s2, pt, grid2, bcs2, 0.5, iterMax=79, iterMin=13, nout=14, verbose=False, λ_relaxation=0.3, viscosity_relaxation=0.05,
                        viscosity_cutoff=(1e18, 1e23), strain_increment=True, free_surface=True, b_width=(4, 4, 0))
    assert (p.iterMax, p.iterMin, p.nout, p.verbose, p.lambda_relaxation, p.viscosity_relaxation, p.cutoff_lo, p.cutoff_hi, p.strain_increment, p.free_surface) == \
        (79, 13, 14, 0, 0.3, 0.05, 1e18, 1e23, 1, 1)
    p = st.vep_params3d(s3, pt, grid3, bcs3, 0.5, iterMax=80, nout=15, verbose=False, λ_relaxation=0.4, viscosity_relaxation=0.06, viscosity_cutoff=(1e17, 1e24),
                        b_width=(2, 3, 6))
    assert (p.iterMax, p.nout, p.verbose, p.lambda_relaxation, p.viscosity_relaxation, p.cutoff_lo, p.cutoff_hi, tuple(p.b_width)) == \
        (80, 15, 0, 0.4, 0.06, 1e17, 1e24, (2, 3, 6))
    # the Julia extension: every solve! method reads the kwargs of its reference signature
    txt = (Path(__file__).resolve().parent.parent / "ext" / "JustRelaxHIPNativeExt.jl").read_text()
    methods = re.split(r"\nfunction (?=JR[23]D\.solve!\(::Trait)", txt)[1:]
    assert len(methods) == 5
    body = {}
    for m in methods:
        head = m.split("\n", 1)[0]
        key = ("3D" if "JR3D" in head else "2D") + ("_phases" if "phase_ratios" in m.split("kwargs)")[0] else ("_material" if "MaterialParams" in m.split("kwargs)")[0] else ""))
        body[key] = m.split("\nend\n")[0]
    helper = txt.split("function vep_params2d(")[1].split("\nend\n")[0]
    for k in ("iterMax", "nout", "b_width", "verbose"):
        assert re.search(rf"kw\.{k}\b", body["3D"]), k
    for k in ("iterMax", "nout", "verbose"):
        assert re.search(rf"kw\.{k}\b", body["2D"]), k
    for k in ("iterMax", "nout", "b_width", "verbose", "λ_relaxation", "viscosity_relaxation", "viscosity_cutoff"):
        assert re.search(rf"kw\.{k}\b", body["3D_phases"]), k
    for k in ("iterMax", "iterMin", "nout", "verbose", "λ_relaxation", "viscosity_relaxation", "viscosity_cutoff", "strain_increment", "free_surface"):
        assert re.search(rf"\b{k}\b", helper), k
    assert "vep_params2d(" in body["2D_phases"] and "kwargs..." in body["2D_phases"]
    assert "vep_params2d(" in body["2D_material"] and "kwargs..." in body["2D_material"]


def test_carts_of_a_2x2x2_process_grid_are_mutually_consistent(jr):
    """halo.make_carts (jrx_cart_create for every rank of the grid the in-process transport and the RCCL transport both run on): neighbour relations are symmetric,
    a 2 x 2 x 2 block has exactly one neighbour per dimension, periodic dimensions wrap, and the plan is the one north_star names (dims = (2, 2, 2) for eight ranks)."""
    from justrelax_jl_amd import halo, _lib
    L = _lib.load()
    carts = halo.make_carts((16, 16, 16), (2, 2, 2))
    assert len(carts) == 8
    for r, c in enumerate(carts):
        assert (c.rank, c.nprocs, tuple(c.dims)) == (r, 8, (2, 2, 2))
        nbs = [(d, s, c.neighbor[d][s]) for d in range(3) for s in range(2) if c.neighbor[d][s] >= 0]
        assert len(nbs) == 3 and {d for d, _, _ in nbs} == {0, 1, 2}
        for d, s, nb in nbs:
            assert carts[nb].neighbor[d][1 - s] == r                       # my right neighbour's left neighbour is me
            co, cn = tuple(c.coords), tuple(carts[nb].coords)
            assert sum(abs(a - b) for a, b in zip(co, cn)) == 1 and cn[d] - co[d] == (1 if s else -1)
    # balanced factorisation of eight ranks (what IGG / MPI_Dims_create gives the reference): (2, 2, 2)
    auto = _lib.Cart()
    import ctypes as C
    assert L.jrx_cart_create(C.c_int32(5), C.c_int32(8), (C.c_int64 * 3)(16, 16, 16), (C.c_int32 * 3)(0, 0, 0), (C.c_int32 * 3)(0, 0, 0), C.byref(auto)) == 0
    assert tuple(auto.dims) == (2, 2, 2) and tuple(auto.coords) == tuple(carts[5].coords)
    per = halo.make_carts((16, 16, 16), (2, 1, 1), periods=(1, 0, 0))
    assert per[0].neighbor[0][0] == 1 and per[0].neighbor[0][1] == 1 and per[1].neighbor[0][0] == 0      # two ranks, periodic: both neighbours are the other rank
    assert per[0].neighbor[1][0] == -1
