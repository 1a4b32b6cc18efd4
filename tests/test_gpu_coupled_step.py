"""GPU: two thermo-mechanical time steps of the thermal-convection script of test/test_WENO5.jl:216-282 chained through the operator API exactly as the
script chains them -- solve! (single-phase non-linear driver, update_ρg! and the viscosity relaxation inside) -> compute_dt -> compute_shear_heating! ->
heatdiffusion_PT! (rheology form) -> center2vertex! / velocity2vertex! / vertex2center! (the grid side of the WENO advection, which itself is particles /
advection and out of scope) -> VTK output -- against the same chain on the CPU oracle.  What one operator writes is what the next one reads, on the device."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cp(a):
    return {k: (v.copy(order="F") if isinstance(v, np.ndarray) else v) for k, v in a.items()}


def test_two_coupled_time_steps_match_the_oracle_chain(jr, oracle, tmp_path):
    import torch
    from justrelax_jl_amd.arrays import from_numpy
    from justrelax_jl_amd.checks import max_rel_diff
    from test_gpu_vep2d import VEP_MAP, _get
    from test_gpu_vep_extras import _nl_params
    dev = torch.device("cuda", torch.cuda.current_device())
    up, dn = (lambda a: from_numpy(a, dev)), jr.to_numpy
    s = jr.miniapps.thermal_convection2d(32, ar=1, iterMax=199, nout=100)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-300
    nx, ny = s.ni
    di, li = s.extra["di"], s.extra["li"]
    ph = dict(s.extra["rheology"], shear_heat=1.0)                                         # Stokes table + ConstantShearheating(1.0)
    trh = dict(k=3.0, Cp=1.2e3, rho0=3.1e3, alpha=1.5e-5, T0=0.0)                          # the same material for the heat solver
    tbc = s.extra["thermal_bc"]
    κ = 3.0 / (1.2e3 * 3.1e3)
    dt_diff = 0.5 * min(di) ** 2 / κ / 2.01
    # ---------------- device state
    st = jr.StokesArrays(jr.AMDGPUBackend, s.ni)
    for k, path in VEP_MAP.items():
        _get(st, path).copy_(up(s.arrays[k]))
    ρg = (up(s.arrays["fx"]), up(s.arrays["fy"]))
    thermal = jr.ThermalArrays(jr.AMDGPUBackend, s.ni)
    thermal.T.copy_(up(s.arrays["T"]))
    Kd, ρCpd = jr.fzeros(s.ni, dev, 3.0), jr.fzeros(s.ni, dev, 1.2e3 * 3.1e3)
    ptt = jr.PTThermalCoeffs(jr.AMDGPUBackend, Kd, ρCpd, s.dt, di, li, CFL=1.0e-3 / np.sqrt(2.1), ϵ=1e-300)
    # ---------------- oracle state
    ref = _cp(s.arrays)
    rh = oracle.rheology_struct([ph])
    from justrelax_jl_amd.miniapps.thermal2d import thermal_shapes2d
    th = {k: np.zeros(shp, order="F") for k, shp in thermal_shapes2d(nx, ny).items()}
    th["T"] = ref["T"]
    th["K"][...], th["rhoCp"][...], th["thetar_dtau"], th["dtau_rho"] = 3.0, 1.2e3 * 3.1e3, dn(ptt.θr_dτ), dn(ptt.dτ_ρ)
    dt_g = dt_o = s.dt
    for step in range(2):
        # Stokes: solve!(stokes, pt_stokes, grid, flow_bcs, ρg, rheology, (; T = thermal.T, P = stokes.P), dt, igg)
        r = jr.solve_(st, s.pt, s.grid, s.flow_bcs, ρg, ph, dict(T=thermal.T, P=st.P), dt_g, None, kwargs=s.kwargs)
        s.dt = dt_o
        r_ref = oracle.stokes2d_nonlinear_solve(ref, rh, _nl_params(oracle, s))
        assert r.iter == r_ref["iter"] == 200
        assert np.allclose(r.err_evo1, r_ref["err_evo1"], rtol=1e-7)
        # dt = compute_dt(stokes, di, dt_diff)
        dt_g = jr.compute_dt_(st, di, dt_diff)
        dt_o = min(dt_diff, 0.9 * min(di[0] / np.abs(ref["Vx"]).max(), di[1] / np.abs(ref["Vy"]).max()))
        assert dt_g == pytest.approx(dt_o, rel=1e-7)
        # compute_shear_heating!(thermal, stokes, rheology, dt)
        jr.compute_shear_heating_(thermal, st, ph, dt_g)
        th["shear_heating"][...] = oracle.compute_shear_heating([ref["txx"], ref["tyy"], ref["txy_c"]], [ref["toxx"], ref["toyy"], ref["toxy_c"]],
                                                                [ref["exx"], ref["eyy"], ref["exy"]], rh, [1.0], dt_o)
        assert max_rel_diff(dn(thermal.shear_heating), th["shear_heating"]) <= 1e-6 and th["shear_heating"].max() > 0
        # heatdiffusion_PT!(thermal, pt_thermal, thermal_bc, rheology, args, dt, grid)
        rt = jr.heatdiffusion_PT_(thermal, ptt, tbc, trh, None, dt_g, s.grid, kwargs=dict(iterMax=300, nout=100, verbose=False))
        p = oracle.thermal_params2d(s.ni, s.grid._di["center"], dt_o, 1e-300, iterMax=300, nout=100, no_flux=tbc.no_flux, constant_value=tbc.constant_value,
                                    constant_flux=tbc.constant_flux, periodic=tbc.periodic, rheology=trh)
        rt_ref = oracle.heatdiffusion_PT2d(th, p)
        assert list(rt.iter_count) == list(rt_ref["iter_count"])
        assert max_rel_diff(dn(thermal.T), th["T"]) <= 1e-9
        ref["T"] = th["T"]
        # grid side of the advection step: center2vertex!(T_WENO, T[2:end-1, 2:end-1]); velocity2vertex!(Vx_v, Vy_v, @velocity(stokes)...)
        Tin = thermal.T[1:-1, 1:-1].permute(1, 0).contiguous().permute(1, 0)              # the copy Julia's non-view slice makes
        T_v, Vx_v, Vy_v = (jr.fzeros((nx + 1, ny + 1), dev) for _ in range(3))
        jr.center2vertex_(T_v, Tin)
        jr.velocity2vertex_(Vx_v, Vy_v, st.V.Vx, st.V.Vy)
        o_vx, o_vy = oracle.velocity2vertex(ref["Vx"], ref["Vy"])
        assert max_rel_diff(dn(Vx_v), o_vx) <= 1e-7 and max_rel_diff(dn(Vy_v), o_vy) <= 1e-7
        Tc = th["T"][1:-1, 1:-1]
        inner = 0.25 * (Tc[:-1, :-1] + Tc[1:, :-1] + Tc[:-1, 1:] + Tc[1:, 1:])                 # @inn(vertex) = @av(center), Interpolations.jl:111-114
        assert np.allclose(dn(T_v)[1:-1, 1:-1], inner, rtol=1e-12)
        back = jr.fzeros(s.ni, dev)
        jr.vertex2center_(back, T_v)
        tv = dn(T_v)
        assert np.array_equal(dn(back), 0.25 * (tv[:-1, :-1] + tv[1:, :-1] + tv[:-1, 1:] + tv[1:, 1:]))
    out = {k: dn(_get(st, path)) for k, path in VEP_MAP.items()}
    for k in ("P", "Vx", "Vy", "txx", "tyy", "txy", "eta", "eta_vep", "exx", "exy", "tII"):
        assert max_rel_diff(out[k], ref[k]) <= 1e-6, k
    # the step's output file: save_vtk of the centre fields and the vertex velocities
    f = tmp_path / "step2"
    jr.save_vtk(str(f), s.grid.xvi, s.grid.xci, dict(T_v=dn(T_v)), dict(T=dn(thermal.T)[1:-1, 1:-1], eta=out["eta"]), (dn(Vx_v), dn(Vy_v)), t=dt_g)
    assert f.with_suffix(".vtr").stat().st_size > 10000
