"""The Julia side of the drop-in boundary (ext/JustRelaxHIPNativeExt.jl) cannot be executed here (no Julia in the build image,
SURVEY F2), so it is machine-checked against include/jrx.h instead:
  * every C struct has a Julia mirror with the same fields in the same order, the same element types and array lengths -- a silent
    field-order drift would corrupt every pointer passed through `Ref{Jrx...}`;
  * every `ccall` names an exported function and passes the argument types of its C prototype;
  * the extension defines the complete backend method table of the reference (src/ext/AMDGPU/2D.jl, 3D.jl, ext/JustRelaxAMDGPUExt.jl)."""
import re
import subprocess
import sys

import pytest

from _abi_parse import C2J, JULIA_EXT, ROOT, c_prototypes, c_structs, julia_ccalls, julia_name, julia_structs, same_arg


def test_every_c_struct_has_an_identical_julia_mirror():
    cs, js = c_structs(), julia_structs()
    assert len(cs) >= 15
    for cname, cfields in cs.items():
        jname = julia_name(cname)
        assert jname in js, f"{cname} has no Julia mirror {jname}"
        jfields = js[jname]
        assert [f[0] for f in jfields] == [f[0] for f in cfields], f"{cname}: field names / order differ"
        for (n, ctype, cptr, ccount), (_, jtype, jptr, jcount) in zip(cfields, jfields):
            assert C2J[ctype] == jtype and cptr == jptr and ccount == jcount, (cname, n, (ctype, cptr, ccount), (jtype, jptr, jcount))
    extra = set(js) - {julia_name(c) for c in cs}
    assert not extra, f"Julia structs without a C counterpart: {extra}"


def test_struct_sizes_agree_with_the_ctypes_binding(jr):
    """third view of the same layouts: the ctypes structs the GPU tests run through (natural alignment on both sides)"""
    import ctypes as C
    from justrelax_jl_amd import _lib
    size = {"Float64": 8, "Int64": 8, "Int32": 4, "UInt32": 4, "UInt8": 1}

    def jl_sizeof(fields):
        off, amax = 0, 1
        for _, ty, ptr, count in fields:
            a = 8 if ptr else size[ty]
            off = (off + a - 1) // a * a + a * count
            amax = max(amax, a)
        return (off + amax - 1) // amax * amax
    js = julia_structs()
    pairs = dict(JrxStokes3dFields=_lib.Stokes3DFields, JrxStokes3dParams=_lib.Stokes3DParams, JrxStokes2dFields=_lib.Stokes2DFields,
                 JrxStokes2dParams=_lib.Stokes2DParams, JrxRheology=_lib.Rheology, JrxVep2dFields=_lib.VEP2DFields, JrxVep2dParams=_lib.VEP2DParams,
                 JrxVep3dFields=_lib.VEP3DFields, JrxVep3dParams=_lib.VEP3DParams, JrxThermal2dFields=_lib.Thermal2DFields,
                 JrxThermal2dParams=_lib.Thermal2DParams, JrxThermal3dFields=_lib.Thermal3DFields, JrxThermal3dParams=_lib.Thermal3DParams,
                 JrxSolveResult=_lib.SolveResult, JrxCart=_lib.Cart)
    for jname, ct in pairs.items():
        assert jl_sizeof(js[jname]) == C.sizeof(ct), jname
        assert [f[0] for f in js[jname]] == [f[0] for f in ct._fields_], jname


def test_every_ccall_matches_its_c_prototype():
    protos, calls = c_prototypes(), julia_ccalls()
    assert len(calls) >= 30
    for fn, sites in calls.items():
        assert fn in protos, f"ccall of {fn}, which include/jrx.h does not declare"
        for ret, args in sites:
            want = protos[fn]
            assert len(args) == len(want), (fn, args, want)
            for w, g in zip(want, args):
                assert same_arg(w, g), (fn, "expected", w, "got", g)
            assert ret in ("Cint", "Cstring", "Int64", "Int32"), (fn, ret)


def test_the_extension_covers_the_reference_method_table():
    txt = JULIA_EXT.read_text()
    calls = julia_ccalls()
    needed_entry_points = ["jrx_create", "jrx_destroy", "jrx_last_error", "jrx_set_option", "jrx_cart_create", "jrx_comm_unique_id", "jrx_comm_init",
                           "jrx_update_halo", "jrx_stokes3d_solve", "jrx_stokes2d_solve", "jrx_stokes2d_vep_solve", "jrx_stokes2d_nonlinear_solve",
                           "jrx_stokes3d_vep_solve", "jrx_heatdiffusion_PT2d", "jrx_heatdiffusion_PT3d", "jrx_thermal_bcs2d", "jrx_thermal_bcs3d",
                           "jrx_flow_bcs2d", "jrx_flow_bcs3d", "jrx_velocity2displacement", "jrx_displacement2velocity", "jrx_compute_dt",
                           "jrx_tensor_invariant2d", "jrx_tensor_invariant3d", "jrx_shear2center2d", "jrx_shear2center3d", "jrx_accumulate_tensor2d",
                           "jrx_accumulate_tensor3d", "jrx_accumulate_vol", "jrx_compute_maxloc", "jrx_center2vertex2d", "jrx_compute_vorticity2d",
                           "jrx_compute_vorticity3d", "jrx_velocity2vertex2d", "jrx_velocity2vertex3d", "jrx_velocity2center2d", "jrx_velocity2center3d",
                           "jrx_vertex2center", "jrx_center2vertex3d", "jrx_center2vertex_harm2d", "jrx_compute_rhog", "jrx_compute_shear_heating", "jrx_compute_viscosity_single",
                           "jrx_vep2d_compute_viscosity", "jrx_vep3d_compute_viscosity", "jrx_compute_lithostatic_pressure"]
    missing = [f for f in needed_entry_points if f not in calls]
    assert not missing, missing
    # the generics the reference's AMDGPU extension adds methods to (src/ext/AMDGPU/2D.jl:48-403, 3D.jl:46-412, ext/JustRelaxAMDGPUExt.jl:5-10)
    for pat in (r"PTArray\(::Type\{AMDGPUBackend\}\)\s*=\s*ROCArray", r"backend\(::ROCArray\)\s*=\s*AMDGPUBackendTrait\(\)",
                r"JR2D\.StokesArrays\(::Type\{AMDGPUBackend\}", r"JR3D\.StokesArrays\(::Type\{AMDGPUBackend\}", r"JR2D\.ThermalArrays\(::Type\{AMDGPUBackend\}",
                r"JR3D\.ThermalArrays\(::Type\{AMDGPUBackend\}", r"JR2D\.PTThermalCoeffs\(::Type\{AMDGPUBackend\}", r"JR3D\.PTThermalCoeffs\(::Type\{AMDGPUBackend\}", r"\$JR\.PTThermalCoeffs\(::Type\{AMDGPUBackend\}, rheology, phase_ratios",
                r"JR2D\.heatdiffusion_PT!\(::Trait", r"JR3D\.heatdiffusion_PT!\(::Trait", r"JR2D\.thermal_bcs!\(::Trait", r"JR3D\.thermal_bcs!\(::Trait",
                r"JR2D\.center2vertex!", r"JR3D\.center2vertex!", r"JR2D\.velocity2vertex!", r"JR3D\.velocity2vertex!", r"JR2D\.velocity2center!",
                r"JR3D\.velocity2center!", r"JR2D\.vertex2center!", r"JR3D\.vertex2center!", r"\$JR\.compute_ρg!", r"\$JR\.compute_shear_heating!\(::Trait", r"\$JR\.compute_viscosity!\(::Trait"):
        assert re.search(pat, txt), pat
    assert len(re.findall(r"function JR2D\.solve!\(::Trait", txt)) == 3          # G, K | phase_ratios | MaterialParams
    assert len(re.findall(r"function JR3D\.solve!\(::Trait", txt)) == 2          # K, G | phase_ratios
    assert len(re.findall(r"heatdiffusion_PT!\(::Trait", txt)) == 4              # 2D/3D x array / rheology form
    for generic in ("flow_bcs!", "velocity2displacement!", "displacement2velocity!", "compute_dt", "tensor_invariant!", "shear2center!",
                    "accumulate_tensor!", "accumulate_vol!", "compute_maxloc!"):
        assert re.search(r"\$JR\." + re.escape(generic), txt), generic          # defined for JR2D and JR3D by the @eval loop
    assert "for JR in (:JR2D, :JR3D)" in txt


def test_generated_struct_block_is_in_sync_with_the_header():
    txt = JULIA_EXT.read_text()
    block = txt.split("# GENERATED by scripts/gen_julia_structs.py from include/jrx.h -- do not edit by hand -- BEGIN\n")[1].split("# GENERATED -- END")[0]
    gen = subprocess.run([sys.executable, str(ROOT / "scripts" / "gen_julia_structs.py")], capture_output=True, text=True, check=True).stdout
    assert block == gen, "run scripts/gen_julia_structs.py and paste its output between the GENERATED markers"
