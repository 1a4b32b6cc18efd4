"""Array(stokes) / PTArray(backend, stokes) / copy(stokes) conversions (src/types/type_conversions.jl:17-68) and the checkpoint / restart functions built on
them (src/IO/JLD2.jl:37-149), here with an .npz container.  The CPU test uses host containers; the GPU test round-trips device containers."""
import numpy as np
import pytest


def _fill(jr, backend, ni, seed):
    import torch
    st = jr.StokesArrays(backend, ni)
    th = jr.ThermalArrays(backend, ni)
    g = torch.Generator().manual_seed(seed)
    for t in (st.P, st.V.Vx, st.V.Vy, st.τ.xx, st.τ.xy, st.ε.xy, st.viscosity.η, st.R.Rx, st.EII_pl, th.T, th.qTx, th.ResT):
        t.copy_(torch.rand(tuple(reversed(t.shape)), generator=g, dtype=torch.float64).permute(*range(t.dim() - 1, -1, -1)))
    return st, th


def _same(jr, a, b):
    ha, hb = jr.Array_(a), jr.Array_(b)

    def walk(x, y, path):
        for k, v in vars(x).items():
            if k.startswith("_"):
                continue
            if isinstance(v, np.ndarray):
                assert np.array_equal(v, getattr(y, k)), path + k
                assert v.flags.f_contiguous
            else:
                walk(v, getattr(y, k), path + k + "/")
    walk(ha, hb, "")


@pytest.mark.parametrize("ni", [(9, 7), (6, 5, 4)])
def test_checkpoint_roundtrip_host(jr, tmp_path, ni):
    st, th = _fill(jr, jr.CPUBackend, ni, 3)
    igg = type("I", (), {"me": 3})()
    jr.checkpointing_npz(str(tmp_path), st, th, 1.25, 0.5, igg, extra_field=st.P, pair=(st.V.Vx, st.V.Vy))
    assert (tmp_path / "checkpoint0003.npz").exists()                       # checkpoint_name(dst, igg) -- JLD2.jl:38
    hs, ht, t, dt = jr.load_checkpoint_npz(str(tmp_path), igg)
    assert (t, dt) == (1.25, 0.5) and hs._ni == tuple(ni) and ht._ni == tuple(ni)
    assert np.array_equal(hs.τ.xy, jr.to_numpy(st.τ.xy)) and np.array_equal(ht.qTx, jr.to_numpy(th.qTx))
    st2, th2 = jr.PTArray_(jr.CPUBackend, hs), jr.PTArray_(jr.CPUBackend, ht)
    assert isinstance(st2, jr.StokesArrays) and isinstance(th2, jr.ThermalArrays)
    _same(jr, st, st2)
    _same(jr, th, th2)
    # without thermal, without igg (JLD2.jl:52-56); a second write replaces the first
    jr.checkpointing_npz(str(tmp_path), st, None, 2.0, 0.25)
    jr.checkpointing_npz(str(tmp_path), st, None, 3.0, 0.25)
    hs3, none, t3, _ = jr.load_checkpoint_npz(str(tmp_path))
    assert none is None and t3 == 3.0
    # copy(stokes) is independent of its source
    c = jr.copy_(st)
    _same(jr, st, c)
    c.P.fill_(0.0)
    assert float(st.P.abs().max()) > 0


@pytest.mark.gpu
def test_checkpoint_roundtrip_device(jr, tmp_path):
    import torch
    ni = (12, 9, 7)
    st_h, th_h = _fill(jr, jr.CPUBackend, ni, 5)
    st = jr.PTArray_(jr.AMDGPUBackend, jr.Array_(st_h))            # host -> device
    th = jr.PTArray_(jr.AMDGPUBackend, jr.Array_(th_h))
    assert st.P.is_cuda and th.T.is_cuda
    jr.checkpointing_npz(str(tmp_path), st, th, 7.0, 0.1)
    hs, ht, t, dt = jr.load_checkpoint_npz(str(tmp_path))
    _same(jr, st_h, jr.PTArray_(jr.CPUBackend, hs))
    _same(jr, th_h, jr.PTArray_(jr.CPUBackend, ht))
    st2 = jr.PTArray_(jr.AMDGPUBackend, hs)
    assert torch.equal(st2.τ.xy, st.τ.xy) and st2.τ.xy.stride() == st.τ.xy.stride()
