"""Worker of tests/test_gpu_ipc_two_processes.py: ONE RANK PER PROCESS through the cross-process copy-engine transport (jrx_comm_init_ipc).

Two of these processes are started by tests/conftest.py at the start of a `-m gpu` session -- before the pytest process itself touches the GPU: a process
that has initialised the GPU may not start children on the GPU boxes -- and share device 0, which RCCL cannot do (it refuses duplicate devices).  This is the
reference's own process model (mpiexec -n 2, test/runtests.jl:73-90) and the one the Julia extension's MPI ranks would use.

Every rank, for every case: builds the same global problem (seeded), runs it UNDECOMPOSED on its own plain handle with the simplest kernels, then runs its block
of the ImplicitGlobalGrid decomposition through solve! on a handle joined to the other process by jrx_comm_init_ipc, and requires its block to equal the
undecomposed run bit for bit (state, residuals, strain rates; update_halo!(V) inside @hide_communication: src/stokes/Stokes3D.jl:104-121, update_halo!(ητ): :57,
norm_mpi with the doubly counted overlap: :127-142).  Results go to $JRX_IPC_OUT/rank<r>.json; the pytest process only reads them.
"""
import ctypes as C
import json
import os
import sys
import time
import traceback
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

CASES = [((2, 1, 1), (70, 13, 12)), ((1, 1, 2), (70, 13, 12)), ((2, 1, 1), (130, 14, 40)), ((1, 1, 2), (130, 14, 40)), ((1, 2, 1), (70, 13, 12))]
PIPES = ["fused", "fused_overlap", "fused_early", "split_sweeps"]
# dt = Inf: the default pipeline of the viscous limit -- the kernel finishes the faces with a neighbour itself behind a device-side flag (fused_overlap = 3)
VISC_CASES = [((2, 1, 1), (70, 13, 12)), ((1, 1, 2), (130, 14, 40)), ((1, 2, 1), (70, 40, 12))]


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out_dir = Path(os.environ["JRX_IPC_OUT"])
    res = {"rank": rank, "cases": [], "error": None, "t_start": time.time()}

    def flush():
        tmp = out_dir / f"rank{rank}.json.tmp"
        tmp.write_text(json.dumps(res))
        tmp.replace(out_dir / f"rank{rank}.json")

    try:
        import numpy as np
        import torch
        if not torch.cuda.is_available():
            res["skipped"] = "no GPU"
            flush()
            return 0
        torch.cuda.set_device(0)
        from __graft_entry__ import load_package
        jr = load_package()
        import _blocks as B
        import test_gpu_two_blocks as T
        import justrelax_jl_amd.grid as g
        from justrelax_jl_amd import _lib, halo
        from justrelax_jl_amd.checks import interior_mask3d
        from justrelax_jl_amd.miniapps.common import Setup, download_stokes, upload_stokes
        h0 = _lib.default_handle()
        hc = _lib.Handle(0)                       # the handle that joins the group
        hc.set_option("comm_timeout_ms", 60000)

        # the undecomposed runs first, while this process is still alone: init_global_grid takes rank / size from torch.distributed once that is
        # initialised (as ImplicitGlobalGrid takes them from MPI), and the undecomposed problem is a one-rank problem
        kw = dict(iterMax=23, nout=8, verbose=False)
        undecomposed = {}
        for dims, n in CASES:
            ng = B.n_global(n, dims)
            if ng in undecomposed:
                continue
            g.finalize_global_grid()
            S = T._global_setup(jr, ng, True, 23, 8)
            T._set(h0, kernel_variant=1)
            stokes, ρg, K, G = upload_stokes(S, jr.AMDGPUBackend)
            rg = jr.solve_(stokes, S.pt, S.grid, S.flow_bcs, ρg, K, G, S.dt, None, kwargs=kw, handle=h0)
            undecomposed[ng] = (S, download_stokes(stokes), int(rg.iter))
            T._set(h0, kernel_variant=0)
            del stokes, ρg, K, G
        for dims, n in VISC_CASES:
            ng = B.n_global(n, dims)
            g.finalize_global_grid()
            S = T._global_setup(jr, ng, True, 23, 8, seed=31, dt=float("inf"))
            T._set(h0, kernel_variant=1, viscous_limit=0)
            stokes, ρg, K, G = upload_stokes(S, jr.AMDGPUBackend)
            rg = jr.solve_(stokes, S.pt, S.grid, S.flow_bcs, ρg, K, G, S.dt, None, kwargs=kw, handle=h0)
            undecomposed[("visc", ng)] = (S, download_stokes(stokes), int(rg.iter))
            T._set(h0, kernel_variant=0, viscous_limit=1)
            del stokes, ρg, K, G
        g.finalize_global_grid()
        import torch.distributed as dist
        from datetime import timedelta
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(minutes=20))

        # 0. the transport alone: update_halo! of a staggered array and the norm all-reduce
        n = (20, 9, 8)
        g.init_global_grid(*n, rank=rank, nprocs=world, dimx=2, dimy=1, dimz=1)
        halo.init_comm_ipc(hc)
        cnt = C.c_int32(0)
        hc.call("jrx_comm_count", C.byref(cnt))
        assert cnt.value == world, cnt.value
        A = jr.fzeros((n[0] + 1, n[1] + 2, n[2] + 2), torch.device("cuda", 0))          # Vx-shaped, column-major
        A += float(rank + 1)
        A += torch.arange(n[0] + 1, dtype=torch.float64, device="cuda")[:, None, None] * 0.01
        for rep in range(3):                      # repeated: the sent / unpacked flags pace the reuse of the receive buffer
            halo.update_halo_(A, ni=n, handle=hc)
        torch.cuda.synchronize()
        a = A.cpu().numpy()
        other = 2 - rank                          # the neighbour's fill value
        if rank == 0:                             # my right ghost plane <- the neighbour's plane ol = 3 (0-based 2)
            assert np.allclose(a[-1], other + 0.02) and np.allclose(a[0], 1.0)
        else:                                     # my left ghost plane <- the neighbour's plane nA - ol = 21 - 3 (0-based 18)
            assert np.allclose(a[0], other + 0.18) and np.allclose(a[-1], 2.0 + 0.20)
        res["cases"].append({"case": "update_halo", "ok": True})
        g.finalize_global_grid()
        flush()

        for dims, n in CASES:
            ng = B.n_global(n, dims)
            S, glob, rg_iter = undecomposed[ng]
            g.finalize_global_grid()
            g.init_global_grid(*n, rank=rank, nprocs=world, dimx=dims[0], dimy=dims[1], dimz=dims[2])
            cart = halo.make_cart()
            halo.init_comm_ipc(hc, cart=cart)     # a fresh group (and control segment) per decomposition
            co = B.coords_of(cart)
            grid = jr.Geometry(n, S.extra["li"])
            for pipe in PIPES:
                if n[0] > 100 and pipe not in ("fused", "fused_early"):
                    continue                      # the multi-tile blocks: default pipeline + in-order pipeline only (time)
                T._set(hc, **T.PIPELINES[pipe])
                loc = Setup(ni=n, arrays={k: B.local_block(v, n, ng, co) for k, v in S.arrays.items()})
                st, rg_, K_, G_ = upload_stokes(loc, jr.AMDGPUBackend)
                f0 = hc.get_option("stat_fused3d")
                r = jr.solve_(st, S.pt, grid, S.flow_bcs, rg_, K_, G_, S.dt, None, kwargs=kw, handle=hc)
                out = download_stokes(st)
                bad = []
                for k in T.STATE + ("Rx", "Ry", "Rz", "RP", "exx", "exy", "divV"):
                    want = B.local_block(glob[k], n, ng, co)
                    m = interior_mask3d(k, want.shape)
                    if not np.array_equal(out[k][m], want[m]):
                        bad.append((k, float(np.abs(out[k] - want)[m].max())))
                res["cases"].append({"case": f"dims={dims} n={n} {pipe}", "ok": not bad and r.iter == rg_iter == 24, "bad": bad, "iter": int(r.iter),
                                     "fused_launches": int(hc.get_option("stat_fused3d") - f0), "norm_Rx": [float(x) for x in r.norm_Rx]})
                flush()
                del st, rg_, K_, G_
            g.finalize_global_grid()
        for dims, n in VISC_CASES:
            ng = B.n_global(n, dims)
            S, glob, rg_iter = undecomposed[("visc", ng)]
            g.finalize_global_grid()
            g.init_global_grid(*n, rank=rank, nprocs=world, dimx=dims[0], dimy=dims[1], dimz=dims[2])
            cart = halo.make_cart()
            halo.init_comm_ipc(hc, cart=cart)
            co = B.coords_of(cart)
            grid = jr.Geometry(n, S.extra["li"])
            T._set(hc, **T.PIPELINES["fused_inkernel"])
            loc = Setup(ni=n, arrays={k: B.local_block(v, n, ng, co) for k, v in S.arrays.items()})
            st, rg_, K_, G_ = upload_stokes(loc, jr.AMDGPUBackend)
            f0 = hc.get_option("stat_fused3d_inkernel")
            r = jr.solve_(st, S.pt, grid, S.flow_bcs, rg_, K_, G_, S.dt, None, kwargs=kw, handle=hc)
            out = download_stokes(st)
            bad = []
            for k in T.STATE + ("Rx", "Ry", "Rz", "RP", "exx", "exy", "divV"):
                want = B.local_block(glob[k], n, ng, co)
                m = interior_mask3d(k, want.shape)
                if not np.array_equal(out[k][m], want[m]):
                    bad.append((k, float(np.abs(out[k] - want)[m].max())))
            res["cases"].append({"case": f"viscous limit dims={dims} n={n} fused_inkernel", "ok": not bad and r.iter == rg_iter == 24, "bad": bad, "iter": int(r.iter),
                                 "inkernel_launches": int(hc.get_option("stat_fused3d_inkernel") - f0), "norm_Rx": [float(x) for x in r.norm_Rx]})
            flush()
            del st, rg_, K_, G_
            g.finalize_global_grid()
        # every rank must hold the same norm bits
        mine = [c["norm_Rx"] for c in res["cases"] if "norm_Rx" in c]
        obj = [None] * world
        dist.all_gather_object(obj, mine)
        res["norm_bits_equal_across_ranks"] = all(o == obj[0] for o in obj)
        hc.close()
        dist.barrier()
        dist.destroy_process_group()
    except BaseException as e:        # noqa: BLE001 -- reported through the result file
        res["error"] = f"{type(e).__name__}: {e}\n{traceback.format_exc()}"
    res["t_end"] = time.time()
    res["done"] = True
    flush()
    return 0 if res["error"] is None else 1


if __name__ == "__main__":
    sys.exit(main())
