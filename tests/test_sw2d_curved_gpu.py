"""Parity of the curved / over-integrated HIP path (bdg_sw2d_curved_*, through the C ABI) against
  (a) the outputs of the reference's swhelpers.rhs.sw2dComputeRHS_curved stored in
      tests/golden/sw2d_rhs_curved_*.npz (N = 2, 3, 4, 6, 8; curvedEls a strict subset of the elements; one case
      with curvedEls a subset of the deformed elements; one with a periodic rewiring of gmapP),
  (b) the NumPy restatement (oracle/oracle_np.py::sw2d_rhs_curved, bit-identical to (a)) on larger seeded inputs
      and through the time loop of the reference's driver (sw2d_curved.py:246-277).
Tolerance (fp64): the kernels evaluate the same contractions in another summation order (4-deep MFMA steps,
fused multiply-adds, W*rx folded once): single RHS <= 1e-12 of max|RHS|, states after 20 steps <= 1e-11.
"""
import glob
import os
import types

import numpy as np
import pytest

import blitzdg_amd.pyblitzdg as dg
from blitzdg_amd._capi import BdgError
from blitzdg_amd.sw2d_curved import Sw2dCurvedSolver
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

RHS_TOL = 1e-12
STATE_TOL = 1e-11
CURVED = sorted(glob.glob(os.path.join(GOLDEN, "sw2d_rhs_curved_*.npz")))
IDS = [os.path.basename(p)[16:-4] for p in CURVED]


def contexts_from_fixture(d):
    ctx = types.SimpleNamespace(numLocalPoints=d["V"].shape[0], numElements=d["J"].shape[1], V=d["V"], filter=d["Filter"])
    cub = types.SimpleNamespace(V=d["cubV"], Dr=d["cubDr"], Ds=d["cubDs"], W=d["cubW"], rx=d["cubrx"], ry=d["cubry"],
                                sx=d["cubsx"], sy=d["cubsy"], MMChol=d["MMChol"])
    gauss = types.SimpleNamespace(Interp=d["gInterp"], W=d["gW"], nx=d["gnx"], ny=d["gny"],
                                  BCmap={3: [int(i) for i in d["gmapW"]]})
    return ctx, cub, gauss


def solver_from_fixture(d):
    ctx, cub, gauss = contexts_from_fixture(d)
    return Sw2dCurvedSolver(ctx, cub, gauss, d["curvedEls"], d["J"], d["gmapM"], d["gmapP"], g=float(d["g"]),
                            zx=d["zx"], zy=d["zy"], f=float(d["f"]), CD=d["CD"])


@pytest.fixture(params=["nodal-trace", "general"], autouse=True)
def form(request, monkeypatch):
    """Every test of this file runs on both kernel forms: the default nodal-trace form (one launch, the neighbours' traces
    are products of their face-node values) and the general form with Gauss-trace planes (BDG_SW2D_CURVED_GENERAL=1)."""
    if request.param == "general":
        monkeypatch.setenv("BDG_SW2D_CURVED_GENERAL", "1")
    else:
        monkeypatch.delenv("BDG_SW2D_CURVED_GENERAL", raising=False)
    return request.param


def relerr(got, ref):
    scale = max(np.abs(r).max() for r in ref)
    return max(np.abs(a - b).max() for a, b in zip(got, ref)) / scale


@pytest.mark.parametrize("path", CURVED, ids=IDS)
def test_curved_rhs_matches_the_reference_function(path, form):
    d = np.load(path)
    s = solver_from_fixture(d)
    assert s.usesNodalTraces == (form == "nodal-trace")          # every fixture (the periodic one too) has the structure
    ref = [d[f"rhs{i}"] for i in (1, 2, 3, 4)]
    got = s.computeRHS(d["h"], d["hu"], d["hv"], d["hN"])
    assert relerr(got, ref) < RHS_TOL
    curved = set(int(k) for k in d["curvedEls"])
    assert 0 < len(curved) < d["J"].shape[1]                      # both mass-matrix branches are exercised
    # the driver's next step, Filter * RHS (sw2d_curved.py:250-253), fused into the kernels
    gotf = s.computeRHS(d["h"], d["hu"], d["hv"], d["hN"], filter=True)
    assert relerr(gotf, [d["Filter"] @ r for r in ref]) < RHS_TOL


def test_drop_in_signature_of_the_reference_function():
    """blitzdg_amd.swhelpers.rhs.sw2dComputeRHS_curved takes the reference's 17 arguments (swhelpers/rhs.py:6) and
    objects with the reference contexts' attribute names -- here real contexts from this repo's builders."""
    from blitzdg_amd.swhelpers.rhs import sw2dComputeRHS_curved
    d = np.load(os.path.join(GOLDEN, "sw2d_rhs_curved_coarse_box_N4.npz"))
    mesh = dg.MeshManager()
    mesh.readMesh(os.path.join(GOLDEN, "coarse_box.msh"))
    nodes = dg.TriangleNodesProvisioner(4, mesh)
    nodes.buildFilter(0.9 * 4, 4)
    ctx = nodes.dgContext()
    nodes.setCoordinates(d["x"], d["y"])
    gauss_ctx = nodes.buildGaussFaceNodes(2 * (4 + 1))
    cub_ctx = nodes.buildCubatureVolumeMesh(3 * (4 + 1))
    J = d["J"]
    gmapM, gmapP = gauss_ctx.mapM, gauss_ctx.mapP
    H = d["H"]
    curvedEls = [int(k) for k in d["curvedEls"]]
    r = sw2dComputeRHS_curved(d["h"], d["hu"], d["hv"], d["hN"], d["zx"], d["zy"], float(d["g"]), H, float(d["f"]), d["CD"],
                              ctx, cub_ctx, gauss_ctx, curvedEls, J, gmapM, gmapP)
    assert relerr(r, [d[f"rhs{i}"] for i in (1, 2, 3, 4)]) < RHS_TOL
    r2 = sw2dComputeRHS_curved(d["h"], 2 * d["hu"], d["hv"], d["hN"], d["zx"], d["zy"], float(d["g"]), H, float(d["f"]),
                               d["CD"], ctx, cub_ctx, gauss_ctx, curvedEls, J, gmapM, gmapP)   # cached device image
    assert relerr(r2, r) > 1e-3


def big_problem(order, nx, ny, seed=7):
    """A deformed box with many elements, contexts from this repo's builders, and the oracle's tables."""
    mesh = dg.MeshManager()
    mesh.buildBoxMesh(nx, ny)
    nodes = dg.TriangleNodesProvisioner(order, mesh)
    nodes.buildFilter(0.9 * order, order)
    ctx = nodes.dgContext()
    x0, y0 = ctx.x, ctx.y
    rho2 = ((x0 - 0.4) ** 2 + (y0 + 1.0) ** 2) / 0.8 ** 2
    b = np.where(rho2 < 1.0, (1.0 - rho2) ** 3, 0.0)
    x, y = x0 + 0.02 * b * np.cos(1.3 * y0), y0 + 0.05 * b * np.sin(1.7 * x0 + 0.4)
    curvedEls = np.where((np.abs(x - x0) + np.abs(y - y0)).max(axis=0) > 0)[0].astype(np.int32)
    nodes.setCoordinates(x, y)
    J = (ctx.Dr @ x) * (ctx.Ds @ y) - (ctx.Ds @ x) * (ctx.Dr @ y)
    gauss = nodes.buildGaussFaceNodes(2 * (order + 1))
    cub = nodes.buildCubatureVolumeMesh(3 * (order + 1))
    rng = np.random.default_rng(seed)
    h = 1.0 + 0.3 * np.exp(-8 * x * x - 8 * y * y)
    hu, hv = 0.05 * rng.standard_normal(x.shape), 0.05 * rng.standard_normal(x.shape)
    hN = h * (0.5 + 0.3 * np.sin(2 * x) * np.cos(3 * y))
    zx, zy = 0.05 + 0 * x, -0.04 * y
    CD = 2.5e-3 * (1.0 + 0.5 * np.cos(x))
    t = dict(cubV=cub.V, cubDr=cub.Dr, cubDs=cub.Ds, cubW=cub.W, cubrx=cub.rx, cubry=cub.ry, cubsx=cub.sx, cubsy=cub.sy,
             gInterp=gauss.Interp, gW=gauss.W, gnx=gauss.nx, gny=gauss.ny, gmapM=gauss.mapM, gmapP=gauss.mapP,
             gmapW=np.array(gauss.BCmap[3], dtype=np.int32), V=ctx.V, J=J, MMChol=cub.MMChol, curvedEls=curvedEls,
             Filter=ctx.filter)
    solver = Sw2dCurvedSolver(ctx, cub, gauss, curvedEls, J, gauss.mapM, gauss.mapP, g=0.0245, zx=zx, zy=zy, f=0.0788, CD=CD)
    return solver, t, (h, hu, hv, hN), dict(zx=zx, zy=zy, g=0.0245, f=0.0788, CD=CD)


@pytest.mark.parametrize("order,nx,ny", [(1, 23, 17), (4, 40, 33), (5, 21, 16), (7, 9, 8)])
def test_curved_rhs_matches_the_oracle_on_ragged_meshes(order, nx, ny):
    """Element counts that are not a multiple of the 16-element tile or of the 64-element padding."""
    from oracle import oracle_np
    s, t, q, ph = big_problem(order, nx, ny)
    ref = oracle_np.sw2d_rhs_curved(*q, ph["zx"], ph["zy"], ph["g"], ph["f"], ph["CD"], t)
    assert relerr(s.computeRHS(*q), ref) < RHS_TOL
    assert 0 < len(t["curvedEls"]) < 2 * nx * ny


def test_driver_loop_rk2_with_filter_matches_the_oracle():
    """sw2d_curved.py:246-277 -- RHS, filter, predictor, RHS, filter, corrector -- 20 steps resident on the device
    against the same loop in NumPy on the oracle."""
    from oracle import oracle_np
    s, t, q, ph = big_problem(3, 12, 10)
    dt, Filt = 2e-3, t["Filter"]
    s.setState(*q)
    s.stepRK2(dt, 20, filter=True)
    got = s.getState()
    ref = [a.copy() for a in q]
    rhs = lambda qq: [Filt @ r for r in oracle_np.sw2d_rhs_curved(*qq, ph["zx"], ph["zy"], ph["g"], ph["f"], ph["CD"], t)]  # noqa: E731
    for _ in range(20):
        r = rhs(ref)
        q1 = [a + 0.5 * dt * b for a, b in zip(ref, r)]
        r = rhs(q1)
        ref = [a + dt * b for a, b in zip(ref, r)]
    assert relerr(got, ref) < STATE_TOL
    assert np.abs(ref[1] - q[1]).max() > 1e-4                    # the state did move
    # LSERK4 stages on the same state: 5 stages = one step of the reference's low-storage scheme (include/LSERK4.hpp)
    a = dg.LSERK4.rk4a
    bcoef = dg.LSERK4.rk4b
    s.setState(*q)
    s.lserk4Stages(dt, 5)
    got = s.getState()
    ref, res = [x.copy() for x in q], [np.zeros_like(x) for x in q]
    for i in range(5):
        r = oracle_np.sw2d_rhs_curved(*ref, ph["zx"], ph["zy"], ph["g"], ph["f"], ph["CD"], t)
        res = [a[i] * x + dt * y for x, y in zip(res, r)]
        ref = [x + bcoef[i] * y for x, y in zip(ref, res)]
    assert relerr(got, ref) < STATE_TOL


def test_straight_mesh_without_curved_elements_and_constant_sources():
    """curvedEls empty, scalar f and CD, no bed slope, identity maps: the plain over-integrated RHS."""
    from oracle import oracle_np
    mesh = dg.MeshManager()
    mesh.buildBoxMesh(7, 5, shuffleSeed=99)
    nodes = dg.TriangleNodesProvisioner(4, mesh)
    ctx = nodes.dgContext()
    gauss, cub = nodes.buildGaussFaceNodes(10), nodes.buildCubatureVolumeMesh(15)
    x, y = ctx.x, ctx.y
    rng = np.random.default_rng(11)
    h = 2.0 + 0.3 * np.sin(2 * x) * np.cos(y)
    hu, hv = 0.2 * rng.standard_normal(x.shape), 0.2 * rng.standard_normal(x.shape)
    hN = 0.7 * h
    s = Sw2dCurvedSolver(ctx, cub, gauss, [], ctx.J, gauss.mapM, gauss.mapP, g=9.81, f=1e-2, CD=3e-3)
    t = dict(cubV=cub.V, cubDr=cub.Dr, cubDs=cub.Ds, cubW=cub.W, cubrx=cub.rx, cubry=cub.ry, cubsx=cub.sx, cubsy=cub.sy,
             gInterp=gauss.Interp, gW=gauss.W, gnx=gauss.nx, gny=gauss.ny, gmapM=gauss.mapM, gmapP=gauss.mapP,
             gmapW=np.array(gauss.BCmap[3], dtype=np.int32), V=ctx.V, J=ctx.J, MMChol=cub.MMChol, curvedEls=[])
    z = np.zeros_like(h)
    ref = oracle_np.sw2d_rhs_curved(h, hu, hv, hN, z, z, 9.81, 1e-2, 3e-3, t)
    assert relerr(s.computeRHS(h, hu, hv, hN), ref) < RHS_TOL
    with pytest.raises(BdgError, match="Filter"):
        s.computeRHS(h, hu, hv, hN, filter=True)                 # ctx.filter was never built


def test_contexts_without_face_structure_fall_back_to_the_general_form(form):
    """The nodal-trace form needs gmapM = identity and a gmapP that pairs whole faces; anything else must be served by
    the general form, with the same answer as the oracle (which follows whatever maps it is given)."""
    from oracle import oracle_np
    d = dict(np.load(os.path.join(GOLDEN, "sw2d_rhs_curved_coarse_box_N3.npz")))
    NG = int(d["NGauss"])
    # (i) two Gauss points of one interior face exchange their partners: the face is no longer paired as a whole
    gmapP = d["gmapP"].copy()
    inner = np.where(gmapP != np.arange(gmapP.size))[0]
    i0 = int(inner[0]) // NG * NG
    gmapP[i0], gmapP[i0 + 1] = gmapP[i0 + 1], gmapP[i0]
    # (ii) an interior-side map that is not the identity
    gmapM = d["gmapM"].copy()
    gmapM[i0 + 2], gmapM[i0 + 3] = gmapM[i0 + 3], gmapM[i0 + 2]
    for maps in (dict(gmapP=gmapP), dict(gmapM=gmapM)):
        t = dict(d, **maps)
        ctx, cub, gauss = contexts_from_fixture(t)
        s = Sw2dCurvedSolver(ctx, cub, gauss, t["curvedEls"], t["J"], t["gmapM"], t["gmapP"], g=float(t["g"]), zx=t["zx"],
                             zy=t["zy"], f=float(t["f"]), CD=t["CD"])
        assert not s.usesNodalTraces
        ref = oracle_np.sw2d_rhs_curved(t["h"], t["hu"], t["hv"], t["hN"], t["zx"], t["zy"], float(t["g"]), float(t["f"]), t["CD"], t)
        assert relerr(s.computeRHS(t["h"], t["hu"], t["hv"], t["hN"]), ref) < RHS_TOL
        assert relerr(ref, [d[f"rhs{i}"] for i in (1, 2, 3, 4)]) > 1e-6      # the rewiring does change the answer


def test_curved_driver_example_runs_the_reference_loop(form):
    """examples/sw2d_curved.py = the reference's sw2d_curved.py driver (curved wall, periodic ends, wall-layer drag, tracer,
    RK2 + filter) on this repository's API: 40 steps stay finite, the periodic rewiring keeps the nodal-trace kernels,
    the tracer stays within its initial bounds to the filter's overshoot. Mass: the reference's own helpers leave the mesh slightly
    non-conforming along the curved wall (adjustStraightEdges snaps a wall vertex onto the nearest of 4096 spline samples for the
    elements that have a wall FACE there, while the elements that touch the wall in that vertex only keep the old position, and
    deformAndBlendElements does not move a face's end node with vr = 1: gaps of a fraction of a sample spacing, 0.1 m in 8 km), so
    the total mass is conserved to that, not to round-off (measured 4e-7 over 40 steps; a conforming mesh gives 1e-13, as the
    partitioned-solver tests of this file check)."""
    import re
    import sys

    from conftest import ROOT, launch
    env = dict(os.environ)
    if form == "general":
        env["BDG_SW2D_CURVED_GENERAL"] = "1"
    else:
        env.pop("BDG_SW2D_CURVED_GENERAL", None)
    out = launch([sys.executable, os.path.join(ROOT, "examples", "sw2d_curved.py"), "box:24x6", "3", "40"], env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "done: steps=40" in out.stdout
    assert f"nodal-trace kernels={form == 'nodal-trace'}" in out.stdout
    drift = float(re.findall(r"mass drift=([-+.\deE]+)", out.stdout)[-1])
    assert abs(drift) < 5e-6
    lo, hi = (float(v) for v in re.findall(r"N in \[([-+.\deE]+), ([-+.\deE]+)\]", out.stdout)[-1])
    assert -0.05 < lo and hi < 1.05


# ---- element-partitioned curved solver: one process per rank on this one GPU, ghost columns over gloo

DIST_MESH, DIST_ORDER, DIST_STEPS, DIST_DT = (14, 10), 3, 4, 2e-3


def _dist_deform(x0, y0):
    blend = np.clip(1.0 - (y0 + 1.0) / 0.35, 0.0, 1.0) ** 3
    return x0 + 0.01 * blend * np.sin(2 * y0 + 1), y0 + 0.04 * blend * np.sin(np.pi * x0)


def _dist_state(x, y):
    h = 1.0 + 0.3 * np.exp(-6 * x * x - 6 * (y + 0.3) ** 2)
    return h, 0.05 * np.sin(3 * x + 1) * np.cos(2 * y), 0.05 * np.cos(2 * x) * np.sin(3 * y - 1), h * (0.5 + 0.3 * np.sin(2 * x) * np.cos(3 * y))


def _dist_sources(x, y):
    return {"zx": 0.05 + 0 * x, "zy": -0.04 * y, "f": 0.0788, "CD": 2.5e-3 * (1.0 + 0.5 * np.cos(x))}


def _curved_rank_worker(rank, world, port, out_dir, general, native_env=None, stepper="rk2"):
    import sys
    from blitzdg_amd.halo import build_plan
    from blitzdg_amd.sw2d_curved import DistributedSw2dCurved, NativeDistributedSw2dCurved
    if general:
        os.environ["BDG_SW2D_CURVED_GENERAL"] = "1"
    mesh = dg.MeshManager()
    mesh.buildBoxMesh(*DIST_MESH, shuffleSeed=5)
    mesh.partitionMesh(world)
    plan = build_plan(mesh.elements, mesh.vertices, mesh.EToE, mesh.elementPartitionMap, rank, world, bctype=mesh.bcType)
    kw = dict(g=0.0245, filter_args=(0.9 * DIST_ORDER, DIST_ORDER), sources=_dist_sources)
    dist = None
    if native_env is not None:     # the library's own exchange (pack kernel, grouped send / receive, unpack kernel) through tests/mock_rccl
        os.environ.update(native_env)
        os.environ.update({"RANK": str(rank), "LOCAL_RANK": "0", "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                           "MASTER_PORT": str(port)})
        d = NativeDistributedSw2dCurved(plan, DIST_ORDER, _dist_deform, **kw)
        assert "torch" not in sys.modules
    else:
        import torch
        import torch.distributed as dist
        backend = "nccl" if world == 1 else "gloo"   # one rank: the RCCL backend's device-side staging (buffers on the solver's GPU)
        if backend == "nccl":
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            torch.cuda.set_device(0)
        pg = {"device_id": torch.device("cuda", 0)} if backend == "nccl" else {}
        dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, **pg)
        d = DistributedSw2dCurved(plan, DIST_ORDER, _dist_deform, dist, device=0, **kw)
        assert d._on_device == (backend == "nccl") and d.device == 0
    try:
        d.set_initial_state(_dist_state)
        if stepper == "rk2":
            d.step_rk2(DIST_DT, 1)                # several calls
            d.step_rk2(DIST_DT, DIST_STEPS - 1)
        else:
            d.lserk4_stages(DIST_DT, 3)
            d.lserk4_stages(DIST_DT, 5 * DIST_STEPS - 3)
        out = d.owned_state()
        if native_env is not None:
            d.barrier()
        np.savez(os.path.join(out_dir, f"curved{rank}.npz"), ids=out[0], ghosts=plan.num_halo, **{f"q{i}": a for i, a in enumerate(out[1:])})
    finally:
        if dist is not None:
            dist.destroy_process_group()
    sys.stdout.flush()


@pytest.mark.parametrize("world,transport,stepper", [(2, "gloo", "rk2"), (3, "gloo", "rk2"), (2, "native", "rk2"), (3, "native", "rk2"),
                                                     (2, "gloo", "lserk4"), (3, "native", "lserk4"), (1, "nccl", "rk2")])
def test_partitioned_curved_solver_matches_the_single_domain_run(tmp_path, world, transport, stepper, form, mock_rccl):
    """DistributedSw2dCurved / NativeDistributedSw2dCurved: each rank owns a part of a deformed, shuffled box mesh plus a ghost
    layer, refreshes the ghost columns before every RHS evaluation -- host-staged over gloo, or by the library itself
    (bdg_sw2d_curved_comm_init / _step_rk2_exchanged: pack kernel, grouped send / receive, unpack kernel, with only librccl.so
    replaced by tests/mock_rccl) -- and runs the driver's RK2 + filter steps, or LSERK4 stages with an exchange in front of each;
    the owned states equal the single-domain solver's to round-off (the tiling of the elements differs between the runs, so not
    bit for bit)."""
    import socket
    from conftest import launch_ranks
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    launch_ranks("test_sw2d_curved_gpu", "_curved_rank_worker", world,
                 (world, port, str(tmp_path), form == "general", mock_rccl if transport == "native" else None, stepper))
    mesh = dg.MeshManager()
    mesh.buildBoxMesh(*DIST_MESH, shuffleSeed=5)
    nodes = dg.TriangleNodesProvisioner(DIST_ORDER, mesh)
    nodes.buildFilter(0.9 * DIST_ORDER, DIST_ORDER)
    ctx = nodes.dgContext()
    x, y = _dist_deform(ctx.x, ctx.y)
    curved = np.where((np.abs(x - ctx.x) + np.abs(y - ctx.y)).max(axis=0) > 0)[0]
    nodes.setCoordinates(x, y)
    J = (ctx.Dr @ x) * (ctx.Ds @ y) - (ctx.Ds @ x) * (ctx.Dr @ y)
    gauss, cub = nodes.buildGaussFaceNodes(2 * (DIST_ORDER + 1)), nodes.buildCubatureVolumeMesh(3 * (DIST_ORDER + 1))
    src = _dist_sources(x, y)
    s = Sw2dCurvedSolver(ctx, cub, gauss, curved, J, gauss.mapM, gauss.mapP, g=0.0245, zx=src["zx"], zy=src["zy"], f=src["f"], CD=src["CD"])
    q0 = _dist_state(x, y)
    s.setState(*q0)
    if stepper == "rk2":
        s.stepRK2(DIST_DT, DIST_STEPS, filter=True)
    else:
        s.lserk4Stages(DIST_DT, 5 * DIST_STEPS)
    ref = s.getState()
    seen = np.zeros(mesh.numElements, dtype=int)
    for r in range(world):
        p = np.load(tmp_path / f"curved{r}.npz")
        seen[p["ids"]] += 1
        assert int(p["ghosts"]) > 0 or world == 1
        for i, full in enumerate(ref):
            assert np.abs(p[f"q{i}"] - full[:, p["ids"]]).max() <= STATE_TOL * np.abs(full).max(), f"field {i} differs on rank {r}"
    assert (seen == 1).all() and 0 < curved.size < mesh.numElements
    assert np.abs(ref[1] - q0[1]).max() > 1e-5                   # the state did move


@pytest.mark.parametrize("overlap", [True, False])
def test_native_curved_exchange_through_real_rccl_loopback(overlap, monkeypatch):
    """The library's exchange through the REAL librccl.so on this one GPU: rank 0's share of a 2-way split with every neighbour
    exchange a send-to-self (NativeDistributedSw2dCurved(loopback=True)): pack kernel, ncclSend / ncclRecv to the own rank, unpack
    kernel. The ghosts then hold this rank's own boundary elements; the same copies done by hand through the host-staged
    interface (get / set elements, rk2Phase) must give the same owned state -- bit for bit when every element is evaluated in
    stream order (BDG_SW2D_CURVED_NO_OVERLAP), to round-off on the two-chain schedule of the nodal-trace form (interior and
    partition-boundary elements are tiled separately there, and a straight element in a tile with a curved one takes the
    general branch: other instructions, same function)."""
    if not overlap:
        monkeypatch.setenv("BDG_SW2D_CURVED_NO_OVERLAP", "1")
    from blitzdg_amd.halo import build_plan
    from blitzdg_amd.sw2d_curved import DistributedSw2dCurved, NativeDistributedSw2dCurved
    mesh = dg.MeshManager()
    mesh.buildBoxMesh(*DIST_MESH)
    mesh.partitionMesh(2)
    plan = build_plan(mesh.elements, mesh.vertices, mesh.EToE, mesh.elementPartitionMap, 0, 2, bctype=mesh.bcType)
    kw = dict(g=0.0245, filter_args=(0.9 * DIST_ORDER, DIST_ORDER), sources=_dist_sources)
    nat = NativeDistributedSw2dCurved(plan, DIST_ORDER, _dist_deform, loopback=True, **kw)
    pr, ss, sc, rs, rc = nat.peer_table
    if int(rc.sum()) != plan.num_halo:
        pytest.skip("this split has a neighbour that sends fewer elements than it receives: a self-exchange leaves ghosts unset")
    nat.set_initial_state(_dist_state)
    nat.step_rk2(DIST_DT, 3)
    nat.barrier()
    got = nat.solver.getState()

    class ByHand(DistributedSw2dCurved):
        def _exchange(self, intermediate):
            n_own = self.plan.num_owned
            for s0, cnt, r0 in zip(ss, sc, rs):
                for i in range(int(cnt)):
                    col = self.solver.getElements(int(self.plan.send_local[s0 + i]), 1, intermediate)
                    self.solver.setElements(n_own + int(r0) + i, col, intermediate)

    class _NoDist:
        @staticmethod
        def get_backend():
            return "none"
    ref = ByHand(plan, DIST_ORDER, _dist_deform, _NoDist(), **kw)
    ref.set_initial_state(_dist_state)
    ref.step_rk2(DIST_DT, 3)
    want = ref.solver.getState()
    n = plan.num_owned            # (ghost elements are not evaluated by the overlapped schedule: owned columns only)
    assert 0 < plan.num_interior < n
    for a, b in zip(got, want):
        assert np.isfinite(b[:, :n]).all()
        if overlap and nat.solver.usesNodalTraces:
            assert np.abs(a[:, :n] - b[:, :n]).max() <= STATE_TOL * np.abs(b).max()
        else:
            assert np.array_equal(a[:, :n], b[:, :n])


@pytest.mark.parametrize("order,cells", [(2, (200, 100)), (4, (160, 100)), (7, (80, 48))])
def test_full_launch_scheduling_does_not_change_the_result(order, cells, form, monkeypatch):
    """What only a launch that fills the chip exercises (the fixtures are a few tiles): the list that deals the general tiles
    to the eight XCDs, the priority window of a CU's second workgroup (N <= 4), the streamed volume tiles in lockstep (N = 7).
    On a deformed box of 32 000 / 40 000 / 7 680 elements: the nodal-trace form with and without the tile list / the priority
    agrees bit for bit (scheduling only), and with the general form -- other kernels, validated on the fixtures -- for one RHS
    and after three RK2 + filter steps. Tolerance of that comparison: g = 9.81 on a fine mesh is a badly conditioned RHS (the
    pressure term's volume and surface integrals are ~500 and cancel to ~5), and on straight-sided elements both forms replace
    the builders' per-point metric tables -- constant up to their own round-off, ~1e-14 relative -- by one number per element
    (differently: 14 numbers here, the cubature geometry only there), so the forms differ by that noise times the
    conditioning: measured 2e-11 .. 1e-10 of max|RHS| on straight elements (against the NumPy restatement on the noisy
    tables: 1.7e-11 and 6e-12), 2.6e-13 on the curved ones; bound asserted: 5e-10."""
    if form == "general":
        pytest.skip("one comparison of the two forms is enough")

    def run(env):
        for k in ("BDG_SW2D_CURVED_GENERAL", "BDG_SW2D_CURVED_NO_TILE_ORDER", "BDG_SW2D_CURVED_PRIO"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        mesh = dg.MeshManager()
        mesh.buildBoxMesh(*cells)
        nodes = dg.TriangleNodesProvisioner(order, mesh)
        nodes.buildFilter(0.9 * order, order)
        ctx = nodes.dgContext()
        x0, y0 = ctx.x, ctx.y
        blend = np.clip(1.0 - (y0 + 1.0) / 0.15, 0.0, 1.0) ** 3
        x, y = x0 + 0.004 * blend * np.sin(2 * y0 + 1), y0 + 0.02 * blend * np.sin(3 * x0)
        curved = np.where((np.abs(x - x0) + np.abs(y - y0)).max(axis=0) > 0)[0]
        nodes.setCoordinates(x, y)
        J = (ctx.Dr @ x) * (ctx.Ds @ y) - (ctx.Ds @ x) * (ctx.Dr @ y)
        gauss, cub = nodes.buildGaussFaceNodes(2 * (order + 1)), nodes.buildCubatureVolumeMesh(3 * (order + 1))
        s = Sw2dCurvedSolver(ctx, cub, gauss, curved, J, gauss.mapM, gauss.mapP, g=9.81, zx=0.01 * np.cos(x), zy=0.02 * np.sin(y),
                             f=0.3, CD=2.5e-3 * (1.0 + 0.5 * np.cos(x)))
        h = 1.0 + 0.1 * np.exp(-10 * x * x - 10 * (y + 0.5) ** 2)
        q = (h, 0.05 * h * np.sin(3 * x), 0.04 * h * np.cos(2 * y), h * (0.5 + 0.3 * np.sin(2 * x)))
        rhs = s.computeRHS(*q, filter=True)
        s.setState(*q)
        s.stepRK2(2e-4, 3, filter=True)
        return s.usesNodalTraces, 0 < curved.size < ctx.numElements, rhs, s.getState()

    nt, mixed, rhs, state = run({})
    assert nt and mixed
    for env in ({"BDG_SW2D_CURVED_NO_TILE_ORDER": "1"}, {"BDG_SW2D_CURVED_PRIO": "0"}):
        _, _, rhs2, state2 = run(env)
        assert all(np.array_equal(a, b) for a, b in zip(rhs, rhs2)) and all(np.array_equal(a, b) for a, b in zip(state, state2)), env
    nt3, _, rhs3, state3 = run({"BDG_SW2D_CURVED_GENERAL": "1"})
    assert not nt3
    assert relerr(rhs, rhs3) <= 5e-10 and relerr(state, state3) <= 5e-10
    assert np.isfinite(state[0]).all()


def test_bad_tables_are_refused_before_anything_runs():
    d = np.load(CURVED[0])
    ctx, cub, gauss = contexts_from_fixture(d)
    bad = d["gmapP"].copy()
    bad[5] = bad.size                                            # one past the last Gauss node
    with pytest.raises(BdgError, match="out of range"):
        Sw2dCurvedSolver(ctx, cub, gauss, d["curvedEls"], d["J"], d["gmapM"], bad, g=1.0)
    with pytest.raises(BdgError, match="curvedEls"):
        Sw2dCurvedSolver(ctx, cub, gauss, [d["J"].shape[1]], d["J"], d["gmapM"], d["gmapP"], g=1.0)
    with pytest.raises(BdgError, match="positive"):
        Sw2dCurvedSolver(ctx, cub, gauss, d["curvedEls"], -d["J"], d["gmapM"], d["gmapP"], g=1.0)


def test_set_partition_refuses_a_plan_that_would_race(form):
    """bdg_sw2d_curved_set_partition: the two-chain schedule evaluates [0, num_interior) beside the ghost exchange, so an
    'interior' element with a ghost neighbour or one that is packed for a neighbour is an argument error, not a silent race;
    and the exchanged steppers refuse to run without a communicator."""
    from ctypes import POINTER, c_int
    from blitzdg_amd._capi import check, lib
    d = np.load(os.path.join(GOLDEN, "sw2d_rhs_curved_coarse_box_N3.npz"))
    s = solver_from_fixture(d)
    K = d["J"].shape[1]
    gm = d["gmapP"].reshape(K, -1) // (d["gmapP"].size // K)      # neighbour element of every Gauss point
    inner = int(np.flatnonzero(gm.max(axis=1) >= K - 5).min())    # first element with a neighbour among the last five
    none = np.zeros(1, dtype=np.int32)
    ptr = lambda a: a.ctypes.data_as(POINTER(c_int))              # noqa: E731
    with pytest.raises(BdgError, match="ghost neighbour"):
        check(lib.bdg_sw2d_curved_set_partition(s._h, inner + 1, K - 5, ptr(none), 0))
    send = np.array([inner], dtype=np.int32)
    with pytest.raises(BdgError, match="interior range"):
        check(lib.bdg_sw2d_curved_set_partition(s._h, inner + 1, K, ptr(send), 1))
    assert 0 < inner < K - 5
    check(lib.bdg_sw2d_curved_set_partition(s._h, inner, K - 5, ptr(none), 0))   # a consistent plan
    held = s.deviceBytes
    check(lib.bdg_sw2d_curved_set_partition(s._h, inner, K - 5, ptr(none), 0))
    assert s.deviceBytes == held                                   # a replaced buffer is not counted twice
    with pytest.raises(BdgError, match="no communicator"):
        check(lib.bdg_sw2d_curved_step_rk2_exchanged(s._h, 1e-3, 1, 0))


# ---- the curved RHS at production scale and gravity (round 4): g = 9.81, depth 10..11, 2080 elements, 553 of them curved

BIG = sorted(glob.glob(os.path.join(GOLDEN, "sw2d_bigcurved_*.npz")))
# Measured on the GPU against the reference function's stored output (printed by the test): 3.7e-13 of max|RHS|
# on the nodal-trace form, 1.7e-13 on the general form, straight and curved elements alike, with and without the straight-element
# compression -- the <= 1e-12 of the small fixtures holds at g = 9.81 on 2080 elements (pressure flux g h^2 / 2 = 490).
BIG_TOL = 2e-12


def rebuilt_contexts(d):
    """The contexts the reference function was given when the fixture was made, from the stored coordinates with the same builders."""
    order = int(d["order"])
    mesh = dg.MeshManager()
    mesh.buildBoxMesh(int(d["nx"]), int(d["ny"]), shuffleSeed=int(d["seed"]))
    nodes = dg.TriangleNodesProvisioner(order, mesh)
    nodes.buildFilter(0.9 * order, order)
    ctx = nodes.dgContext()
    x, y = d["x"], d["y"]
    nodes.setCoordinates(x, y)
    J = np.dot(ctx.Dr, x) * np.dot(ctx.Ds, y) - np.dot(ctx.Ds, x) * np.dot(ctx.Dr, y)
    gauss = nodes.buildGaussFaceNodes(2 * (order + 1))
    cub = nodes.buildCubatureVolumeMesh(3 * (order + 1))
    return nodes, ctx, cub, gauss, J


@pytest.mark.parametrize("path", BIG, ids=[os.path.basename(p)[15:-4] for p in BIG])
def test_curved_rhs_at_production_gravity_matches_the_reference_function(path, form, monkeypatch):
    """Both kernel forms (and the nodal-trace form with its straight-element compression switched off) against the output of the
    reference's swhelpers.rhs.sw2dComputeRHS_curved at g = 9.81 on a 2080-element deformed mesh: the tolerance the solver is
    documented with at production scale."""
    d = np.load(path)
    nodes, ctx, cub, gauss, J = rebuilt_contexts(d)
    assert len(gauss.BCmap[3]) == int(d["num_wall"])
    ref = [d[f"rhs{i}"] for i in (1, 2, 3, 4)]
    errs = {}
    for label, env in (("default", None), ("tables", "1")):
        if env:
            monkeypatch.setenv("BDG_SW2D_CURVED_NO_AFFINE", env)
        else:
            monkeypatch.delenv("BDG_SW2D_CURVED_NO_AFFINE", raising=False)
        s = Sw2dCurvedSolver(ctx, cub, gauss, d["curvedEls"], J, gauss.mapM, gauss.mapP, g=float(d["g"]), zx=d["zx"], zy=d["zy"],
                             f=float(d["f"]), CD=d["CD"])
        assert s.usesNodalTraces == (form == "nodal-trace")
        got = s.computeRHS(d["h"], d["hu"], d["hv"], d["hN"])
        errs[label] = relerr(got, ref)
        straight = np.setdiff1d(np.arange(J.shape[1]), d["curvedEls"])
        scale = max(np.abs(r).max() for r in ref)
        errs[label + " straight"] = max(np.abs(a[:, straight] - b[:, straight]).max() for a, b in zip(got, ref)) / scale
        errs[label + " curved"] = max(np.abs(a[:, d["curvedEls"]] - b[:, d["curvedEls"]]).max() for a, b in zip(got, ref)) / scale
    print(f"curved RHS, g = 9.81, K = {J.shape[1]}, form {form}: relative to max|RHS| " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    assert max(errs.values()) < BIG_TOL, errs
