"""Multi-rank sw2d on the GPU box. Every child process (rank workers, bench.py, hipcc) is started by
tests/launcher.py (conftest.launch / launch_ranks), never forked from this GPU-initialised pytest process, and PyTorch
is imported by rank workers only. Only ONE GPU is available to tests, and RCCL refuses two
ranks on one device, so:
  * two ranks share cuda:0 and exchange ghosts through `gloo` (host-staged): exercises the
    device-side partition logic -- pack / unpack kernels, interior / boundary launches, ghost
    slots, dt reduction over owned elements -- against a single-domain run;
  * one rank with the real `nccl` (RCCL) backend exercises the production code path
    (ExternalStream, batch_isend_irecv plumbing with an empty neighbour list, all_reduce).
The 2/4/8-GPU RCCL runs themselves are the driver's SCALE run.
"""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import launch, launch_group, launch_ranks

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NX, NY, ORDER, NSTAGES = 48, 36, 4, 12


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fields(x, y):
    h = 10.0 + np.exp(-10 * x * x - 10 * y * y)
    hu = 0.1 * np.sin(3 * x + 1) * np.cos(2 * y)
    hv = 0.1 * np.cos(2 * x) * np.sin(3 * y - 1)
    return h, hu, hv


def _worker(rank, world, port, backend, out_dir):
    import faulthandler
    faulthandler.enable()
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    from blitzdg_amd.halo import DistributedSw2d

    torch.cuda.set_device(0)
    kw = {"device_id": torch.device("cuda", 0)} if backend == "nccl" else {}
    dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, **kw)
    try:
        d = DistributedSw2d.box(NX, NY, ORDER, device=0)
        d.set_initial_state(_fields)
        dt = 0.5 * d.compute_dt(0.65)
        for _ in range(NSTAGES):
            d.lserk4_stage(dt)
        d.solver.synchronize()
        ids, h, hu, hv = d.owned_state()
        np.savez(os.path.join(out_dir, f"{backend}{rank}.npz"), ids=ids, h=h, hu=hu, hv=hv, dt=dt,
                 total=d.global_elements, **{k: v for k, v in d.halo_counts().items()})
    finally:
        dist.destroy_process_group()


def _single_domain(dt, order=ORDER):
    import blitzdg_amd.pyblitzdg as dg
    from blitzdg_amd import sw2d
    mesh = dg.MeshManager()
    mesh.buildBoxMesh(NX, NY)
    nodes = dg.TriangleNodesProvisioner(order, mesh)
    ctx = nodes.dgContext()
    s = sw2d.Sw2dSolver(nodes=nodes)
    s.setState(*_fields(ctx.x, ctx.y))
    dt_single = 0.5 * s.computeDt(0.65)[0]
    assert dt_single == dt  # global min over ranks == single-domain value, bit for bit
    s.lserk4Stages(dt, NSTAGES)
    return s.getState()


@pytest.mark.parametrize("backend,world", [("gloo", 2), ("gloo", 3), ("nccl", 1)])
def test_distributed_matches_single_domain(tmp_path, backend, world):
    launch_ranks("test_dist_gpu", "_worker", world, (world, _free_port(), backend, str(tmp_path)))
    parts = [np.load(tmp_path / f"{backend}{r}.npz") for r in range(world)]
    ref = _single_domain(float(parts[0]["dt"]))
    seen = np.zeros(2 * NX * NY, dtype=int)
    for p in parts:
        ids = p["ids"]
        seen[ids] += 1
        assert int(p["total"]) == 2 * NX * NY
        if world > 1:
            assert p["ghost"] > 0 and p["sent"] > 0 and p["interior"] > 0
        for name, r in zip(("h", "hu", "hv"), ref):
            # same kernels, same per-element arithmetic: identical bits
            assert np.array_equal(p[name], r[:, ids]), name
    assert (seen == 1).all()


_NATIVE_SCRIPT = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np
from blitzdg_amd.halo import NativeDistributedSw2d
from test_dist_gpu import NX, NY, ORDER, NSTAGES, _fields
assert "torch" not in sys.modules
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
d = NativeDistributedSw2d.box(NX, NY, ORDER, rank, world, device=0)
d.set_initial_state(_fields)
dt = 0.5 * d.compute_dt(0.65)
d.lserk4_stages(dt, NSTAGES)
d.barrier()
assert d.allreduce_max(3.5) == 3.5
ids, h, hu, hv = d.owned_state()
np.savez(sys.argv[2], ids=ids, h=h, hu=hu, hv=hv, dt=dt, total=d.global_elements)
d.close()
"""


def test_native_rccl_single_rank(tmp_path):
    """The production transport (RCCL bound by the C++ library, no PyTorch in the process) with
    one rank: communicator init from a file rendezvous, empty neighbour group, stream/event
    choreography of the exchanged stage, device all-reduce."""
    out = tmp_path / "native.npz"
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()))
    r = launch([sys.executable, "-c", _NATIVE_SCRIPT, ROOT, str(out)], env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    p = np.load(out)
    ref = _single_domain(float(p["dt"]))
    assert int(p["total"]) == 2 * NX * NY and np.array_equal(p["ids"], np.arange(2 * NX * NY))
    for name, a in zip(("h", "hu", "hv"), ref):
        assert np.array_equal(p[name], a), name


@pytest.mark.parametrize("order,world,shape,halo_kernels", [
    (4, 2, (48, 36), False), (4, 3, (48, 36), False), (4, 5, (40, 40), False), (2, 8, (64, 32), False),
    (6, 4, (24, 20), False), (8, 2, (12, 10), False), (5, 3, (30, 24), False), (4, 4, (48, 36), True),
    (3, 3, (40, 30), True), (6, 3, (24, 20), True)])
def test_overlapped_two_chain_schedule_matches_single_domain(order, world, shape, halo_kernels, monkeypatch):
    """The production stage schedule (interior elements on the compute stream, pack / exchange /
    unpack / boundary elements on the exchange stream, meeting one stage later through alternating
    events) with every part of the split on this one GPU and device-to-device copies as the
    transport: after 23 stages (not a multiple of 5, odd) the assembled state must equal the
    single-domain run bit for bit -- any missing dependency between the chains shows up as a
    difference. The boundary kernel does the halo staging itself (reads the received records, writes the
    send records); halo_kernels=True (BDG_SW2D_HALO_KERNELS=1) keeps the separate pack / unpack kernels."""
    if halo_kernels:
        monkeypatch.setenv("BDG_SW2D_HALO_KERNELS", "1")
    # N >= 5: the strip runs on the throughput form of the kernel the interior uses (identical arithmetic, so the comparison
    # can be bit for bit); the default latency form of the strip kernel (one field per wave) is covered by
    # test_strip_kernel_matches_single_domain_to_round_off
    monkeypatch.setenv("BDG_SW2D_STRIP_THROUGHPUT", "1")
    import blitzdg_amd.pyblitzdg as dg
    from blitzdg_amd import sw2d
    from blitzdg_amd.halo import LocalGroupSw2d
    mesh = dg.MeshManager()
    mesh.buildBoxMesh(*shape)
    nodes = dg.TriangleNodesProvisioner(order, mesh)
    ctx = nodes.dgContext()
    single = sw2d.Sw2dSolver(nodes=nodes)
    q0 = _fields(ctx.x, ctx.y)
    single.setState(*q0)
    dt = 0.5 * single.computeDt(0.65)[0]
    group = LocalGroupSw2d(mesh, order, world)
    try:
        assert sum(p.num_owned for p in group.plans) == mesh.numElements
        assert all(p.num_halo > 0 and p.num_interior < p.num_owned for p in group.plans)
        group.set_global_state(*q0)
        for chunk in (1, 2, 7, 13):                 # several calls: the chains re-join between them
            group.lserk4_stages(dt, chunk)
            single.lserk4Stages(dt, chunk)
        got, ref = group.gather_state(), single.getState()
        for a, b in zip(got, ref):
            assert np.array_equal(a, b)
        assert np.abs(ref[1] - q0[1]).max() > 1e-4
    finally:
        group.close()


@pytest.mark.parametrize("order,world,shape", [(5, 3, (30, 24)), (6, 4, (24, 20)), (7, 2, (16, 12)), (8, 2, (12, 10)), (8, 5, (20, 16))])
def test_strip_kernel_matches_single_domain_to_round_off(order, world, shape):
    """Default partition-boundary kernel at N >= 5: sw2d_strip_mfma3_kernel, a tile shared by three waves (one conserved
    field each) with the halo staging folded in. Its operands are formed by other instruction sequences than the interior
    kernel's (FMA contraction differs), so the assembled state agrees with the single-domain run to round-off, not bit for
    bit: <= 1e-12 after 23 stages. (At N = 8 a first version was off by 1e-6: the compiler had placed the destination of a
    chain's first matrix instruction over its A operand -- see mfma_zero and tests/test_isa_hazards.py.)"""
    import blitzdg_amd.pyblitzdg as dg
    from blitzdg_amd import sw2d
    from blitzdg_amd.halo import LocalGroupSw2d
    mesh = dg.MeshManager()
    mesh.buildBoxMesh(*shape)
    nodes = dg.TriangleNodesProvisioner(order, mesh)
    ctx = nodes.dgContext()
    single = sw2d.Sw2dSolver(nodes=nodes)
    q0 = _fields(ctx.x, ctx.y)
    single.setState(*q0)
    dt = 0.5 * single.computeDt(0.65)[0]
    group = LocalGroupSw2d(mesh, order, world)
    try:
        group.set_global_state(*q0)
        for chunk in (1, 2, 7, 13):
            group.lserk4_stages(dt, chunk)
            single.lserk4Stages(dt, chunk)
        got, ref = group.gather_state(), single.getState()
        for a, b in zip(got, ref):
            assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max()
        assert any(not np.array_equal(a, b) for a, b in zip(got, ref))   # it IS another kernel than the interior's
        assert np.abs(ref[1] - q0[1]).max() > 1e-4
    finally:
        group.close()


def _run_bench(args, env_extra):
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    out = launch([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_distributed_entry_point_with_native_rccl_single_rank():
    """bench.py the way the driver launches it for N > 1 (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment), here with one rank: the library's own RCCL communicator, file rendezvous, two-stream stage
    loop, global dt reduction, one JSON line from rank 0."""
    d = _run_bench(["--gpus", "1", "--steps", "20", "--warmup", "5", "--cells", "60x40"],
                   {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(_free_port()), "BDG_BENCH_FORCE_DISTRIBUTED": "1"})
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["config"]["elements"] == 4800
    assert abs(d["config"]["mass_relative_drift"]) < 1e-13  # walls everywhere: mass is conserved to round-off


def test_bench_loopback_rehearsal_of_an_eight_way_split():
    """The RCCL exchange path with real message sizes and neighbour counts, on one GPU: rank r's share of an
    8-way split with every neighbour exchange a send-to-self (timing aid; the state is not physical)."""
    d = _run_bench(["--rehearse-world", "8", "--steps", "30", "--warmup", "5", "--cells", "160x80"],
                   {"BDG_REHEARSE_RANKS": "0,5"})
    assert d["rehearsal"] and d["world"] == 8 and [r["rank"] for r in d["ranks"]] == [0, 5]
    for r in d["ranks"]:
        assert r["owned"] == 3200 and r["ghost"] > 0 and r["peers"] >= 1 and 0 < r["ms_per_stage"] < 5


@pytest.mark.parametrize("order", [4, 6])
def test_loopback_rccl_result_is_independent_of_event_flags_and_halo_staging(order, monkeypatch):
    """Real RCCL (send-to-self of the true message sizes; rank 5's share of an 8-way split) through the asynchronous
    two-chain schedule: the final state must be bit-identical whether the four cross-stream events carry the
    system-scope fence (default) or not (BDG_SW2D_EVENT_NOFENCE=1), and whether the halo staging is folded into the
    boundary kernel or done by separate pack / unpack kernels -- a missing dependency between the chains would show as a
    difference between these runs (the ghost values are this rank's own elements: not physical, but deterministic)."""
    from blitzdg_amd.halo import NativeDistributedSw2d
    monkeypatch.setenv("BDG_SW2D_STRIP_THROUGHPUT", "1")       # N >= 5: both stagings on the same arithmetic
    monkeypatch.setenv("BDG_SW2D_EVENT_SYNC", "1")             # (this test is about the event form; the counters: next test)
    results = {}
    for nofence in ("0", "1"):
        for halo_kernels in ("", "1"):
            monkeypatch.setenv("BDG_SW2D_EVENT_NOFENCE", nofence)
            if halo_kernels:
                monkeypatch.setenv("BDG_SW2D_HALO_KERNELS", "1")
            else:
                monkeypatch.delenv("BDG_SW2D_HALO_KERNELS", raising=False)
            results[(nofence, halo_kernels)] = _loopback_run(order)
    ref = results[("0", "")]
    assert np.isfinite(ref[0]).all() and np.abs(ref[1]).max() > 0
    for key, got in results.items():
        for a, b in zip(got, ref):
            assert np.array_equal(a, b), key


def _loopback_run(order, shape=(160, 80), rank=5, chunks=(3, 11, 23)):
    from blitzdg_amd.halo import NativeDistributedSw2d
    d = NativeDistributedSw2d.box(shape[0], shape[1], order, rank, 8, device=0, loopback=True)
    try:
        d.set_initial_state(_fields)
        dt = 0.25 * d.compute_dt(0.65)
        for chunk in chunks:
            d.lserk4_stages(dt, chunk)
        d.barrier()
        return d.owned_state()[1:]
    finally:
        d.close()


@pytest.mark.parametrize("order,shape", [(2, (160, 80)), (4, (160, 80)), (4, (400, 200)), (5, (160, 80)), (6, (160, 80)), (8, (120, 80))])
def test_loopback_rccl_in_kernel_dependencies_equal_the_event_form(order, shape, monkeypatch):
    """Round 4: the two chains of an exchanged stage meet through device counters polled INSIDE the kernels (the ring of interior
    tiles next to the partition boundary waits for the previous boundary launch, the boundary launch for the previous ring) instead
    of through events on the queues. Through real RCCL (send-to-self, rank 5's share of an 8-way split; several calls, so that the
    counters carry over): the final state must equal the event form's bit for bit -- the same kernels' arithmetic, only the
    dependencies differ; a wait that is missing, or a hand-off that reads stale lines, shows as a difference. (400 x 200 cells:
    a 20 000-element share whose interior launch fills the chip for several rounds while the boundary launches run beside it.)"""
    monkeypatch.delenv("BDG_SW2D_STRIP_THROUGHPUT", raising=False)
    monkeypatch.delenv("BDG_SW2D_HALO_KERNELS", raising=False)
    monkeypatch.setenv("BDG_SW2D_EVENT_SYNC", "1")
    ref = _loopback_run(order, shape)
    monkeypatch.delenv("BDG_SW2D_EVENT_SYNC", raising=False)
    for attempt in range(3):                     # (the interleaving of the two chains differs from run to run)
        got = _loopback_run(order, shape)
        assert np.isfinite(ref[0]).all() and np.abs(ref[1]).max() > 0
        for a, b in zip(got, ref):
            assert np.array_equal(a, b), (order, attempt)


_BOUNDED_WAIT_SCRIPT = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np
from blitzdg_amd._capi import BdgError
from test_dist_gpu import _loopback_run
try:
    got = _loopback_run(8, (120, 80), chunks=(2,))
    np.savez(sys.argv[2], *got)
    print("finished")
except BdgError as e:
    print("reported:", e)
"""


def test_in_kernel_wait_is_bounded_and_reported(tmp_path):
    """The safety net of the in-kernel dependencies: with the interior launch pinned to ALL 256 compute units at N = 8 (BDG_SW2D_INTERIOR_CAP=256;
    its workgroups hold 120 KB of LDS each, so the boundary launch's workgroups fit nowhere) the ring tiles can wait for a boundary launch that
    cannot start. The wait is bounded (sync_wait gives up after 2^21 polls, marks the run and goes on), so the device never hangs: either the
    hardware found room anyway and the result equals the event form's, or the next synchronisation reports the time-out as an error.
    (Child processes: the cap is read once per process.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("BDG_SW2D_STRIP_THROUGHPUT", "BDG_SW2D_HALO_KERNELS", "BDG_SW2D_EVENT_SYNC")}
    ref = launch([sys.executable, "-c", _BOUNDED_WAIT_SCRIPT, ROOT, str(tmp_path / "ref.npz")], env=dict(env, BDG_SW2D_EVENT_SYNC="1"), timeout=300)
    assert ref.returncode == 0 and "finished" in ref.stdout, ref.stdout + ref.stderr
    out = launch([sys.executable, "-c", _BOUNDED_WAIT_SCRIPT, ROOT, str(tmp_path / "capped.npz")], env=dict(env, BDG_SW2D_INTERIOR_CAP="256"), timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    if "reported:" in out.stdout:
        assert "timed out" in out.stdout
        return
    a, b = np.load(tmp_path / "ref.npz"), np.load(tmp_path / "capped.npz")
    for key in a.files:
        assert np.array_equal(a[key], b[key])


# ---- the real multi-process path (one process per rank, the library's own communicator and stage loop) with a
# ---- file-based stand-in for librccl.so, so that the ranks can share the single GPU of a test box

# (fixture mock_rccl: tests/conftest.py)


def _native_worker(rank, world, port, env, out_dir, order=ORDER, strip="throughput"):
    import faulthandler
    faulthandler.enable()
    sys.path.insert(0, ROOT)
    os.environ.update(env)
    if strip == "throughput":                       # N >= 5: the strip on the interior's own kernel form (bit-for-bit comparison)
        os.environ["BDG_SW2D_STRIP_THROUGHPUT"] = "1"
    else:                                           # the default: sw2d_strip_mfma3_kernel (round-off comparison)
        os.environ.pop("BDG_SW2D_STRIP_THROUGHPUT", None)
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": "0", "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port)})
    from blitzdg_amd.halo import NativeDistributedSw2d
    d = NativeDistributedSw2d.box(NX, NY, order, rank, world, device=0)
    try:
        d.set_initial_state(_fields)
        dt = 0.5 * d.compute_dt(0.65)          # global minimum through the communicator's all-reduce
        mass0 = d.allreduce_sum(d.owned_mass())
        d.lserk4_stages(dt, 3)                 # several calls: the two chains re-join between them
        d.lserk4_stages(dt, NSTAGES - 3)
        d.barrier()
        mass1 = d.allreduce_sum(d.owned_mass())
        ids, h, hu, hv = d.owned_state()
        np.savez(os.path.join(out_dir, f"native{rank}.npz"), ids=ids, h=h, hu=hu, hv=hv, dt=dt, mass0=mass0, mass1=mass1,
                 total=d.global_elements, **d.halo_counts())
    finally:
        d.close()


@pytest.mark.parametrize("world,order,strip", [(2, 4, "throughput"), (3, 4, "throughput"), (4, 4, "throughput"),
                                               (2, 7, "throughput"), (3, 6, "default"), (2, 8, "default"),
                                               (3, 4, "default"), (4, 3, "default"), (4, 5, "default")])
def test_native_multi_process_path_matches_single_domain(tmp_path, world, order, strip, mock_rccl):
    """One process per rank exactly as under torchrun -- file rendezvous of the communicator id,
    bdg_sw2d_comm_init, the library's two-chain stage loop with grouped send / receive on the exchange
    stream, all-reduces for dt and mass -- with only librccl.so replaced (tests/mock_rccl). Owned states must
    equal the single-domain run bit for bit, and total mass must be conserved to round-off. strip="default" runs the
    partition-boundary strips of N >= 5 on the kernel a production run uses (sw2d_strip_mfma3_kernel: other instruction
    sequences than the interior's, so the comparison is to round-off, <= 1e-12)."""
    launch_ranks("test_dist_gpu", "_native_worker", world, (world, _free_port(), mock_rccl, str(tmp_path), order, strip))
    parts = [np.load(tmp_path / f"native{r}.npz") for r in range(world)]
    dt = float(parts[0]["dt"])
    assert all(float(p["dt"]) == dt for p in parts)
    ref = _single_domain(dt, order)
    seen = np.zeros(int(parts[0]["total"]), dtype=int)
    for p in parts:
        ids = p["ids"]
        seen[ids] += 1
        assert int(p["ghost"]) > 0 and int(p["interior"]) < int(p["owned"])
        for name, full in zip(("h", "hu", "hv"), ref):
            if strip == "throughput":
                assert np.array_equal(p[name], full[:, ids]), f"{name} differs on a rank"
            elif order <= 4:      # in-kernel dependencies (the default), matrix-core kernel on both sides: still bit for bit
                assert np.array_equal(p[name], full[:, ids]), f"{name} differs on a rank"
            else:
                assert np.abs(p[name] - full[:, ids]).max() <= 1e-12 * np.abs(full).max(), f"{name} differs on a rank"
        assert abs(float(p["mass1"]) - float(p["mass0"])) < 1e-13 * abs(float(p["mass0"]))
    assert (seen == 1).all()


# ---- variants B and D partitioned: one process per rank, the library's own communicator; variant B's global speed is
# ---- one ncclAllReduce(max) per RHS evaluation on the solver's stream

def _open_left_edge(mesh):
    verts = np.asarray(mesh.vertices).reshape(-1, 3)
    etov = np.asarray(mesh.elements).reshape(-1, 3)
    bc = np.asarray(mesh.bcType).reshape(-1, 3).copy()
    xmin = verts[:, 0].min()
    for f, (a, b) in enumerate(((0, 1), (1, 2), (2, 0))):
        left = (np.abs(verts[etov[:, a], 0] - xmin) < 1e-12) & (np.abs(verts[etov[:, b], 0] - xmin) < 1e-12)
        bc[left & (bc[:, f] == 3), f] = 2
    mesh.setBCType(bc)


def _bed(x, y):
    return 12.0 + 1.5 * x - 0.8 * y * y + 0.3 * np.sin(3 * x) * np.cos(2 * y)


def _b_state(x, y):
    return _bed(x, y) + 0.4 * np.exp(-6 * x * x - 6 * y * y), 0.8 * np.sin(3 * x + 1) * np.cos(2 * y), 0.8 * np.cos(2 * x - y)


def _d_state(x, y):
    h, hu, hv = _fields(x, y)
    return h, hu, hv, h * (1.0 + 0.3 * np.sin(2 * x) * np.cos(3 * y))


def _d_sources(x, y):
    return {"zx": -0.5 + 0 * x, "zy": 0.5 * y, "f": 1e-1 * (1.0 + 0.5 * y), "CD": 2.5e-2}


B_KW = dict(CD=2.5e-3, f=1.0070e-4)
B_TIME = 0.37 * 3600 * 12.42
VARIANT_MESH = (26, 18)


def _variant_worker(rank, world, port, env, out_dir, variant, order):
    import faulthandler
    faulthandler.enable()
    sys.path.insert(0, ROOT)
    os.environ.update(env)
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": "0", "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port)})
    import blitzdg_amd.pyblitzdg as dg
    from blitzdg_amd.halo import NativeDistributedSw2d, build_plan
    mesh = dg.MeshManager()
    mesh.buildBoxMesh(*VARIANT_MESH)
    if variant == "B":
        _open_left_edge(mesh)
    mesh.partitionMesh(world)
    plan = build_plan(mesh.elements, mesh.vertices, mesh.EToE, mesh.elementPartitionMap, rank, world, bctype=mesh.bcType)
    del mesh
    if variant == "B":
        d = NativeDistributedSw2d(plan, order, device=0)
    else:
        d = NativeDistributedSw2d(plan, order, device=0, filter_args=(0.9 * order, order), fields=4, sources=_d_sources)
    try:
        ctx = d.nodes.dgContext()
        if variant == "B":
            d.enable_variant_b(_bed, **B_KW)
            d.solver.time = B_TIME
            d.solver.setState(*_b_state(ctx.x, ctx.y))
            dt = 0.5 * d.compute_dt(0.25)
            d.step_ssprk2(dt, 2, sponge=3.0)
            d.lserk4_stages(dt, 7)                 # LSERK4 with the all-reduced speed too
            d.step_ssprk2(dt, 1)
        else:
            d.solver.setState4(*_d_state(ctx.x, ctx.y))
            dt = 0.5 * d.compute_dt(0.4)
            d.step_rk2(dt, 3, filter=True)
            d.lserk4_stages(dt, 6)                 # the overlapped two-chain schedule with four fields
        d.barrier()
        out = d.owned_state()
        np.savez(os.path.join(out_dir, f"{variant}{rank}.npz"), ids=out[0], dt=dt, **{f"q{i}": a for i, a in enumerate(out[1:])},
                 **d.halo_counts())
    finally:
        d.close()


@pytest.mark.parametrize("variant,world,order", [("B", 2, 4), ("B", 3, 6), ("D", 2, 3), ("D", 3, 6)])
def test_variants_b_and_d_partitioned_match_single_domain(tmp_path, variant, world, order, mock_rccl, monkeypatch):
    """Variant B (star states, tide boundary, ONE global Lax-Friedrichs speed = an 8-byte all-reduce per evaluation)
    under Heun + sponge and LSERK4, and variant D (tracer, Coriolis array, drag, bed slope, filter) under midpoint RK2
    and the overlapped LSERK4 schedule, on 2 and 3 rank processes through the stand-in transport: owned states equal
    the single-domain run bit for bit (unrolled kernels at N <= 4, matrix-core kernels at N = 6)."""
    import blitzdg_amd.pyblitzdg as dg
    from blitzdg_amd import sw2d
    # single-domain reference with the separate speed pass (the same reduction kernel the ranks run before their
    # all-reduce; the unrolled kernel's fused next-evaluation speed agrees with it to round-off only)
    monkeypatch.setenv("BDG_SW2D_SPEED_PASS", "1")
    launch_ranks("test_dist_gpu", "_variant_worker", world, (world, _free_port(), mock_rccl, str(tmp_path), variant, order))
    parts = [np.load(tmp_path / f"{variant}{r}.npz") for r in range(world)]
    dt = float(parts[0]["dt"])
    assert all(float(p["dt"]) == dt for p in parts)
    mesh = dg.MeshManager()
    mesh.buildBoxMesh(*VARIANT_MESH)
    if variant == "B":
        _open_left_edge(mesh)
        nodes = dg.TriangleNodesProvisioner(order, mesh)
        ctx = nodes.dgContext()
        s = sw2d.Sw2dSolver(nodes=nodes, flags=sw2d.KEEP_ORDER)
        H = _bed(ctx.x, ctx.y)
        Hx, Hy = nodes.bedSlopes(H)
        s.enableVariantB(H, Hx, Hy, mapO=np.array(ctx.BCmap.get(2, []), dtype=np.int32), **B_KW)
        s.time = B_TIME
        s.setState(*_b_state(ctx.x, ctx.y))
        assert 0.5 * s.computeDt(0.25)[0] == dt
        s.stepSSPRK2(dt, 2, sponge=3.0)
        s.lserk4Stages(dt, 7)
        s.stepSSPRK2(dt, 1)
        ref = s.getState()
    else:
        nodes = dg.TriangleNodesProvisioner(order, mesh)
        nodes.buildFilter(0.9 * order, order)
        ctx = nodes.dgContext()
        s = sw2d.Sw2dSolver(nodes=nodes, flags=sw2d.KEEP_ORDER, fields=4, sources=_d_sources(ctx.x, ctx.y))
        s.setState4(*_d_state(ctx.x, ctx.y))
        assert 0.5 * s.computeDt(0.4)[0] == dt
        s.stepRK2(dt, 3, filter=True)
        s.lserk4Stages(dt, 6)
        ref = s.getState4()
    seen = np.zeros(mesh.numElements, dtype=int)
    for p in parts:
        ids = p["ids"]
        seen[ids] += 1
        assert int(p["ghost"]) > 0
        for i, full in enumerate(ref):
            assert np.array_equal(p[f"q{i}"], full[:, ids]), f"field {i} differs on a rank"
    assert (seen == 1).all()
    assert np.abs(ref[1] - (_b_state(ctx.x, ctx.y) if variant == "B" else _d_state(ctx.x, ctx.y))[1]).max() > 1e-6


def test_bench_two_ranks_through_the_mock_transport(mock_rccl):
    """bench.py as the driver launches it for N = 2 (two processes, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*),
    both on this GPU through the stand-in transport: rank 0 prints the one JSON line, rank 1 nothing."""
    import json
    port = str(_free_port())
    cmds = []
    for rank in (0, 1):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, **mock_rccl)
        cmds.append(([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "15", "--warmup", "5",
                      "--cells", "60x40"], env, ROOT))
    outs = launch_group(cmds, timeout=600)
    assert all(o.returncode == 0 for o in outs), "".join(o.stdout[-1500:] + o.stderr[-1500:] for o in outs)
    lines0 = [ln for ln in outs[0].stdout.splitlines() if ln.startswith("{")]
    assert len(lines0) == 1 and not [ln for ln in outs[1].stdout.splitlines() if ln.startswith("{")]
    d = json.loads(lines0[0])
    assert d["n_gpus"] == 2 and d["steps"] == 15 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["config"]["rank0_partition"]["ghost"] > 0
    assert abs(d["config"]["mass_relative_drift"]) < 1e-13


def test_bench_plain_gpus_flag_starts_its_own_ranks(mock_rccl):
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: the parent starts the two rank processes itself
    (LOCAL_RANK 1 folded onto the only device, stand-in transport) and relays rank 0's one line with n_gpus = 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    import json
    out = launch([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "4",
                  "--cells", "60x40"], timeout=900, cwd=ROOT, env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="0", **mock_rccl))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 12 and d["config"]["elements"] == 4800
    assert d["config"]["rank0_partition"]["ghost"] > 0 and abs(d["config"]["mass_relative_drift"]) < 1e-13


def test_bench_under_torch_distributed_run_exact_driver_command(mock_rccl):
    """The driver's own N > 1 command line: python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W (here N = 2, LOCAL_RANK 1
    folded onto the only device, stand-in transport)."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **mock_rccl)
    out = launch([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                  "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2",
                  "--steps", "12", "--warmup", "4", "--cells", "80x50"], timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 12 and d["warmup"] == 4 and d["config"]["elements"] == 8000
    assert abs(d["config"]["mass_relative_drift"]) < 1e-13
