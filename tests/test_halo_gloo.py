"""The N>1 path on CPU: world_size-2 (and 3) `gloo` runs of the halo plan + exchange used by
blitzdg_amd.halo.DistributedSw2d, with the CPU oracle standing in for the HIP kernels.

Each rank owns a part of the mesh plus one layer of ghost elements; per LSERK4 stage it packs
its partition-boundary elements, exchanges them with torch.distributed (the same
exchange_ops the GPU path hands to RCCL), unpacks into the ghost slots and advances. The
owned results must equal a single-domain run BIT FOR BIT (per-element arithmetic does not
depend on the numbering), which pins the partition, the local renumbering, the ghost ordering
on both sides and the wall flags of the local meshes. The rank processes are started by tests/launcher.py
(conftest.launch_ranks): PyTorch lives in the workers only, never in the pytest process.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fields(x, y):
    h = 10.0 + np.exp(-10 * x * x - 10 * y * y)
    hu = 0.1 * np.sin(3 * x + 1) * np.cos(2 * y)
    hv = 0.1 * np.cos(2 * x) * np.sin(3 * y - 1)
    return h, hu, hv


def _tables(nodes):
    ctx = nodes.dgContext()
    t = {k: getattr(ctx, k) for k in
         ("Dr", "Ds", "Lift", "rx", "sx", "ry", "sy", "nx", "ny", "Fscale", "vmapM", "vmapP", "x", "y")}
    t["mapW"] = np.array(ctx.BCmap.get(3, []), dtype=np.int32)
    return t


def _worker(rank, world, port, mesh_args, order, nstages, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import blitzdg_amd.pyblitzdg as dg
    from blitzdg_amd.halo import build_local_mesh, build_plan, exchange_ops
    from oracle import Sw2dOracle

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        mesh = dg.MeshManager()
        mesh.buildBoxMesh(*mesh_args[:2], shuffleSeed=mesh_args[2])
        mesh.partitionMesh(world)
        plan = build_plan(mesh.elements, mesh.vertices, mesh.EToE, mesh.elementPartitionMap, rank, world,
                          bctype=mesh.bcType)
        local = build_local_mesh(plan)
        nodes = dg.TriangleNodesProvisioner(order, local)
        t = _tables(nodes)
        o = Sw2dOracle(t["Dr"], t["Ds"], t["Lift"], t["rx"], t["sx"], t["ry"], t["sy"], t["nx"], t["ny"],
                       t["Fscale"], t["vmapM"], t["vmapP"], t["mapW"])
        q = list(_fields(t["x"], t["y"]))
        res = [np.zeros_like(q[0]) for _ in range(3)]
        Np, n_own, n_halo = q[0].shape[0], plan.num_owned, plan.num_halo
        # poison the ghosts: only the exchange may make them right
        for f in q:
            f[:, n_own:] = np.nan
        dt = 2e-3
        for s in range(nstages):
            state = np.concatenate(q, axis=0)                            # (3Np, K_loc)
            sendbuf = torch.from_numpy(np.ascontiguousarray(state[:, plan.send_local].T))   # element-major
            recvbuf = torch.zeros((max(n_halo, 1), 3 * Np), dtype=torch.float64)
            ops = exchange_ops(plan, sendbuf, recvbuf, dist)
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            ghosts = recvbuf.numpy()[:n_halo].T                          # (3Np, K_halo)
            for c in range(3):
                q[c][:, n_own:] = ghosts[c * Np:(c + 1) * Np]
            h, hu, hv, res = o.lserk4_stages(q[0], q[1], q[2], res, dt, s, 1)
            q = [h, hu, hv]
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ids=plan.own_global, h=q[0][:, :n_own],
                 hu=q[1][:, :n_own], hv=q[2][:, :n_own], interior=plan.num_interior, halo=n_halo,
                 sent=plan.send_local.size)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mesh_args,order", [(2, (6, 4, 0), 3), (2, (5, 7, 12345), 2), (3, (8, 6, 77), 4)])
def test_partitioned_lserk4_equals_single_domain(tmp_path, world, mesh_args, order):
    import blitzdg_amd.pyblitzdg as dg
    from conftest import launch_ranks
    from oracle import Sw2dOracle

    nstages = 7
    port = _free_port()
    launch_ranks("test_halo_gloo", "_worker", world, (world, port, mesh_args, order, nstages, str(tmp_path)))

    mesh = dg.MeshManager()
    mesh.buildBoxMesh(*mesh_args[:2], shuffleSeed=mesh_args[2])
    nodes = dg.TriangleNodesProvisioner(order, mesh)
    t = _tables(nodes)
    o = Sw2dOracle(t["Dr"], t["Ds"], t["Lift"], t["rx"], t["sx"], t["ry"], t["sy"], t["nx"], t["ny"], t["Fscale"],
                   t["vmapM"], t["vmapP"], t["mapW"])
    h, hu, hv = _fields(t["x"], t["y"])
    zero = [np.zeros_like(h) for _ in range(3)]
    rh, rhu, rhv, _ = o.lserk4_stages(h, hu, hv, zero, 2e-3, 0, nstages)

    seen = np.zeros(mesh.numElements, dtype=int)
    for r in range(world):
        d = np.load(tmp_path / f"rank{r}.npz")
        ids = d["ids"]
        seen[ids] += 1
        assert d["halo"] > 0 and d["sent"] > 0 and 0 <= d["interior"] < ids.size
        assert np.array_equal(d["h"], rh[:, ids])
        assert np.array_equal(d["hu"], rhu[:, ids])
        assert np.array_equal(d["hv"], rhv[:, ids])
    assert (seen == 1).all()  # every element owned exactly once


def _open_left_edge(mesh):
    """Faces on x = xmin re-tagged Out (2), as the variant-B driver does before its second buildBCHash."""
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


B_PHYS = dict(g=9.81, f=1.0070e-4, CD=2.5e-3, time=0.37 * 3600 * 12.42)


def _b_worker(rank, world, port, mesh_args, order, nsteps, dt, out_dir):
    """Variant B on a partition: per RHS evaluation the ghosts of the evaluated state are exchanged and the one global
    Lax-Friedrichs speed is the all-reduced maximum of the ranks' owned-element maxima (gloo all_reduce MAX)."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import blitzdg_amd.pyblitzdg as dg
    from blitzdg_amd.halo import build_local_mesh, build_plan, exchange_ops
    from oracle import oracle_np as onp

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        mesh = dg.MeshManager()
        mesh.buildBoxMesh(*mesh_args[:2], shuffleSeed=mesh_args[2])
        _open_left_edge(mesh)
        mesh.partitionMesh(world)
        plan = build_plan(mesh.elements, mesh.vertices, mesh.EToE, mesh.elementPartitionMap, rank, world,
                          bctype=mesh.bcType)
        nodes = dg.TriangleNodesProvisioner(order, build_local_mesh(plan))
        ctx = nodes.dgContext()
        t = _tables(nodes)
        mapO = np.array(ctx.BCmap.get(2, []), dtype=np.int32)
        H = _bed(t["x"], t["y"])
        Hx, Hy = nodes.bedSlopes(H)
        q = list(_b_state(t["x"], t["y"]))
        Np, n_own, n_halo = q[0].shape[0], plan.num_owned, plan.num_halo

        def refresh(state):
            for a in state:
                a[:, n_own:] = np.nan
            full = np.concatenate(state, axis=0)
            sendbuf = torch.from_numpy(np.ascontiguousarray(full[:, plan.send_local].T))
            recvbuf = torch.zeros((max(n_halo, 1), 3 * Np), dtype=torch.float64)
            for w in dist.batch_isend_irecv(exchange_ops(plan, sendbuf, recvbuf, dist)):
                w.wait()
            ghosts = recvbuf.numpy()[:n_halo].T
            for c in range(3):
                state[c][:, n_own:] = ghosts[c * Np:(c + 1) * Np]

        def allmax(v):
            tv = torch.tensor([v], dtype=torch.float64)
            dist.all_reduce(tv, op=dist.ReduceOp.MAX)
            return float(tv.item())

        def rhs(state):
            refresh(state)
            return onp.sw2d_rhs_b(*state, H, Hx, Hy, B_PHYS["g"], B_PHYS["f"], B_PHYS["CD"], B_PHYS["time"], t, mapO,
                                  owned=n_own, reduce_speed=allmax)
        for _ in range(nsteps):                                          # Heun, both evaluations at the old time level
            r = rhs(q)
            q1 = [a + dt * b for a, b in zip(q, r)]
            r = rhs(q1)
            q = [0.5 * (a + a1 + dt * b) for a, a1, b in zip(q, q1, r)]
        np.savez(os.path.join(out_dir, f"rankB{rank}.npz"), ids=plan.own_global, h=q[0][:, :n_own], hu=q[1][:, :n_own],
                 hv=q[2][:, :n_own], nout=mapO.size)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mesh_args,order", [(2, (6, 5, 0), 3), (3, (7, 6, 31), 2)])
def test_partitioned_variant_b_with_all_reduced_speed_equals_single_domain(tmp_path, world, mesh_args, order):
    """Variant B's ONE global speed (reference src/sw2d/main.cpp:414) in a partitioned run = max over ranks of the
    owned-element maxima: Heun steps on 2 and 3 ranks equal the single-domain restatement bit for bit."""
    import blitzdg_amd.pyblitzdg as dg
    from conftest import launch_ranks
    from oracle import oracle_np as onp

    nsteps, dt = 3, 1e-3
    launch_ranks("test_halo_gloo", "_b_worker", world, (world, _free_port(), mesh_args, order, nsteps, dt, str(tmp_path)))
    mesh = dg.MeshManager()
    mesh.buildBoxMesh(*mesh_args[:2], shuffleSeed=mesh_args[2])
    _open_left_edge(mesh)
    nodes = dg.TriangleNodesProvisioner(order, mesh)
    ctx = nodes.dgContext()
    t = _tables(nodes)
    mapO = np.array(ctx.BCmap.get(2, []), dtype=np.int32)
    H = _bed(t["x"], t["y"])
    Hx, Hy = nodes.bedSlopes(H)
    q = list(_b_state(t["x"], t["y"]))
    for _ in range(nsteps):
        r = onp.sw2d_rhs_b(*q, H, Hx, Hy, B_PHYS["g"], B_PHYS["f"], B_PHYS["CD"], B_PHYS["time"], t, mapO)
        q1 = [a + dt * b for a, b in zip(q, r)]
        r = onp.sw2d_rhs_b(*q1, H, Hx, Hy, B_PHYS["g"], B_PHYS["f"], B_PHYS["CD"], B_PHYS["time"], t, mapO)
        q = [0.5 * (a + a1 + dt * b) for a, a1, b in zip(q, q1, r)]
    seen, nout = np.zeros(mesh.numElements, dtype=int), 0
    for r in range(world):
        d = np.load(tmp_path / f"rankB{r}.npz")
        seen[d["ids"]] += 1
        nout += int(d["nout"])
        for name, full in zip(("h", "hu", "hv"), q):
            assert np.array_equal(d[name], full[:, d["ids"]]), name
    assert (seen == 1).all() and nout >= mapO.size and mapO.size > 0


def test_plan_is_consistent_between_ranks():
    """Send lists and ghost lists of every pair of ranks name the same elements in the same
    order; interior elements have no remote neighbour."""
    import blitzdg_amd.pyblitzdg as dg
    from blitzdg_amd.halo import build_plan

    mesh = dg.MeshManager()
    mesh.buildBoxMesh(12, 9, shuffleSeed=5)
    world = 4
    mesh.partitionMesh(world)
    ep, E = mesh.elementPartitionMap, mesh.EToE
    plans = [build_plan(mesh.elements, mesh.vertices, E, ep, r, world, bctype=mesh.bcType) for r in range(world)]
    for p in plans:
        assert (ep[p.own_global] == p.rank).all() and (ep[p.halo_global] != p.rank).all()
        interior = p.own_global[:p.num_interior]
        assert (ep[E[interior]] == p.rank).all()
        boundary = p.own_global[p.num_interior:]
        assert (ep[E[boundary]] != p.rank).any(axis=1).all()
        for peer, start, count in p.send_slices:
            sent = p.own_global[p.send_local[start:start + count]]
            q = plans[peer]
            match = [(s, c) for (pr, s, c) in q.recv_slices if pr == p.rank]
            assert len(match) == 1 and match[0][1] == count
            assert np.array_equal(q.halo_global[match[0][0]:match[0][0] + count], sent)
        # local mesh is orientation-preserving and compact
        assert p.local_EToV.max() == p.local_verts.shape[0] - 1


@pytest.mark.parametrize("world,mesh_args", [(2, (9, 7, 0)), (3, (11, 8, 5)), (4, (12, 10, 0)), (8, (24, 12, 7))])
def test_plan_orders_owned_elements_deep_ring_boundary(world, mesh_args):
    """The invariants the partitioned stage builds on (halo.build_plan), for every rank of a split: the owned
    elements come as [deep interior | ring | partition boundary]; a boundary element has a ghost neighbour, an interior one has none
    (bdg_sw2d_set_partition refuses otherwise); ring elements are exactly the interior ones next to a boundary element, so a deep
    element touches neither a ghost nor a boundary element; the send list holds boundary elements only, and every rank's receive
    ranges are its neighbours' send ranges in the same order."""
    sys.path.insert(0, ROOT)
    import blitzdg_amd.pyblitzdg as dg
    from blitzdg_amd.halo import build_plan
    mesh = dg.MeshManager()
    mesh.buildBoxMesh(*mesh_args[:2], shuffleSeed=mesh_args[2])
    mesh.partitionMesh(world)
    EToE = np.asarray(mesh.EToE).reshape(-1, 3)
    epart = np.asarray(mesh.elementPartitionMap).reshape(-1)
    plans = [build_plan(mesh.elements, mesh.vertices, mesh.EToE, epart, r, world, bctype=mesh.bcType) for r in range(world)]
    assert sorted(np.concatenate([p.own_global for p in plans]).tolist()) == list(range(EToE.shape[0]))
    for p in plans:
        own = p.own_global
        assert (epart[own] == p.rank).all()
        remote = epart[EToE[own]] != p.rank                                  # (K_own, 3)
        is_boundary = remote.any(axis=1)
        assert not is_boundary[:p.num_interior].any() and is_boundary[p.num_interior:].all()
        boundary_ids = set(own[p.num_interior:].tolist())
        touches = np.array([[int(n) in boundary_ids for n in row] for row in EToE[own[:p.num_interior]]]).reshape(-1, 3).any(axis=1)
        first_ring = int(np.argmax(touches)) if touches.any() else p.num_interior
        assert not touches[:first_ring].any() and touches[first_ring:].all()    # [deep | ring]: one switch, no mixing
        assert set(p.send_local.tolist()) <= set(range(p.num_interior, p.num_owned))
        assert set(p.halo_global.tolist()) == set(EToE[own][remote].tolist())
        for peer, start, count in p.recv_slices:
            back = [(s, c) for (q, s, c) in plans[peer].send_slices if q == p.rank]
            assert len(back) == 1 and back[0][1] == count
            sent = plans[peer].own_global[plans[peer].send_local[back[0][0]:back[0][0] + count]]
            assert np.array_equal(sent, p.halo_global[start:start + count])
