"""The N>1 path on CPU: world_size-2 (and 3) `gloo` runs of the halo plan + exchange used by
blitzdg_amd.halo.DistributedSw2d, with the CPU oracle standing in for the HIP kernels.

Each rank owns a part of the mesh plus one layer of ghost elements; per LSERK4 stage it packs
its partition-boundary elements, exchanges them with torch.distributed (the same
exchange_ops the GPU path hands to RCCL), unpacks into the ghost slots and advances. The
owned results must equal a single-domain run BIT FOR BIT (per-element arithmetic does not
depend on the numbering), which pins the partition, the local renumbering, the ghost ordering
on both sides and the wall flags of the local meshes.
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
    import torch.multiprocessing as mp

    import blitzdg_amd.pyblitzdg as dg
    from oracle import Sw2dOracle

    nstages = 7
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, mesh_args, order, nstages, str(tmp_path)), nprocs=world,
                       join=True, start_method="spawn")

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
