"""Multi-GPU execution of the sw2d path: element partition + ghost-element halo.

The reference has no parallel path: ``MeshManager::partitionMesh`` only produces METIS
partition vectors that nothing consumes (reference src/MeshManager.cpp:491-544). This module
is the MI355X-native consumer of such a vector: one process per GPU, each owning the elements
``epart == rank`` plus one layer of ghost elements, renumbered locally as

    [ interior | partition-boundary | ghost ]

so that (a) ``vmapP`` of an owned element never needs a branch -- ghosts are ordinary elements
of the local mesh -- and (b) a stage is: pack boundary state -> RCCL send/recv to the
neighbour ranks (xGMI is point-to-point: each pair has its own link, messages are tens of KB,
latency-bound) -> meanwhile the interior elements compute -> unpack ghosts -> boundary
elements compute. No collective sits on the per-stage path; the adaptive time step needs one
8-byte all-reduce per step.

``HaloPlan`` / ``build_plan`` are pure NumPy (tested on CPU with gloo, world_size 2);
``DistributedSw2d`` drives the HIP solver through the C ABI and torch.distributed.
"""
from dataclasses import dataclass, field

import numpy as np


@dataclass
class HaloPlan:
    rank: int
    world: int
    own_global: np.ndarray        # (K_own,)  global ids of owned elements, local order [interior|boundary]
    halo_global: np.ndarray       # (K_halo,) global ids of ghost elements, grouped by owner, ascending
    num_interior: int
    send_local: np.ndarray        # (n_send,) local slots of owned elements to send, grouped by peer
    send_slices: list = field(default_factory=list)   # [(peer, start, count)] into send_local
    recv_slices: list = field(default_factory=list)   # [(peer, start, count)] into the ghost slots
    local_EToV: np.ndarray = None  # (K_own + K_halo, 3) local vertex ids
    local_verts: np.ndarray = None  # (Nv_loc, 3)
    local_bctype: np.ndarray = None  # (K_loc, 3) BC tags of the GLOBAL mesh for owned elements, 0 for ghosts

    @property
    def num_owned(self):
        return int(self.own_global.size)

    @property
    def num_halo(self):
        return int(self.halo_global.size)

    @property
    def local_to_global(self):
        return np.concatenate([self.own_global, self.halo_global])


def build_plan(EToV, Vert, EToE, epart, rank, world, bctype=None):
    """Halo plan of `rank` from the global mesh tables and an element partition vector."""
    EToV = np.asarray(EToV).reshape(-1, 3)
    EToE = np.asarray(EToE).reshape(-1, 3)
    Vert = np.asarray(Vert, dtype=np.float64)
    epart = np.asarray(epart).reshape(-1)
    own = np.flatnonzero(epart == rank)
    nb = EToE[own]                    # (K_own, 3) neighbour elements (self on physical boundaries)
    nb_owner = epart[nb]
    remote = nb_owner != rank
    is_boundary = remote.any(axis=1)
    # interior elements that touch a partition-boundary element (the "ring") come last among the interior ones: they are the only
    # interior elements whose traces depend on the boundary launch of the previous stage, and a partitioned stage lets its
    # interior launch start on the others while that launch may still be running
    boundary_flag = np.zeros(epart.size, dtype=bool)
    boundary_flag[own[is_boundary]] = True
    is_ring = ~is_boundary & boundary_flag[nb].any(axis=1)
    own_order = np.concatenate([own[~is_boundary & ~is_ring], own[is_ring], own[is_boundary]])
    num_interior = int((~is_boundary).sum())
    slot_of = np.full(epart.size, -1, dtype=np.int64)
    slot_of[own_order] = np.arange(own_order.size)

    # ghosts: remote neighbours, grouped by owner rank then ascending global id
    ghost_ids = np.unique(nb[remote])
    ghost_owner = epart[ghost_ids]
    order = np.lexsort((ghost_ids, ghost_owner))
    ghost_ids, ghost_owner = ghost_ids[order], ghost_owner[order]
    recv_slices, start = [], 0
    for peer in np.unique(ghost_owner):
        cnt = int((ghost_owner == peer).sum())
        recv_slices.append((int(peer), start, cnt))
        start += cnt

    # what each peer needs from us: our elements adjacent to one of theirs, ascending global id
    send_local, send_slices, start = [], [], 0
    for peer in np.unique(nb_owner[remote]):
        mine = np.unique(own[(nb_owner == peer).any(axis=1)])
        send_local.append(slot_of[mine])
        send_slices.append((int(peer), start, int(mine.size)))
        start += mine.size
    send_local = np.concatenate(send_local).astype(np.int32) if send_local else np.zeros(0, dtype=np.int32)

    loc2glob = np.concatenate([own_order, ghost_ids])
    ev = EToV[loc2glob]
    verts_used, inv = np.unique(ev, return_inverse=True)
    local_EToV = inv.reshape(ev.shape).astype(np.int32)
    local_verts = Vert.reshape(-1, Vert.shape[-1] if Vert.ndim == 2 else 3)[verts_used]
    local_bc = np.zeros((loc2glob.size, 3), dtype=np.int32)
    if bctype is not None:
        local_bc[:own_order.size] = np.asarray(bctype).reshape(-1, 3)[own_order]
    return HaloPlan(rank=rank, world=world, own_global=own_order.astype(np.int64), halo_global=ghost_ids.astype(np.int64),
                    num_interior=num_interior, send_local=send_local, send_slices=send_slices,
                    recv_slices=recv_slices, local_EToV=local_EToV, local_verts=local_verts, local_bctype=local_bc)


def exchange_ops(plan, sendbuf, recvbuf, dist):
    """P2P ops of one halo exchange. sendbuf: (n_send, width), recvbuf: (K_halo, width) tensors
    (CPU with gloo, CUDA with nccl/RCCL). Returns the list for dist.batch_isend_irecv."""
    ops = []
    for peer, start, count in plan.recv_slices:
        ops.append(dist.P2POp(dist.irecv, recvbuf[start:start + count], peer))
    for peer, start, count in plan.send_slices:
        ops.append(dist.P2POp(dist.isend, sendbuf[start:start + count], peer))
    return ops


def build_local_mesh(plan):
    """MeshManager of the rank-local mesh (owned + ghost elements). Ghost elements' outer faces
    become walls of the local mesh; they are never updated, so that is immaterial. Owned
    elements keep the BC tags of the global mesh."""
    from . import pyblitzdg as dg
    mesh = dg.MeshManager()
    mesh.buildMesh(plan.local_EToV, plan.local_verts)
    got = mesh.elements
    if not np.array_equal(got, plan.local_EToV):
        raise RuntimeError("local mesh was re-oriented: the global mesh must be counter-clockwise")
    bc = mesh.bcType
    bc[:plan.num_owned] = plan.local_bctype[:plan.num_owned]
    mesh.setBCType(bc)
    return mesh


def _launcher_start_time(pid):
    """Wall-clock start of process `pid` (the launcher all ranks are children of), or 0.0."""
    import os
    try:
        with open(f"/proc/{pid}/stat") as f:
            ticks = int(f.read().rsplit(")", 1)[1].split()[19])
        with open("/proc/stat") as f:
            btime = next(int(ln.split()[1]) for ln in f if ln.startswith("btime"))
        return btime + ticks / os.sysconf("SC_CLK_TCK")
    except (OSError, ValueError, StopIteration, IndexError):
        return 0.0


def file_rendezvous(rank, world, make_id, timeout=300.0):
    """Hands rank 0's 128-byte RCCL id to every rank of a single-node job through a file in a
    private (0700, owned by this user) directory under /tmp. The name is unique per launch: all ranks
    are children of one launcher process (its pid) and share MASTER_PORT and, where the launcher sets
    one, a nonce (bench.py's own launcher: BDG_LAUNCH_NONCE; torchrun: TORCHELASTIC_RUN_ID). Rank 0
    unlinks whatever an earlier, crashed launch left under that name and creates the file with
    O_EXCL, mode 0600; the record carries rank 0's clock, and a reader only accepts a record written
    after its launcher started (a recycled launcher pid cannot match a stale record). Rank 0 removes
    the file once every rank has joined (see NativeDistributedSw2d)."""
    import os
    import struct
    import time
    base = os.path.join(os.environ.get("BDG_RENDEZVOUS_DIR", "/tmp"), f"bdg_rccl_{os.getuid()}")
    os.makedirs(base, mode=0o700, exist_ok=True)
    st = os.lstat(base)
    if st.st_uid != os.getuid() or (st.st_mode & 0o077) or not os.path.isdir(base) or os.path.islink(base):
        raise RuntimeError(f"{base} is not a private directory of this user; refusing to exchange the RCCL id there")
    nonce = os.environ.get("BDG_LAUNCH_NONCE") or os.environ.get("TORCHELASTIC_RUN_ID", "none")
    path = os.path.join(base, f"id_{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}_{world}_{nonce}")
    if rank == 0:
        uid = make_id()
        try:
            os.unlink(path)
        except FileNotFoundError:
            pass
        tmp = f"{path}.{os.getpid()}.tmp"
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(struct.pack("<d", time.time()) + uid)
        os.replace(tmp, path)  # atomic: readers never see a partial record
        return uid, path
    not_before = _launcher_start_time(os.getppid()) - 1.0
    deadline = time.time() + timeout
    while time.time() < deadline:
        try:
            with open(path, "rb") as f:
                rec = f.read()
            if len(rec) == 136 and struct.unpack("<d", rec[:8])[0] >= not_before:
                return rec[8:], path
        except FileNotFoundError:
            pass
        time.sleep(0.02)
    raise TimeoutError(f"rank {rank}: no fresh RCCL id at {path} after {timeout} s")


class NativeDistributedSw2d:
    """sw2d on `world` GPUs with the exchange driven entirely by the C++ library: grouped
    ncclSend/ncclRecv on a communication stream, overlapped with the interior elements, whole
    stage loops issued by one C call (no per-stage Python). PyTorch is not involved."""

    def __init__(self, plan, order, g=9.81, device=0, unique_id=None, loopback=False, filter_args=None, fields=3,
                 sources=None):
        """loopback=True: schedule rehearsal on ONE GPU -- this process computes `plan.rank`'s share
        of a `plan.world`-way split, and every neighbour exchange is a send-to-self of the same size
        (ghost values are then not the neighbours' -- timing only, never results)."""
        import ctypes
        import os

        from . import pyblitzdg as dg
        from . import sw2d
        from ._capi import byref, c_double, check, lib, ptr

        self._lib, self._check, self._byref, self._c_double = lib, check, byref, c_double
        self.plan, self.order = plan, order
        self.mesh = build_local_mesh(plan)
        self.nodes = dg.TriangleNodesProvisioner(order, self.mesh)
        if filter_args is not None:
            self.nodes.buildFilter(*filter_args)
        # fields = 4 / sources: variants C / D (tracer; Coriolis, drag, bed slope), sources as a function
        # (x, y) -> dict(zx=, zy=, f=, CD=) of the rank-local coordinates
        if fields == 4 or sources is not None:
            ctx = self.nodes.dgContext()
            src = sources(ctx.x, ctx.y) if callable(sources) else (sources or {"f": 0.0, "CD": 0.0})
            self.solver = sw2d.Sw2dSolver(nodes=self.nodes, g=g, device=device, flags=sw2d.KEEP_ORDER, fields=fields,
                                          sources=src)
        else:
            self.solver = sw2d.Sw2dSolver(nodes=self.nodes, g=g, device=device, flags=sw2d.KEEP_ORDER)
        self.Np = self.solver.Np
        send = np.ascontiguousarray(plan.send_local, dtype=np.int32)
        check(lib.bdg_sw2d_set_partition(self.solver._h, plan.num_interior, plan.num_owned, ptr(send), send.size))

        id_path = None
        comm_rank, comm_world = plan.rank, plan.world
        if loopback:
            comm_rank, comm_world = 0, 1
            buf = ctypes.create_string_buffer(128)
            check(lib.bdg_comm_unique_id(buf, 128))
            unique_id = buf.raw
        if unique_id is None:
            def make_id():
                buf = ctypes.create_string_buffer(128)
                check(lib.bdg_comm_unique_id(buf, 128))
                return buf.raw
            unique_id, id_path = file_rendezvous(plan.rank, plan.world, make_id)
        recv_of = {peer: (start, count) for peer, start, count in plan.recv_slices}
        send_of = {peer: (start, count) for peer, start, count in plan.send_slices}
        peers = sorted(set(recv_of) | set(send_of))
        arr = lambda vals: np.ascontiguousarray(vals, dtype=np.int32)  # noqa: E731
        pr = arr(peers)
        ss, sc = arr([send_of.get(p, (0, 0))[0] for p in peers]), arr([send_of.get(p, (0, 0))[1] for p in peers])
        rs, rc = arr([recv_of.get(p, (0, 0))[0] for p in peers]), arr([recv_of.get(p, (0, 0))[1] for p in peers])
        if loopback:
            pr = arr([0] * len(peers))
            sc = rc = np.minimum(sc, rc)
        idbuf = ctypes.create_string_buffer(unique_id, 128)
        check(lib.bdg_sw2d_comm_init(self.solver._h, comm_rank, comm_world, idbuf, ptr(pr), ptr(ss), ptr(sc), ptr(rs),
                                     ptr(rc), len(peers)))
        self.barrier()
        if id_path is not None and plan.rank == 0:
            try:
                os.remove(id_path)
            except OSError:
                pass
        self.global_elements = None

    @classmethod
    def box(cls, nx, ny, order, rank, world, g=9.81, device=0, x0=-1.0, x1=1.0, y0=-1.0, y1=1.0, loopback=False):
        from . import pyblitzdg as dg
        mesh = dg.MeshManager()
        mesh.buildBoxMesh(nx, ny, x0, x1, y0, y1)
        mesh.partitionMesh(world)
        plan = build_plan(mesh.elements, mesh.vertices, mesh.EToE, mesh.elementPartitionMap, rank, world,
                          bctype=mesh.bcType)
        total = mesh.numElements
        del mesh
        self = cls(plan, order, g=g, device=device, loopback=loopback)
        self.global_elements = total
        return self

    def close(self):
        solver, self.solver = getattr(self, "solver", None), None
        if solver is not None:
            solver.close()

    def halo_counts(self):
        return {"owned": self.plan.num_owned, "interior": self.plan.num_interior, "ghost": self.plan.num_halo,
                "sent": int(self.plan.send_local.size), "peers": len(self.plan.recv_slices)}

    def set_initial_state(self, fn):
        ctx = self.nodes.dgContext()
        self.solver.setState(*fn(ctx.x, ctx.y))

    def compute_dt(self, CFL):
        dt, em = self._c_double(), self._c_double()
        self._check(self._lib.bdg_sw2d_compute_dt_global(self.solver._h, float(CFL), self._byref(dt), self._byref(em)))
        return dt.value

    def lserk4_stages(self, dt, nstages):
        self._check(self._lib.bdg_sw2d_lserk4_stages_exchanged(self.solver._h, float(dt), int(nstages)))

    def step_rk2(self, dt, nsteps=1, filter=True):
        """Midpoint RK2 (+ filter) of the sw2d drivers, ghosts refreshed before each of the two evaluations."""
        self._check(self._lib.bdg_sw2d_step_rk2_exchanged(self.solver._h, float(dt), int(nsteps), int(bool(filter))))

    def step_ssprk2(self, dt, nsteps=1, filter=False, sponge=0.0):
        """Heun + sponge of the variant-B driver; with variant B enabled every evaluation also reduces the global
        Lax-Friedrichs speed over all ranks."""
        self._check(self._lib.bdg_sw2d_step_ssprk2_exchanged(self.solver._h, float(dt), int(nsteps), int(bool(filter)),
                                                             float(sponge)))

    def enable_variant_b(self, fields_fn, **kwargs):
        """Variant B on this rank's part: fields_fn(x, y) -> H of the rank-local nodes (owned and ghost elements);
        bed slopes from this rank's operators, open boundary = this rank's BCmap[2] (owned elements keep the global
        mesh's tags). kwargs: CD, f, tide*, as Sw2dSolver.enableVariantB."""
        ctx = self.nodes.dgContext()
        H = fields_fn(ctx.x, ctx.y)
        Hx, Hy = self.nodes.bedSlopes(H)
        mapO = np.array(ctx.BCmap.get(2, []), dtype=np.int32)
        self.solver.enableVariantB(H, Hx, Hy, mapO=mapO, **kwargs)
        return H

    def allreduce_max(self, value):
        out = self._c_double()
        self._check(self._lib.bdg_sw2d_allreduce_max(self.solver._h, float(value), self._byref(out)))
        return out.value

    def allreduce_sum(self, value):
        out = self._c_double()
        self._check(self._lib.bdg_sw2d_allreduce_sum(self.solver._h, float(value), self._byref(out)))
        return out.value

    def barrier(self):
        self._check(self._lib.bdg_sw2d_barrier(self.solver._h))

    def owned_mass(self, fn=None):
        """Integral of h over this rank's owned elements (nodal quadrature with the mass matrix' row sums);
        of the resident state, or of ``fn(x, y)[0]`` when a state function is given."""
        ctx = self.nodes.dgContext()
        V = ctx.V
        w = np.linalg.inv(V @ V.T).sum(axis=0)
        n = self.plan.num_owned
        h = (self.solver.getState()[0] if fn is None else fn(ctx.x, ctx.y)[0])[:, :n]
        return float((w @ h * ctx.J[0, :n]).sum())

    def owned_state(self):
        q = self.solver.getState4() if self.solver.fields == 4 else self.solver.getState()
        n = self.plan.num_owned
        return (self.plan.own_global,) + tuple(a[:, :n] for a in q)


class DistributedSw2d:
    """sw2d on `world` GPUs (one process each): owned elements on this rank's device, ghosts
    refreshed every stage over torch.distributed (backend "nccl" = RCCL over xGMI).
    Kept as the torch-plumbed alternative to NativeDistributedSw2d and as the way to run
    several ranks on ONE GPU in tests (host-staged `gloo`)."""

    def __init__(self, plan, order, g=9.81, device=0, filter_args=None):
        import torch
        import torch.distributed as dist

        from . import pyblitzdg as dg
        from . import sw2d
        from ._capi import check, lib, ptr

        self._torch, self._dist, self._lib, self._check = torch, dist, lib, check
        self.plan = plan
        self.order = order
        self.mesh = build_local_mesh(plan)
        self.nodes = dg.TriangleNodesProvisioner(order, self.mesh)
        if filter_args is not None:
            self.nodes.buildFilter(*filter_args)
        self.solver = sw2d.Sw2dSolver(nodes=self.nodes, g=g, device=device, flags=sw2d.KEEP_ORDER)
        self.Np, self.K_loc = self.solver.Np, self.solver.K
        send = np.ascontiguousarray(plan.send_local, dtype=np.int32)
        check(lib.bdg_sw2d_set_partition(self.solver._h, plan.num_interior, plan.num_owned, ptr(send), send.size))
        width = 3 * self.Np
        dev = torch.device("cuda", device)
        self.sendbuf = torch.zeros((max(send.size, 1), width), dtype=torch.float64, device=dev)
        self.recvbuf = torch.zeros((max(plan.num_halo, 1), width), dtype=torch.float64, device=dev)
        # Transport: RCCL moves the device buffers directly. With the CPU-only `gloo` backend (used
        # to exercise several ranks on ONE GPU in tests) the buffers are staged through host memory.
        self.host_staged = dist.get_backend() != "nccl"
        if self.host_staged:
            self.send_host = torch.zeros_like(self.sendbuf, device="cpu").pin_memory()
            self.recv_host = torch.zeros_like(self.recvbuf, device="cpu").pin_memory()
        # the solver launches on its own stream: make it torch's current stream around the
        # exchange so RCCL orders itself after the pack and before the unpack
        self.stream = torch.cuda.ExternalStream(lib.bdg_sw2d_stream(self.solver._h), device=dev)
        counts = torch.tensor([plan.num_owned], dtype=torch.int64, device="cpu" if self.host_staged else dev)
        dist.all_reduce(counts)
        self.global_elements = int(counts.item())

    @classmethod
    def box(cls, nx, ny, order, g=9.81, device=0, x0=-1.0, x1=1.0, y0=-1.0, y1=1.0):
        """Every rank builds the (cheap) global box mesh and the same RCB partition, then keeps
        only its part."""
        import torch.distributed as dist

        from . import pyblitzdg as dg
        rank, world = dist.get_rank(), dist.get_world_size()
        mesh = dg.MeshManager()
        mesh.buildBoxMesh(nx, ny, x0, x1, y0, y1)
        mesh.partitionMesh(world)
        plan = build_plan(mesh.elements, mesh.vertices, mesh.EToE, mesh.elementPartitionMap, rank, world,
                          bctype=mesh.bcType)
        del mesh
        return cls(plan, order, g=g, device=device)

    def close(self):
        """Release in a safe order: torch's wrapper of the solver's stream and the exchange
        buffers first, the solver (which owns and destroys that stream) last."""
        solver = getattr(self, "solver", None)
        if solver is None:
            return
        try:
            solver.synchronize()
        except Exception:
            pass
        self.stream = None
        self.sendbuf = self.recvbuf = None
        if getattr(self, "host_staged", False):
            self.send_host = self.recv_host = None
        self.solver = None
        solver.close()

    def __del__(self):
        self.close()

    def halo_counts(self):
        return {"owned": self.plan.num_owned, "interior": self.plan.num_interior, "ghost": self.plan.num_halo,
                "sent": int(self.plan.send_local.size)}

    def set_initial_state(self, fn):
        ctx = self.nodes.dgContext()
        h, hu, hv = fn(ctx.x, ctx.y)   # owned and ghost elements alike: ghosts start current
        self.solver.setState(h, hu, hv)

    def compute_dt(self, CFL):
        torch, dist = self._torch, self._dist
        dt, _ = self.solver.computeDt(CFL)   # reduction over owned elements only
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if self.host_staged else self.sendbuf.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t.item())

    def lserk4_stage(self, dt):
        torch, dist, lib, check = self._torch, self._dist, self._lib, self._check
        h = self.solver._h
        if self.host_staged:
            return self._lserk4_stage_host_staged(dt)
        with torch.cuda.stream(self.stream):
            check(lib.bdg_sw2d_halo_pack(h, self.sendbuf.data_ptr()))
            ops = exchange_ops(self.plan, self.sendbuf, self.recvbuf, dist)
            works = dist.batch_isend_irecv(ops) if ops else []
            check(lib.bdg_sw2d_lserk4_stage_part(h, dt, 0))       # interior: overlaps the exchange
            for w in works:
                w.wait()                                            # stream-level wait, host does not block
            check(lib.bdg_sw2d_halo_unpack(h, self.recvbuf.data_ptr()))
            check(lib.bdg_sw2d_lserk4_stage_part(h, dt, 1))       # partition-boundary elements, advance

    def _lserk4_stage_host_staged(self, dt):
        torch, dist, lib, check = self._torch, self._dist, self._lib, self._check
        h = self.solver._h
        with torch.cuda.stream(self.stream):
            check(lib.bdg_sw2d_halo_pack(h, self.sendbuf.data_ptr()))
            self.send_host.copy_(self.sendbuf, non_blocking=True)
            check(lib.bdg_sw2d_lserk4_stage_part(h, dt, 0))
            self.stream.synchronize()
            ops = exchange_ops(self.plan, self.send_host, self.recv_host, dist)
            for w in (dist.batch_isend_irecv(ops) if ops else []):
                w.wait()
            self.recvbuf.copy_(self.recv_host, non_blocking=True)
            check(lib.bdg_sw2d_halo_unpack(h, self.recvbuf.data_ptr()))
            check(lib.bdg_sw2d_lserk4_stage_part(h, dt, 1))

    def owned_state(self):
        """(global ids, h, hu, hv) of the owned elements."""
        h, hu, hv = self.solver.getState()
        n = self.plan.num_owned
        return self.plan.own_global, h[:, :n], hu[:, :n], hv[:, :n]


class LocalGroupSw2d:
    """All parts of an element split inside ONE process (one per GPU of the node via `devices`, or
    several on one GPU): same rank-local meshes and the same overlapped two-chain stage schedule as
    NativeDistributedSw2d, with device-to-device copies instead of RCCL. Besides being the
    single-process multi-GPU mode, this is how the schedule is verified against a single-domain
    run on a one-GPU box."""

    def __init__(self, mesh, order, world, g=9.81, devices=None):
        import ctypes

        from . import pyblitzdg as dg
        from . import sw2d
        from ._capi import c_void_p, check, lib, ptr

        self._lib, self._check = lib, check
        mesh.partitionMesh(world)
        epart = mesh.elementPartitionMap
        self.world, self.order = world, order
        self.global_elements = mesh.numElements
        self.plans, self.meshes, self.nodes, self.solvers = [], [], [], []
        arr = lambda vals: np.ascontiguousarray(vals, dtype=np.int32)  # noqa: E731
        for r in range(world):
            plan = build_plan(mesh.elements, mesh.vertices, mesh.EToE, epart, r, world, bctype=mesh.bcType)
            lm = build_local_mesh(plan)
            nd = dg.TriangleNodesProvisioner(order, lm)
            dev = devices[r] if devices is not None else 0
            s = sw2d.Sw2dSolver(nodes=nd, g=g, device=dev, flags=sw2d.KEEP_ORDER)
            send = arr(plan.send_local)
            check(lib.bdg_sw2d_set_partition(s._h, plan.num_interior, plan.num_owned, ptr(send), send.size))
            recv_of = {peer: (start, count) for peer, start, count in plan.recv_slices}
            send_of = {peer: (start, count) for peer, start, count in plan.send_slices}
            peers = sorted(set(recv_of) | set(send_of))
            pr = arr(peers)
            ss, sc = arr([send_of.get(p, (0, 0))[0] for p in peers]), arr([send_of.get(p, (0, 0))[1] for p in peers])
            rs, rc = arr([recv_of.get(p, (0, 0))[0] for p in peers]), arr([recv_of.get(p, (0, 0))[1] for p in peers])
            check(lib.bdg_sw2d_local_peers(s._h, r, ptr(pr), ptr(ss), ptr(sc), ptr(rs), ptr(rc), len(peers)))
            self.plans.append(plan)
            self.meshes.append(lm)
            self.nodes.append(nd)
            self.solvers.append(s)
        self._handles = (c_void_p * world)(*[s._h for s in self.solvers])
        self._ctypes = ctypes

    def close(self):
        for s in self.solvers:
            s.close()
        self.solvers = []

    def set_initial_state(self, fn):
        for nd, s in zip(self.nodes, self.solvers):
            ctx = nd.dgContext()
            s.setState(*fn(ctx.x, ctx.y))

    def set_global_state(self, h, hu, hv):
        for plan, s in zip(self.plans, self.solvers):
            ids = plan.local_to_global
            s.setState(h[:, ids], hu[:, ids], hv[:, ids])

    def lserk4_stages(self, dt, nstages):
        self._check(self._lib.bdg_sw2d_group_lserk4_stages(self._handles, self.world, float(dt), int(nstages)))

    def synchronize(self):
        for s in self.solvers:
            s.synchronize()

    def gather_state(self):
        """Global (Np, K) arrays assembled from every part's owned elements."""
        Np = self.solvers[0].Np
        out = [np.empty((Np, self.global_elements)) for _ in range(3)]
        for plan, s in zip(self.plans, self.solvers):
            q = s.getState()
            n = plan.num_owned
            for o, a in zip(out, q):
                o[:, plan.own_global] = a[:, :n]
        return tuple(out)
