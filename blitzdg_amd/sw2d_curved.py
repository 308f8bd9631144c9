"""Curved / over-integrated sw2d on the MI355X: host-side mirror of the reference's curved RHS path.

``Sw2dCurvedSolver`` wraps the ``bdg_sw2d_curved_*`` C ABI (include/blitzdg_hip.h): every table the reference's
``swhelpers.rhs.sw2dComputeRHS_curved`` (swhelpers/rhs.py:6-176) reads from its context arguments is an input.
``examples``-style use, the reference driver's loop (sw2d_curved.py:246-277) with the state resident::

    solver = Sw2dCurvedSolver(ctx, cub_ctx, gauss_ctx, curvedEls, J, gmapM, gmapP, g=g, zx=zx, zy=zy, f=f, CD=CD)
    solver.setState(h, hu, hv, hN)
    solver.stepRK2(dt, nsteps, filter=True)
    h, hu, hv, hN = solver.getState()

There is no CPU fallback: without the HIP library / a GPU, construction raises.
"""
import numpy as np

from . import _capi as C
from ._capi import POINTER, byref, c_double, c_float, c_void_p, check, lib


class Sw2dCurvedSolver:
    def __init__(self, ctx, cub_ctx, gauss_ctx, curvedEls, J, gmapM, gmapP, g=9.81, zx=None, zy=None, f=0.0, CD=0.0,
                 device=0):
        """ctx: DGContext2D-like (V, numLocalPoints, numElements, optional filter); cub_ctx: V, Dr, Ds, W, rx, ry, sx,
        sy, MMChol; gauss_ctx: Interp, W, nx, ny, BCmap (tag 3 = reflective wall); J: (Np, K) nodal Jacobian;
        gmapM / gmapP: flat Gauss-node maps; f and CD: scalars or (Np, K) arrays."""
        Np, K = int(ctx.numLocalPoints), int(ctx.numElements)
        order = int(round((np.sqrt(8 * Np + 1) - 3) / 2))
        if (order + 1) * (order + 2) // 2 != Np:
            raise ValueError(f"numLocalPoints = {Np} is not a triangle node count")
        cubV = C.as_f64(cub_ctx.V)
        Ncub = cubV.shape[0]
        gI = C.as_f64(gauss_ctx.Interp)
        if gI.shape[0] % 3:
            raise ValueError("gauss_ctx.Interp must have 3*NGauss rows")
        NG = gI.shape[0] // 3
        keep = []  # arrays must outlive the create call

        def f64(a, shape, name):
            arr = C.as_f64(a, shape, name)
            keep.append(arr)
            return C.ptr(arr)

        def i32(a, n, name):
            arr = C.as_i32(np.asarray(a).reshape(-1))
            if n is not None and arr.size != n:
                raise ValueError(f"{name}: expected {n} entries, got {arr.size}")
            keep.append(arr)
            return C.ptr(arr) if arr.size else None, arr.size

        filt = getattr(ctx, "filter", None)
        d = C.Sw2dCurvedDesc()
        d.order, d.num_elements, d.num_cub, d.num_gauss = order, K, Ncub, NG
        d.V = f64(ctx.V, (Np, Np), "ctx.V")
        # a provisioner whose buildFilter was never called hands out an all-zero (or empty) table: no filter
        d.Filter = f64(filt, (Np, Np), "ctx.filter") if filt is not None and np.size(filt) == Np * Np and np.any(filt) else None
        d.J = f64(J, (Np, K), "J")
        d.cubV = f64(cubV, (Ncub, Np), "cub_ctx.V")
        d.cubDr, d.cubDs = f64(cub_ctx.Dr, (Ncub, Np), "cub_ctx.Dr"), f64(cub_ctx.Ds, (Ncub, Np), "cub_ctx.Ds")
        for name in ("W", "rx", "ry", "sx", "sy"):
            setattr(d, "cub" + name, f64(getattr(cub_ctx, name), (Ncub, K), "cub_ctx." + name))
        d.gaussInterp = f64(gI, (3 * NG, Np), "gauss_ctx.Interp")
        d.gaussW, d.gaussnx, d.gaussny = (f64(getattr(gauss_ctx, n), (3 * NG, K), "gauss_ctx." + n) for n in ("W", "nx", "ny"))
        d.gmapM, _ = i32(gmapM, 3 * NG * K, "gmapM")
        d.gmapP, _ = i32(gmapP, 3 * NG * K, "gmapP")
        d.gmapW, d.num_wall = i32(gauss_ctx.BCmap.get(3, []), None, "gauss_ctx.BCmap[3]")
        d.curvedEls, d.num_curved = i32(list(curvedEls), None, "curvedEls")
        d.MMChol = f64(cub_ctx.MMChol, (Np, Np, K), "cub_ctx.MMChol") if d.num_curved else None
        d.zx = f64(zx, (Np, K), "zx") if zx is not None else None
        d.zy = f64(zy, (Np, K), "zy") if zy is not None else None
        if np.ndim(f) == 0:
            d.coriolis, d.coriolis_const = None, float(f)
        else:
            d.coriolis, d.coriolis_const = f64(f, (Np, K), "f"), 0.0
        if np.ndim(CD) == 0:
            d.drag, d.drag_const = None, float(CD)
        else:
            d.drag, d.drag_const = f64(CD, (Np, K), "CD"), 0.0
        d.g, d.device, d.flags = float(g), int(device), 0
        h = c_void_p()
        check(lib.bdg_sw2d_curved_create(byref(d), byref(h)))
        self._h = h
        self.Np, self.K, self.order, self.Ncub, self.NGauss = Np, K, order, Ncub, NG
        self.hasFilter = d.Filter is not None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.bdg_sw2d_curved_destroy(h)

    close = __del__

    def _fields(self, arrays, names=("h", "hu", "hv", "hN")):
        return [C.as_f64(a, (self.Np, self.K), n) for a, n in zip(arrays, names)]

    def computeRHS(self, h, hu, hv, hN, filter=False):
        """(RHS1, RHS2, RHS3, RHS4) of the reference function for host fields (upload, three kernels, download)."""
        q = self._fields((h, hu, hv, hN))
        out = [np.empty((self.Np, self.K)) for _ in range(4)]
        check(lib.bdg_sw2d_curved_rhs(self._h, *[C.ptr(a) for a in q], *[C.ptr(o) for o in out], int(bool(filter))))
        return tuple(out)

    def setState(self, h, hu, hv, hN):
        q = self._fields((h, hu, hv, hN))
        check(lib.bdg_sw2d_curved_set_state(self._h, *[C.ptr(a) for a in q]))

    def getState(self):
        out = [np.empty((self.Np, self.K)) for _ in range(4)]
        check(lib.bdg_sw2d_curved_get_state(self._h, *[C.ptr(o) for o in out]))
        return tuple(out)

    def stepRK2(self, dt, nsteps=1, filter=True):
        """The reference driver's loop body (sw2d_curved.py:246-277), nsteps times, state resident."""
        check(lib.bdg_sw2d_curved_step_rk2(self._h, float(dt), int(nsteps), int(bool(filter))))

    def lserk4Stages(self, dt, nstages):
        check(lib.bdg_sw2d_curved_lserk4_stages(self._h, float(dt), int(nstages)))

    def timeRK2(self, dt, nsteps, filter=True):
        """Milliseconds per RHS evaluation (HIP events on the solver's stream), averaged over 2*nsteps."""
        ms = c_float()
        check(lib.bdg_sw2d_curved_time_rk2(self._h, float(dt), int(nsteps), int(bool(filter)), byref(ms)))
        return ms.value

    def synchronize(self):
        check(lib.bdg_sw2d_curved_synchronize(self._h))

    # ---- element-partitioned runs: ghost columns in and out, the two halves of a step
    def getElements(self, first, count, intermediate=False):
        """Columns [first, first + count) of the resident state (or of the RK2 intermediate state) as a (4 Np, count) array."""
        out = np.empty((4 * self.Np, int(count)))
        check(lib.bdg_sw2d_curved_get_elements(self._h, int(bool(intermediate)), int(first), int(count),
                                               out.ctypes.data_as(POINTER(c_double))))
        return out

    def setElements(self, first, values, intermediate=False):
        v = np.ascontiguousarray(values, dtype=np.float64)
        if v.ndim != 2 or v.shape[0] != 4 * self.Np:
            raise ValueError(f"expected a (4 Np = {4 * self.Np}, count) array")
        check(lib.bdg_sw2d_curved_set_elements(self._h, int(bool(intermediate)), int(first), v.shape[1],
                                               v.ctypes.data_as(POINTER(c_double))))

    def rk2Phase(self, dt, phase, filter=True):
        """phase 0: intermediate = state + dt/2 RHS(state); phase 1: state += dt RHS(intermediate)."""
        check(lib.bdg_sw2d_curved_rk2_phase(self._h, float(dt), int(phase), int(bool(filter))))

    deviceBytes = property(lambda self: lib.bdg_sw2d_curved_device_bytes(self._h))
    bytesPerElement = property(lambda self: lib.bdg_sw2d_curved_bytes_per_element(self._h))
    usesNodalTraces = property(lambda self: lib.bdg_sw2d_curved_form(self._h) == 1)


class DistributedSw2dCurved:
    """The curved / over-integrated solver on an element partition, one process per rank (no reference analogue: the reference is
    single-process). Each rank holds its owned elements plus one layer of ghost elements of the mesh (``halo.build_plan``); the
    contexts are built by the provisioner on that local mesh from the caller's deformation of the node coordinates, and before
    EVERY RHS evaluation the ghost columns of the state that evaluation reads are refreshed from their owners over
    ``torch.distributed`` (gloo: host tensors; nccl = RCCL: device tensors). Ghost elements are computed like any other (their
    outer faces are walls of the local mesh) and their results are discarded -- overwritten by the next exchange.

    A functional path (host-staged columns); NativeDistributedSw2dCurved below drives the same exchange from the library over
    RCCL, device to device."""

    def __init__(self, plan, order, deform, dist, g=9.81, filter_args=None, sources=None, device=0):
        """deform(x0, y0) -> (x, y): node coordinates of the curved mesh from the straight ones (elements it moves are listed in
        curvedEls); sources(x, y) -> dict with any of zx, zy, f, CD (arrays or scalars)."""
        from . import pyblitzdg as dg
        from .halo import build_local_mesh
        self.plan, self.dist, self.order = plan, dist, order
        mesh = build_local_mesh(plan)
        nodes = dg.TriangleNodesProvisioner(order, mesh)
        if filter_args is not None:
            nodes.buildFilter(*filter_args)
        ctx = nodes.dgContext()
        x0, y0 = ctx.x, ctx.y
        x, y = deform(x0, y0)
        curved = np.where((np.abs(x - x0) + np.abs(y - y0)).max(axis=0) > 0)[0]
        nodes.setCoordinates(x, y)
        J = (ctx.Dr @ x) * (ctx.Ds @ y) - (ctx.Ds @ x) * (ctx.Dr @ y)
        gauss = nodes.buildGaussFaceNodes(2 * (order + 1))
        cub = nodes.buildCubatureVolumeMesh(3 * (order + 1))
        src = sources(x, y) if sources else {}
        self.solver = Sw2dCurvedSolver(ctx, cub, gauss, curved, J, gauss.mapM, gauss.mapP, g=g, zx=src.get("zx"), zy=src.get("zy"),
                                       f=src.get("f", 0.0), CD=src.get("CD", 0.0), device=device)
        self.nodes, self.ctx, self.cub, self.x, self.y = nodes, ctx, cub, x, y
        self.filtered = filter_args is not None
        self.device = device                      # the exchange buffers of the nccl backend live on the solver's GPU
        self._on_device = dist.get_backend() == "nccl"

    def set_initial_state(self, fn):
        self.solver.setState(*fn(self.x, self.y))

    def _exchange(self, intermediate):
        """Ghost columns of the state the next evaluation reads, from their owners."""
        import torch
        plan, dist = self.plan, self.dist
        n_own, n_int = plan.num_owned, plan.num_interior
        rows = 4 * self.solver.Np
        boundary = self.solver.getElements(n_int, n_own - n_int, intermediate)          # the partition-boundary block
        send = np.ascontiguousarray(boundary[:, plan.send_local - n_int].T)               # (n_send, rows): one record per element
        dev = torch.device("cuda", self.device) if self._on_device else torch.device("cpu")
        sendbuf = torch.from_numpy(send).to(dev)
        recvbuf = torch.zeros((max(plan.num_halo, 1), rows), dtype=torch.float64, device=dev)
        from .halo import exchange_ops
        ops = exchange_ops(plan, sendbuf, recvbuf, dist)
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if plan.num_halo:
            self.solver.setElements(n_own, recvbuf[:plan.num_halo].cpu().numpy().T, intermediate)

    def step_rk2(self, dt, nsteps=1):
        """The driver's loop body (sw2d_curved.py:246-277) with an exchange in front of each of its two evaluations."""
        for _ in range(nsteps):
            self._exchange(False)
            self.solver.rk2Phase(dt, 0, self.filtered)
            self._exchange(True)
            self.solver.rk2Phase(dt, 1, self.filtered)

    def lserk4_stages(self, dt, nstages):
        """Sw2dCurvedSolver.lserk4Stages with an exchange in front of every stage."""
        for _ in range(nstages):
            self._exchange(False)
            self.solver.lserk4Stages(dt, 1)

    def owned_state(self):
        n = self.plan.num_owned
        return (self.plan.own_global,) + tuple(a[:, :n] for a in self.solver.getState())


class NativeDistributedSw2dCurved(DistributedSw2dCurved):
    """DistributedSw2dCurved with the ghost exchange driven by the C++ library: pack kernel, grouped ncclSend / ncclRecv with
    every neighbour on the solver's stream, unpack kernel -- device to device over RCCL (xGMI), whole step loops in one C call,
    no PyTorch. One process per rank; rank 0's RCCL id reaches the others through halo.file_rendezvous (or pass unique_id).
    On the nodal-trace form the elements without a ghost neighbour are evaluated beside the exchange (two streams)."""

    def __init__(self, plan, order, deform, g=9.81, filter_args=None, sources=None, device=0, unique_id=None, loopback=False):
        """loopback=True: this one process computes plan.rank's share of a plan.world-way split and every neighbour exchange
        is a send-to-self of the same size through the real transport (ghost values are then this rank's own boundary
        elements, not the neighbours': a rehearsal of the exchange on one GPU, not a partitioned result)."""
        import ctypes
        import os

        from ._capi import ptr
        from .halo import file_rendezvous

        class _NoDist:                    # the base class only asks its transport for the backend name
            @staticmethod
            def get_backend():
                return "native"
        super().__init__(plan, order, deform, _NoDist(), g=g, filter_args=filter_args, sources=sources, device=device)
        h = self.solver._h
        send = np.ascontiguousarray(plan.send_local, dtype=np.int32)
        check(lib.bdg_sw2d_curved_set_partition(h, plan.num_interior, plan.num_owned, ptr(send), send.size))
        id_path = None
        comm_rank, comm_world = plan.rank, plan.world

        def make_id():
            buf = ctypes.create_string_buffer(128)
            check(lib.bdg_comm_unique_id(buf, 128))
            return buf.raw
        if loopback:
            comm_rank, comm_world, unique_id = 0, 1, make_id()
        if unique_id is None:
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
        self.peer_table = (pr, ss, sc, rs, rc)
        idbuf = ctypes.create_string_buffer(unique_id, 128)
        check(lib.bdg_sw2d_curved_comm_init(h, comm_rank, comm_world, idbuf, ptr(pr), ptr(ss), ptr(sc), ptr(rs), ptr(rc), len(peers)))
        self.barrier()
        if id_path is not None and plan.rank == 0:
            try:
                os.remove(id_path)
            except OSError:
                pass

    def _exchange(self, intermediate):
        check(lib.bdg_sw2d_curved_exchange(self.solver._h, int(bool(intermediate))))

    def step_rk2(self, dt, nsteps=1):
        check(lib.bdg_sw2d_curved_step_rk2_exchanged(self.solver._h, float(dt), int(nsteps), int(self.filtered)))

    def lserk4_stages(self, dt, nstages):
        """Sw2dCurvedSolver.lserk4Stages with an exchange in front of every stage."""
        check(lib.bdg_sw2d_curved_lserk4_stages_exchanged(self.solver._h, float(dt), int(nstages)))

    def barrier(self):
        check(lib.bdg_sw2d_curved_barrier(self.solver._h))
