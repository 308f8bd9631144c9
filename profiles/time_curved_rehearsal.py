#!/usr/bin/env python3
"""Loop-back rehearsal of a partitioned run of the curved / over-integrated solver on ONE GPU:
    python3 profiles/time_curved_rehearsal.py [order] [cellsX] [cellsY] [world] [rank] [steps]
This process computes `rank`'s share of a `world`-way split of the deformed box of profiles/time_curved.py; every neighbour
exchange is a real RCCL send-to-self of the true size (NativeDistributedSw2dCurved(loopback=True)). Timing only: the ghosts
then hold this rank's own boundary elements. Prints one JSON line: ms per RHS evaluation on the two-chain schedule, with
every element in stream order (BDG_SW2D_CURVED_NO_OVERLAP), and of the whole mesh on the same GPU."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blitzdg_amd.pyblitzdg as dg  # noqa: E402
from blitzdg_amd.halo import build_plan  # noqa: E402
from blitzdg_amd.sw2d_curved import NativeDistributedSw2dCurved, Sw2dCurvedSolver  # noqa: E402


def deform(x0, y0):
    b = np.clip(1.0 - (y0 + 1.0) / 0.1, 0.0, 1.0) ** 3
    return x0, y0 + 0.02 * b * np.sin(3 * x0)


def state(x, y):
    h = 1.0 + 0.1 * np.exp(-10 * x * x - 10 * y * y)
    z = np.zeros_like(h)
    return h, z, z.copy(), 0.5 * h


def sources(x, y):
    z = np.zeros_like(x)
    return {"zx": z, "zy": z.copy(), "f": 1e-4, "CD": 2.5e-3 + z}


def timed(run, sync, steps):
    run(5)
    sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        run(steps)
        sync()
        best = min(best, (time.perf_counter() - t0) / (2 * steps) * 1e3)
    return best


def main():
    a = [int(v) for v in sys.argv[1:]]
    order, nx, ny, world, rank, steps = (a + [4, 500, 250, 8, 1, 50][len(a):])[:6]
    mesh = dg.MeshManager()
    mesh.buildBoxMesh(nx, ny)
    total = mesh.numElements
    mesh.partitionMesh(world)
    plan = build_plan(mesh.elements, mesh.vertices, mesh.EToE, mesh.elementPartitionMap, rank, world, bctype=mesh.bcType)
    d = NativeDistributedSw2dCurved(plan, order, deform, g=9.81, filter_args=(0.9 * order, order), sources=sources, loopback=True)
    d.set_initial_state(state)
    dt = 1e-5
    out = {"order": order, "elements": total, "world": world, "rank": rank, "owned": int(plan.num_owned),
           "interior": int(plan.num_interior), "ghost": int(plan.num_halo), "nodal_trace_form": bool(d.solver.usesNodalTraces)}
    out["ms_per_rhs_two_chains"] = timed(lambda n: d.step_rk2(dt, n), d.solver.synchronize, steps)
    os.environ["BDG_SW2D_CURVED_NO_OVERLAP"] = "1"
    out["ms_per_rhs_stream_order"] = timed(lambda n: d.step_rk2(dt, n), d.solver.synchronize, steps)
    del os.environ["BDG_SW2D_CURVED_NO_OVERLAP"]
    del d
    # the whole mesh on this GPU
    nodes = dg.TriangleNodesProvisioner(order, mesh)
    nodes.buildFilter(0.9 * order, order)
    ctx = nodes.dgContext()
    x, y = deform(ctx.x, ctx.y)
    curved = np.where(np.abs(y - ctx.y).max(axis=0) > 0)[0].astype(np.int32)
    nodes.setCoordinates(x, y)
    J = (ctx.Dr @ x) * (ctx.Ds @ y) - (ctx.Ds @ x) * (ctx.Dr @ y)
    gauss, cub = nodes.buildGaussFaceNodes(2 * (order + 1)), nodes.buildCubatureVolumeMesh(3 * (order + 1))
    src = sources(x, y)
    s = Sw2dCurvedSolver(ctx, cub, gauss, curved, J, gauss.mapM, gauss.mapP, g=9.81, zx=src["zx"], zy=src["zy"], f=src["f"], CD=src["CD"])
    s.setState(*state(x, y))
    out["ms_per_rhs_whole_mesh"] = timed(lambda n: s.stepRK2(dt, n, True), s.synchronize, steps)
    out["speedup_two_chains"] = out["ms_per_rhs_whole_mesh"] / out["ms_per_rhs_two_chains"]
    out["speedup_stream_order"] = out["ms_per_rhs_whole_mesh"] / out["ms_per_rhs_stream_order"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
