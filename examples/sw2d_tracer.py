#!/usr/bin/env python3
"""The reference's sw2d.py driver (sw2d.py:148-260: reduced-gravity shallow water with a passive
tracer and f-plane Coriolis, midpoint RK2 with the modal filter on every RHS, fixed dt from the
initial state, blow-up check on max|h|) on the MI355X path, state resident in HBM.

    python examples/sw2d_tracer.py [mesh.msh | box:NXxNY] [order] [steps] [outputDir]

Differences from the script: the mesh defaults to tests/golden/coarse_box.msh (the script's
input/R_8km_circle.msh is not shipped here), lengths are scaled to that box, and the eta / u / v / N *.vtu
files (sw2d.py:250-259) are written only when an output directory is given -- by the VTK-free writer, from
fields computed on the device.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import blitzdg_amd.pyblitzdg as dg  # noqa: E402
from blitzdg_amd import sw2d  # noqa: E402


def setup(mesh_arg, NOrder):
    meshManager = dg.MeshManager()
    if mesh_arg.startswith("box:"):
        nx, ny = (int(v) for v in mesh_arg[4:].split("x"))
        meshManager.buildBoxMesh(nx, ny)
    else:
        meshManager.readMesh(mesh_arg)
    nodes = dg.TriangleNodesProvisioner(NOrder, meshManager)
    nodes.buildFilter(0.9 * NOrder, 4)
    ctx = nodes.dgContext()
    x, y = ctx.x, ctx.y
    drho = 1.0025 - 1.000
    g = drho * 9.81                      # reduced gravity (sw2d.py:150-152)
    f = 7.88e-5
    eta = -2.5 * (x / 8.0)               # the script tilts the interface across an 8 km basin
    H = 10 * np.ones_like(x)
    N = np.exp(-((y - 0.25) / 0.2) ** 2)
    h = H + eta
    hu, hv, hN = np.zeros_like(h), np.zeros_like(h), h * N
    c = np.sqrt(g * h)
    CFL = 0.8
    dt = CFL / np.max(((NOrder + 1) ** 2) * 0.5 * np.abs(ctx.Fscale.flatten("F")) * (c.flatten("F")[ctx.vmapM]))
    return nodes, ctx, (h, hu, hv, hN), H, g, f, dt


def main():
    mesh_arg = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "coarse_box.msh")
    NOrder = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    outdir = sys.argv[4] if len(sys.argv) > 4 else None
    nodes, ctx, q, H, g, f, dt = setup(mesh_arg, NOrder)
    solver = sw2d.Sw2dSolver(nodes=nodes, g=g, fields=4, sources=dict(f=f, CD=0.0))
    solver.setState4(*q)
    solver.setBathymetry(H)
    outputter = dg.VtkOutputter(nodes)
    if outdir:
        os.makedirs(outdir, exist_ok=True)
        outputter.writeSolverFields(solver, 0, directory=outdir)
    t, step = 0.0, 0
    while step < steps:
        n = min(50, steps - step)
        solver.stepRK2(dt, n, filter=True)          # predictor + corrector, Filt on both RHS (sw2d.py:218-244)
        step += n
        t += n * dt
        h, hu, hv, hN = solver.getState4()
        h_max = np.max(np.abs(h))
        if h_max > 1e8 or np.isnan(h_max):
            raise Exception("A numerical instability has occurred.")
        if outdir:
            outputter.writeSolverFields(solver, step, directory=outdir)
        print(f"t={t:.6g} step={step} eta_max={np.abs(h - H).max():.6g} |u|max={np.abs(hu / h).max():.6g} "
              f"N in [{(hN / h).min():.4f}, {(hN / h).max():.4f}]")
    return solver.getState4(), t


if __name__ == "__main__":
    main()
