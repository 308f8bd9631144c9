#!/usr/bin/env python3
"""The reference's sw2d_curved.py driver (sw2d_curved.py:96-290: a geostrophic jet with a passive tracer in a channel
whose walls are curved; over-integrated RHS -- cubature volume term, Gauss-face surface term, the listed elements'
own Cholesky mass matrices -- with the modal filter on every RHS; midpoint RK2; a drag coefficient that damps near the
walls; the open ends of the channel wired periodically) on the MI355X path, state resident in HBM.

    python examples/sw2d_curved.py [box:NXxNY] [order] [steps]

Differences from the script: its channel mesh (input/channel_curved.msh) and the spline through its top wall are not
shipped here, so the channel is a box [-1, 1]^2 whose wall y = -1 is bent into a smooth curve (the elements within 0.2 of
it are deformed with a blend that vanishes away from the wall, and listed in curvedEls, like the script's
deformAndBlendElements output); the x = -1 / x = +1 ends are rewired periodically the way swhelpers.maps.makeMapsPeriodic
does it (nodes paired by their y); lengths are scaled to that box. The loop body is sw2d_curved.py:246-277 -- here one
call per 50 steps, the RHS, filter, predictor and corrector all on the device (bdg_sw2d_curved_step_rk2).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import blitzdg_amd.pyblitzdg as dg  # noqa: E402
from blitzdg_amd.sw2d_curved import Sw2dCurvedSolver  # noqa: E402


def make_periodic_in_x(gauss, gmapP):
    """The Gauss-node map with the wall faces on x = xmin and x = xmax paired (makeMapsPeriodic): each boundary Gauss node
    takes the node of the opposite end with the same y as its exterior node and leaves the wall list."""
    gx, gy = gauss.x.flatten("F"), gauss.y.flatten("F")
    walls = np.array(gauss.BCmap.get(3, []), dtype=np.int64)
    left, right = walls[np.abs(gx[walls] - gx.min()) < 1e-9], walls[np.abs(gx[walls] - gx.max()) < 1e-9]
    gmapP = gmapP.copy()
    for a, b in ((left, right), (right, left)):
        order = b[np.argsort(gy[b])]
        pos = np.clip(np.searchsorted(gy[order], gy[a]), 1, order.size - 1)
        near = np.where(np.abs(gy[order[pos - 1]] - gy[a]) < np.abs(gy[order[pos]] - gy[a]), order[pos - 1], order[pos])
        assert np.abs(gy[near] - gy[a]).max() < 1e-9, "the two ends of the channel do not carry matching face nodes"
        gmapP[a] = near
    keep = np.setdiff1d(walls, np.concatenate([left, right]))
    bcmap = dict(gauss.BCmap)
    bcmap[3] = [int(i) for i in keep]
    return gmapP, bcmap


def setup(mesh_arg, NOrder):
    mesh = dg.MeshManager()
    nx, ny = (int(v) for v in mesh_arg[4:].split("x")) if mesh_arg.startswith("box:") else (24, 16)
    mesh.buildBoxMesh(nx, ny)
    nodes = dg.TriangleNodesProvisioner(NOrder, mesh)
    nodes.buildFilter(0.9 * NOrder, 4)
    ctx = nodes.dgContext()
    x0, y0 = ctx.x, ctx.y
    # the wall y = -1 becomes a smooth curve; the deformation is blended out over the first 0.2 above it
    blend = np.clip(1.0 - (y0 + 1.0) / 0.2, 0.0, 1.0) ** 3
    x, y = x0, y0 + 0.03 * blend * np.sin(np.pi * x0)
    curvedEls = np.where(np.abs(y - y0).max(axis=0) > 0)[0]
    nodes.setCoordinates(x, y)
    J = (ctx.Dr @ x) * (ctx.Ds @ y) - (ctx.Ds @ x) * (ctx.Dr @ y)          # sw2d_curved.py:112-118
    gauss_ctx = nodes.buildGaussFaceNodes(2 * (NOrder + 1))
    cub_ctx = nodes.buildCubatureVolumeMesh(3 * (NOrder + 1))
    gmapM = gauss_ctx.mapM
    gmapP, gbc = make_periodic_in_x(gauss_ctx, gauss_ctx.mapP)

    g, f, H0 = 9.81 * 0.0025, 0.5, 1.0                                       # reduced gravity; f scaled to the box
    amp, L, W = 0.03 * H0, 0.0, 0.25
    eta = amp * np.exp(-((y - L) / W) ** 2)                                  # a jet in geostrophic balance (:160-165)
    u = (-g / f) * (-2 * amp * (y - L) * np.exp(-((y - L) / W) ** 2) / W ** 2)
    H = H0 * np.ones_like(x)
    N = np.exp(-(((x + 0.3) / 0.2) ** 2 + ((y + 0.4) / 0.2) ** 2))
    h = H + eta
    hu, hv, hN = h * u, np.zeros_like(h), h * N
    # drag that damps within a wall layer (:171-192); the walls are y = const here
    dist = np.minimum(np.abs(y - y.min()), np.abs(y - y.max()))
    CD = 2.5e-3 * 0.5 * (1 - np.tanh((dist - 0.05) / 0.01))
    z = -H
    zx = ctx.rx * (ctx.Dr @ z) + ctx.sx * (ctx.Ds @ z)
    zy = ctx.ry * (ctx.Dr @ z) + ctx.sy * (ctx.Ds @ z)
    c = np.sqrt(g * H.mean())
    CFL = 0.75
    spd = c + np.hypot(u, 0 * u)
    dt = CFL / np.max(((NOrder + 1) ** 2) * 0.5 * np.abs(ctx.Fscale.flatten("F")) * spd.flatten("F")[ctx.vmapM])   # :231
    gauss = type("GaussCtx", (), dict(Interp=gauss_ctx.Interp, W=gauss_ctx.W, nx=gauss_ctx.nx, ny=gauss_ctx.ny, BCmap=gbc))()
    solver = Sw2dCurvedSolver(ctx, cub_ctx, gauss, curvedEls, J, gmapM, gmapP, g=g, zx=zx, zy=zy, f=f, CD=CD)
    w = cub_ctx.W.sum(axis=0)                                                 # element areas (for the mass check below)
    return solver, ctx, cub_ctx, (h, hu, hv, hN), H, dt, w, len(curvedEls)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    mesh_arg = argv[0] if len(argv) > 0 else "box:24x16"
    NOrder = int(argv[1]) if len(argv) > 1 else 4
    steps = int(argv[2]) if len(argv) > 2 else 200
    solver, ctx, cub, q, H, dt, w, ncurved = setup(mesh_arg, NOrder)
    print(f"K={ctx.numElements} Np={ctx.numLocalPoints} Ncub={cub.NumCubaturePoints} curved elements={ncurved} dt={dt:.6g} "
          f"nodal-trace kernels={solver.usesNodalTraces}")
    mass = lambda hh: float(((cub.V @ hh) * cub.W).sum())                     # noqa: E731  (cubature integral of h)
    mass0 = mass(q[0])
    solver.setState(*q)
    t, step = 0.0, 0
    while step < steps:
        n = min(50, steps - step)
        solver.stepRK2(dt, n, filter=True)            # RHS, Filter, predictor, RHS, Filter, corrector (sw2d_curved.py:246-277)
        step += n
        t += n * dt
        h, hu, hv, hN = solver.getState()
        h_max = np.max(np.abs(h))
        if h_max > 1e8 or np.isnan(h_max):
            raise Exception("A numerical instability has occurred.")
        print(f"t={t:.6g} step={step} eta_max={np.abs(h - H).max():.6g} |u|max={np.abs(hu / h).max():.6g} "
              f"N in [{(hN / h).min():.4f}, {(hN / h).max():.4f}] mass drift={(mass(h) - mass0) / mass0:.3e}")
    print(f"done: steps={step} t={t:.6g}")
    return solver.getState(), t


if __name__ == "__main__":
    main()
