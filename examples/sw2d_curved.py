#!/usr/bin/env python3
"""The reference's sw2d_curved.py driver (a geostrophic jet with a passive tracer in an 8 km channel whose top wall carries a
headland; over-integrated RHS -- cubature volume term, Gauss-face surface term, the curved elements' own Cholesky mass
matrices -- with the modal filter on every RHS; midpoint RK2; a drag coefficient that damps near the walls; the open ends of
the channel wired periodically) on the MI355X path, state resident in HBM.

    python examples/sw2d_curved.py [box:NXxNY] [order] [steps]

The set-up is the script's own, line for line in ITS order with only the import lines changed (sw2d_curved.py:7-17 ->
blitzdg_amd.pyblitzdg, blitzdg_amd.swhelpers.maps, blitzdg_amd.meshhelpers.curved): correctBCTable tags the channel ends
(:43), the wall curve is the spline through the top wall's nodes (:64-99), adjustStraightEdges / deformAndBlendElements curve
the elements along it (:105-106), makeMapsPeriodic rewires the nodal and the Gauss maps of the two ends (:144-145), fields
and the wall-layer drag as at :155-215. Two things differ: the script's mesh (input/headlands_highres.msh) is not shipped with
the reference, so the channel [0, 8000] x [0, 1000] is built here with its wall vertices on an analytic headland; and the
context handed to the two curved helpers HOLDS its coordinate arrays -- with a context that copies on every access, which is what
the reference's own pyblitzdg does, their in-place blending is lost (blitzdg_amd/meshhelpers/curved.py). The loop body is
sw2d_curved.py:246-277 -- here one call per 50 steps, the RHS, filter, predictor and corrector all on the device
(bdg_sw2d_curved_step_rk2).
"""
import os
import sys
import types

import numpy as np
from scipy.interpolate import splev, splrep

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import blitzdg_amd.pyblitzdg as dg  # noqa: E402
from blitzdg_amd.meshhelpers.curved import adjustStraightEdges, deformAndBlendElements  # noqa: E402
from blitzdg_amd.sw2d_curved import Sw2dCurvedSolver  # noqa: E402
from blitzdg_amd.swhelpers.maps import correctBCTable, makeMapsPeriodic  # noqa: E402


def headland(xv):
    """The top wall: y = 1000 with a headland of 60 m between x = 3250 and x = 4750."""
    return 1000.0 - 60.0 * np.exp(-((xv - 4000.0) / 350.0) ** 2) * ((xv >= 3250.0) & (xv <= 4750.0))


def channel_mesh(nx, ny):
    """[0, 8000] x [0, 1000] in nx x ny cells of two triangles, the columns compressed under the headland so that the wall
    vertices lie on it (stand-in for input/headlands_highres.msh)."""
    box = dg.MeshManager()
    box.buildBoxMesh(nx, ny, 0.0, 8000.0, 0.0, 1000.0)
    verts = box.vertices
    verts[:, 1] *= headland(verts[:, 0]) / 1000.0
    mesh = dg.MeshManager()
    mesh.buildMesh(box.elements, verts)
    return mesh


def setup(mesh_arg, NOrder):
    nx, ny = (int(v) for v in mesh_arg[4:].split("x")) if mesh_arg.startswith("box:") else (32, 8)
    drho = 1.00100 - 1.000                                                    # sw2d_curved.py:21-28
    g = drho * 9.81
    H0 = 7.5

    meshManager = channel_mesh(nx, ny)
    Verts, EToV, bcType = meshManager.vertices, meshManager.elements, meshManager.bcType
    bcType = correctBCTable(bcType, EToV, Verts, 2)                           # :42-44  2 = outflow
    meshManager.setBCType(bcType)

    nodes = dg.TriangleNodesProvisioner(NOrder, meshManager)
    nodes.buildFilter(0.9 * NOrder, 4)
    ctx = nodes.dgContext()
    x, y = ctx.x, ctx.y
    xFlat, yFlat = x.flatten("F"), y.flatten("F")

    vmapW = ctx.vmapM[ctx.BCmap[3]]                                           # :64-79 the wall nodes under the headland ...
    xW, yW = xFlat[vmapW], yFlat[vmapW]
    top = np.logical_and(yW > 200, np.logical_and(xW > 3250, xW < 4750))
    isort = np.argsort(xW[top])
    xtop, ytop = xW[top][isort], yW[top][isort]
    keep = np.concatenate([[True], np.hypot(np.diff(xtop), np.diff(ytop)) > 1e-9])   # (a vertex node is listed by both its faces)
    xtop, ytop = xtop[keep], ytop[keep]
    s = np.concatenate([[0.0], np.cumsum(np.hypot(np.diff(xtop), np.diff(ytop)))])   # :81-86 ... parametrised by arc length
    s128, ss = np.linspace(s[0], s[-1], 128), np.linspace(s[0], s[-1], 4096)
    splx, sply = splrep(s128, np.interp(s128, s, xtop)), splrep(s128, np.interp(s128, s, ytop))   # :91-96
    xTopSmooth, yTopSmooth = splev(ss, splx, ext=2), splev(ss, sply, ext=2)

    bcInds = np.where(bcType.flatten("F") > 0)                                # :101-106
    bcFaces = np.transpose(np.unravel_index(bcInds, (ctx.numElements, ctx.numFaces), order="F"))
    held = types.SimpleNamespace(x=x, y=y, r=ctx.r, s=ctx.s, Fmask=ctx.Fmask, numFaces=ctx.numFaces)
    Verts, modifiedVerts, curvedFaces = adjustStraightEdges(Verts, EToV, bcFaces, xTopSmooth, yTopSmooth, held)
    x, y, curvedEls = deformAndBlendElements(Verts, EToV, curvedFaces, xTopSmooth, yTopSmooth, ss, splx, sply, held, NOrder)
    curvedEls = np.unique(np.asarray(curvedEls, dtype=np.int32))

    nodes.setCoordinates(x, y)                                                # :109-119
    J = np.dot(ctx.Dr, x) * np.dot(ctx.Ds, y) - np.dot(ctx.Ds, x) * np.dot(ctx.Dr, y)
    gauss_ctx = nodes.buildGaussFaceNodes(2 * (NOrder + 1))

    mapW, mapO = ctx.BCmap[3], ctx.BCmap[2]                                   # :121-145
    vmapO, vmapW = ctx.vmapM[mapO], ctx.vmapM[mapW]
    xFlat, yFlat = x.flatten("F"), y.flatten("F")
    gxFlat, gyFlat = np.dot(gauss_ctx.Interp, x).flatten("F"), np.dot(gauss_ctx.Interp, y).flatten("F")
    gmapO = np.array(gauss_ctx.BCmap[2])
    vmapM, vmapP = makeMapsPeriodic(ctx.vmapM, ctx.vmapP, vmapO, xFlat, yFlat, xFlat[vmapO], yFlat[vmapO])
    gmapM, gmapP = makeMapsPeriodic(gauss_ctx.mapM, gauss_ctx.mapP, gmapO, gxFlat, gyFlat, gxFlat[gmapO], gyFlat[gmapO])
    cub_ctx = nodes.buildCubatureVolumeMesh(3 * (NOrder + 1))

    f = 7.8825e-5                                                             # :155-165 a jet in geostrophic balance
    amp, L, W = 0.5 * .065 * H0, 500, 200
    eta = amp * np.exp(-((y - L) / W) ** 2)
    u = (-g / f) * (-2 * amp * (y - L) * np.exp(-((y - L) / W) ** 2) / W ** 2)
    xW, yW = xFlat[vmapW], yFlat[vmapW]                                       # :171-192 drag within 250 m of a wall
    CD_max, length_tol = 2.5e-3, 2.5e2
    min_dist = np.concatenate([np.hypot(xFlat[i:i + 4096, None] - xW[None, :], yFlat[i:i + 4096, None] - yW[None, :]).min(axis=1)
                               for i in range(0, xFlat.size, 4096)])
    CD = np.array(np.reshape(CD_max * 0.5 * (1 - np.tanh((min_dist - 0.5 * length_tol) / (0.1 * length_tol))), x.shape, order="F"), order="C")
    H = H0 * np.ones_like(x)                                                  # :201-223
    z = -H
    zx = ctx.rx * np.dot(ctx.Dr, z) + ctx.sx * np.dot(ctx.Ds, z)
    zy = ctx.ry * np.dot(ctx.Dr, z) + ctx.sy * np.dot(ctx.Ds, z)
    N = np.exp(-(((x - 4000.0) / 3e2) ** 2 + ((y - 350.0) / 3e2) ** 2))
    h = H + eta
    h[h < 1e-3] = 1e-3
    hu, hv, hN = h * u, np.zeros_like(h), h * N
    c = np.sqrt(g * np.mean(H))                                               # :236-239
    CFL = 0.75
    dt = CFL / np.max(((NOrder + 1) ** 2) * 0.5 * np.abs(ctx.Fscale.flatten("F")) * (c + np.abs(u.flatten("F")[vmapM])))
    solver = Sw2dCurvedSolver(ctx, cub_ctx, gauss_ctx, curvedEls, J, gmapM, gmapP, g=g, zx=zx, zy=zy, f=f, CD=CD)
    w = cub_ctx.W.sum(axis=0)
    return solver, ctx, cub_ctx, (h, hu, hv, hN), H, dt, w, len(curvedEls)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    mesh_arg = argv[0] if len(argv) > 0 else "box:32x8"
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
