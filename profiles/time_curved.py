#!/usr/bin/env python3
"""Times the curved / over-integrated RHS (bdg_sw2d_curved_*) on a deformed box mesh:
    python3 profiles/time_curved.py [order] [cellsX] [cellsY] [steps]
Prints one JSON line: ms per RHS evaluation (HIP events), compulsory bytes per element, achieved GB/s.
Under `rocprofv3 --kernel-trace --stats` the same command gives the per-kernel split."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blitzdg_amd.pyblitzdg as dg  # noqa: E402
from blitzdg_amd.sw2d_curved import Sw2dCurvedSolver  # noqa: E402


def main():
    order = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    nx = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    ny = int(sys.argv[3]) if len(sys.argv) > 3 else 250
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    t0 = time.perf_counter()
    mesh = dg.MeshManager()
    mesh.buildBoxMesh(nx, ny)
    nodes = dg.TriangleNodesProvisioner(order, mesh)
    nodes.buildFilter(0.9 * order, order)
    ctx = nodes.dgContext()
    x0, y0 = ctx.x, ctx.y
    # the wall y = -1 becomes a smooth curve; elements within 0.1 of it are deformed (and listed in curvedEls)
    b = np.clip(1.0 - (y0 + 1.0) / 0.1, 0.0, 1.0) ** 3
    x, y = x0, y0 + 0.02 * b * np.sin(3 * x0)
    curvedEls = np.where(np.abs(y - y0).max(axis=0) > 0)[0].astype(np.int32)
    nodes.setCoordinates(x, y)
    J = (ctx.Dr @ x) * (ctx.Ds @ y) - (ctx.Ds @ x) * (ctx.Dr @ y)
    gauss = nodes.buildGaussFaceNodes(2 * (order + 1))
    cub = nodes.buildCubatureVolumeMesh(3 * (order + 1))
    h = 1.0 + 0.1 * np.exp(-10 * x * x - 10 * y * y)
    z = np.zeros_like(h)
    s = Sw2dCurvedSolver(ctx, cub, gauss, curvedEls, J, gauss.mapM, gauss.mapP, g=9.81, zx=z, zy=z, f=1e-4,
                         CD=2.5e-3 + z)
    setup = time.perf_counter() - t0
    s.setState(h, z, z, 0.5 * h)
    dt = 1e-5
    s.stepRK2(dt, 20, True)
    s.synchronize()
    ms = min(s.timeRK2(dt, steps, True) for _ in range(3))
    K, Np = ctx.numElements, ctx.numLocalPoints
    hN = s.getState()[0]
    assert np.isfinite(hN).all()
    print(json.dumps({"order": order, "elements": K, "Np": Np, "Ncub": cub.NumCubaturePoints, "NGauss": gauss.NGauss,
                      "curved_elements": int(curvedEls.size), "ms_per_rhs": ms,
                      "bytes_per_element": s.bytesPerElement, "GBps": s.bytesPerElement * K / (ms * 1e-3) / 1e9,
                      "element_dof_updates_per_s": Np * K / (ms * 1e-3), "setup_seconds": round(setup, 1),
                      "device_bytes": s.deviceBytes}))


if __name__ == "__main__":
    main()
