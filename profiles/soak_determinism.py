"""Run-to-run determinism of the step kernels at full size: the same initial state advanced twice by the same calls must give the
same bits (every element is computed by one lane from values of the previous stage; any difference means a race or a read of
something not yet written). Variants A, D (tracer + sources) and B, LSERK4 stages and midpoint RK2 + filter / SSP-RK2 steps.
  python3 profiles/soak_determinism.py [stages]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import blitzdg_amd.pyblitzdg as dg  # noqa: E402
from blitzdg_amd import sw2d  # noqa: E402

stages = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ok = True
for order, cells in ((4, (1000, 500)), (8, (500, 250)), (6, (1000, 250)), (3, (1000, 500))):
    mesh = dg.MeshManager()
    mesh.buildBoxMesh(*cells)
    nodes = dg.TriangleNodesProvisioner(order, mesh)
    nodes.buildFilter(0.9 * order, order)
    ctx = nodes.dgContext()
    x, y = ctx.x, ctx.y
    h = 10 + np.exp(-10 * x * x - 10 * y * y)
    hu, hv = 0.1 * np.sin(3 * x), 0.1 * np.cos(2 * y)
    for name in ("A", "D4", "B"):
        finals = []
        for rep in range(2):
            if name == "A":
                s = sw2d.Sw2dSolver(nodes=nodes)
                s.setState(h, hu, hv)
            elif name == "D4":
                s = sw2d.Sw2dSolver(nodes=nodes, fields=4, sources=dict(f=1e-4, CD=2.5e-3, zx=0.01 + 0 * x, zy=0 * x))
                s.setState4(h, hu, hv, 0.5 * h)
            else:
                s = sw2d.Sw2dSolver(nodes=nodes)
                H = 10 + 0.2 * x
                s.enableVariantB(H, *nodes.bedSlopes(H), CD=2.5e-3, f=1e-4)
                s.setState(h, hu, hv)
            dt = 0.2 * s.computeDt(0.5)[0]
            s.lserk4Stages(dt, stages)
            if name == "B":
                s.stepSSPRK2(dt, stages // 10, False, 1e-3)
            else:
                s.stepRK2(dt, stages // 10, True)
            finals.append(s.getState4() if name == "D4" else s.getState())
            s.close()
        same = all(np.array_equal(a, b) for a, b in zip(*finals))
        finite = all(np.isfinite(a).all() for a in finals[0])
        ok = ok and same and finite
        print(json.dumps({"order": order, "elements": ctx.numElements, "variant": name, "lserk4_stages": stages, "rk2_steps": stages // 10,
                          "bit_identical_across_runs": bool(same), "finite": bool(finite)}), flush=True)
print(json.dumps({"soak": "run-to-run determinism", "all_ok": bool(ok)}), flush=True)
sys.exit(0 if ok else 1)
