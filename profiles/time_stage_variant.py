#!/usr/bin/env python3
"""LSERK4 stages of one solver variant at a given order / mesh, for timing and for the rocprofv3 collections of
profiles/collect_curved.sh (BDG_COLLECT_CMD):
    python3 profiles/time_stage_variant.py <A|D3|D4|B> <order> <NXxNY> [stages]
Prints one JSON line: order, elements, variant, ms per stage (HIP-event free: wall clock around a synchronised block)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blitzdg_amd.pyblitzdg as dg  # noqa: E402
from blitzdg_amd import sw2d  # noqa: E402

variant, order = sys.argv[1], int(sys.argv[2])
nx, ny = (int(v) for v in sys.argv[3].split("x"))
stages = int(sys.argv[4]) if len(sys.argv) > 4 else 40
m = dg.MeshManager()
m.buildBoxMesh(nx, ny)
nodes = dg.TriangleNodesProvisioner(order, m)
nodes.buildFilter(0.9 * order, order)
ctx = nodes.dgContext()
x, y = ctx.x, ctx.y
h, hu, hv = 10 + np.exp(-10 * x * x - 10 * y * y), 0.1 * np.sin(3 * x), 0.1 * np.cos(2 * y)
src = dict(f=1e-4, CD=2.5e-3, zx=0.01 + 0 * x, zy=0 * x)
if variant == "B":
    s = sw2d.Sw2dSolver(nodes=nodes)
    H = 10 + 0.2 * x
    Hx, Hy = nodes.bedSlopes(H)
    s.enableVariantB(H, Hx, Hy, CD=2.5e-3, f=1e-4)
    s.setState(h, hu, hv)
elif variant == "D4":
    s = sw2d.Sw2dSolver(nodes=nodes, fields=4, sources=src)
    s.setState4(h, hu, hv, 0.5 * h)
elif variant == "D3":
    s = sw2d.Sw2dSolver(nodes=nodes, fields=3, sources=src)
    s.setState(h, hu, hv)
else:
    s = sw2d.Sw2dSolver(nodes=nodes)
    s.setState(h, hu, hv)
dt = 0.2 * s.computeDt(0.5)[0]
s.lserk4Stages(dt, 10)
s.synchronize()
t0 = time.perf_counter()
s.lserk4Stages(dt, stages)
s.synchronize()
ms = (time.perf_counter() - t0) / stages * 1e3
print(json.dumps({"order": order, "elements": ctx.numElements, "variant": variant, "stages": stages, "ms_per_stage": ms}), flush=True)
