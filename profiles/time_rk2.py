"""Per-evaluation time of the midpoint RK2 + filter step the reference's Python drivers run (sw2d.py, examples): two RHS
evaluations per step, fused with the combine (MODE_COMBINE), against the LSERK4 stage of the same solver.
  python3 profiles/time_rk2.py <order> <NXxNY>"""
import sys, time, json, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blitzdg_amd.pyblitzdg as dg
from blitzdg_amd import sw2d
order = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nx, ny = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1000x500").split("x"))
m = dg.MeshManager(); m.buildBoxMesh(nx, ny)
nodes = dg.TriangleNodesProvisioner(order, m); nodes.buildFilter(0.9 * order, order)
ctx = nodes.dgContext(); x, y = ctx.x, ctx.y
h = 10 + np.exp(-10 * x * x - 10 * y * y); hu = 0.1 * np.sin(3 * x); hv = 0.1 * np.cos(2 * y)


def best(fn, per):
    b = 1e9
    for _ in range(3):
        fn(5); s.synchronize()
        t0 = time.perf_counter(); fn(25); s.synchronize()
        b = min(b, (time.perf_counter() - t0) / (25 * per) * 1e3)
    return b


out = {"order": order, "elements": ctx.numElements}
for name, kw, four in (("A", {}, False), ("D3", dict(fields=3, sources=dict(f=1e-4, CD=2.5e-3, zx=0.01 + 0 * x, zy=0 * x)), False),
                       ("D4", dict(fields=4, sources=dict(f=1e-4, CD=2.5e-3, zx=0.01 + 0 * x, zy=0 * x)), True)):
    s = sw2d.Sw2dSolver(nodes=nodes, **kw)
    (s.setState4(h, hu, hv, 0.5 * h) if four else s.setState(h, hu, hv))
    dt = 0.2 * s.computeDt(0.5)[0]
    out[name] = {"lserk_ms_per_stage": best(lambda n: s.lserk4Stages(dt, 5 * n), 5),
                 "rk2_filter_ms_per_evaluation": best(lambda n: s.stepRK2(dt, n, True), 2),
                 "ssprk2_ms_per_evaluation": best(lambda n: s.stepSSPRK2(dt, n, False, 1e-3), 2)}
    s.close()
s = sw2d.Sw2dSolver(nodes=nodes); H = 10 + 0.2 * x
Hx, Hy = nodes.bedSlopes(H)
s.enableVariantB(H, Hx, Hy, CD=2.5e-3, f=1e-4); s.setState(h, hu, hv)
dt = 0.2 * s.computeDt(0.5)[0]
out["B"] = {"lserk_ms_per_stage": best(lambda n: s.lserk4Stages(dt, 5 * n), 5),
            "ssprk2_sponge_ms_per_evaluation": best(lambda n: s.stepSSPRK2(dt, n, False, 1e-3), 2)}
s.close()
print(json.dumps(out))
