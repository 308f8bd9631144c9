import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blitzdg_amd.pyblitzdg as dg
from blitzdg_amd import sw2d
order = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nx, ny = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1000x500").split("x"))
m = dg.MeshManager(); m.buildBoxMesh(nx, ny)
nodes = dg.TriangleNodesProvisioner(order, m); nodes.buildFilter(0.9*order, order)
ctx = nodes.dgContext(); x, y = ctx.x, ctx.y
K = ctx.numElements
h = 10 + np.exp(-10*x*x-10*y*y); hu = 0.1*np.sin(3*x); hv = 0.1*np.cos(2*y)
def timeit(s, n=100):
    """best of three blocks of n stages (allocations of the previous solver being released can disturb one)"""
    dt = 0.2*s.computeDt(0.5)[0]
    best = 1e9
    for _ in range(3):
        s.lserk4Stages(dt, 20); s.synchronize()
        t0 = time.perf_counter(); s.lserk4Stages(dt, n); s.synchronize()
        best = min(best, (time.perf_counter()-t0)/n*1e3)
    return best
s = sw2d.Sw2dSolver(nodes=nodes); s.setState(h, hu, hv)
print(f"N={order} K={K}: variant A   {timeit(s):.4f} ms/stage"); s.close()
s = sw2d.Sw2dSolver(nodes=nodes, fields=3, sources=dict(f=1e-4, CD=2.5e-3, zx=0.01+0*x, zy=0*x)); s.setState(h, hu, hv)
print(f"N={order} K={K}: variant D3  {timeit(s):.4f} ms/stage (3 fields + sources)"); s.close()
s = sw2d.Sw2dSolver(nodes=nodes, fields=4, sources=dict(f=1e-4, CD=2.5e-3, zx=0.01+0*x, zy=0*x)); s.setState4(h, hu, hv, 0.5*h)
print(f"N={order} K={K}: variant D4  {timeit(s):.4f} ms/stage (tracer + sources)"); s.close()
s = sw2d.Sw2dSolver(nodes=nodes); H = 10 + 0.2*x
Hx, Hy = nodes.bedSlopes(H)
s.enableVariantB(H, Hx, Hy, CD=2.5e-3, f=1e-4); s.setState(h, hu, hv)
print(f"N={order} K={K}: variant B   {timeit(s):.4f} ms/stage (speed pass + stage)"); s.close()
