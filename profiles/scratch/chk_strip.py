import sys, numpy as np
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import blitzdg_amd.pyblitzdg as dg
from blitzdg_amd import sw2d
from blitzdg_amd.halo import LocalGroupSw2d
from test_dist_gpu import _fields
for order, world, shape in ((6, 4, (24, 20)), (8, 2, (12, 10)), (5, 3, (30, 24))):
    mesh = dg.MeshManager(); mesh.buildBoxMesh(*shape)
    nodes = dg.TriangleNodesProvisioner(order, mesh); ctx = nodes.dgContext()
    single = sw2d.Sw2dSolver(nodes=nodes)
    q0 = _fields(ctx.x, ctx.y); single.setState(*q0)
    dt = 0.5 * single.computeDt(0.65)[0]
    group = LocalGroupSw2d(mesh, order, world)
    group.set_global_state(*q0)
    for chunk in (1, 2, 7, 13):
        group.lserk4_stages(dt, chunk); single.lserk4Stages(dt, chunk)
    got, ref = group.gather_state(), single.getState()
    print(order, world, [float(np.abs(a-b).max()/np.abs(b).max()) for a, b in zip(got, ref)], [int((a!=b).sum()) for a,b in zip(got,ref)], got[0].size)
    group.close()
