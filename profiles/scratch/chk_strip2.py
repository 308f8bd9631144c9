import sys, numpy as np
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import blitzdg_amd.pyblitzdg as dg
from blitzdg_amd import sw2d
from blitzdg_amd.halo import LocalGroupSw2d
from test_dist_gpu import _fields
order, world, shape = 8, 2, (12, 10)
mesh = dg.MeshManager(); mesh.buildBoxMesh(*shape)
nodes = dg.TriangleNodesProvisioner(order, mesh); ctx = nodes.dgContext()
single = sw2d.Sw2dSolver(nodes=nodes)
q0 = _fields(ctx.x, ctx.y); single.setState(*q0)
dt = 0.5 * single.computeDt(0.65)[0]
group = LocalGroupSw2d(mesh, order, world)
group.set_global_state(*q0)
group.lserk4_stages(dt, 1); single.lserk4Stages(dt, 1)
got, ref = group.gather_state(), single.getState()
for c,(a,b) in enumerate(zip(got,ref)):
    d=np.abs(a-b)
    bad=np.argwhere(d>1e-12*np.abs(b).max())
    print('field',c,'max',d.max(),'nbad',len(bad), 'nodes', sorted(set(bad[:,0].tolist()))[:50], 'elements', sorted(set(bad[:,1].tolist()))[:40])
for p in group.plans:
    print('plan', p.rank, 'interior', p.num_interior, 'owned', p.num_owned, 'boundary global ids', p.own_global[p.num_interior:][:40])
group.close()
