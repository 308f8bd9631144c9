"""What a GPU-initialised Python process of this repository holds when it is about to start a child: the shared
libraries of the ROCm / OpenMP families that are mapped, and its threads. `python tests/tools/process_census.py [torch]`
(with `torch`: PyTorch imported first, as the round-2 pytest parent had it). Evidence for DESIGN section 6.1."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def census(tag):
    libs = collections.OrderedDict()
    with open("/proc/self/maps") as f:
        for ln in f:
            path = ln.split()[-1]
            base = os.path.basename(path)
            if any(k in base for k in ("omp", "amdhip", "hsa-runtime", "rccl", "blitzdg", "oracle", "preload", "guard")):
                libs[path] = True
    threads = collections.Counter()
    for t in os.listdir("/proc/self/task"):
        try:
            threads[open(f"/proc/self/task/{t}/comm").read().strip()] += 1
        except OSError:
            pass
    print(f"[{tag}] LD_PRELOAD={os.environ.get('LD_PRELOAD')!r} threads={sum(threads.values())} {dict(threads)}")
    for p in libs:
        print(f"[{tag}]   {p}")


if __name__ == "__main__":
    if "torch" in sys.argv[1:]:
        import torch  # noqa: F401
        census("torch imported")
    import blitzdg_amd.pyblitzdg as dg
    from blitzdg_amd import sw2d
    census("library loaded")
    mesh = dg.MeshManager()
    mesh.buildBoxMesh(16, 16)
    nodes = dg.TriangleNodesProvisioner(3, mesh)
    s = sw2d.Sw2dSolver(nodes=nodes)
    ctx = nodes.dgContext()
    s.setState(10.0 + 0 * ctx.x, 0 * ctx.x, 0 * ctx.x)
    s.lserk4Stages(1e-3, 5)
    s.getState()
    census("solver ran")
    if "oracle" in sys.argv[1:]:
        from oracle import Sw2dOracle  # noqa: F401
        census("oracle loaded")
