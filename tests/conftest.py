import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# ---- child processes: never forked from this (soon GPU-initialised) process, but by tests/launcher.py, which is
# ---- started here, at session start, before any test has loaded the HIP library

_launcher = None


def _hip_is_mapped():
    try:
        with open("/proc/self/maps") as f:
            return any("libamdhip64" in ln or "libhsa-runtime64" in ln for ln in f)
    except OSError:
        return False


def _start_launcher():
    global _launcher
    if _launcher is not None and _launcher.poll() is None:
        return _launcher
    assert not _hip_is_mapped(), "tests/launcher.py must be started before this process loads the HIP runtime"
    import subprocess
    _launcher = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "launcher.py")], stdin=subprocess.PIPE,
                                 stdout=subprocess.PIPE, text=True, bufsize=1)
    return _launcher


def pytest_sessionstart(session):
    _start_launcher()


def pytest_sessionfinish(session, exitstatus):
    global _launcher
    if _launcher is not None:
        try:
            _launcher.stdin.close()         # EOF ends its loop
            _launcher.wait(timeout=30)
        except Exception:
            _launcher.kill()
        _launcher = None


class Completed:
    """What subprocess.run(capture_output=True, text=True) returns, for a command tests/launcher.py ran."""

    def __init__(self, argv, returncode, stdout, stderr):
        self.args, self.returncode, self.stdout, self.stderr = argv, returncode, stdout, stderr


def launch_group(cmds, timeout=600):
    """Run commands side by side through the launcher; cmds = [(argv, env or None, cwd or None), ...]."""
    import json
    proc = _start_launcher()
    req = {"cmds": [{"argv": [str(a) for a in argv], "env": env, "cwd": cwd} for argv, env, cwd in cmds], "timeout": timeout}
    proc.stdin.write(json.dumps(req) + "\n")
    proc.stdin.flush()
    line = proc.stdout.readline()
    assert line, "tests/launcher.py went away"
    reply = json.loads(line)
    assert "error" not in reply, reply.get("error")
    return [Completed(c[0], r["returncode"], r["stdout"], r["stderr"]) for c, r in zip(cmds, reply["results"])]


def launch(argv, env=None, cwd=None, timeout=600):
    """subprocess.run(argv, capture_output=True, text=True, ...) without a fork of this process."""
    return launch_group([(argv, env, cwd)], timeout)[0]


def launch_ranks(module, function, nprocs, args, env=None, timeout=900):
    """function(rank, *args) of tests/<module>.py in `nprocs` fresh interpreters, side by side (what
    torch.multiprocessing.start_processes(..., start_method="spawn", join=True) did from inside this process).
    args must survive JSON. Fails the test with the workers' output when one of them fails."""
    import json
    worker = os.path.join(ROOT, "tests", "rank_worker.py")
    env = dict(os.environ if env is None else env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    done = launch_group([([sys.executable, worker, module, function, str(r), json.dumps(list(args))], env, ROOT)
                         for r in range(nprocs)], timeout)
    bad = [(r, d) for r, d in enumerate(done) if d.returncode != 0]
    assert not bad, "".join(f"\n--- rank {r} exit {d.returncode}\n{d.stdout[-1500:]}{d.stderr[-3000:]}" for r, d in bad)
    return done


@pytest.fixture(autouse=True)
def _gpu_tests_fork_nothing_and_keep_torch_out(request, monkeypatch):
    """A GPU test runs in a process that holds the HIP runtime: it may neither fork (subprocess / multiprocessing go
    through launch() / launch_ranks() above) nor pull PyTorch -- a second copy of the ROCm runtime libraries and an
    OpenMP runtime -- into this process."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import subprocess

    def refuse(*a, **k):
        raise AssertionError("a GPU test must not fork the pytest process: use conftest.launch / launch_ranks")
    had_torch = "torch" in sys.modules
    monkeypatch.setattr(subprocess, "Popen", refuse)
    monkeypatch.setattr(os, "fork", refuse)
    yield
    assert had_torch or "torch" not in sys.modules, "a GPU test imported torch into the pytest process"


@pytest.fixture(scope="session")
def known():
    """Known-answer vectors parsed from the reference's unit tests (tests/golden/make_golden.py)."""
    return np.load(os.path.join(GOLDEN, "reference_known_answers.npz"))


@pytest.fixture(scope="session")
def coarse_mesh():
    import blitzdg_amd.pyblitzdg as dg
    m = dg.MeshManager()
    m.readMesh(os.path.join(GOLDEN, "coarse_box.msh"))
    return m


def load_case(name):
    return np.load(os.path.join(GOLDEN, f"sw2d_rhs_{name}.npz"))


def seeded_fields(x, y, seed=0):
    """BASELINE.md section 3 parity inputs: H=10, Gaussian eta, 0.1*N(0,1) momentum."""
    rng = np.random.default_rng(seed)
    h = 10.0 + np.exp(-10 * x * x - 10 * y * y)
    hu = 0.1 * rng.standard_normal(x.shape)
    hv = 0.1 * rng.standard_normal(x.shape)
    return h, hu, hv


def oracle_from(tables, g=9.81, threads=1):
    """Build the CPU oracle from a mapping with the DG tables (npz fixture or dict)."""
    from oracle import Sw2dOracle
    filt = tables["Filter"] if "Filter" in tables else None
    return Sw2dOracle(tables["Dr"], tables["Ds"], tables["Lift"], tables["rx"], tables["sx"], tables["ry"],
                      tables["sy"], tables["nx"], tables["ny"], tables["Fscale"], tables["vmapM"], tables["vmapP"],
                      tables["mapW"], g=g, Filter=filt, threads=threads)


def tables_from_nodes(nodes):
    """Host tables of a pyblitzdg.TriangleNodesProvisioner as a dict (fresh ndarrays)."""
    ctx = nodes.dgContext()
    t = {k: getattr(ctx, k) for k in
         ("Dr", "Ds", "Lift", "rx", "sx", "ry", "sy", "nx", "ny", "Fscale", "vmapM", "vmapP", "x", "y")}
    t["Filter"] = ctx.filter
    t["mapW"] = np.array(ctx.BCmap.get(3, []), dtype=np.int32)
    t["order"] = ctx.order
    return t


def relmax(a, b):
    scale = max(np.abs(b).max(), 1e-300)
    return np.abs(np.asarray(a) - np.asarray(b)).max() / scale


def variant_b_setup(order, mesh, seed=3):
    """A variant-B problem on `mesh` the way the reference's driver sets it up
    (src/sw2d/main.cpp:60-190): faces on the left edge are re-tagged Out (2) and buildBCHash is
    called AGAIN, which appends (so those nodes stay in the wall list too); sloping depth H, bed
    slopes through Filter, sponge field around the open boundary. Returns (nodes, tables, extras)."""
    import blitzdg_amd.pyblitzdg as dg
    nodes = dg.TriangleNodesProvisioner(order, mesh)
    nodes.buildFilter(0.9 * order, order)
    verts = np.asarray(mesh.vertices).reshape(-1, 3)
    etov = np.asarray(mesh.elements).reshape(-1, 3)
    bc = np.asarray(mesh.bcType).reshape(-1, 3).copy()
    xmin = verts[:, 0].min()
    for f, (a, b) in enumerate(((0, 1), (1, 2), (2, 0))):
        left = (np.abs(verts[etov[:, a], 0] - xmin) < 1e-12) & (np.abs(verts[etov[:, b], 0] - xmin) < 1e-12)
        bc[left & (bc[:, f] == 3), f] = 2
    nodes.buildBCHash(bc.reshape(-1))
    t = tables_from_nodes(nodes)
    ctx = nodes.dgContext()
    mapO = np.array(ctx.BCmap.get(2, []), dtype=np.int32)
    x, y = t["x"], t["y"]
    rng = np.random.default_rng(seed)
    H = 12.0 + 1.5 * x - 0.8 * y * y + 0.3 * np.sin(3 * x) * np.cos(2 * y)
    h = H + 0.4 * np.exp(-6 * x * x - 6 * y * y) + 0.05 * rng.standard_normal(x.shape)
    hu = 0.8 * rng.standard_normal(x.shape)
    hv = 0.8 * rng.standard_normal(x.shape)
    extras = dict(mapO=mapO, H=H, h=h, hu=hu, hv=hv, CD=2.5e-3, f=1.0070e-4, time=0.37 * 3600 * 12.42)
    return nodes, t, extras


@pytest.fixture(scope="session")
def mock_rccl(tmp_path_factory):
    """A file-based stand-in for librccl.so (tests/mock_rccl: the nine entry points the library binds), so that the ranks of a
    multi-process test can share the single GPU of a test box, which RCCL itself refuses: the environment a rank needs."""
    out = tmp_path_factory.mktemp("mock_rccl")
    lib = out / "libmock_rccl.so"
    cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O2", "-fPIC", "-shared", "--offload-arch=gfx950", "-I/opt/rocm/include",
           os.path.join(ROOT, "tests", "mock_rccl", "mock_rccl.cpp"), "-o", str(lib)]
    build = launch(cmd, timeout=600)
    assert build.returncode == 0, build.stderr[-3000:]
    return {"BDG_RCCL_LIBRARY": str(lib), "BDG_MOCK_RCCL_DIR": str(out)}
