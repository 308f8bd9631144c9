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
