"""GaussFaceContext2D / CubatureContext2D builders (reference src/TriangleNodesProvisioner.cpp:81-381) against
the reference's own specs (src/test/TriangleNodesProvisionerTests.cpp:465-543), the curved-RHS oracle against
the reference function's outputs (tests/golden/sw2d_rhs_curved_*.npz), and closed-form properties."""
import glob
import os

import numpy as np
import pytest

import blitzdg_amd.pyblitzdg as dg
from conftest import GOLDEN

EPS = 50 * np.finfo(float).eps
CURVED = sorted(glob.glob(os.path.join(GOLDEN, "sw2d_rhs_curved_*.npz")))


@pytest.fixture(scope="module")
def coarse_mesh():
    m = dg.MeshManager()
    m.readMesh(os.path.join(GOLDEN, "coarse_box.msh"))
    return m


def test_gauss_face_mesh_and_quadrature(coarse_mesh):  # Should_Build_Gauss_Face_Mesh_and_Quadrature :465-491
    N = 3
    nodes = dg.TriangleNodesProvisioner(N, coarse_mesh)
    g = nodes.buildGaussFaceNodes(2 * (N + 1))
    assert g.NGauss == 2 * (N + 1)
    Fx, Fy = nodes._table(dg.C.TRI_FX), nodes._table(dg.C.TRI_FY)
    P = (np.hypot(Fx[N, 0] - Fx[0, 0], Fy[N, 0] - Fy[0, 0])
         + np.hypot(Fx[2 * N + 1, 0] - Fx[N + 1, 0], Fy[2 * N + 1, 0] - Fy[N + 1, 0])
         + np.hypot(Fx[3 * N + 2, 0] - Fx[2 * (N + 1), 0], Fy[3 * N + 2, 0] - Fy[2 * (N + 1), 0]))
    assert abs(P - g.W[:, 0].sum()) < EPS                     # perimeter of element 0 = sum of its face weights


def test_cubature_volume_mesh(coarse_mesh, known):  # Should_Build_Cubature_Volume_Mesh :492-543
    N = 3
    nodes = dg.TriangleNodesProvisioner(N, coarse_mesh)
    c = nodes.buildCubatureVolumeMesh(3 * (N + 1))
    Ncub = c.NumCubaturePoints
    assert Ncub == 36 and c.NCubature == 12                   # the reference's tabulated degree-12 rule (:504)
    fieldcub = c.V @ np.ones(10)
    assert abs(fieldcub.sum() - Ncub) < 1e-11                 # interpolation reproduces constants
    assert abs((fieldcub * c.w).sum() - 2.0) < EPS * 10       # area of the reference triangle
    assert np.abs(known["tri_cholExpected"] - c.MMChol[:, :, 0]).max() < 6e-4   # the reference's MATLAB factor, its tolerance
    U = c.MMChol[:, :, 7]
    assert np.allclose(U.T @ U, c.MM[:, :, 7], rtol=0, atol=1e-15) and np.all(np.tril(U, -1) == 0)


REFERENCE_RULE_SIZES = [1, 3, 6, 6, 7, 12, 15, 16, 19, 25, 28, 36, 40, 46, 54, 58, 66, 73, 82, 85, 93, 100, 106, 118, 126, 138, 145, 225]


def _monomial_errors(c, degree):
    """Integrals of r^a s^b over the reference triangle {r, s >= -1, r + s <= 0} in closed form
    (shift to the unit simplex: int x^a y^b = a! b! / (a + b + 2)!, x = (r+1)/2, scale 4)."""
    from math import factorial
    x, y, worst = (c.r + 1) / 2, (c.s + 1) / 2, 0.0
    for a in range(degree + 1):
        for b in range(degree + 1 - a):
            exact = 4.0 * factorial(a) * factorial(b) / factorial(a + b + 2)
            worst = max(worst, abs((c.w * x ** a * y ** b).sum() - exact))
    return worst


class _Rule:
    def __init__(self, degree):
        rule = dg.TriangleCubatureRules(degree)
        self.r, self.s, self.w, self.n = rule.rCoord(), rule.sCoord(), rule.weights(), rule.NumCubaturePoints()


@pytest.mark.parametrize("degree", range(1, 29))
def test_tabulated_cubature_rules_have_the_reference_sizes_and_are_exact_to_their_degree(degree):
    """Degrees 1..28: the reference's table (include/TriangleCubatureRules.hpp:26-1804), rule NCubature-1. The point
    counts are the reference's; each rule integrates every monomial of its degree to the accuracy of 15-digit
    constants (the degree-3/4 rule included: the reference's own copy of it is damaged, see the class header)."""
    c = _Rule(degree)
    assert c.n == REFERENCE_RULE_SIZES[degree - 1]
    assert abs(c.w.sum() - 2.0) < 1e-13
    assert np.all(c.r >= -1) and np.all(c.s >= -1) and np.all(c.r + c.s <= 1e-15)     # no point outside the triangle
    assert _monomial_errors(c, degree) < 5e-13


@pytest.mark.parametrize("degree", [29, 30, 33, 40])
def test_computed_cubature_rule_beyond_the_table_is_exact_to_its_degree(coarse_mesh, degree):
    """Beyond degree 28 (where the reference indexes past its table): the conical product rule, through the builder."""
    nodes = dg.TriangleNodesProvisioner(1, coarse_mesh)
    c = nodes.buildCubatureVolumeMesh(degree)
    assert np.all(c.w > 0) and np.all(c.r > -1) and np.all(c.s > -1) and np.all(c.r + c.s < 0)
    assert _monomial_errors(c, degree) < 2e-14
    assert c.NumCubaturePoints == ((degree + 2) // 2) ** 2    # n x n conical product, 2n - 1 >= degree
    assert _Rule(degree).n == c.NumCubaturePoints


@pytest.mark.parametrize("order", [1, 2, 4, 6])
def test_gauss_maps_and_geometry_identities(coarse_mesh, order):
    nodes = dg.TriangleNodesProvisioner(order, coarse_mesh)
    ctx = nodes.dgContext()
    NG = order + 2
    g = nodes.buildGaussFaceNodes(NG)
    K, n3 = ctx.numElements, 3 * NG
    mapM, mapP = g.mapM, g.mapP
    assert np.array_equal(mapM, np.arange(n3 * K))             # reference :257-261
    gx, gy = g.x.flatten("F"), g.y.flatten("F")
    assert np.abs(gx[mapM] - gx[mapP]).max() < 1e-13 and np.abs(gy[mapM] - gy[mapP]).max() < 1e-13
    assert np.array_equal(mapP[mapP], mapM)                    # an involution
    inner = mapP != mapM
    nxF, nyF = g.nx.flatten("F"), g.ny.flatten("F")
    assert np.abs(nxF[mapM][inner] + nxF[mapP][inner]).max() < 1e-13   # opposite normals across a face
    assert np.abs(nyF[mapM][inner] + nyF[mapP][inner]).max() < 1e-13
    bc = g.BCmap
    assert sorted(bc) == [1, 2, 3, 4, 5, 6, 7, 8] and bc[3] == sorted(bc[3], key=lambda i: ((i % n3) // NG, i // n3, i % NG))
    assert set(bc[3]) == set(np.where(~inner)[0])
    # straight elements: nodal-to-Gauss interpolation of the linear coordinates is exact, metric terms constant
    assert np.abs(g.Interp @ ctx.x - g.x).max() < 1e-14
    assert np.abs(g.rx - g.rx[0]).max() < 1e-11 and np.abs(g.J - ctx.J[0]).max() < 1e-12
    # closed surface: sum over faces of n * W vanishes on every element
    assert np.abs((g.nx * g.W).sum(axis=0)).max() < 1e-13 and np.abs((g.ny * g.W).sum(axis=0)).max() < 1e-13


def test_cubature_mesh_repairs_the_nodal_metric_terms_after_set_coordinates(coarse_mesh):
    """reference :129-152: buildCubatureVolumeMesh recomputes J, rx, ry, sx, sy at the nodes from the
    current coordinates (setCoordinates alone does not)."""
    d = np.load(CURVED[-1]) if CURVED else None
    nodes = dg.TriangleNodesProvisioner(4, coarse_mesh)
    ctx = nodes.dgContext()
    x0, y0, J0 = ctx.x, ctx.y, ctx.J
    x, y = x0 + 0.05 * np.sin(2 * y0) * (1 - x0 * x0), y0 + 0.04 * np.sin(3 * x0) * (1 - y0 * y0)
    nodes.setCoordinates(x, y)
    assert np.array_equal(ctx.J, J0)
    cub = nodes.buildCubatureVolumeMesh(15)
    xr, xs, yr, ys = ctx.Dr @ x, ctx.Ds @ x, ctx.Dr @ y, ctx.Ds @ y
    J = xr * ys - xs * yr
    assert np.abs(ctx.J - J).max() < 1e-13 and np.abs(ctx.rx - ys / J).max() < 1e-11
    area = (cub.W.sum(axis=0)).sum()                           # total area is that of the (undeformed) box
    assert abs(area - 4.0) < 1e-10
    assert d is None or "cubV" in d


@pytest.mark.parametrize("path", CURVED, ids=[os.path.basename(p)[16:-4] for p in CURVED])
def test_curved_oracle_reproduces_the_reference_function_bit_for_bit(path):
    """oracle/oracle_np.py::sw2d_rhs_curved against the stored output of the reference's
    swhelpers.rhs.sw2dComputeRHS_curved (tests/golden/make_golden.py::curved_case)."""
    from oracle import oracle_np
    d = np.load(path)
    r = oracle_np.sw2d_rhs_curved(d["h"], d["hu"], d["hv"], d["hN"], d["zx"], d["zy"], float(d["g"]), float(d["f"]),
                                  d["CD"], d)
    for c in range(4):
        assert np.array_equal(r[c], d[f"rhs{c + 1}"])


@pytest.mark.parametrize("path", CURVED, ids=[os.path.basename(p)[16:-4] for p in CURVED])
def test_curved_fixture_tables_are_what_the_builders_produce(path):
    """The context tables stored with a fixture are reproduced by today's builders from the stored
    coordinates (so the fixtures pin the builders too, and cannot drift from them silently)."""
    d = np.load(path)
    name = os.path.basename(path)
    mesh = dg.MeshManager()
    if "coarse_box" in name:
        mesh.readMesh(os.path.join(GOLDEN, "coarse_box.msh"))
    elif "box6x5" in name:
        mesh.buildBoxMesh(6, 5, shuffleSeed=12345)
    else:
        mesh.buildBoxMesh(3, 2)
    N = int(d["order"])
    nodes = dg.TriangleNodesProvisioner(N, mesh)
    nodes.setCoordinates(d["x"], d["y"])
    g = nodes.buildGaussFaceNodes(int(d["NGauss"]))
    c = nodes.buildCubatureVolumeMesh(int(d["NCubature"]))
    for key, got in (("cubV", c.V), ("cubDr", c.Dr), ("cubW", c.W), ("cubrx", c.rx), ("cubsy", c.sy), ("MMChol", c.MMChol),
                     ("gInterp", g.Interp), ("gW", g.W), ("gnx", g.nx), ("gny", g.ny), ("gmapM", g.mapM)):
        assert np.array_equal(got, d[key]), key
    if "periodic" not in name:
        assert np.array_equal(g.mapP, d["gmapP"]) and np.array_equal(np.array(g.BCmap[3], dtype=np.int32), d["gmapW"])
