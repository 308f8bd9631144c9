"""blitzdg_amd.swhelpers.maps / blitzdg_amd.meshhelpers.curved against the outputs of the reference's own functions
(reference swhelpers/maps.py:3-65, meshhelpers/curved.py:5-136) stored in tests/golden/curved_helpers_*.npz by
tests/golden/make_golden.py::curved_helpers_case -- a channel of the reference driver's size with open ends and a headland
in the top wall, the helpers called in the driver's order (sw2d_curved.py:43-145). Index outputs exact; coordinates to
1e-13 of the channel length. Host code only: no GPU."""
import glob
import os
import types

import numpy as np
import pytest

import blitzdg_amd.pyblitzdg as dg
from blitzdg_amd.meshhelpers.curved import adjustStraightEdges, deformAndBlendElements
from blitzdg_amd.swhelpers.maps import correctBCTable, makeMapsPeriodic
from conftest import GOLDEN

CASES = sorted(glob.glob(os.path.join(GOLDEN, "curved_helpers_*.npz")))
COORD_TOL = 1e-13 * 8000.0


@pytest.fixture(params=CASES, ids=[os.path.basename(p)[15:-4] for p in CASES])
def case(request):
    return np.load(request.param)


def holder(d):
    return types.SimpleNamespace(x=d["x0"].copy(), y=d["y0"].copy(), r=d["r"], s=d["s"], Fmask=d["Fmask"], numFaces=3)


def test_fixtures_exist():
    assert CASES


def test_correct_bc_table(case):
    bc = case["bcType0"].copy()
    out = correctBCTable(bc, case["EToV"], case["Verts0"], 2)
    assert out is bc                                              # in place, as the reference
    assert np.array_equal(out, case["bcType"]) and (out == 2).sum() > 0 and (out == 3).sum() > 0


def test_adjust_straight_edges(case):
    verts = case["Verts0"].copy()
    out, modified, faces = adjustStraightEdges(verts, case["EToV"], case["bcFaces"], case["xSpline"], case["ySpline"], holder(case))
    assert out is verts
    assert np.array_equal(np.array(faces), case["curvedFaces"]) and len(faces) >= 3
    assert np.array_equal(modified, case["modifiedVerts"])
    assert np.array_equal(out, case["Verts1"])                    # snapped onto sample points: the very same numbers


def test_deform_and_blend_elements(case):
    ctx = holder(case)
    tck = lambda which: (case[f"spl{which}_t"], case[f"spl{which}_c"], int(case["spl_k"]))    # noqa: E731
    x, y, els = deformAndBlendElements(case["Verts1"], case["EToV"], [list(f) for f in case["curvedFaces"]], case["xSpline"],
                                       case["ySpline"], case["ss"], tck("x"), tck("y"), ctx, int(case["order"]))
    assert x is ctx.x and y is ctx.y                              # blended in place in a context that holds its arrays
    assert list(els) == list(case["curvedEls"])
    assert np.abs(x - case["x1"]).max() <= COORD_TOL and np.abs(y - case["y1"]).max() <= COORD_TOL
    moved = np.abs(x - case["x0"]) + np.abs(y - case["y0"]) > 0
    assert moved.any() and set(np.flatnonzero(moved.any(axis=0))) <= set(int(k) for k in els)


def test_deform_and_blend_with_a_context_that_copies(case):
    """With a context whose properties hand out fresh arrays (the reference's own DGContext2D, and this package's) the
    in-place blending is lost and the undeformed coordinates come back -- the reference function's behaviour, kept."""
    class Copies:
        numFaces = 3
        r = property(lambda self: case["r"].copy())
        s = property(lambda self: case["s"].copy())
        Fmask = property(lambda self: case["Fmask"].copy())
        x = property(lambda self: case["x0"].copy())
        y = property(lambda self: case["y0"].copy())
    tck = lambda which: (case[f"spl{which}_t"], case[f"spl{which}_c"], int(case["spl_k"]))    # noqa: E731
    x, y, els = deformAndBlendElements(case["Verts1"], case["EToV"], [list(f) for f in case["curvedFaces"]], case["xSpline"],
                                       case["ySpline"], case["ss"], tck("x"), tck("y"), Copies(), int(case["order"]))
    assert np.array_equal(x, case["x0"]) and np.array_equal(y, case["y0"]) and list(els) == list(case["curvedEls"])


def test_make_maps_periodic_nodal_and_gauss(case):
    x1, y1 = case["x1"], case["y1"]
    xF, yF = x1.flatten("F"), y1.flatten("F")
    vM, vP, vO = case["vmapM"].copy(), case["vmapP"].copy(), case["vmapO"]
    outM, outP = makeMapsPeriodic(vM, vP, vO, xF, yF, xF[vO], yF[vO])
    assert outM is vM and outP is vP                              # vmapP is rewired in place
    assert np.array_equal(outP, case["vmapP_periodic"]) and np.array_equal(outM, case["vmapM_periodic"])
    assert (outP != case["vmapP"]).sum() >= vO.size
    gM, gP, gO = case["gmapM"].copy(), case["gmapP"].copy(), case["gmapO"]
    outM, outP = makeMapsPeriodic(gM, gP, gO, case["gxFlat"], case["gyFlat"], case["gxFlat"][gO], case["gyFlat"][gO])
    assert np.array_equal(outP, case["gmapP_periodic"]) and np.array_equal(outM, case["gmapM_periodic"])
    # every rewired Gauss point looks at a point of the opposite end at the same height
    hit = np.flatnonzero(outP != case["gmapP"])
    assert hit.size == gO.size
    assert np.abs(case["gyFlat"][outP[hit]] - case["gyFlat"][gM[hit]]).max() < 1e-3
    assert np.abs(case["gxFlat"][outP[hit]] - case["gxFlat"][gM[hit]]).min() > 1000


def test_vandermonde_builder_is_the_provisioners_matrix():
    """VandermondeBuilder.buildVandermondeMatrix (pyblitzdg.cpp:92-93): on the 1-D provisioner's nodes it is that provisioner's V
    (pinned against the reference's literals in test_setup_golden.py); the inverse inverts; a column count below the point
    count gives the rectangular matrix deformAndBlendElements uses."""
    n1 = dg.Nodes1DProvisioner(4, 3, -1.0, 1.0)
    n1.buildNodes()
    b = dg.VandermondeBuilder()
    V, Vinv = b.buildVandermondeMatrix(n1.rGrid.reshape(-1), True, 4)
    assert V.shape == (5, 5) and np.abs(V @ Vinv - np.eye(5)).max() < 1e-13
    assert np.array_equal(V, n1.V)
    (Vr,) = b.buildVandermondeMatrix(np.linspace(-1, 1, 9), False, 4)
    assert Vr.shape == (9, 5) and np.allclose(Vr[:, 0], 1 / np.sqrt(2))
    with pytest.raises(Exception):
        b.buildVandermondeMatrix(np.linspace(-1, 1, 9), True, 4)   # the inverse of a rectangular matrix
