"""Curved-wall set-up of the reference's curved shallow-water driver, reference meshhelpers/curved.py:5-136
(call sites sw2d_curved.py:105-106): same names, argument order, in-place behaviour and return values.

``adjustStraightEdges`` snaps the end points of the boundary faces that lie along a sampled wall curve onto the curve's
nearest sample points; ``deformAndBlendElements`` then moves the face nodes of those faces onto the spline and blends
the displacement into the element (Gordon-Hall blending), which makes the element curved.

Both functions read and write ``ctx.x`` / ``ctx.y`` through attribute access exactly where the reference does. That
matters: with a context whose properties hand out FRESH arrays on every access -- the reference's own pyblitzdg
(src/DGContext2D.cpp:45-57) and this package's ``pyblitzdg.DGContext2D`` -- the in-place blending lands in a temporary
and the returned coordinates are the undeformed ones; with a context that HOLDS its arrays (``types.SimpleNamespace(x=...,
y=..., r=..., s=..., Fmask=..., numFaces=3)``) the returned coordinates are the blended ones. ``examples/sw2d_curved.py``
passes such a holder.
"""
import numpy as np
from scipy.interpolate import splev

from .. import pyblitzdg as dg

_SNAP_RADIUS = 200        # curved.py:25   a face end further than this from every curve sample is left alone
_ON_CURVE_TOL = 1.0e-8    # curved.py:83-84
_APEX_TOL = 1.0e-7        # curved.py:128


def adjustStraightEdges(Verts, EToV, bcFaces, xSpline, ySpline, ctx):
    """reference meshhelpers/curved.py:5-51. ``bcFaces``: rows ``[[element, face]]`` (the (n, 1, 2) array
    sw2d_curved.py:103 builds). A face both of whose end points are within 200 of the sampled curve
    ``(xSpline, ySpline)`` has them moved onto their nearest samples (``Verts`` is modified IN PLACE). Returns
    ``(Verts, modifiedVerts, curvedFaces)``: a 0/1 flag per vertex and the list of ``[element, face]`` pairs moved."""
    xSpline, ySpline = np.asarray(xSpline), np.asarray(ySpline)
    moved = np.zeros(Verts.shape[0])
    curvedFaces = []
    for entry in bcFaces:
        el, face = entry[0][0], entry[0][1]
        ends = (EToV[el, face], EToV[el, (face + 1) % ctx.numFaces])
        nearest = []
        for v in ends:
            dist = np.hypot(xSpline - Verts[v, 0], ySpline - Verts[v, 1])
            at = np.argmin(dist)
            nearest.append((at, dist[at]))
        if nearest[0][1] > _SNAP_RADIUS or nearest[1][1] > _SNAP_RADIUS:
            continue
        curvedFaces.append([el, face])
        # the second end is looked up BEFORE the first is moved (both distances above use the old coordinates)
        for v, (at, _) in zip(ends, nearest):
            Verts[v, 0] = xSpline[at]
            Verts[v, 1] = ySpline[at]
            moved[v] = 1
    return Verts, moved, curvedFaces


# (first vertex, second vertex, which reference coordinate runs along the face) per local face: curved.py:60-71
_FACE_PARAM = ((0, 1, "r"), (1, 2, "s"), (0, 2, "s"))


def deformAndBlendElements(Verts, EToV, curvedFaces, xSpline, ySpline, ss, xs, ys, ctx, NOrder):
    """reference meshhelpers/curved.py:53-136. For each ``[k, f]`` of ``curvedFaces``: the face's end points must be
    samples of the curve (``adjustStraightEdges`` put them there), their parameters ``t1, t2`` come from ``ss``; the face
    nodes are sent to ``t = t1 (1 - fr)/2 + t2 (1 + fr)/2`` on the splines ``xs, ys`` (scipy tck, ``splev(..., ext=3)``);
    the displacement of the face nodes is extended along the face coordinate with the 1-D Vandermonde matrices
    (``Vvol Vface^-1``) and blended into the element with ``-(r+s)/(1-vr)`` (faces 0 and 2) or ``(r+1)/(1-vr)`` (face 1),
    skipping the nodes with ``vr = 1``. Returns ``(ctx.x, ctx.y, curvedEls)``; see the module docstring for what the first
    two are. A face whose two ends are the same sample (a degenerate interval) is listed in ``curvedEls`` and not deformed,
    as in the reference."""
    xSpline, ySpline = np.asarray(xSpline), np.asarray(ySpline)
    builder = dg.VandermondeBuilder()
    curvedEls = []
    for k, f in curvedFaces:
        curvedEls.append(k)
        a, b, coord = _FACE_PARAM[f]
        v1, v2 = EToV[k, a], EToV[k, b]
        vr = getattr(ctx, coord)
        fmsk = ctx.Fmask[:, f]
        fr = vr[fmsk]

        def samples_at(v):
            return np.where(np.sqrt((Verts[v, 0] - xSpline) ** 2 + (Verts[v, 1] - ySpline) ** 2) < _ON_CURVE_TOL)[0]
        i1, i2 = samples_at(v1), samples_at(v2)
        # curved.py:95: `if v1s_inds[0] == v2s_inds[0]: continue` on the index ARRAYS (an empty or a longer array makes the
        # comparison's truth value ambiguous there: the same ValueError here)
        if bool(i1 == i2):
            continue
        t1, t2 = ss[i1], ss[i2]
        tface = 0.5 * t1 * (1 - fr) + 0.5 * t2 * (1 + fr)
        dxf = splev(tface, xs, ext=3) - ctx.x[fmsk, k]
        dyf = splev(tface, ys, ext=3) - ctx.y[fmsk, k]

        Vface, Vfinv = builder.buildVandermondeMatrix(fr, True, NOrder)
        Vvol, = builder.buildVandermondeMatrix(vr, False, NOrder)
        vdx = np.dot(Vvol, np.dot(Vfinv, dxf))
        vdy = np.dot(Vvol, np.dot(Vfinv, dyf))

        r, s = ctx.r, ctx.s
        ids = np.where(np.abs(1 - vr) > _APEX_TOL)[0]
        if f == 1:
            blend = (r[ids] + 1) / (1 - vr[ids])
        else:
            blend = -(r[ids] + s[ids]) / (1 - vr[ids])
        ctx.x[ids, k] += blend * vdx[ids]
        ctx.y[ids, k] += blend * vdy[ids]
    return ctx.x, ctx.y, curvedEls
