"""Set-up helpers of the reference's curved shallow-water driver, reference swhelpers/maps.py:3-65
(call sites sw2d_curved.py:43, 144-145): same names, argument order, in-place behaviour and return values.

``makeMapsPeriodic`` rewires the '+' map of the nodes on the two open ends of a channel onto their partners on the
opposite end; ``correctBCTable`` tags the faces on the channel ends x = 0 and x = 8000.
"""
import numpy as np

# geometric constants of the reference functions (its channel is 8 km long, coordinates in metres)
_SAME_ROW_TOL = 1e-3     # maps.py:11   |y_partner - y| below this: same height across the channel
_FAR_SIDE_DIST = 1000    # maps.py:11   |x_partner - x| above this: the other end
_FACE_TOL = 1e-6         # maps.py:59,61
_CHANNEL_ENDS = (0.0, 8000.0)


def makeMapsPeriodic(vmapM, vmapP, vmapO, xFlat, yFlat, xO, yO):
    """reference swhelpers/maps.py:3-45. For every node id in ``vmapO`` (coordinates ``xFlat[i], yFlat[i]``) the candidates
    are the entries of ``vmapO`` whose ``yO`` agrees to 1e-3 and whose ``xO`` lies more than 1000 away. One candidate: that
    is the partner. Several (a node shared by two faces appears twice on the far end): the first candidate if it is
    adjacent (id difference <= 1) to the first candidate of the previous or the next outflow node, else the second.
    Every entry of ``vmapP`` whose ``vmapM`` is an outflow node is then overwritten by its partner -- IN PLACE, as the
    reference does; returns ``(vmapM, vmapP)``.

    Differences from the reference text that cannot change a result: candidates are found with one vectorised comparison
    per node instead of np.where on two masks; the look-up is built directly (the reference fills a dict created with
    dict.fromkeys, which keeps the FIRST position of a repeated node id and assigns partners by position in that
    de-duplicated key order -- reproduced here, including for repeated ids)."""
    vmapO = np.asarray(vmapO)
    xO, yO = np.asarray(xO), np.asarray(yO)
    candidates = []
    for i in vmapO:
        far = (np.abs(yO - yFlat[i]) < _SAME_ROW_TOL) & (np.abs(xO - xFlat[i]) > _FAR_SIDE_DIST)
        candidates.append(vmapO[np.flatnonzero(far)])

    n = len(candidates)

    def near(a, b):
        # the reference's `abs(a - b) <= 1` used as a condition: a is a candidate ARRAY at the two ends of the list
        # (maps.py:26,30) and its first entry in between (maps.py:28); truth of a one-element array = its element,
        # more than one element raises there and here alike
        return bool(abs(a - b) <= 1)

    partners = []
    for i, c in enumerate(candidates):
        if len(c) == 1:
            partners.append(c[0])
        elif i == 0 and near(candidates[i + 1], c[0]):
            partners.append(c[0])
        elif 0 < i < n - 1 and (near(candidates[i - 1][0], c[0]) or near(candidates[i + 1][0], c[0])):
            partners.append(c[0])
        elif i == n - 1 and near(candidates[i - 1], c[0]):
            partners.append(c[0])
        else:
            partners.append(c[1])

    # dict.fromkeys(vmapOM) + "for i, key in enumerate(lookup.keys())": distinct ids in first-seen order get partners[0..]
    lookup = {}
    for key in vmapO.tolist():
        if key not in lookup:
            lookup[key] = None
    for i, key in enumerate(lookup):
        lookup[key] = partners[i]

    for i in range(len(vmapM)):
        key = int(vmapM[i])
        if key in lookup:
            vmapP[i] = lookup[key]
    return vmapM, vmapP


def correctBCTable(bcType, EToV, Verts, bcTag):
    """reference swhelpers/maps.py:49-65: every face (local vertices 0-1, 1-2, 2-0) whose midpoint lies on x = 0 or
    x = 8000 (to 1e-6) gets ``bcTag`` in the (K, 3) table ``bcType`` -- in place; returns ``bcType``."""
    bcType = bcType if isinstance(bcType, np.ndarray) else np.asarray(bcType)
    EToV = np.asarray(EToV)
    vx = np.asarray(Verts)[:, 0]
    K = len(bcType)
    ends = ((0, 1), (1, 2), (2, 0))
    for face, (a, b) in enumerate(ends):
        mid = 0.5 * (vx[EToV[:K, a]] + vx[EToV[:K, b]])
        on_end = np.zeros(K, dtype=bool)
        for x_end in _CHANNEL_ENDS:
            on_end |= np.abs(mid - x_end) < _FACE_TOL
        bcType[on_end, face] = bcTag
    return bcType
