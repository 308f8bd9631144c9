#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/. Run in the BUILD
container only (needs /root/reference); the fixtures it writes are plain data and
travel to the GPU box, the reference does not.

  python tests/golden/make_golden.py

Writes
  coarse_box.msh            copy of the reference's input/coarse_box.msh (the mesh its
                            own tests use; data file)
  reference_known_answers.npz
                            the known-answer vectors of the reference's unit tests,
                            parsed out of the literals in src/test/*.cpp:
                            vmapM/vmapP/vmapB/mapB/mapW (TriangleNodesProvisionerTests.cpp
                            :434-441), Dr/Ds (:310-330), Lift (:146-155), V (:219-228),
                            equilateral nodes (:349-369), cubature mass-matrix Cholesky factor
                            (:526-536), EToV/EToE/EToF/BCType/verts
                            (MeshManagerTests.cpp:202-206), 1-D V/Dr/x/Lift/EToE/EToF/
                            vmapM/vmapP (Nodes1DProvisionerTests.cpp:52-247)
  sw2d_rhs_<case>.npz       inputs (tables built by THIS repo's host code + seeded
                            fields) and the RHS computed by the REFERENCE's
                            swhelpers.rhs.sw2dComputeRHS (swhelpers/rhs.py:178-311) with
                            hN = 0, f = CD = 0, zx = zy = 0  (variant D == variant A up
                            to round-off)
  sw2d_rhs4_<case>.npz      the same function with tracer, Coriolis array, drag and bed slope
  sw2d_rhs4n_<case>.npz     the same on per-node (non-affine) rx .. sy, nx, ny, Fscale of a smoothly deformed mesh
  sw2d_rhs_curved_<case>.npz
                            swhelpers.rhs.sw2dComputeRHS_curved (swhelpers/rhs.py:6-176) on deformed
                            meshes; the Gauss-face and cubature contexts it reads are built by THIS
                            repo's buildGaussFaceNodes / buildCubatureVolumeMesh and stored with it
  sw2d_rhsB_degenerate_<case>.npz
                            the same function on a state (uniform depth and speed, flat bed, Coriolis only)
                            for which the C++ driver's variant B must give the same RHS
  sw2d_rhsB_bed_<case>.npz, sw2d_rhsB_bed_drag_<case>.npz
                            the same function over a continuous, non-flat bed (zx = -Hx, zy = -Hy) on a state with
                            v = 0 and |u| + sqrt(g h) uniform, without and with drag: variant B's star states,
                            bed-slope source and RHS2 drag must reproduce it
  sw2d_bigcurved_<case>.npz the same function at g = 9.81 on a 2080-element deformed mesh (compact: coordinates, fields, outputs; the
                            contexts are rebuilt by the test)
  curved_helpers_<case>.npz the reference's correctBCTable / makeMapsPeriodic (swhelpers/maps.py:3-65) and adjustStraightEdges /
                            deformAndBlendElements (meshhelpers/curved.py:5-136) on a channel with a headland, inputs and outputs
  advec1d_rhs_N4_K100.npz   advec1dComputeRHS(u, c, nodes1d) of the reference SCRIPT advec1d.py:12-39
  sw2d_rhsC_<case>.npz      sw2dComputeRHS(h,hu,hv,hN,g,H,f,ctx) of the reference SCRIPT sw2d.py:37-146
                            ("variant C"), its two function definitions compiled on their own
"""
import os
import re
import shutil
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def literals(src, name, occurrence=0):
    """All numbers of the `occurrence`-th `name = a, b, c, ...;` initialiser in a C++ source."""
    hits = re.findall(r"\b" + re.escape(name) + r"\s*=\s*([^;]*);", src)
    body = hits[occurrence]
    return np.array([float(tok) for tok in re.split(r"[,\s]+", body.strip()) if tok])


def known_answers():
    tri = open(os.path.join(REF, "src/test/TriangleNodesProvisionerTests.cpp")).read()
    msh = open(os.path.join(REF, "src/test/MeshManagerTests.cpp")).read()
    n1d = open(os.path.join(REF, "src/test/Nodes1DProvisionerTests.cpp")).read()
    out = {}
    for key in ("expectedvmapM", "expectedvmapP", "expectedvmapB", "expectedmapB", "expectedmapW"):
        out["tri_" + key[8:]] = literals(tri, key).astype(np.int32)
    out["tri_Dr"] = literals(tri, "expectedDr").reshape(10, 10)
    out["tri_Ds"] = literals(tri, "expectedDs").reshape(10, 10)
    out["tri_Lift"] = literals(tri, "Lift_expected").reshape(10, 12)
    out["tri_V"] = literals(tri, "V_expected", 1).reshape(10, 10)
    out["tri_cholExpected"] = literals(tri, "cholExpected").reshape(10, 10)   # :526-536, asserted to 6e-4 (:542)
    out["tri_eq_x"] = literals(tri, "expectedx")
    out["tri_eq_y"] = literals(tri, "expectedy")
    out["mesh_verts"] = literals(msh, "expectedVerts").reshape(29, 3)
    out["mesh_EToV"] = literals(msh, "expectedElements").astype(np.int32).reshape(40, 3)
    out["mesh_EToE"] = literals(msh, "expectedEToE").astype(np.int32).reshape(40, 3)
    out["mesh_EToF"] = literals(msh, "expectedEToF").astype(np.int32).reshape(40, 3)
    out["mesh_BCType"] = literals(msh, "expectedBcTable").astype(np.int32).reshape(40, 3)
    out["n1d_V"] = literals(n1d, "expectedV").reshape(4, 4)
    out["n1d_Dr"] = literals(n1d, "expectedDr").reshape(4, 4)
    out["n1d_x"] = literals(n1d, "expectedx").reshape(4, 5)
    out["n1d_Lift"] = literals(n1d, "expectedLift").reshape(4, 2)
    out["n1d_EToE"] = literals(n1d, "expectedEToE").astype(np.int32).reshape(5, 2)
    out["n1d_EToF"] = literals(n1d, "expectedEToF").astype(np.int32).reshape(5, 2)
    out["n1d_vmapM"] = literals(n1d, "expectedVmapM").astype(np.int32)
    out["n1d_vmapP"] = literals(n1d, "expectedVmapP").astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "reference_known_answers.npz"), **out)
    print("reference_known_answers.npz:", sorted(out))


def seeded_fields(x, y, seed=0):
    """H = 10, eta = exp(-10x^2-10y^2), momentum perturbed with
    0.1*standard_normal from default_rng(seed)."""
    rng = np.random.default_rng(seed)
    h = 10.0 + np.exp(-10 * x * x - 10 * y * y)
    hu = 0.1 * rng.standard_normal(x.shape)
    hv = 0.1 * rng.standard_normal(x.shape)
    return h, hu, hv


def rhs_case(name, mesh, order, g=9.81):
    import blitzdg_amd.pyblitzdg as dg
    sys.path.insert(0, REF)
    if not hasattr(np, "float"):
        np.float = float  # swhelpers/rhs.py:262 uses the alias NumPy removed in 1.24
    from swhelpers.rhs import sw2dComputeRHS  # the reference's own NumPy RHS

    nodes = dg.TriangleNodesProvisioner(order, mesh)
    nodes.buildFilter(0.9 * order, order)
    ctx = nodes.dgContext()
    tabs = {k: getattr(ctx, k) for k in
            ("Dr", "Ds", "Lift", "rx", "sx", "ry", "sy", "nx", "ny", "Fscale", "vmapM", "vmapP", "x", "y")}
    tabs["Filter"] = ctx.filter
    bcmap = ctx.BCmap
    tabs["mapW"] = np.array(bcmap.get(3, []), dtype=np.int32)
    h, hu, hv = seeded_fields(tabs["x"], tabs["y"])
    ref_ctx = types.SimpleNamespace(BCmap=bcmap, nx=tabs["nx"], ny=tabs["ny"], rx=tabs["rx"], sx=tabs["sx"],
                                    ry=tabs["ry"], sy=tabs["sy"], Dr=tabs["Dr"], Ds=tabs["Ds"],
                                    numFacePoints=ctx.numFacePoints, numElements=ctx.numElements,
                                    numFaces=ctx.numFaces, Lift=tabs["Lift"], Fscale=tabs["Fscale"])
    zeros = np.zeros_like(h)
    r1, r2, r3, r4 = sw2dComputeRHS(h, hu, hv, zeros.copy(), zeros, zeros, g, 10.0 + zeros, 0.0, 0.0, ref_ctx,
                                    tabs["vmapM"], tabs["vmapP"])
    assert np.all(r4 == 0.0)
    np.savez_compressed(os.path.join(HERE, f"sw2d_rhs_{name}.npz"), order=order, g=g, h=h, hu=hu, hv=hv,
                        rhs1=r1, rhs2=r2, rhs3=r3, **tabs)
    print(f"sw2d_rhs_{name}.npz: K={ctx.numElements} Np={ctx.numLocalPoints} |rhs|max="
          f"{max(abs(r1).max(), abs(r2).max(), abs(r3).max()):.6g}")


def deformed_tables(ctx, order, tabs):
    """Per-node metric terms and face geometry of a smoothly deformed mesh by the reference's formulas
    (src/TriangleNodesProvisioner.cpp:810-892) from this repository's Dr / Ds: genuinely non-affine rx .. Fscale."""
    x0, y0 = ctx.x, ctx.y
    x = x0 + 0.06 * np.sin(2.1 * y0) * (1 - x0 * x0)
    y = y0 + 0.05 * np.sin(2.7 * x0 + 0.3) * (1 - y0 * y0)
    Dr, Ds = ctx.Dr, ctx.Ds
    xr, xs, yr, ys = Dr @ x, Ds @ x, Dr @ y, Ds @ y
    J = xr * ys - xs * yr
    assert J.min() > 0
    tabs.update(rx=ys / J, sx=-yr / J, ry=-xs / J, sy=xr / J, x=x, y=y)
    Nfp = order + 1
    Fm = ctx.Fmask.T.reshape(-1) if ctx.Fmask.shape[0] == Nfp else ctx.Fmask.reshape(-1)
    fxr, fxs, fyr, fys = xr[Fm], xs[Fm], yr[Fm], ys[Fm]
    nxf, nyf = np.empty_like(fxr), np.empty_like(fxr)
    nxf[:Nfp], nyf[:Nfp] = fyr[:Nfp], -fxr[:Nfp]
    nxf[Nfp:2 * Nfp], nyf[Nfp:2 * Nfp] = fys[Nfp:2 * Nfp] - fyr[Nfp:2 * Nfp], -fxs[Nfp:2 * Nfp] + fxr[Nfp:2 * Nfp]
    nxf[2 * Nfp:], nyf[2 * Nfp:] = -fys[2 * Nfp:], fxs[2 * Nfp:]
    sJ = np.hypot(nxf, nyf)
    tabs.update(nx=nxf / sJ, ny=nyf / sJ, Fscale=sJ / J[Fm])
    return tabs


def rhs4_case(name, mesh, order, g=9.81, deformed=False):
    """Variant D: tracer + Coriolis (array f) + drag + bed slope, all non-trivial; output of the
    reference's swhelpers.rhs.sw2dComputeRHS itself. deformed: on the per-node geometry tables of a smoothly deformed
    mesh (sw2d_rhs4n_<case>.npz) -- the function takes whatever rx .. Fscale the context holds."""
    import blitzdg_amd.pyblitzdg as dg
    sys.path.insert(0, REF)
    if not hasattr(np, "float"):
        np.float = float
    from swhelpers.rhs import sw2dComputeRHS

    nodes = dg.TriangleNodesProvisioner(order, mesh)
    nodes.buildFilter(0.9 * order, order)
    ctx = nodes.dgContext()
    tabs = {k: getattr(ctx, k) for k in
            ("Dr", "Ds", "Lift", "rx", "sx", "ry", "sy", "nx", "ny", "Fscale", "vmapM", "vmapP", "x", "y")}
    tabs["Filter"] = ctx.filter
    bcmap = ctx.BCmap
    tabs["mapW"] = np.array(bcmap.get(3, []), dtype=np.int32)
    if deformed:
        deformed_tables(ctx, order, tabs)
    x, y = tabs["x"], tabs["y"]
    h, hu, hv = seeded_fields(x, y)
    rng = np.random.default_rng(1)
    hN = h * (1.0 + 0.3 * np.sin(2 * x) * np.cos(3 * y)) + 0.05 * rng.standard_normal(x.shape)
    H = 10.0 - 0.5 * x + 0.25 * y * y
    zx, zy = -0.5 + 0 * x, 0.5 * y
    f = 1e-1 * (1.0 + 0.5 * y)
    CD = 2.5e-2
    ref_ctx = types.SimpleNamespace(BCmap=bcmap, nx=tabs["nx"], ny=tabs["ny"], rx=tabs["rx"], sx=tabs["sx"],
                                    ry=tabs["ry"], sy=tabs["sy"], Dr=tabs["Dr"], Ds=tabs["Ds"],
                                    numFacePoints=ctx.numFacePoints, numElements=ctx.numElements,
                                    numFaces=ctx.numFaces, Lift=tabs["Lift"], Fscale=tabs["Fscale"])
    r = sw2dComputeRHS(h, hu, hv, hN, zx, zy, g, H, f, CD, ref_ctx, tabs["vmapM"], tabs["vmapP"])
    stem = "sw2d_rhs4n" if deformed else "sw2d_rhs4"
    np.savez_compressed(os.path.join(HERE, f"{stem}_{name}.npz"), order=order, g=g, h=h, hu=hu, hv=hv, hN=hN,
                        H=H, zx=zx, zy=zy, f=f, CD=CD, rhs1=r[0], rhs2=r[1], rhs3=r[2], rhs4=r[3], **tabs)
    print(f"{stem}_{name}.npz: K={ctx.numElements} Np={ctx.numLocalPoints} |rhs|max="
          f"{max(abs(a).max() for a in r):.6g}")


def rhsB_degenerate_case(name, mesh, order, g=9.81, f=0.05):
    """Variant B (src/sw2d/main.cpp:279-484) where it degenerates to something the reference's Python RHS can
    produce: flat bed (star states are the identity, no bed-slope source), no open boundary, no drag, and a state
    with |u| and h uniform -- h = 10, (u, v) = 0.8 (cos theta, sin theta) -- so that the speed |u| + sqrt(g h) is the
    same at every face node and the per-face Lax-Friedrichs speed of the Python RHS equals variant B's one global
    speed (to the last bits of the sqrt). Output of swhelpers.rhs.sw2dComputeRHS with hN = 0, CD = 0, zx = zy = 0."""
    import blitzdg_amd.pyblitzdg as dg
    sys.path.insert(0, REF)
    if not hasattr(np, "float"):
        np.float = float
    from swhelpers.rhs import sw2dComputeRHS

    nodes = dg.TriangleNodesProvisioner(order, mesh)
    nodes.buildFilter(0.9 * order, order)
    ctx = nodes.dgContext()
    tabs = {k: getattr(ctx, k) for k in
            ("Dr", "Ds", "Lift", "rx", "sx", "ry", "sy", "nx", "ny", "Fscale", "vmapM", "vmapP", "x", "y")}
    tabs["Filter"] = ctx.filter
    bcmap = ctx.BCmap
    tabs["mapW"] = np.array(bcmap.get(3, []), dtype=np.int32)
    x, y = tabs["x"], tabs["y"]
    theta = 1.3 * x + 0.7 * y * y
    h = 10.0 + 0 * x
    hu, hv = h * 0.8 * np.cos(theta), h * 0.8 * np.sin(theta)
    ref_ctx = types.SimpleNamespace(BCmap=bcmap, nx=tabs["nx"], ny=tabs["ny"], rx=tabs["rx"], sx=tabs["sx"],
                                    ry=tabs["ry"], sy=tabs["sy"], Dr=tabs["Dr"], Ds=tabs["Ds"],
                                    numFacePoints=ctx.numFacePoints, numElements=ctx.numElements,
                                    numFaces=ctx.numFaces, Lift=tabs["Lift"], Fscale=tabs["Fscale"])
    z = np.zeros_like(h)
    r = sw2dComputeRHS(h, hu, hv, z.copy(), z, z, g, h.copy(), f, 0.0, ref_ctx, tabs["vmapM"], tabs["vmapP"])
    np.savez_compressed(os.path.join(HERE, f"sw2d_rhsB_degenerate_{name}.npz"), order=order, g=g, f=f, h=h, hu=hu, hv=hv,
                        H=h.copy(), rhs1=r[0], rhs2=r[1], rhs3=r[2], **tabs)
    print(f"sw2d_rhsB_degenerate_{name}.npz: K={ctx.numElements} Np={ctx.numLocalPoints} |rhs|max="
          f"{max(abs(a).max() for a in r[:3]):.6g}")


def rhsB_bed_case(name, mesh, order, CD, g=9.81, f=0.05):
    """More of variant B (src/sw2d/main.cpp:279-484) pinned by the reference's Python RHS: a CONTINUOUS, NON-FLAT bed
    H(x, y) -- the star states (:356-368) are exercised and must come out as the identity, the bed-slope source
    (:461-469, RHS2 += g h Hx) is active and equals variant D's with zx = -Hx, zy = -Hy -- with v = 0 and
    u = c0 - sqrt(g h), so that |u| + sqrt(g h) = c0 at every face node and B's one global speed (:414) equals every
    face's own maximum. With CD > 0 the drag of RHS2 (:474, -CD u |u|) is pinned as well; v = 0 makes the drag of RHS3,
    whose sign differs between the two sources (swhelpers/rhs.py:307), vanish in both. Coriolis on, walls everywhere.
    Output of swhelpers.rhs.sw2dComputeRHS with hN = 0."""
    import blitzdg_amd.pyblitzdg as dg
    sys.path.insert(0, REF)
    if not hasattr(np, "float"):
        np.float = float
    from swhelpers.rhs import sw2dComputeRHS

    nodes = dg.TriangleNodesProvisioner(order, mesh)
    nodes.buildFilter(0.9 * order, order)
    ctx = nodes.dgContext()
    tabs = {k: getattr(ctx, k) for k in
            ("Dr", "Ds", "Lift", "rx", "sx", "ry", "sy", "nx", "ny", "Fscale", "vmapM", "vmapP", "x", "y")}
    tabs["Filter"] = ctx.filter
    bcmap = ctx.BCmap
    tabs["mapW"] = np.array(bcmap.get(3, []), dtype=np.int32)
    x, y = tabs["x"], tabs["y"]
    H = 10.0 + 1.5 * x - 0.8 * y * y + 0.3 * np.sin(3 * x) * np.cos(2 * y)
    Hx, Hy = nodes.bedSlopes(H)                      # the driver's filtered gradient (src/sw2d/main.cpp:128-133)
    h = H + 0.3 * np.exp(-4 * (x - 0.2) ** 2 - 4 * (y + 0.1) ** 2)
    c0 = 1.25 * np.sqrt(g * h.max())
    hu, hv = h * (c0 - np.sqrt(g * h)), np.zeros_like(h)
    ref_ctx = types.SimpleNamespace(BCmap=bcmap, nx=tabs["nx"], ny=tabs["ny"], rx=tabs["rx"], sx=tabs["sx"],
                                    ry=tabs["ry"], sy=tabs["sy"], Dr=tabs["Dr"], Ds=tabs["Ds"],
                                    numFacePoints=ctx.numFacePoints, numElements=ctx.numElements,
                                    numFaces=ctx.numFaces, Lift=tabs["Lift"], Fscale=tabs["Fscale"])
    r = sw2dComputeRHS(h, hu, hv, np.zeros_like(h), -Hx, -Hy, g, H, f, CD, ref_ctx, tabs["vmapM"], tabs["vmapP"])
    tag = "bed_drag" if CD else "bed"
    np.savez_compressed(os.path.join(HERE, f"sw2d_rhsB_{tag}_{name}.npz"), order=order, g=g, f=f, CD=CD, c0=c0, h=h, hu=hu,
                        hv=hv, H=H, Hx=Hx, Hy=Hy, rhs1=r[0], rhs2=r[1], rhs3=r[2], **tabs)
    print(f"sw2d_rhsB_{tag}_{name}.npz: K={ctx.numElements} Np={ctx.numLocalPoints} c0={c0:.6g} |rhs|max="
          f"{max(abs(a).max() for a in r[:3]):.6g}")


def bump_deformation(x, y, centre, radius, amp):
    """Smooth displacement with compact support (C^2 bump (1 - rho^2)^3 inside a disk, zero outside):
    elements with every node outside the disk stay exactly straight-sided, faces between a deformed and
    an undeformed element stay straight, and the two sides of every face keep the same curve."""
    rho2 = ((x - centre[0]) ** 2 + (y - centre[1]) ** 2) / radius ** 2
    b = np.where(rho2 < 1.0, (1.0 - rho2) ** 3, 0.0)
    return x + amp[0] * b * np.cos(1.3 * y), y + amp[1] * b * np.sin(1.7 * x + 0.4)


def curved_case(name, mesh, order, deform, flag="deformed", periodic_x=False, g=9.81 * 0.0025):
    """Curved / over-integrated RHS: output of the reference's swhelpers.rhs.sw2dComputeRHS_curved
    (swhelpers/rhs.py:6-176) with every context argument built by this repository: nodes moved by
    `deform` + setCoordinates, gauss_ctx = buildGaussFaceNodes(2(N+1)), cub_ctx =
    buildCubatureVolumeMesh(3(N+1)), J = xr*ys - xs*yr as the driver forms it (sw2d_curved.py:112-118).
    flag: "deformed" -> curvedEls = the elements whose nodes moved; "half" -> every second one of them
    (the rest take the standard mass matrix with their nodal, non-constant J, as the function allows).
    periodic_x: the Gauss map gmapP of the x = xmin / x = xmax boundary faces is rewired to the
    opposite side (what swhelpers.maps.makeMapsPeriodic does to the driver's maps)."""
    import blitzdg_amd.pyblitzdg as dg
    sys.path.insert(0, REF)
    if not hasattr(np, "float"):
        np.float = float
    from swhelpers.rhs import sw2dComputeRHS_curved

    nodes = dg.TriangleNodesProvisioner(order, mesh)
    nodes.buildFilter(0.9 * order, order)
    ctx = nodes.dgContext()
    x0, y0 = ctx.x, ctx.y
    x, y = deform(x0, y0)
    moved = np.where((np.abs(x - x0) + np.abs(y - y0)).max(axis=0) > 0)[0]
    curvedEls = moved if flag == "deformed" else moved[::2]
    nodes.setCoordinates(x, y)
    Dr, Ds = ctx.Dr, ctx.Ds
    J = np.dot(Dr, x) * np.dot(Ds, y) - np.dot(Ds, x) * np.dot(Dr, y)
    gauss = nodes.buildGaussFaceNodes(2 * (order + 1))
    cub = nodes.buildCubatureVolumeMesh(3 * (order + 1))
    assert cub.J.min() > 0 and gauss.J.min() > 0, "deformation inverted an element"
    gmapM, gmapP = gauss.mapM, gauss.mapP
    gbc = gauss.BCmap
    gmapW = np.array(gbc.get(3, []), dtype=np.int32)
    if periodic_x:
        gx, gy = gauss.x.flatten("F"), gauss.y.flatten("F")
        left = [i for i in gmapW if abs(gx[i] - gx.min()) < 1e-9]
        right = [i for i in gmapW if abs(gx[i] - gx.max()) < 1e-9]
        for a, b in ((left, right), (right, left)):
            for i in a:
                j = min(b, key=lambda m: abs(gy[m] - gy[i]))
                assert abs(gy[j] - gy[i]) < 1e-9
                gmapP[i] = j
        gmapW = np.array([i for i in gmapW if i not in set(left) | set(right)], dtype=np.int32)
        gbc = dict(gbc)
        gbc[3] = [int(i) for i in gmapW]
    h, hu, hv = seeded_fields(x, y)
    h = h - 9.0                      # depth 1..2: the free-surface hump matters against the mean depth
    rng = np.random.default_rng(3)
    hN = h * (0.5 + 0.3 * np.sin(2 * x) * np.cos(3 * y)) + 0.02 * rng.standard_normal(x.shape)
    H = 1.0 - 0.05 * x + 0.02 * y * y
    zx, zy = 0.05 + 0 * x, -0.04 * y
    f, CD = 7.8825e-5 * 1e3, 2.5e-3 * (1.0 + 0.5 * np.cos(x))   # CD is a nodal array in the driver (sw2d_curved.py:176-192)
    ref_ctx = types.SimpleNamespace(numLocalPoints=ctx.numLocalPoints, numElements=ctx.numElements, V=ctx.V)
    ref_cub = types.SimpleNamespace(V=cub.V, Dr=cub.Dr, Ds=cub.Ds, W=cub.W, rx=cub.rx, ry=cub.ry, sx=cub.sx, sy=cub.sy,
                                    MMChol=cub.MMChol)
    ref_gauss = types.SimpleNamespace(nx=gauss.nx, ny=gauss.ny, BCmap=gbc, Interp=gauss.Interp, W=gauss.W)
    r = sw2dComputeRHS_curved(h, hu, hv, hN, zx, zy, g, H, f, CD, ref_ctx, ref_cub, ref_gauss, curvedEls, J,
                              gmapM, gmapP)
    np.savez_compressed(
        os.path.join(HERE, f"sw2d_rhs_curved_{name}.npz"), order=order, g=g, f=f, CD=CD, h=h, hu=hu, hv=hv, hN=hN,
        H=H, zx=zx, zy=zy, x=x, y=y, x0=x0, y0=y0, J=J, V=ctx.V, Filter=ctx.filter, curvedEls=curvedEls.astype(np.int32),
        NGauss=gauss.NGauss, NCubature=cub.NCubature, cubV=cub.V, cubDr=cub.Dr, cubDs=cub.Ds, cubW=cub.W,
        cubrx=cub.rx, cubry=cub.ry, cubsx=cub.sx, cubsy=cub.sy, MMChol=cub.MMChol, gInterp=gauss.Interp, gW=gauss.W,
        gnx=gauss.nx, gny=gauss.ny, gmapM=gmapM, gmapP=gmapP, gmapW=gmapW,
        rhs1=r[0], rhs2=r[1], rhs3=r[2], rhs4=r[3])
    print(f"sw2d_rhs_curved_{name}.npz: K={ctx.numElements} Np={ctx.numLocalPoints} Ncub={cub.NumCubaturePoints} "
          f"curved={len(curvedEls)}/{len(moved)} moved |rhs|max={max(abs(a).max() for a in r):.6g}")


def curved_big_case(name="box40x26_g981_N4", order=4, nx=40, ny=26, seed=11, g=9.81):
    """The curved RHS at production scale and gravity: 2080 elements, g = 9.81, depth 10..11 (pressure flux g h^2 / 2 = 490,
    whose volume and surface integrals cancel to a few units: the conditioning under which the straight-element compression of
    the nodal-trace kernels was seen to differ from the per-point tables by 1e-11). Output of the REFERENCE's
    swhelpers.rhs.sw2dComputeRHS_curved on contexts built by this repository's builders. Stored compactly: the mesh recipe, the
    deformed node coordinates, the fields and source tables, the four outputs -- the Gauss / cubature contexts are rebuilt by the
    test with the same builders (buildGaussFaceNodes(2(N+1)), buildCubatureVolumeMesh(3(N+1)))."""
    import blitzdg_amd.pyblitzdg as dg
    sys.path.insert(0, REF)
    if not hasattr(np, "float"):
        np.float = float
    from swhelpers.rhs import sw2dComputeRHS_curved
    mesh = dg.MeshManager()
    mesh.buildBoxMesh(nx, ny, shuffleSeed=seed)
    nodes = dg.TriangleNodesProvisioner(order, mesh)
    nodes.buildFilter(0.9 * order, order)
    ctx = nodes.dgContext()
    x0, y0 = ctx.x, ctx.y
    x, y = bump_deformation(x0, y0, (0.3, -1.0), 0.8, (0.02, 0.05))
    curvedEls = np.where((np.abs(x - x0) + np.abs(y - y0)).max(axis=0) > 0)[0]
    nodes.setCoordinates(x, y)
    J = np.dot(ctx.Dr, x) * np.dot(ctx.Ds, y) - np.dot(ctx.Ds, x) * np.dot(ctx.Dr, y)
    gauss = nodes.buildGaussFaceNodes(2 * (order + 1))
    cub = nodes.buildCubatureVolumeMesh(3 * (order + 1))
    assert cub.J.min() > 0 and gauss.J.min() > 0
    h, hu, hv = seeded_fields(x, y)                       # depth 10 .. 11
    rng = np.random.default_rng(3)
    hN = h * (0.5 + 0.3 * np.sin(2 * x) * np.cos(3 * y)) + 0.02 * rng.standard_normal(x.shape)
    H = 10.0 - 0.05 * x + 0.02 * y * y
    zx, zy = 0.05 + 0 * x, -0.04 * y
    f, CD = 7.8825e-5 * 1e3, 2.5e-3 * (1.0 + 0.5 * np.cos(x))
    gmapW = np.array(gauss.BCmap.get(3, []), dtype=np.int32)
    ref_ctx = types.SimpleNamespace(numLocalPoints=ctx.numLocalPoints, numElements=ctx.numElements, V=ctx.V)
    ref_cub = types.SimpleNamespace(V=cub.V, Dr=cub.Dr, Ds=cub.Ds, W=cub.W, rx=cub.rx, ry=cub.ry, sx=cub.sx, sy=cub.sy, MMChol=cub.MMChol)
    ref_gauss = types.SimpleNamespace(nx=gauss.nx, ny=gauss.ny, BCmap=gauss.BCmap, Interp=gauss.Interp, W=gauss.W)
    r = sw2dComputeRHS_curved(h, hu, hv, hN, zx, zy, g, H, f, CD, ref_ctx, ref_cub, ref_gauss, curvedEls, J, gauss.mapM, gauss.mapP)
    np.savez_compressed(os.path.join(HERE, f"sw2d_bigcurved_{name}.npz"), order=order, nx=nx, ny=ny, seed=seed, g=g, f=f, x=x, y=y,
                        curvedEls=curvedEls.astype(np.int32), h=h, hu=hu, hv=hv, hN=hN, zx=zx, zy=zy, CD=CD, num_wall=gmapW.size,
                        rhs1=r[0], rhs2=r[1], rhs3=r[2], rhs4=r[3])
    print(f"sw2d_bigcurved_{name}.npz: K={ctx.numElements} curved={curvedEls.size} |rhs|max={max(abs(a).max() for a in r):.6g} "
          f"file={os.path.getsize(os.path.join(HERE, f'sw2d_bigcurved_{name}.npz')) / 1e6:.2f} MB")


def script_functions(path, names):
    """The named top-level function definitions of a reference SCRIPT (one that cannot be imported
    because its module body needs pyblitzdg and runs a whole simulation), compiled on their own with
    NumPy in scope. Nothing of the script is written anywhere: only the functions' outputs are kept."""
    import ast
    tree = ast.parse(open(path).read(), filename=path)
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(picked) == len(names)
    if not hasattr(np, "float"):
        np.float = float
    scope = {"np": np}
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), scope)
    return scope


def curved_helpers_case(name="channel32x6_N4", order=4, nx=32, ny=6, seed=7):
    """The set-up helpers of the reference's curved driver on a channel of the driver's size ([0, 8000] x [0, 1000], open
    ends at x = 0 and x = 8000, a headland bump in the top wall), in the order sw2d_curved.py:43-145 calls them; every output
    is that of the REFERENCE function:
      correctBCTable, makeMapsPeriodic            imported from /root/reference/swhelpers/maps.py
      adjustStraightEdges, deformAndBlendElements the two function definitions of /root/reference/meshhelpers/curved.py compiled
                                                  on their own (the module imports the compiled pyblitzdg, which does not exist
                                                  here) with numpy, scipy's splev and -- for its one call into pyblitzdg,
                                                  VandermondeBuilder().buildVandermondeMatrix -- this repository's mirror of that
                                                  class in scope (1-D orthonormal Legendre Vandermonde matrix, pinned against the
                                                  reference's literals by tests/test_setup_golden.py)
    The context handed to the two curved helpers HOLDS its x / y arrays (a SimpleNamespace), so the blending they do in place is
    what they return (with the reference's own DGContext2D every access of ctx.x is a fresh copy and the blending is lost)."""
    import blitzdg_amd.pyblitzdg as dg
    from scipy.interpolate import splev, splrep
    sys.path.insert(0, REF)
    from swhelpers.maps import correctBCTable, makeMapsPeriodic
    import ast
    path = os.path.join(REF, "meshhelpers/curved.py")
    tree = ast.parse(open(path).read(), filename=path)
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("adjustStraightEdges", "deformAndBlendElements")]
    assert len(picked) == 2
    scope = {"np": np, "splev": splev, "dg": dg}
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), scope)

    # the wall curve, an analytic headland between x = 3250 and x = 4750; as in the reference's mesh the straight-sided
    # elements' wall VERTICES already lie on it (the columns of the box mesh are compressed under it)
    length = 1500.0
    curve = lambda t: (3250.0 + t, 1000.0 - 60.0 * np.exp(-((3250.0 + t - 4000.0) / 350.0) ** 2))   # noqa: E731
    box = dg.MeshManager()
    box.buildBoxMesh(nx, ny, 0.0, 8000.0, 0.0, 1000.0, shuffleSeed=seed)
    vb = box.vertices
    under = (vb[:, 0] >= 3250.0) & (vb[:, 0] <= 4750.0)
    vb[under, 1] *= curve(vb[under, 0] - 3250.0)[1] / 1000.0
    mesh = dg.MeshManager()
    mesh.buildMesh(box.elements, vb)
    assert np.array_equal(mesh.elements, box.elements)
    Verts0, EToV, bc0 = mesh.vertices, mesh.elements, mesh.bcType
    bcType = correctBCTable(bc0.copy(), EToV, Verts0, 2)
    mesh.setBCType(bcType)
    nodes = dg.TriangleNodesProvisioner(order, mesh)
    ctx = nodes.dgContext()
    x0, y0 = ctx.x, ctx.y

    # its parametrisation: 128-point spline, 4096 samples (sw2d_curved.py:82-99)
    ss = np.linspace(0.0, length, 4096)
    s128 = np.linspace(0.0, length, 128)
    splx, sply = splrep(s128, curve(s128)[0]), splrep(s128, curve(s128)[1])
    xS, yS = splev(ss, splx, ext=2), splev(ss, sply, ext=2)

    bcInds = np.where(bcType.flatten("F") > 0)
    bcFaces = np.transpose(np.unravel_index(bcInds, (ctx.numElements, ctx.numFaces), order="F"))
    holder = types.SimpleNamespace(x=x0.copy(), y=y0.copy(), r=ctx.r, s=ctx.s, Fmask=ctx.Fmask, numFaces=ctx.numFaces)
    Verts1, modified, curvedFaces = scope["adjustStraightEdges"](Verts0.copy(), EToV, bcFaces, xS, yS, holder)
    x1, y1, curvedEls = scope["deformAndBlendElements"](Verts1, EToV, curvedFaces, xS, yS, ss, splx, sply, holder, order)
    assert len(curvedFaces) >= 3 and np.abs(x1 - x0).max() + np.abs(y1 - y0).max() > 1.0

    # periodic ends: nodal maps and Gauss maps (sw2d_curved.py:120-145), on the deformed coordinates
    nodes.setCoordinates(x1, y1)
    gauss = nodes.buildGaussFaceNodes(2 * (order + 1))
    assert gauss.J.min() > 0
    bcmap = ctx.BCmap
    vmapM, vmapP = ctx.vmapM, ctx.vmapP
    vmapO = vmapM[np.array(bcmap[2])]
    xFlat, yFlat = x1.flatten("F"), y1.flatten("F")
    vM1, vP1 = makeMapsPeriodic(vmapM.copy(), vmapP.copy(), vmapO, xFlat, yFlat, xFlat[vmapO], yFlat[vmapO])
    gmapM, gmapP = gauss.mapM, gauss.mapP
    gxFlat, gyFlat = np.dot(gauss.Interp, x1).flatten("F"), np.dot(gauss.Interp, y1).flatten("F")
    gmapO = np.array(gauss.BCmap[2])
    gM1, gP1 = makeMapsPeriodic(gmapM.copy(), gmapP.copy(), gmapO, gxFlat, gyFlat, gxFlat[gmapO], gyFlat[gmapO])
    print('rewired', (vP1 != vmapP).sum(), 'of', vmapO.size, 'nodal;', (gP1 != gmapP).sum(), 'of', gmapO.size, 'gauss')

    np.savez_compressed(
        os.path.join(HERE, f"curved_helpers_{name}.npz"), order=order, nx=nx, ny=ny, seed=seed,
        Verts0=Verts0, EToV=EToV, bcType0=bc0, bcType=bcType, bcFaces=bcFaces, x0=x0, y0=y0, r=ctx.r, s=ctx.s, Fmask=ctx.Fmask,
        ss=ss, xSpline=xS, ySpline=yS, splx_t=splx[0], splx_c=splx[1], sply_t=sply[0], sply_c=sply[1], spl_k=splx[2],
        Verts1=Verts1, modifiedVerts=modified, curvedFaces=np.array(curvedFaces, dtype=np.int64), x1=x1, y1=y1,
        curvedEls=np.array(curvedEls, dtype=np.int64),
        vmapM=vmapM, vmapP=vmapP, vmapO=vmapO, vmapP_periodic=vP1, vmapM_periodic=vM1,
        gmapM=gmapM, gmapP=gmapP, gmapO=gmapO, gxFlat=gxFlat, gyFlat=gyFlat, gmapP_periodic=gP1, gmapM_periodic=gM1)
    print(f"curved_helpers_{name}.npz: K={ctx.numElements} outflow faces tagged={(bcType == 2).sum()} curved faces={len(curvedFaces)} "
          f"max displacement={max(np.abs(x1 - x0).max(), np.abs(y1 - y0).max()):.4g} periodic nodes={vmapO.size} gauss={gmapO.size}")


def rhsC_case(name, mesh, order, g=9.81 * 0.0025, f=7.88e-5):
    """Variant C: sw2dComputeRHS(h, hu, hv, hN, g, H, f, ctx) of the reference's sw2d.py:37-146 (reduced
    gravity and f-plane Coriolis as in its driver, :150-155), output of the reference function itself."""
    import blitzdg_amd.pyblitzdg as dg
    scope = script_functions(os.path.join(REF, "sw2d.py"), ("sw2dComputeFluxes", "sw2dComputeRHS"))
    nodes = dg.TriangleNodesProvisioner(order, mesh)
    nodes.buildFilter(0.9 * order, order)
    ctx = nodes.dgContext()
    tabs = {k: getattr(ctx, k) for k in
            ("Dr", "Ds", "Lift", "rx", "sx", "ry", "sy", "nx", "ny", "Fscale", "vmapM", "vmapP", "x", "y")}
    tabs["Filter"] = ctx.filter
    bcmap = ctx.BCmap
    tabs["mapW"] = np.array(bcmap.get(3, []), dtype=np.int32)
    x, y = tabs["x"], tabs["y"]
    h, hu, hv = seeded_fields(x, y)
    rng = np.random.default_rng(2)
    hN = h * np.exp(-((y - 0.3) / 0.4) ** 2) + 0.05 * rng.standard_normal(x.shape)
    H = 10.0 + 0 * x
    ref_ctx = types.SimpleNamespace(BCmap=bcmap, nx=tabs["nx"], ny=tabs["ny"], rx=tabs["rx"], sx=tabs["sx"],
                                    ry=tabs["ry"], sy=tabs["sy"], Dr=tabs["Dr"], Ds=tabs["Ds"],
                                    numFacePoints=ctx.numFacePoints, numElements=ctx.numElements,
                                    numFaces=ctx.numFaces, Lift=tabs["Lift"], Fscale=tabs["Fscale"],
                                    vmapM=tabs["vmapM"], vmapP=tabs["vmapP"])
    scope["K"] = ctx.numElements  # the function reads the script-global K (sw2d.py:114)
    r = scope["sw2dComputeRHS"](h, hu, hv, hN, g, H, f, ref_ctx)
    np.savez_compressed(os.path.join(HERE, f"sw2d_rhsC_{name}.npz"), order=order, g=g, f=f, h=h, hu=hu, hv=hv, hN=hN,
                        H=H, rhs1=r[0], rhs2=r[1], rhs3=r[2], rhs4=r[3], **tabs)
    print(f"sw2d_rhsC_{name}.npz: K={ctx.numElements} Np={ctx.numLocalPoints} |rhs|max="
          f"{max(abs(a).max() for a in r):.6g}")


def advec1d_case(order=4, K=100, xmin=-1.0, xmax=4.0, c=0.1):
    """advec1dComputeRHS(u, c, nodes1d) of the reference SCRIPT advec1d.py:12-39 on BASELINE config 1
    (N=4, K=100, [-1, 4]), fed with this repo's Nodes1DProvisioner tables; a Gaussian and a seeded field."""
    import blitzdg_amd.pyblitzdg as dg
    scope = script_functions(os.path.join(REF, "advec1d.py"), ("advec1dComputeRHS",))
    nodes = dg.Nodes1DProvisioner(order, K, xmin, xmax)
    nodes.buildNodes()
    nodes.computeJacobian()
    tabs = {k: getattr(nodes, k) for k in ("Dr", "Lift", "rx", "Fscale", "nx", "vmapM", "vmapP", "xGrid")}
    ref_nodes = types.SimpleNamespace(mapI=nodes.mapI, mapO=nodes.mapO, **tabs)
    x = tabs["xGrid"]
    rng = np.random.default_rng(5)
    u1 = np.exp(-10 * x * x)
    u2 = u1 + 0.1 * rng.standard_normal(x.shape)
    r1 = scope["advec1dComputeRHS"](u1, c, ref_nodes)
    r2 = scope["advec1dComputeRHS"](u2, c, ref_nodes)
    np.savez_compressed(os.path.join(HERE, "advec1d_rhs_N4_K100.npz"), order=order, K=K, xmin=xmin, xmax=xmax, c=c,
                        u1=u1, u2=u2, rhs1=r1, rhs2=r2, mapI=nodes.mapI, mapO=nodes.mapO, **tabs)
    print(f"advec1d_rhs_N4_K100.npz: |rhs|max={max(abs(r1).max(), abs(r2).max()):.6g}")


def main():
    import blitzdg_amd.pyblitzdg as dg
    shutil.copyfile(os.path.join(REF, "input/coarse_box.msh"), os.path.join(HERE, "coarse_box.msh"))
    known_answers()
    coarse = dg.MeshManager()
    coarse.readMesh(os.path.join(HERE, "coarse_box.msh"))
    for order in (1, 2, 3, 4, 5, 6):
        rhs_case(f"coarse_box_N{order}", coarse, order)
    box = dg.MeshManager()
    box.buildBoxMesh(2, 2)
    rhs_case("box2x2_N8", box, 8)  # high-order fixture for the oracle (and the later N=8 device path)
    shuffled = dg.MeshManager()
    shuffled.buildBoxMesh(6, 5, shuffleSeed=12345)
    rhs_case("box6x5_shuffled_N4", shuffled, 4)
    for order in (2, 4, 6):
        rhs4_case(f"coarse_box_N{order}", coarse, order)
    rhs4_case("box6x5_shuffled_N3", shuffled, 3)
    rhs4_case("box6x5_shuffled_N5", shuffled, 5)  # N = 5, 7: the state-once kernel with sources and tracer (round 2)
    rhs4_case("box6x5_shuffled_N7", shuffled, 7)
    rhs4_case("box2x2_N8", box, 8)
    rhsC_case("coarse_box_N3", coarse, 3)
    rhsC_case("box6x5_shuffled_N6", shuffled, 6)
    advec1d_case()
    curved_cases()
    degenerate_b_cases()
    nodal_cases()


def nodal_cases():
    import blitzdg_amd.pyblitzdg as dg
    for order, (nx, ny) in ((2, (7, 6)), (4, (6, 5)), (6, (5, 4)), (8, (3, 2))):
        mesh = dg.MeshManager()
        mesh.buildBoxMesh(nx, ny, shuffleSeed=77)
        rhs4_case(f"box{nx}x{ny}_N{order}", mesh, order, deformed=True)


def degenerate_b_cases():
    import blitzdg_amd.pyblitzdg as dg
    coarse = dg.MeshManager()
    coarse.readMesh(os.path.join(HERE, "coarse_box.msh"))
    shuffled = dg.MeshManager()
    shuffled.buildBoxMesh(6, 5, shuffleSeed=12345)
    rhsB_degenerate_case("coarse_box_N3", coarse, 3)
    rhsB_degenerate_case("box6x5_shuffled_N6", shuffled, 6)
    box = dg.MeshManager()
    box.buildBoxMesh(3, 2)
    for CD in (0.0, 2.5e-2):
        rhsB_bed_case("coarse_box_N3", coarse, 3, CD)
        rhsB_bed_case("box6x5_shuffled_N6", shuffled, 6, CD)
        rhsB_bed_case("box3x2_N8", box, 8, CD)


def curved_cases():
    import blitzdg_amd.pyblitzdg as dg
    coarse = dg.MeshManager()
    coarse.readMesh(os.path.join(HERE, "coarse_box.msh"))
    shuffled = dg.MeshManager()
    shuffled.buildBoxMesh(6, 5, shuffleSeed=12345)
    box = dg.MeshManager()
    box.buildBoxMesh(3, 2)
    wall_bump = lambda x, y: bump_deformation(x, y, (0.55, -1.0), 0.9, (0.03, 0.07))      # noqa: E731
    inner_bump = lambda x, y: bump_deformation(x, y, (-0.2, 0.1), 0.8, (0.05, -0.04))     # noqa: E731
    curved_case("coarse_box_N3", coarse, 3, wall_bump)
    curved_case("coarse_box_N4", coarse, 4, wall_bump)
    curved_case("box6x5_shuffled_N6", shuffled, 6, inner_bump, flag="half")
    curved_case("box6x5_periodic_N2", shuffled, 2, lambda x, y: bump_deformation(x, y, (0.0, 1.0), 0.7, (0.0, 0.06)),
                periodic_x=True)
    curved_case("box3x2_N8", box, 8, wall_bump)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "curved":
        curved_cases()
    elif len(sys.argv) > 1 and sys.argv[1] == "curved_big":
        curved_big_case()
    elif len(sys.argv) > 1 and sys.argv[1] == "curved_helpers":
        curved_helpers_case()
    elif len(sys.argv) > 1 and sys.argv[1] == "variant_b":
        degenerate_b_cases()
    elif len(sys.argv) > 1 and sys.argv[1] == "nodal":
        nodal_cases()
    elif len(sys.argv) > 1 and sys.argv[1] == "degenerate_b":
        degenerate_b_cases()
    else:
        main()
