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
                            equilateral nodes (:349-369), EToV/EToE/EToF/BCType/verts
                            (MeshManagerTests.cpp:202-206), 1-D V/Dr/x/Lift/EToE/EToF/
                            vmapM/vmapP (Nodes1DProvisionerTests.cpp:52-247)
  sw2d_rhs_<case>.npz       inputs (tables built by THIS repo's host code + seeded
                            fields) and the RHS computed by the REFERENCE's
                            swhelpers.rhs.sw2dComputeRHS (swhelpers/rhs.py:178-311) with
                            hN = 0, f = CD = 0, zx = zy = 0  (variant D == variant A up
                            to round-off)
  sw2d_rhs4_<case>.npz      the same function with tracer, Coriolis array, drag and bed slope
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


def rhs4_case(name, mesh, order, g=9.81):
    """Variant D: tracer + Coriolis (array f) + drag + bed slope, all non-trivial; output of the
    reference's swhelpers.rhs.sw2dComputeRHS itself."""
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
    np.savez_compressed(os.path.join(HERE, f"sw2d_rhs4_{name}.npz"), order=order, g=g, h=h, hu=hu, hv=hv, hN=hN,
                        H=H, zx=zx, zy=zy, f=f, CD=CD, rhs1=r[0], rhs2=r[1], rhs3=r[2], rhs4=r[3], **tabs)
    print(f"sw2d_rhs4_{name}.npz: K={ctx.numElements} Np={ctx.numLocalPoints} |rhs|max="
          f"{max(abs(a).max() for a in r):.6g}")


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
    rhs4_case("box2x2_N8", box, 8)
    rhsC_case("coarse_box_N3", coarse, 3)
    rhsC_case("box6x5_shuffled_N6", shuffled, 6)
    advec1d_case()


if __name__ == "__main__":
    main()
