"""Pins the CPU oracle (oracle/oracle_sw2d.c) -- the checker of the HIP path.

 * against the RHS computed by the REFERENCE's own NumPy implementation
   (swhelpers/rhs.py:178-311), committed as tests/golden/sw2d_rhs_*.npz by
   tests/golden/make_golden.py;
 * through analytic properties for the pieces that have no importable reference
   (steppers, dt, advec1d): lake at rest, mass conservation, LSERK4 order, exactness.
Tolerance: relative max-norm 1e-12 against the reference fixtures (measured ~3e-14; the two
differ only in rounding: (hu*hu)/h vs hu*(hu/h), FMA-free summation in both).
"""
import numpy as np
import pytest

import blitzdg_amd.pyblitzdg as dg
from conftest import load_case, oracle_from, relmax, seeded_fields, tables_from_nodes
from oracle import advec1d_rhs, advec1d_steps, lserk4_coefficients

CASES = ["coarse_box_N1", "coarse_box_N2", "coarse_box_N3", "coarse_box_N4", "coarse_box_N5", "coarse_box_N6",
         "box2x2_N8", "box6x5_shuffled_N4"]


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_reference_rhs(case):
    d = load_case(case)
    o = oracle_from(d, g=float(d["g"]))
    r = o.rhs(d["h"], d["hu"], d["hv"])
    scale = max(np.abs(d[f"rhs{i}"]).max() for i in (1, 2, 3))
    for i in range(3):
        assert np.abs(r[i] - d[f"rhs{i + 1}"]).max() / scale < 1e-12


def test_oracle_threads_do_not_change_results():
    d = load_case("coarse_box_N4")
    a = oracle_from(d, threads=1).rhs(d["h"], d["hu"], d["hv"])
    b = oracle_from(d, threads=4).rhs(d["h"], d["hu"], d["hv"])
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_lake_at_rest_rhs_vanishes():
    d = load_case("coarse_box_N3")
    o = oracle_from(d)
    h = np.full_like(d["h"], 10.0)
    z = np.zeros_like(h)
    r = o.rhs(h, z, z)
    # momentum equations carry g h^2/2 ~ 490: round-off of Dr*const, relative to that
    assert max(np.abs(x).max() for x in r) < 1e-10


def test_filtered_rhs_is_filter_times_rhs():
    d = load_case("coarse_box_N4")
    o = oracle_from(d)
    r = o.rhs(d["h"], d["hu"], d["hv"])
    rf = o.rhs(d["h"], d["hu"], d["hv"], filter=True)
    for a, b in zip(r, rf):
        assert relmax(d["Filter"] @ a, b) < 1e-14




def test_mass_is_conserved_by_both_steppers(coarse_mesh):
    nodes = dg.TriangleNodesProvisioner(3, coarse_mesh)
    nodes.buildFilter(0.9 * 3, 3)
    t = tables_from_nodes(nodes)
    ctx = nodes.dgContext()
    w = np.linalg.inv(ctx.V @ ctx.V.T) @ np.ones(10)  # exact for degree <= N integrands
    J = ctx.J
    o = oracle_from(t)
    h, hu, hv = seeded_fields(t["x"], t["y"])
    mass0 = (w[:, None] * J * h).sum()
    dt = 0.5 * o.dt(h, hu, hv, 0.65, 3)
    h1, _, _ = o.step_lserk4(h, hu, hv, dt, 3)
    h2, _, _ = o.step_rk2(h, hu, hv, dt, 3, filter=True)
    assert abs((w[:, None] * J * h1).sum() - mass0) / mass0 < 1e-13
    assert abs((w[:, None] * J * h2).sum() - mass0) / mass0 < 1e-13
    assert np.abs(h1 - h).max() > 1e-4  # the state did move


def test_lserk4_coefficients_satisfy_order_conditions():
    """Convert the 2N-storage (a, b) pairs to Butcher form and check orders 1-4
    (include/LSERK4.hpp:15-29 are Carpenter & Kennedy's 5-stage scheme)."""
    a, b = lserk4_coefficients()
    s = 5
    # res_i = a_i res_{i-1} + dt k_i; u += b_i res_i  =>  weight of k_j in stage increment i
    W = np.zeros((s, s))
    for i in range(s):
        for j in range(i + 1):
            W[i, j] = b[i] * np.prod(a[j + 1:i + 1])
    Bw = W.sum(axis=0)                    # final weights
    A = np.zeros((s, s))
    for i in range(s):
        A[i, :i] = W[:i, :i].sum(axis=0)  # u at stage i = u0 + dt sum_j A_ij k_j
    c = A.sum(axis=1)
    assert abs(Bw.sum() - 1) < 1e-14
    assert abs(Bw @ c - 1 / 2) < 1e-14
    assert abs(Bw @ c ** 2 - 1 / 3) < 1e-14 and abs(Bw @ (A @ c) - 1 / 6) < 1e-14
    assert abs(Bw @ c ** 3 - 1 / 4) < 1e-13 and abs((Bw * c) @ (A @ c) - 1 / 8) < 1e-13
    assert abs(Bw @ (A @ c ** 2) - 1 / 12) < 1e-13 and abs(Bw @ (A @ (A @ c)) - 1 / 24) < 1e-13


def test_lserk4_temporal_order_on_sw2d(coarse_mesh):
    nodes = dg.TriangleNodesProvisioner(2, coarse_mesh)
    t = tables_from_nodes(nodes)
    o = oracle_from(t)
    h, hu, hv = seeded_fields(t["x"], t["y"])
    T = 0.02
    ref = o.step_lserk4(h, hu, hv, T / 64, 64)
    errs = []
    for n in (2, 4, 8):
        cur = o.step_lserk4(h, hu, hv, T / n, n)
        errs.append(max(np.abs(a - b).max() for a, b in zip(cur, ref)))
    rates = [np.log2(errs[i] / errs[i + 1]) for i in range(2)]
    assert min(rates) > 3.5, (errs, rates)


def test_dt_formula(coarse_mesh):
    nodes = dg.TriangleNodesProvisioner(3, coarse_mesh)
    t = tables_from_nodes(nodes)
    o = oracle_from(t)
    h, hu, hv = seeded_fields(t["x"], t["y"])
    fm, em = o.fsc_eta_max(h, hu, hv, H=np.full_like(h, 10.0))
    spd = np.sqrt((hu / h) ** 2 + (hv / h) ** 2) + np.sqrt(9.81 * h)
    expect = (np.abs(t["Fscale"].flatten("F")) * spd.flatten("F")[t["vmapM"]]).max()
    assert fm == expect
    assert em == np.abs(h - 10.0).max()
    hbad = h.copy()
    hbad[3, 7] = np.nan
    fm, em = o.fsc_eta_max(hbad, hu, hv)
    assert np.isnan(fm) and np.isnan(em)


def test_advec1d_oracle_matches_host_plumbing_and_converges():
    errs = []
    for K in (10, 20):
        n = dg.Nodes1DProvisioner(4, K, -1.0, 4.0)
        n.buildNodes()
        n.computeJacobian()
        x = n.xGrid
        u0 = np.exp(-10 * (x - 1.5) ** 2)  # centred: no inflow clipping
        c = 0.1
        dt = 0.8 * (x[1, 0] - x[0, 0]) / c
        args = (n.Dr, n.Lift, n.rx, n.Fscale, n.nx, n.vmapM, n.vmapP, n.mapI, n.mapO, c)
        rhs = advec1d_rhs(*args, u0)
        # interior: -c du/dx of the interpolant plus the upwind jump (tiny for smooth data)
        assert np.abs(rhs + c * (-20 * (x - 1.5) * u0)).max() < (0.3 if K == 10 else 0.03)
        nsteps = int(round(2.0 / dt))
        u = advec1d_steps(*args, dt, nsteps, u0)
        errs.append(np.abs(u - np.exp(-10 * (x - 1.5 - c * dt * nsteps) ** 2)).max())
    assert errs[0] / errs[1] > 8, errs


RHS4_CASES = ["coarse_box_N2", "coarse_box_N4", "coarse_box_N6", "box6x5_shuffled_N3", "box6x5_shuffled_N5",
              "box6x5_shuffled_N7", "box2x2_N8"]


@pytest.mark.parametrize("case", RHS4_CASES)
def test_variant_d_numpy_oracle_reproduces_the_reference_output(case):
    """oracle/oracle_np.py (tracer + Coriolis + drag + bed slope) against the output of the
    reference's swhelpers.rhs.sw2dComputeRHS itself (tests/golden/sw2d_rhs4_*.npz). The
    restatement performs the same NumPy operations in the same order: exact equality."""
    import os

    from conftest import GOLDEN
    from oracle.oracle_np import sw2d_rhs4
    d = np.load(os.path.join(GOLDEN, f"sw2d_rhs4_{case}.npz"))
    r = sw2d_rhs4(d["h"], d["hu"], d["hv"], d["hN"], d["zx"], d["zy"], float(d["g"]), d["f"], float(d["CD"]), d)
    for i in range(4):
        assert np.array_equal(r[i], d[f"rhs{i + 1}"])


def test_variant_d_reduces_to_variant_a():
    """With hN = 0, f = CD = 0, zx = zy = 0 the four-field RHS equals the C oracle's three-field
    RHS to round-off."""
    from oracle.oracle_np import sw2d_rhs4
    d = load_case("coarse_box_N4")
    z = np.zeros_like(d["h"])
    r4 = sw2d_rhs4(d["h"], d["hu"], d["hv"], z, z, z, float(d["g"]), 0.0, 0.0, d)
    r3 = oracle_from(d).rhs(d["h"], d["hu"], d["hv"])
    scale = max(np.abs(x).max() for x in r3)
    assert max(np.abs(a - b).max() for a, b in zip(r4[:3], r3)) / scale < 1e-13
    assert np.all(r4[3] == 0)


# ---- variant B (src/sw2d/main.cpp): restatement in oracle/oracle_np.py

@pytest.mark.parametrize("case", ["coarse_box_N3", "coarse_box_N6", "box6x5_shuffled_N4"])
def test_variant_b_oracle_with_local_speed_reproduces_the_reference_fixture(case):
    """Flat bottom, no open boundary, no sources, per-face instead of global Lax-Friedrichs speed:
    variant B is then variant A, and must reproduce the reference's own RHS output (the parts of
    main.cpp:279-484 shared with the Python RHS are pinned this way; the rest is unpinned)."""
    from oracle.oracle_np import sw2d_rhs_b
    d = load_case(case)
    z = np.zeros_like(d["h"])
    r = sw2d_rhs_b(d["h"], d["hu"], d["hv"], 10.0 + z, z, z, float(d["g"]), 0.0, 0.0, 0.0, d, global_lf=False)
    scale = max(np.abs(d[f"rhs{i}"]).max() for i in (1, 2, 3))
    for i in range(3):
        assert np.abs(r[i] - d[f"rhs{i + 1}"]).max() / scale < 1e-14


@pytest.mark.parametrize("case", ["coarse_box_N3", "box6x5_shuffled_N6"])
def test_variant_b_oracle_with_global_speed_reproduces_the_reference_on_a_uniform_speed_state(case):
    """Where variant B degenerates to what the reference's Python RHS computes -- flat bed, no open boundary, no
    drag, and a state of uniform depth and speed, for which the ONE global Lax-Friedrichs speed of the C++ driver
    (src/sw2d/main.cpp:414) equals every face's own maximum -- the restatement WITH its global speed reproduces the
    reference function's output (tests/golden/sw2d_rhsB_degenerate_*.npz; Coriolis on)."""
    import os
    from conftest import GOLDEN
    from oracle import oracle_np as onp
    d = np.load(os.path.join(GOLDEN, f"sw2d_rhsB_degenerate_{case}.npz"))
    z = np.zeros_like(d["h"])
    r = onp.sw2d_rhs_b(d["h"], d["hu"], d["hv"], d["H"], z, z, float(d["g"]), float(d["f"]), 0.0, 0.0, d, ())
    scale = max(np.abs(d[f"rhs{i}"]).max() for i in (1, 2, 3))
    for c in range(3):
        assert np.abs(r[c] - d[f"rhs{c + 1}"]).max() / scale < 1e-14


@pytest.mark.parametrize("tag", ["bed", "bed_drag"])
@pytest.mark.parametrize("case", ["coarse_box_N3", "box6x5_shuffled_N6", "box3x2_N8"])
def test_variant_b_oracle_reproduces_the_reference_over_a_continuous_bed(case, tag):
    """Variant B's star states (the identity over a continuous bed, but computed), bed-slope source and RHS2 drag
    against the reference function's output: v = 0 and |u| + sqrt(g h) = c0 everywhere, so that the one global speed
    equals every face's maximum and the RHS3 drag (whose sign differs between the two reference sources) vanishes
    (tests/golden/sw2d_rhsB_bed*_*.npz, make_golden.py::rhsB_bed_case)."""
    import os
    from conftest import GOLDEN
    from oracle import oracle_np as onp
    d = np.load(os.path.join(GOLDEN, f"sw2d_rhsB_{tag}_{case}.npz"))
    assert (float(d["CD"]) > 0) == (tag == "bed_drag") and np.abs(d["Hx"]).max() > 0.5 and np.ptp(d["H"]) > 1
    r = onp.sw2d_rhs_b(d["h"], d["hu"], d["hv"], d["H"], d["Hx"], d["Hy"], float(d["g"]), float(d["f"]), float(d["CD"]),
                       0.0, d, ())
    scale = max(np.abs(d[f"rhs{i}"]).max() for i in (1, 2, 3))
    for c in range(3):
        assert np.abs(r[c] - d[f"rhs{c + 1}"]).max() / scale < 1e-14
    if tag == "bed_drag":      # the drag is resolved by the comparison: without it the restatement is off by 1e-6
        r0 = onp.sw2d_rhs_b(d["h"], d["hu"], d["hv"], d["H"], d["Hx"], d["Hy"], float(d["g"]), float(d["f"]), 0.0, 0.0, d, ())
        assert np.abs(r0[1] - d["rhs2"]).max() / scale > 1e-7


@pytest.mark.parametrize("case", ["box7x6_N2", "box6x5_N4", "box5x4_N6", "box3x2_N8"])
def test_variant_d_oracle_reproduces_the_reference_on_per_node_geometry(case):
    """The reference function on genuinely non-affine tables (metric terms and normals of a smoothly deformed mesh per
    node; tests/golden/sw2d_rhs4n_*.npz, make_golden.py::rhs4_case(deformed=True)): the NumPy restatement, which
    performs the same operations in the same order, reproduces its output bit for bit."""
    import os
    from conftest import GOLDEN
    from oracle.oracle_np import sw2d_rhs4
    d = np.load(os.path.join(GOLDEN, f"sw2d_rhs4n_{case}.npz"))
    assert np.ptp(d["rx"], axis=0).max() > 1e-3                 # the metric does vary inside the elements
    r = sw2d_rhs4(d["h"], d["hu"], d["hv"], d["hN"], d["zx"], d["zy"], float(d["g"]), d["f"], float(d["CD"]), d)
    for c in range(4):
        assert np.array_equal(r[c], d[f"rhs{c + 1}"])


def test_variant_b_sources_agree_with_variant_d_fixture():
    """Bed slope and Coriolis of variant B (RHS2 += g h Hx + f hv, RHS3 += g h Hy - f hu) are variant
    D's with zx = -Hx, zy = -Hy (drag differs by D's sign quirk, so CD = 0 here); the tracer is ignored."""
    import os

    from conftest import GOLDEN
    from oracle.oracle_np import sw2d_rhs4, sw2d_rhs_b
    d = np.load(os.path.join(GOLDEN, "sw2d_rhs4_coarse_box_N4.npz"))
    g = float(d["g"])
    r4 = sw2d_rhs4(d["h"], d["hu"], d["hv"], d["hN"], d["zx"], d["zy"], g, d["f"], 0.0, d)
    rb = sw2d_rhs_b(d["h"], d["hu"], d["hv"], 10.0 + 0 * d["h"], -d["zx"], -d["zy"], g, d["f"], 0.0, 0.0, d,
                    global_lf=False)
    scale = max(np.abs(x).max() for x in r4[:3])
    assert max(np.abs(a - b).max() for a, b in zip(rb, r4[:3])) / scale < 1e-14


def test_variant_b_global_speed_star_states_and_tide(coarse_mesh):
    from conftest import variant_b_setup
    from oracle import oracle_np as onp
    nodes, t, e = variant_b_setup(3, coarse_mesh)
    Hx, Hy = onp.bed_slopes(e["H"], t)
    assert len(e["mapO"]) > 0 and set(e["mapO"]) <= set(t["mapW"])  # buildBCHash appended
    args = (e["h"], e["hu"], e["hv"], e["H"], Hx, Hy, 9.81, e["f"], e["CD"])
    r_glob = onp.sw2d_rhs_b(*args, e["time"], t, e["mapO"])
    r_loc = onp.sw2d_rhs_b(*args, e["time"], t, e["mapO"], global_lf=False)
    assert max(np.abs(a - b).max() for a, b in zip(r_glob, r_loc)) > 1e-3        # the global speed matters
    r_wall = onp.sw2d_rhs_b(*args, e["time"], t, ())
    assert max(np.abs(a - b).max() for a, b in zip(r_glob, r_wall)) > 1e-3       # and so does the open boundary
    r_t2 = onp.sw2d_rhs_b(*args, e["time"] + 3600.0, t, e["mapO"])
    assert np.abs(r_glob[0] - r_t2[0]).max() > 1e-4                               # tide phase enters
    # tide formula (main.cpp:352) at a quarter period is ~0 and ramps in with tanh
    T = onp.TIDE_PERIOD
    assert abs(onp.tide_elevation(0.25 * T)) < 1e-12
    assert abs(onp.tide_elevation(50 * T) - 3.0) < 1e-9
    # lake at rest over a plane bed: continuous H => star states are the traces themselves, and the
    # pressure gradient (H^2 is quadratic: differentiated exactly) balances the bed-slope source
    Hl = 12.0 + 1.5 * t["x"] - 0.8 * t["y"]
    rest = onp.sw2d_rhs_b(Hl, 0 * Hl, 0 * Hl, Hl, *onp.bed_slopes(Hl, t), 9.81, 0.0, 0.0, 0.0, t, ())
    assert max(np.abs(x).max() for x in rest) < 1e-10


def test_variant_b_still_water_over_a_bed_that_jumps_between_elements_stays_still(coarse_mesh):
    """A property of the reference's scheme that needs no vector (src/sw2d/main.cpp:357-368): with the hydrostatic star states
    h* = max(0, h - H + min(H-, H+)) both sides of a face see the same depth when the surface is level, so still water over a bed that
    is constant per element and jumps at every face has a vanishing right-hand side -- and a bump in one element does not."""
    from conftest import variant_b_setup
    from oracle import oracle_np as onp
    nodes, t, e = variant_b_setup(4, coarse_mesh)
    K = t["x"].shape[1]
    H = np.tile(9.0 + 3.0 * np.random.default_rng(4).random(K), (t["x"].shape[0], 1))
    zero = 0 * H
    moving = onp.sw2d_rhs_b(e["h"], e["hu"], e["hv"], e["H"], *onp.bed_slopes(e["H"], t), 9.81, e["f"], e["CD"], 0.0, t, ())
    scale = max(np.abs(x).max() for x in moving)
    rest = onp.sw2d_rhs_b(H + 0.25, zero, zero, H, zero, zero, 9.81, e["f"], e["CD"], 0.0, t, ())
    assert max(np.abs(x).max() for x in rest) < 1e-13 * scale
    h = H + 0.25
    h[:, K // 2] += 0.01
    bump = onp.sw2d_rhs_b(h, zero, zero, H, zero, zero, 9.81, e["f"], e["CD"], 0.0, t, ())
    assert max(np.abs(x).max() for x in bump) > 1e-6 * scale


def test_variant_b_host_helpers_match_the_restatement(coarse_mesh):
    from conftest import variant_b_setup
    from oracle import oracle_np as onp
    nodes, t, e = variant_b_setup(4, coarse_mesh)
    Hx, Hy = nodes.bedSlopes(e["H"])
    rx, ry = onp.bed_slopes(e["H"], t)
    assert relmax(Hx, rx) < 1e-13 and relmax(Hy, ry) < 1e-13
    sp = nodes.buildSpongeCoeff(e["mapO"], 10.0, 0.6)
    ref = onp.build_sponge_coeff(t, e["mapO"], 10.0, 0.6)
    assert np.array_equal(sp, ref) and sp.max() == 10.0 and (sp == 0).any()
    assert np.all(nodes.buildSpongeCoeff([], 10.0, 0.6) == 0)


@pytest.mark.parametrize("case", ["coarse_box_N3", "box6x5_shuffled_N6"])
def test_variant_c_numpy_oracle_reproduces_the_reference_script_function(case):
    """oracle_np.sw2d_rhs_c against the output of sw2dComputeRHS of the reference's sw2d.py script
    (tests/golden/sw2d_rhsC_*.npz): same NumPy operations in the same order, exact equality."""
    import os

    from conftest import GOLDEN
    from oracle.oracle_np import sw2d_rhs_c
    d = np.load(os.path.join(GOLDEN, f"sw2d_rhsC_{case}.npz"))
    r = sw2d_rhs_c(d["h"], d["hu"], d["hv"], d["hN"], float(d["g"]), float(d["f"]), d)
    for i in range(4):
        assert np.array_equal(r[i], d[f"rhs{i + 1}"])


def test_advec1d_rhs_matches_the_reference_script_function():
    """BASELINE config 1 (advec1d, N=4, K=100): the host advec1d::computeRHS (C++ and its Python twin
    pyblitzdg.advec1dComputeRHS) and the C oracle against the output of advec1dComputeRHS of the reference's
    advec1d.py script (tests/golden/advec1d_rhs_N4_K100.npz)."""
    import os

    import blitzdg_amd.pyblitzdg as dg
    from conftest import GOLDEN
    from oracle import advec1d_rhs
    d = np.load(os.path.join(GOLDEN, "advec1d_rhs_N4_K100.npz"))
    nodes = dg.Nodes1DProvisioner(int(d["order"]), int(d["K"]), float(d["xmin"]), float(d["xmax"]))
    nodes.buildNodes()
    nodes.computeJacobian()
    assert np.array_equal(nodes.vmapM, d["vmapM"]) and np.array_equal(nodes.vmapP, d["vmapP"])
    for u, ref in ((d["u1"], d["rhs1"]), (d["u2"], d["rhs2"])):
        scale = np.abs(ref).max()
        got = dg.advec1dComputeRHS(u, float(d["c"]), nodes)
        assert np.abs(got - ref).max() / scale < 1e-13
        orc = advec1d_rhs(d["Dr"], d["Lift"], d["rx"], d["Fscale"], d["nx"], d["vmapM"], d["vmapP"], int(d["mapI"]),
                          int(d["mapO"]), float(d["c"]), u)
        assert np.abs(orc - ref).max() / scale < 1e-13


def test_numpy_restatement_reproduces_the_reference_at_production_gravity():
    """The curved restatement on the 2080-element, g = 9.81 fixture (tests/golden/sw2d_bigcurved_*.npz: output of the reference's
    sw2dComputeRHS_curved; the contexts are rebuilt here from the stored coordinates with this repository's builders, as the GPU
    test does): to round-off of the rebuilt tables (bit for bit when the builders have not changed since the fixture was made)."""
    import glob
    import blitzdg_amd.pyblitzdg as dg
    from oracle import oracle_np
    from conftest import GOLDEN
    import os
    paths = sorted(glob.glob(os.path.join(GOLDEN, "sw2d_bigcurved_*.npz")))
    assert paths
    for path in paths:
        d = np.load(path)
        order = int(d["order"])
        mesh = dg.MeshManager()
        mesh.buildBoxMesh(int(d["nx"]), int(d["ny"]), shuffleSeed=int(d["seed"]))
        nodes = dg.TriangleNodesProvisioner(order, mesh)
        ctx = nodes.dgContext()
        x, y = d["x"], d["y"]
        nodes.setCoordinates(x, y)
        J = np.dot(ctx.Dr, x) * np.dot(ctx.Ds, y) - np.dot(ctx.Ds, x) * np.dot(ctx.Dr, y)
        gauss, cub = nodes.buildGaussFaceNodes(2 * (order + 1)), nodes.buildCubatureVolumeMesh(3 * (order + 1))
        t = dict(cubV=cub.V, cubDr=cub.Dr, cubDs=cub.Ds, cubW=cub.W, cubrx=cub.rx, cubry=cub.ry, cubsx=cub.sx, cubsy=cub.sy,
                 gInterp=gauss.Interp, gW=gauss.W, gnx=gauss.nx, gny=gauss.ny, gmapM=gauss.mapM, gmapP=gauss.mapP,
                 gmapW=np.array(gauss.BCmap[3], dtype=np.int32), V=ctx.V, J=J, MMChol=cub.MMChol, curvedEls=d["curvedEls"])
        got = oracle_np.sw2d_rhs_curved(d["h"], d["hu"], d["hv"], d["hN"], d["zx"], d["zy"], float(d["g"]), float(d["f"]), d["CD"], t)
        ref = [d[f"rhs{i}"] for i in (1, 2, 3, 4)]
        scale = max(np.abs(r).max() for r in ref)
        assert max(np.abs(a - b).max() for a, b in zip(got, ref)) <= 1e-13 * scale
