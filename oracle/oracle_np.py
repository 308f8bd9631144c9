"""NumPy restatement of the reference's four-field sw2d RHS with source terms ("variant D") --
TEST INFRASTRUCTURE ONLY (same rules as oracle_sw2d.c).

Follows swhelpers/rhs.py:178-311 and swhelpers/flux.py:1-22 step by step, including the
re-formation huM = hM*(hu/h)[vmapM] (rhs.py:229-233) and the sign pattern of the drag term in RHS3
(rhs.py:307: ``RHS3 -= f*hu - CD|u| v``). Pinned by tests/golden/sw2d_rhs4_*.npz, which hold the
output of the reference function itself (tests/golden/make_golden.py).
"""
import numpy as np


def fluxes(h, hu, hv, hN, g):
    """flux.py:1-22"""
    u, v = hu / h, hv / h
    return (hu, hu * u + 0.5 * g * h * h, hv * u, hN * u), (hv, hu * v, hv * v + 0.5 * g * h * h, hN * v)


def sw2d_rhs4(h, hu, hv, hN, zx, zy, g, f, CD, t):
    """t: mapping with Dr Ds Lift rx sx ry sy nx ny Fscale vmapM vmapP mapW (reference numbering)."""
    Nfp = t["nx"].shape[0] // 3
    K = t["rx"].shape[1]
    vM, vP, mapW = np.asarray(t["vmapM"]), np.asarray(t["vmapP"]), np.asarray(t["mapW"], dtype=np.int64)
    col = lambda a: a.flatten("F")  # noqa: E731  column-wise numbering n + Np*k
    hC, huC, hvC, hNC, nxC, nyC = col(h), col(hu), col(hv), col(hN), col(t["nx"]), col(t["ny"])
    hM, hP = hC[vM], hC[vP]
    uM, uP = huC[vM] / hC[vM], huC[vP] / hC[vP]
    vMv, vPv = hvC[vM] / hC[vM], hvC[vP] / hC[vP]
    hNM, hNP = hNC[vM], hNC[vP]
    huM, hvM, huP, hvP = hM * uM, hM * vMv, hP * uP, hP * vPv
    nxW, nyW = nxC[mapW], nyC[mapW]
    un = huM[mapW] * nxW + hvM[mapW] * nyW
    huP[mapW] = huM[mapW] - 2 * nxW * un
    hvP[mapW] = hvM[mapW] - 2 * nyW * un
    dq = (hM - hP, huM - huP, hvM - hvP, hNM - hNP)
    FM, GM = fluxes(hM, huM, hvM, hNM, g)
    FP, GP = fluxes(hP, huP, hvP, hNP, g)
    F, G = fluxes(h, hu, hv, hN, g)
    uM, vMv, uP, vPv = huM / hM, hvM / hM, huP / hP, hvP / hP
    spd = np.maximum(np.sqrt(uM * uM + vMv * vMv) + np.sqrt(g * hM), np.sqrt(uP * uP + vPv * vPv) + np.sqrt(g * hP))
    lam = np.repeat(spd.reshape(3 * K, Nfp).max(axis=1), Nfp)  # per-face maximum, broadcast to its nodes
    out = []
    for c in range(4):
        dflux = 0.5 * ((FM[c] - FP[c]) * nxC + (GM[c] - GP[c]) * nyC - lam * dq[c])
        surf = t["Fscale"] * dflux.reshape((3 * Nfp, K), order="F")
        r = -(t["rx"] * (t["Dr"] @ F[c]) + t["sx"] * (t["Ds"] @ F[c]))
        r += -(t["ry"] * (t["Dr"] @ G[c]) + t["sy"] * (t["Ds"] @ G[c]))
        out.append(r + t["Lift"] @ surf)
    u, v = hu / h, hv / h
    cdn = CD * np.hypot(u, v)
    out[1] += f * hv - cdn * u
    out[2] -= f * hu - cdn * v
    out[1] -= g * h * zx
    out[2] -= g * h * zy
    return tuple(out)
