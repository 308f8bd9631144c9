"""burgers1d (SURVEY 8f.4's tail: the 1-D LSERK4 path on Nodes1DProvisioner beside advec1d; reference src/burgers1d/main.cpp).
Host code only. The reference holds no known answer for this solver and its C++ cannot be built here: parity unpinned. What is
checked: the host RHS against the NumPy restatement of the same file (two independent readings), the driver's own measure -- the
max-norm error against the travelling wave it is initialised with -- at the driver's constants and under refinement, and the
bin/burgers1d front end."""
import os
import subprocess
import sys

import numpy as np

import blitzdg_amd.pyblitzdg as dg
from conftest import ROOT
from oracle import oracle_np


def test_rhs_equals_the_numpy_restatement():
    N, K = 6, 40                                                   # the driver's numbers (main.cpp:42-44)
    n1 = dg.Nodes1DProvisioner(N, K, -5.0, 5.0)
    n1.buildNodes()
    n1.computeJacobian()
    x = n1.xGrid
    rng = np.random.default_rng(4)
    for t in (0.0, 0.037):
        u = oracle_np.burgers2(x, t, 1.0, 0.1, 0.5) + 0.05 * rng.standard_normal(x.shape)
        got = dg.burgers1dComputeRHS(u, t, 0.5, 1.0, 0.1, n1)
        ref = oracle_np.burgers1d_rhs(u, x, t, 0.5, 1.0, 0.1, n1.Dr, n1.rx, n1.Lift, n1.Fscale, n1.nx, n1.vmapM, n1.vmapP,
                                      n1.mapI, n1.mapO, 0, (N + 1) * K - 1)
        assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
        assert np.abs(ref).max() > 1e-2


def test_the_travelling_wave_is_followed_and_the_error_falls_under_refinement():
    err, steps = dg.burgers1dRun()                                 # N = 6, K = 40, T = 0.1: the reference driver's run
    assert steps > 0 and err < 1e-5
    coarse, _ = dg.burgers1dRun(N=3, K=20, finalTime=0.05)
    fine, _ = dg.burgers1dRun(N=3, K=40, finalTime=0.05)
    assert fine < coarse / 4 and fine < 1e-3                       # a smooth wave: well beyond first order


def test_bin_burgers1d_prints_the_error_like_the_reference():
    exe = os.path.join(ROOT, "bin", "burgers1d")
    out = subprocess.run([exe, "6", "40", "0.1"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    err = float(out.stdout.strip().split("Error:")[1])
    assert abs(err - dg.burgers1dRun()[0]) <= 1e-5 * err          # (printed with six significant digits)
