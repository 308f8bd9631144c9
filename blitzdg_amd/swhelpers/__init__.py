"""Device-backed counterpart of the reference's ``swhelpers`` package: ``swhelpers.rhs`` (the RHS functions of the straight-sided
and the curved sw2d path, on the GPU) and ``swhelpers.maps`` (the curved driver's host-side map helpers)."""
from .maps import correctBCTable, makeMapsPeriodic  # noqa: F401
from .rhs import sw2dComputeRHS  # noqa: F401
