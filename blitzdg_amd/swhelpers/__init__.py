"""Device-backed counterpart of the reference's ``swhelpers`` package (only the RHS of the
straight-sided sw2d path: ``swhelpers.rhs.sw2dComputeRHS``)."""
from .rhs import sw2dComputeRHS  # noqa: F401
