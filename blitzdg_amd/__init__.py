"""blitzdg_amd -- MI355X-native implementation of blitzdg's sw2d DG right-hand-side path.

  blitzdg_amd.pyblitzdg   pyblitzdg-shaped setup objects (MeshManager, TriangleNodesProvisioner,
                          DGContext2D, Nodes1DProvisioner, LSERK4, BCType) over the C ABI
  blitzdg_amd.sw2d        computeRHS drop-in and the device-resident Sw2dSolver
  blitzdg_amd.halo        element partition + ghost-face halo plan for multi-GPU runs

The compute path is libblitzdg_hip.so (hand-written HIP for gfx950); importing this package
fails if the library has not been built -- there is no CPU fallback.
"""
from . import _capi  # noqa: F401  (loads the shared library or raises)
from . import pyblitzdg, sw2d  # noqa: F401

__all__ = ["pyblitzdg", "sw2d"]
