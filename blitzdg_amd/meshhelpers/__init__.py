"""Mesh set-up helpers of the reference's curved driver (reference meshhelpers/curved.py)."""
from .curved import adjustStraightEdges, deformAndBlendElements  # noqa: F401
