"""CPU oracle for the sw2d / advec1d hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package, and only as the checker; blitzdg_amd/ never does. See oracle_sw2d.c and
oracle_np.py for the reference file:line each function restates.
"""
from .oracle import Sw2dOracle, advec1d_rhs, advec1d_steps, lserk4_coefficients  # noqa: F401
