#!/usr/bin/env python3
"""Reads a phase-clock dump of the curved nodal-trace kernel (a -DBDG_PHASE_CLOCK build of sw2d_curved_order.hip loaded
through BDG_HIP_LIBRARY, BDG_PHASE_CLOCK_FILE=<file>) and prints the share of each phase in a wave's cycles:
    python3 profiles/phase_report.py <file>"""
import sys

import numpy as np

NAMES = ["requests of the first round trip", "volume term", "surface term", "sources", "mass, update, stores"]


def main():
    rows = np.loadtxt(sys.argv[1], dtype=np.float64)
    c = rows[:, 1:6]
    c = c[c.sum(axis=1) > 0]
    tot = c.sum(axis=1)
    print(f"waves {len(c)}  cycles per wave: mean {tot.mean():.0f}  min {tot.min():.0f}  max {tot.max():.0f}")
    ok = rows[:, 8] > 0
    if ok.any():  # the loop in shader cycles over the same span in 100 MHz ticks: the clock the chip held in the kernel
        ghz = rows[ok, 7] / rows[ok, 8] * 0.1
        print(f"in-kernel clock: median {np.median(ghz):.2f} GHz  (min {ghz.min():.2f}, max {ghz.max():.2f})")
    if rows.shape[1] >= 11:
        live = rows[:, 10] > 0
        t0, t1 = rows[live, 9], rows[live, 10]
        span = (t1.max() - t0.min()) * 10e-3  # 100 MHz ticks -> us
        print(f"first entry to last exit: {span:.1f} us;  entries spread over {(t0.max() - t0.min()) * 10e-3:.1f} us;  "
              f"exits spread over {(t1.max() - t1.min()) * 10e-3:.1f} us;  prologue mean {rows[live, 6].mean():.0f} cycles")
        print("  exit time after first entry, percentiles 10/50/90/100 (us):",
              " ".join(f"{v:.1f}" for v in np.percentile((t1 - t0.min()) * 10e-3, [10, 50, 90, 100])))
    for i, n in enumerate(NAMES):
        print(f"  {n:32s} {100 * c[:, i].sum() / tot.sum():5.1f} %   mean {c[:, i].mean():10.0f}")


if __name__ == "__main__":
    main()
