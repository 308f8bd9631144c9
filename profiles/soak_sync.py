"""Soak run of the in-kernel stage dependencies (round 4): thousands of exchanged LSERK4 stages of one rank's share of an 8-way split
(loop-back exchanges through real RCCL) with the dependencies polled inside the kernels, against the same run with event waits on the
queues -- same kernels, same arithmetic, so the owned state must come out the same BIT FOR BIT; a missing wait or a stale line read at
a hand-off shows as a difference somewhere along the way. Full-size shares (the sizes of the rehearsal), several calls per run so that
the counters carry over.   python3 profiles/soak_sync.py [stages]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from blitzdg_amd.halo import NativeDistributedSw2d  # noqa: E402


def fields(x, y):
    h = 10.0 + np.exp(-10 * x * x - 10 * y * y)
    return h, 0.1 * np.sin(3 * x + 1) * np.cos(2 * y), 0.1 * np.cos(2 * x) * np.sin(3 * y - 1)


def run(order, shape, rank, chunks):
    d = NativeDistributedSw2d.box(shape[0], shape[1], order, rank, 8, device=0, loopback=True)
    try:
        d.set_initial_state(fields)
        dt = 0.25 * d.compute_dt(0.65)
        for c in chunks:
            d.lserk4_stages(dt, c)
        d.barrier()
        return d.owned_state()[1:], d.halo_counts()
    finally:
        d.close()


def main():
    stages = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    chunks = (7, stages // 3, stages - 7 - stages // 3)
    out = []
    for order, shape in ((4, (1000, 500)), (3, (400, 200)), (5, (800, 400)), (6, (1000, 250)), (7, (500, 250)), (8, (500, 250))):
        for rank in (1, 4):
            os.environ["BDG_SW2D_EVENT_SYNC"] = "1"
            ref, counts = run(order, shape, rank, chunks)
            os.environ.pop("BDG_SW2D_EVENT_SYNC")
            got, _ = run(order, shape, rank, chunks)
            same = all(np.array_equal(a, b) for a, b in zip(got, ref))
            moved = float(np.abs(ref[1]).max())
            rec = {"order": order, "cells": list(shape), "rank": rank, "stages": stages, "bit_identical": bool(same),
                   "finite": bool(np.isfinite(ref[0]).all()), "max_abs_hu": moved, **counts}
            print(json.dumps(rec), flush=True)
            out.append(rec)
    ok = all(r["bit_identical"] and r["finite"] for r in out)
    print(json.dumps({"soak": "in-kernel dependencies against event waits", "runs": len(out), "all_bit_identical": ok}), flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
