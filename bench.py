#!/usr/bin/env python3
"""bench.py -- sw2d DG right-hand side + LSERK4 stage on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path: ONE fused RHS + LSERK4 stage over the whole mesh.
Workload (configs[2] of BASELINE.json): synthetic box [-1,1]^2, 1000 x 500 cells = 10^6
triangles, N=4 (Np=15), H=10 + Gaussian free surface, fp64, state resident in HBM.
For --gpus N > 1 the SAME mesh is partitioned into N parts (strong scaling); ghost elements
are exchanged every stage over RCCL (torch.distributed, one process per GPU) while the
interior elements compute.

Prints one JSON line (rank 0): metric = element-DOF updates/s (Np*K*stages / wall time), plus
  roofline     algorithmic HBM bytes (2400 B/element/stage at N=4) over the
               average stage-kernel duration measured with HIP events on the solver's stream
  cpu_baseline the CPU oracle (port of the reference algorithm) on a bounded sample, rank 0, N=1
  also         (default headline run only; outside its timed region) BASELINE config 5 -- 250 k triangles at N=8 -- and
               config 3 on the seed-12345 shuffled mesh, as given and renumbered: {"config5_n8": {...}, "config3_shuffled": {...}}
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ORDER = 4
NX, NY = 1000, 500
G = 9.81
CFL = 0.65
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def algorithmic_bytes_per_element(order):
    """RHS + LSERK4 stage, fp64 values / int32 indices: 128*Np + 96*Nfp."""
    np_, nfp = (order + 1) * (order + 2) // 2, order + 1
    return 128 * np_ + 96 * nfp


def actual_bytes_per_element(order, affine):
    """Compulsory HBM bytes of the kernel actually run (vmapM is implicit; affine geometry keeps
    4 + 9 doubles per element instead of 4*Np + 9*Nfp)."""
    np_, nfp = (order + 1) * (order + 2) // 2, order + 1
    geo = 13 * 8 if affine else (4 * np_ + 9 * nfp) * 8
    return 3 * np_ * 8 * 4 + geo + 3 * nfp * 4


_COMMENT_OR_STRING = None


def strip_comments(text):
    """C++ source without comments and with runs of white space collapsed (string literals kept as they are)."""
    import re
    global _COMMENT_OR_STRING
    if _COMMENT_OR_STRING is None:
        _COMMENT_OR_STRING = re.compile(r'//[^\n]*|/\*.*?\*/|"(?:\\.|[^"\\\n])*"|\'(?:\\.|[^\'\\\n])*\'', re.S)
    code = _COMMENT_OR_STRING.sub(lambda m: m.group(0) if m.group(0)[0] in "\"'" else " ", text)
    return " ".join(code.split())


def kernel_source_sha():
    """Hash of the device sources of the straight-element kernels (everything under csrc/hip but sw2d_curved_*) WITHOUT their
    comments and layout: a committed PMC summary is only quoted while it
    still describes the kernels that are being timed (profiles/summarize.py records the same hash), and a comment or
    re-indentation does not invalidate a collection (in round 2 a header comment cost a full PMC run)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "blitzdg_amd", "csrc", "hip", "*"))):
        if os.path.basename(f).startswith("sw2d_curved"):
            continue        # the curved solver's own translation units: not part of what this benchmark times
        h.update(os.path.basename(f).encode())
        h.update(strip_comments(open(f, errors="replace").read()).encode())
    return h.hexdigest()[:16]


def committed_traffic(order, elements):
    """HBM bytes per stage-kernel launch from the PMC passes of this same command
    (profiles/collect.sh -> profiles/<tag>_pmc_summary.json: FETCH_SIZE and WRITE_SIZE collected in
    separate rocprofv3 --pmc runs, reads = 2 x FETCH_SIZE on gfx950). None when no committed summary
    matches this workload AND the device sources as they are now (a stale figure is not quoted)."""
    import glob
    sha = kernel_source_sha()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), reverse=True):
        try:
            d = json.load(open(f))
            if d.get("kernel_source_sha") != sha or d.get("order") != order or d.get("elements") != elements:
                continue
            # only the default three-field stage kernels (profiles/ also holds collections of variant B, the tracer phase, A/B variants)
            kern = d.get("kernel") or ""
            if not ("sw2d_stage_affine_kernel<" in kern or "sw2d_stage_mfma3_kernel<" in kern or "sw2d_stage_mfma_kernel<" in kern):
                continue
            return d["hbm_traffic_per_launch"]["total_bytes"], os.path.basename(f)
        except (KeyError, ValueError, OSError):
            continue
    return None, None


def initial_state(x, y):
    h = 10.0 + np.exp(-10 * x * x - 10 * y * y)
    z = np.zeros_like(h)
    return h, z, z.copy()


def cpu_baseline(order):
    """Times the CPU oracle (oracle/oracle_sw2d.c: pass-by-pass port of the reference's
    computeRHS + LSERK4 update, gcc -O2, single thread like blitzdg) on a bounded sample of the
    same workload: a 400 x 125-cell box (10^5 triangles) at the same order, 1 warm-up + 40 timed
    stages (about 10 s of CPU work at N=4). Also reports the same port with OpenMP on all host cores."""
    import blitzdg_amd.pyblitzdg as dg
    from oracle import Sw2dOracle

    mesh = dg.MeshManager()
    mesh.buildBoxMesh(400, 125)
    nodes = dg.TriangleNodesProvisioner(order, mesh)
    ctx = nodes.dgContext()
    K, Np = ctx.numElements, ctx.numLocalPoints
    tabs = dict(Dr=ctx.Dr, Ds=ctx.Ds, Lift=ctx.Lift, rx=ctx.rx, sx=ctx.sx, ry=ctx.ry, sy=ctx.sy, nx=ctx.nx,
                ny=ctx.ny, Fscale=ctx.Fscale, vmapM=ctx.vmapM, vmapP=ctx.vmapP,
                mapW=np.array(ctx.BCmap[3], dtype=np.int32))
    h, hu, hv = initial_state(ctx.x, ctx.y)
    host_cores = os.cpu_count() or 1
    out = {}
    cores = min(host_cores, 32)  # a 1-GPU box shares its host; more threads than that only add OpenMP overhead
    nstages = max(4, int(40 * (15.0 / Np) ** 2))  # ~10 s single-threaded at every order (cost per node ~ Np)
    for label, threads, stages in (("single", 1, nstages), ("all", cores, nstages)):
        o = Sw2dOracle(g=G, threads=threads, **tabs)
        dt = 1e-4
        res = [np.zeros_like(h) for _ in range(3)]
        q = o.lserk4_stages(h, hu, hv, res, dt, 0, 1)  # warm-up (page faults, caches)
        t0 = time.perf_counter()
        o.lserk4_stages(q[0], q[1], q[2], q[3], dt, 1, stages)
        sec = time.perf_counter() - t0
        out[label] = Np * K * stages / sec
    return {"value": out["single"], "unit": "element-DOF updates/s", "cores": 1, "kind": "port",
            "sample": f"CPU oracle (C port of sw2d-simple computeRHS + LSERK4 stage, gcc -O2, 1 thread), "
                      f"400x125-cell box = {K} triangles, N={order}, {nstages} timed stages",
            "value_all_cores": out["all"], "cores_all": cores, "host_cores": host_cores,
            "gbps_algorithmic_single": out["single"] / Np * algorithmic_bytes_per_element(order) / 1e9}


def stage_kernel_name(order, elements, affine):
    """Name of the kernel a default LSERK4 stage launch of this size runs (blitzdg_amd/csrc/hip/sw2d_device.hip:
    unrolled vector kernel up to N=4 for large launches, matrix cores otherwise)."""
    if not affine:
        if os.environ.get("BDG_SW2D_NODAL_VECTOR") and order <= 6:
            return f"sw2d_stage_kernel<{order}, MODE_LSERK, false>"
        return f"sw2d_stage_mfma3_kernel<{order}, MODE_LSERK, false, NODAL>"
    forced = os.environ.get("BDG_SW2D_AFFINE_VARIANT")
    if forced is not None:
        return f"BDG_SW2D_AFFINE_VARIANT={forced} <{order}, MODE_LSERK>"
    small = {1: 4000, 2: 10000, 3: 160000, 4: 160000}
    if order >= 5:
        return f"sw2d_stage_mfma3_kernel<{order}, MODE_LSERK>"
    if elements < small[order]:
        return f"sw2d_stage_mfma_kernel<{order}, MODE_LSERK>"
    return f"sw2d_stage_affine_kernel<{order}, MODE_LSERK>"


# the `also` measurements: the mesh of each is built on the host first (the clocks fall back meanwhile), hence a ramp like the headline's
ALSO_RAMP, ALSO_STAGES = 150, 100


def also_measure(order, cells, shuffle_seed=0, keep_order=False, ramp=ALSO_RAMP, stages=ALSO_STAGES):
    """One more configuration of BASELINE.json, measured in the same process AFTER the headline's timed region (its time
    is not part of `value` / `ms_per_step`): build, a short untimed ramp, then `stages` fused LSERK4 stage launches timed
    with HIP events on the solver's stream. Returns (ms per launch, elements, Np, solver flags of interest)."""
    import blitzdg_amd.pyblitzdg as dg
    from blitzdg_amd import sw2d
    mesh = dg.MeshManager()
    mesh.buildBoxMesh(cells[0], cells[1], shuffleSeed=shuffle_seed)
    nodes = dg.TriangleNodesProvisioner(order, mesh)
    ctx = nodes.dgContext()
    solver = sw2d.Sw2dSolver(nodes=nodes, g=G, device=0, flags=sw2d.KEEP_ORDER if keep_order else 0)
    solver.setState(*initial_state(ctx.x, ctx.y))
    dt, _ = solver.computeDt(CFL)
    solver.timeLSERK4Stages(dt, ramp)
    ms = solver.timeLSERK4Stages(dt, stages)
    solver.computeDt(CFL)  # raises if the run blew up
    return ms, ctx.numElements, ctx.numLocalPoints, solver.isRenumbered, solver.usesAffineGeometry


def also_block():
    """BASELINE.md section 3 rows C5 and C3-shuffled beside the headline (C3, natural order) in the driver's ONE line:
    250 k triangles at N=8 on the matrix-core kernel, and the seed-12345 shuffled 10^6-triangle mesh at N=4, once in the
    caller's order (BDG_SW2D_KEEP_ORDER) and once with the solver's automatic renumbering."""
    out = {}
    ms, K, Np, _, affine = also_measure(8, (500, 250))
    gbps = algorithmic_bytes_per_element(8) * K / (ms * 1e-3) / 1e9
    traffic, source = committed_traffic(8, K)   # PMC passes of `bench.py --order 8 --cells 500x250` at these sources, or None
    out["config5_n8"] = {"ms": ms, "elements": K, "order": 8, "updates_per_s": Np * K / (ms * 1e-3), "achieved_GBps": gbps,
                         "frac": gbps / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": source,
                         "kernel": stage_kernel_name(8, K, affine), "stages_timed": ALSO_STAGES, "ramp_stages_untimed": ALSO_RAMP}
    ms_given, K, Np, renum_given, _ = also_measure(4, (1000, 500), shuffle_seed=12345, keep_order=True)
    ms_renum, _, _, renum, affine = also_measure(4, (1000, 500), shuffle_seed=12345)
    gbps = algorithmic_bytes_per_element(4) * K / (ms_renum * 1e-3) / 1e9
    out["config3_shuffled"] = {"ms_as_given": ms_given, "ms_renumbered": ms_renum, "elements": K, "order": 4, "seed": 12345,
                               "renumbered_internally": [renum_given, renum], "updates_per_s_renumbered": Np * K / (ms_renum * 1e-3),
                               "frac_renumbered": gbps / HBM_PEAK_GBPS, "kernel": stage_kernel_name(4, K, affine), "stages_timed": ALSO_STAGES, "ramp_stages_untimed": ALSO_RAMP}
    return out


def run_single(args):
    os.environ.setdefault("OMP_NUM_THREADS", str(min(32, os.cpu_count() or 8)))
    import blitzdg_amd.pyblitzdg as dg
    from blitzdg_amd import sw2d

    t_setup = time.perf_counter()
    mesh = dg.MeshManager()
    mesh.buildBoxMesh(NX, NY, shuffleSeed=args.shuffle_seed)
    nodes = dg.TriangleNodesProvisioner(ORDER, mesh)
    ctx = nodes.dgContext()
    K, Np = ctx.numElements, ctx.numLocalPoints
    solver = sw2d.Sw2dSolver(nodes=nodes, g=G, device=0, flags=(sw2d.REORDER if args.reorder else 0)
                             | (sw2d.NODAL_GEOMETRY if args.nodal_geometry else 0))
    h, hu, hv = initial_state(ctx.x, ctx.y)
    solver.setState(h, hu, hv)
    dt, _ = solver.computeDt(CFL)
    t_setup = time.perf_counter() - t_setup

    # A fresh box ramps its clocks during the first few hundred launches (round 1: 0.39 ms cold against
    # 0.35 ms warm for the same kernel). Untimed ramp: blocks of 50 stages until two consecutive blocks
    # agree within 2 % (at most 60 blocks); the requested warm-up and the timed steps follow.
    ramp_blocks, prev = 0, None
    while not args.no_clock_ramp and ramp_blocks < 60:
        cur = solver.timeLSERK4Stages(dt, 50)
        ramp_blocks += 1
        if prev is not None and abs(cur - prev) <= 0.02 * prev:
            break
        prev = cur
    solver.lserk4Stages(dt, args.warmup)
    solver.synchronize()
    t0 = time.perf_counter()
    ms_per_launch = solver.timeLSERK4Stages(dt, args.steps)  # exactly K launches, HIP events on their stream
    solver.synchronize()
    wall = time.perf_counter() - t0
    _, eta_max = solver.computeDt(CFL)  # raises if the run blew up
    # total mass before / after (walls everywhere: conserved to round-off). Computed only now: NumPy's BLAS
    # threads keep spinning for a while after a product and would compete with the launch thread above.
    V = ctx.V
    quad = np.linalg.inv(V @ V.T).sum(axis=0)  # nodal quadrature weights of the reference triangle

    def total_mass(hh):
        return float((quad @ hh * ctx.J[0]).sum())
    mass0 = total_mass(h)
    mass_drift = (total_mass(solver.getState()[0]) - mass0) / mass0
    probe_ms = solver.probeStageTraffic(20) if solver.usesAffineGeometry else None
    triad = sw2d.streamTriadGBps(0)

    bytes_elem = algorithmic_bytes_per_element(ORDER)
    achieved = bytes_elem * K / (ms_per_launch * 1e-3) / 1e9
    traffic, traffic_src = committed_traffic(ORDER, K)
    actual_bytes = actual_bytes_per_element(ORDER, solver.usesAffineGeometry) * K
    line = {
        "metric": f"element-DOF updates/sec (sw2d RHS + LSERK4 stage, N={ORDER}, {K / 1e6:g}M tris)",
        "value": Np * K * args.steps / wall,
        "unit": "element-DOF updates/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"sw2d RHS + fused LSERK4 stage, synthetic box {NX}x{NY} cells = {K} triangles, "
                               f"N={ORDER} (Np={Np}), " + (f"element order shuffled (seed {args.shuffle_seed})"
                               
                               if args.shuffle_seed else "natural element order") + ", walls on all sides",
                   "order": ORDER, "elements": K, "fields": 3, "step": "one fused RHS+LSERK4 stage launch",
                   "geometry": "affine (one metric set per element, one normal/scale per face)"
                               if solver.usesAffineGeometry else "nodal",
                   "actual_hbm_bytes_per_element": actual_bytes_per_element(ORDER, solver.usesAffineGeometry),
                   "renumbered_internally": solver.isRenumbered,
                   "clock_ramp_stages_untimed": 50 * ramp_blocks,
                   "dt": dt, "eta_max_after": eta_max, "mass_relative_drift": mass_drift,
                   "setup_seconds": round(t_setup, 2),
                   "device_bytes": solver.deviceBytes},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": bytes_elem * K, "kernel_ms": ms_per_launch,
                     # what the kernel really moves (affine geometry is compressed): compulsory bytes of the
                     # kernel as written, and the PMC-measured bytes where a current summary exists
                     "kernel_compulsory_bytes_per_launch": actual_bytes,
                     "actual_GBps": (traffic or actual_bytes) / (ms_per_launch * 1e-3) / 1e9,
                     "actual_frac_of_peak": (traffic or actual_bytes) / (ms_per_launch * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "measured_stream_triad_GBps": triad, "same_access_pattern_no_compute_ms": probe_ms,
                     "kernel": stage_kernel_name(ORDER, K, solver.usesAffineGeometry)},
    }
    headline = ORDER == 4 and (NX, NY) == (1000, 500) and not (args.shuffle_seed or args.reorder or args.nodal_geometry)
    if headline and not args.no_also and not os.environ.get("BDG_SW2D_AFFINE_VARIANT"):
        del solver              # (its 2.6 GB are not needed beside the next meshes)
        line["also"] = also_block()
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(ORDER)
    print(json.dumps(line), flush=True)


def distributed_line(world, steps, warmup, wall, K, Np, counts, transport, mass_drift=None, handoff=None):
    bytes_elem = algorithmic_bytes_per_element(ORDER)
    achieved = bytes_elem * K * steps / wall / 1e9
    return {
        "metric": f"element-DOF updates/sec (sw2d RHS + LSERK4 stage, N={ORDER}, {K / 1e6:g}M tris)",
        "value": Np * K * steps / wall,
        "unit": "element-DOF updates/s",
        "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": wall / steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"sw2d RHS + fused LSERK4 stage, synthetic box {NX}x{NY} cells = {K} "
                               f"triangles, N={ORDER}, partitioned into {world} parts (RCB), ghost-element "
                               "halo over RCCL overlapped with interior elements",
                   "order": ORDER, "elements": K, "fields": 3, "parallelism": f"elem-partition x{world}",
                   "transport": transport, "stage_dependencies": handoff, "rank0_partition": counts,
                   # walls everywhere: total mass is conserved to round-off only if every ghost trace is right
                   "mass_relative_drift": mass_drift},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS * world, "unit": "GB/s",
                     "frac": achieved / (HBM_PEAK_GBPS * world), "traffic": None,
                     "note": "whole-job wall time incl. halo exchange, not a per-kernel figure"},
    }


def cap_host_threads(world):
    """Host-side setup runs its element loops on up to OMP_NUM_THREADS (or BDG_NUM_THREADS) workers; with one process
    per GPU keep the total thread count sane."""
    cores = os.cpu_count() or 8
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(16, cores // max(world, 1)))))


def run_distributed_native(args):
    """One process per GPU; RCCL driven from the C++ library (no PyTorch in the workers). The
    timed region is bracketed by an all-rank barrier + device synchronisation on both sides and
    the maximum over ranks is taken."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    cap_host_threads(world)
    from blitzdg_amd.halo import NativeDistributedSw2d

    from blitzdg_amd._capi import lib
    # a launcher may expose one GPU per process (then it is ordinal 0) or all of them (ordinal LOCAL_RANK)
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank))) % max(1, lib.bdg_device_count())
    d = NativeDistributedSw2d.box(NX, NY, ORDER, rank, world, g=G, device=local_rank)
    try:
        # The two launches of a stage meet through counters polled inside the kernels; those waits are bounded, and one that gave
        # up anywhere is reported by every rank's barrier. Then -- all ranks together -- the measurement starts over from the
        # initial state with the event form of the dependencies (BDG_SW2D_EVENT_SYNC=1), and the line says so.
        handoff = "events (pinned)" if os.environ.get("BDG_SW2D_EVENT_SYNC", "0") not in ("", "0") else "in-kernel counters where the kernels have them"
        for attempt in (0, 1):
            try:
                d.set_initial_state(initial_state)
                dt = d.compute_dt(CFL)
                d.lserk4_stages(dt, args.warmup)
                d.barrier()
                t0 = time.perf_counter()
                d.lserk4_stages(dt, args.steps)
                d.barrier()
                break
            except Exception as exc:  # noqa: BLE001  (the binding's error type carries the library's message)
                if attempt == 1 or "timed out" not in str(exc):
                    raise
                os.environ["BDG_SW2D_EVENT_SYNC"] = "1"
                handoff = "events (an in-kernel wait timed out in the first attempt; measured again from the initial state)"
        wall = d.allreduce_max(time.perf_counter() - t0)
        d.compute_dt(CFL)  # blow-up check (global)
        # mass before / after, only now (NumPy's BLAS threads would disturb the launch thread of the timed loop)
        mass1 = d.allreduce_sum(d.owned_mass())
        mass0 = d.allreduce_sum(d.owned_mass(initial_state))
        if rank == 0:
            print(json.dumps(distributed_line(world, args.steps, args.warmup, wall, d.global_elements, d.Np,
                                              d.halo_counts(), "native RCCL (ncclSend/ncclRecv groups)",
                                              mass_drift=(mass1 - mass0) / mass0, handoff=handoff)), flush=True)
    finally:
        d.close()


def run_rehearsal(args):
    """Schedule rehearsal on ONE GPU: rank 0's share of a --rehearse-world split with loop-back
    exchanges of the real message sizes. Prints the per-stage time of the overlapped pipeline; the
    state is not physical (ghosts are this rank's own elements), so no throughput is claimed."""
    cap_host_threads(1)
    from blitzdg_amd.halo import NativeDistributedSw2d
    world = args.rehearse_world
    out = []
    ranks = os.environ.get("BDG_REHEARSE_RANKS")
    for r in ([int(v) for v in ranks.split(",")] if ranks else sorted({0, world // 2, world - 1})):
        d = NativeDistributedSw2d.box(NX, NY, ORDER, r, world, g=G, device=0, loopback=True)
        try:
            d.set_initial_state(initial_state)
            dt = 0.25 * d.compute_dt(CFL)
            d.lserk4_stages(dt, args.warmup)
            d.barrier()
            t0 = time.perf_counter()
            d.lserk4_stages(dt, args.steps)
            issued = time.perf_counter() - t0
            d.barrier()
            wall = time.perf_counter() - t0
            out.append({"rank": r, "ms_per_stage": wall / args.steps * 1e3,
                        "host_issue_ms_per_stage": issued / args.steps * 1e3, **d.halo_counts()})
        finally:
            d.close()
    print(json.dumps({"rehearsal": True, "world": world, "order": ORDER, "cells": [NX, NY], "steps": args.steps,
                      "ranks": out}), flush=True)


def run_distributed_torch(args):
    import torch
    import torch.distributed as dist

    from blitzdg_amd.halo import DistributedSw2d

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank))) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    try:
        d = DistributedSw2d.box(NX, NY, ORDER, g=G, device=local_rank)
        d.set_initial_state(initial_state)
        dt = d.compute_dt(CFL)
        for _ in range(args.warmup):
            d.lserk4_stage(dt)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            d.lserk4_stage(dt)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        wall = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        dist.all_reduce(wall, op=dist.ReduceOp.MAX)
        wall = wall.item()
        d.compute_dt(CFL)  # blow-up check
        counts = d.halo_counts()
        if rank == 0:
            print(json.dumps(distributed_line(world, args.steps, args.warmup, wall, d.global_elements, d.Np, counts,
                                              "torch.distributed nccl (RCCL) batch_isend_irecv")), flush=True)
        d.close()
    finally:
        dist.destroy_process_group()


def launch_ranks(args, script=None, argv=None):
    """`python bench.py --gpus N` started plainly (no launcher, WORLD_SIZE unset): this parent -- which
    never touches the GPU -- starts N rank processes of this same script, one per GPU, with the
    environment torch.distributed.run would give them, relays rank 0's single JSON line and returns the
    worst exit code. Refuses a line whose n_gpus is not the N that was asked for."""
    import secrets
    import socket
    import subprocess
    n = args.gpus
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs, nonce = [], secrets.token_hex(8)
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BDG_LAUNCH_NONCE=nonce)
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)]
                                      + (sys.argv[1:] if argv is None else argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = p.wait() or rc
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    if rc != 0 or len(lines) != 1:
        sys.stderr.write(out)
        raise SystemExit(rc or 1)
    if json.loads(lines[0]).get("n_gpus") != n:
        raise SystemExit(f"bench.py: rank 0 reported n_gpus={json.loads(lines[0]).get('n_gpus')}, asked for {n}")
    print(lines[0], flush=True)


def run_distributed(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and os.environ.get("BDG_BENCH_FORCE_DISTRIBUTED") != "1":
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; refusing to mislabel the line")
    if os.environ.get("BDG_TRANSPORT", "native") == "torch":
        run_distributed_torch(args)
    else:
        run_distributed_native(args)


def main():
    global ORDER, NX, NY
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clock-ramp", action="store_true", help="skip the untimed clock-ramp blocks")
    ap.add_argument("--no-also", action="store_true", help="skip the untimed extra configurations (config 5, config 3 shuffled)")
    ap.add_argument("--order", type=int, default=ORDER, help="polynomial order (default: the BASELINE metric's N=4)")
    ap.add_argument("--cells", default=f"{NX}x{NY}", help="box cells NXxNY, 2 triangles each (default 1000x500)")
    ap.add_argument("--shuffle-seed", type=int, default=0, help="Fisher-Yates element shuffle (adversarial ordering)")
    ap.add_argument("--reorder", action="store_true", help="let the solver renumber elements internally (BFS)")
    ap.add_argument("--nodal-geometry", action="store_true", help="force the per-node-geometry kernels")
    ap.add_argument("--rehearse-world", type=int, default=0,
                    help="one-GPU schedule rehearsal of an N-way split (loop-back exchanges; timing only)")
    args = ap.parse_args()
    ORDER = args.order
    NX, NY = (int(v) for v in args.cells.lower().split("x"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.rehearse_world > 1:
        run_rehearsal(args)
    elif args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)
    elif args.gpus > 1 or world > 1 or os.environ.get("BDG_BENCH_FORCE_DISTRIBUTED") == "1":
        run_distributed(args)
    else:
        run_single(args)


if __name__ == "__main__":
    main()
