#!/usr/bin/env python3
"""Condenses the raw rocprofv3 CSVs written by profiles/collect.sh into the small files that
are committed under profiles/:  <tag>_kernel_stats.csv (verbatim --stats table) and
<tag>_pmc_summary.json (per-launch means of every counter for the stage kernel, plus the HBM
traffic derived as MI355X_MICROARCH.md prescribes: FETCH_SIZE/WRITE_SIZE are in KiB, and on
gfx950 FETCH_SIZE counts 64 B per 128-B request for coalesced streams, so reads = 2 x FETCH_SIZE
-- calibrated here on scatter_rows_kernel<double>, whose byte count is known exactly).

  python profiles/summarize.py <tag> <gpurun_out dir> [--into profiles/] [--kernel sw2d_curved_stage]
(--kernel: substring of the kernel the counters are summarized for; default the straight-element stage kernels)
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def newest(pattern):
    """gpurun merges a call's files into gpurun_out/ without removing those of earlier calls: only the newest CSV of a
    directory belongs to the collection being summarized."""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]


def counters(path, kernel_substr):
    agg = defaultdict(list)
    extra = {}
    for f in newest(os.path.join(path, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                extra = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                                           "Scratch_Size", "Workgroup_Size", "Grid_Size") if k in r}
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}, extra


def main():
    tag, out = sys.argv[1], sys.argv[2]
    opts = dict(zip(sys.argv[3::2], sys.argv[4::2]))
    into = opts.get("--into")
    stage = opts.get("--kernel", "sw2d_stage")
    summary = {"tag": tag, "kernel": None, "counters": {}, "launches_sampled": {}}
    # what the collection ran: bench.py only quotes a summary that matches its workload and the device sources of today
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        summary["kernel_source_sha"] = bench.kernel_source_sha()
        for line in open(os.path.join(out, f"prof_{tag}_warm.log")):
            if line.startswith("{"):
                d = json.loads(line)
                cfg = d.get("config", d)        # bench.py nests them under "config", the profiles/time_*.py lines do not
                summary["order"], summary["elements"] = cfg["order"], cfg["elements"]
                summary["workload_line"] = {k: v for k, v in d.items() if k not in ("config", "roofline", "cpu_baseline")}
    except Exception as e:  # noqa: BLE001  (a summary without the tie is still a summary; bench.py then ignores it)
        summary["kernel_source_sha_error"] = repr(e)
    for group in ("fetch", "write", "tcc", "sq", "grbm", "mfma", "mem"):
        vals, n, extra = counters(os.path.join(out, f"prof_{tag}_{group}"), stage)
        summary["counters"].update(vals)
        summary["launches_sampled"].update(n)
        if extra:
            summary["resources"] = extra
    # calibration of the FETCH_SIZE factor on a kernel whose bytes are known: the (15, K) upload
    cal, _, cx = counters(os.path.join(out, f"prof_{tag}_fetch"), "scatter_rows_kernel<double>")
    stats = newest(os.path.join(out, f"prof_{tag}_trace", "*", "*_kernel_stats.csv"))
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        for r in rows:
            if stage in r["Name"]:
                summary["kernel"] = r["Name"]
                summary["kernel_trace"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                           "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]),
                                           "stddev_ns": float(r["StdDev"]), "percent_of_gpu_time": float(r["Percentage"])}
                break
    # the timed region of bench.py = the LAST 200 stage launches of the traced run (before them: the untimed
    # clock-ramp blocks and the warm-up, during which a fresh box is still raising its clocks)
    traces = newest(os.path.join(out, f"prof_{tag}_trace", "*", "*_kernel_trace.csv"))
    if traces:
        durs = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                for r in csv.DictReader(open(traces[0])) if stage in r["Kernel_Name"]]
        durs = [d for _, d in sorted(durs)][-200:]
        if durs:
            summary["kernel_trace_timed_region"] = {"launches": len(durs), "avg_ns": sum(durs) / len(durs),
                                                    "min_ns": min(durs), "max_ns": max(durs)}
    c = summary["counters"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        summary["hbm_traffic_per_launch"] = {
            "fetch_size_kib": c["FETCH_SIZE"], "write_size_kib": c["WRITE_SIZE"],
            "read_bytes": 2 * c["FETCH_SIZE"] * 1024, "write_bytes": c["WRITE_SIZE"] * 1024,
            "total_bytes": 2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024,
            "note": "reads = 2 x FETCH_SIZE (gfx950 counts 64 B per 128-B request on coalesced streams)"}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:
        # MfmaUtil as rocprofv3 defines it: busy cycles summed over SIMDs / (active cycles * 1024 SIMDs);
        # GRBM_GUI_ACTIVE is the sum over the 8 XCDs (MI355X_MICROARCH.md, DVFS note)
        summary["mfma_util_percent"] = 100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / ((c["GRBM_GUI_ACTIVE"] / 8.0) * 1024.0)
        if "SQ_INSTS_VALU_MFMA_MOPS_F64" in c and summary.get("kernel_trace"):
            flops = c["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0
            summary["mfma_f64_tflops"] = flops / (summary["kernel_trace"]["avg_ns"] * 1e-9) / 1e12
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        summary["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    print(json.dumps(summary, indent=1))
    if into:
        os.makedirs(into, exist_ok=True)
        with open(os.path.join(into, f"{tag}_pmc_summary.json"), "w") as f:
            json.dump(summary, f, indent=1)
        if stats:
            shutil.copyfile(stats[0], os.path.join(into, f"{tag}_kernel_stats.csv"))


if __name__ == "__main__":
    main()
