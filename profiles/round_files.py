#!/usr/bin/env python3
"""Turns what profiles/collect_round.sh left under gpurun_out/<tag>final/ (and the PMC summaries under gpurun_out/summaries/)
into the committed profiles/<tag>_* files:  python3 profiles/round_files.py r03"""
import json
import os
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
O, P = os.path.join(R, "gpurun_out", TAG + "final"), os.path.join(R, "profiles")


def jsonl(path):
    return [json.loads(ln) for ln in open(path) if ln.startswith("{")] if os.path.exists(path) else []


rows = jsonl(os.path.join(O, "curved_timings.jsonl"))
old = os.path.join(P, f"{TAG}_curved_timings.json")
keep = json.load(open(old)).get("round2_kernels_on_reference_rules", []) if os.path.exists(old) else []
if rows:     # (a round that did not touch the curved kernels collects nothing for them)
  json.dump({"note": "ms per RHS evaluation of the curved / over-integrated solver (profiles/time_curved.py: deformed box, 5 % of the elements in curvedEls, "
                   "midpoint RK2 + filter, tracer, drag array, bed slope), one box, final sources of the round. form = nodal-trace (default, "
                   "sw2d_curved_nt_kernel.hpp) or general (BDG_SW2D_CURVED_GENERAL=1: the round-2 stage kernels with this round's fix-up kernel). "
                   "round2_kernels_on_reference_rules: the round-2 code on the reference's cubature rules, measured at the start of round 3.",
           "final": rows, "round2_kernels_on_reference_rules": keep}, open(old, "w"), indent=1)
json.dump({"note": "ms per RHS evaluation of the fused step kernels (profiles/time_rk2.py), final sources of the round, one box",
           "orders": jsonl(os.path.join(O, "rk2_timings.jsonl"))}, open(os.path.join(P, f"{TAG}_rk2_timings.json"), "w"), indent=1)
cr = jsonl(os.path.join(O, "curved_rehearsal.jsonl"))
if cr:
    json.dump({"note": "loop-back rehearsal of a partitioned run of the curved solver (profiles/time_curved_rehearsal.py): ONE GPU computes rank 1's "
                       "share of a world-way split, every neighbour exchange a real RCCL send-to-self; ms per RHS evaluation on the two-chain schedule, "
                       "with every element in stream order, and of the whole mesh on the same GPU", "runs": cr},
              open(os.path.join(P, f"{TAG}_curved_rehearsal.json"), "w"), indent=1)
for t in (f"{TAG}_curved_n4", f"{TAG}_curved_n8"):
    for suf in ("_pmc_summary.json", "_kernel_stats.csv"):
        src = os.path.join(R, "gpurun_out", "summaries", t + suf)
        if os.path.exists(src):
            shutil.copyfile(src, os.path.join(P, t + suf))

reh = jsonl(os.path.join(O, "rehearsal.jsonl"))
lines = ["Loop-back rehearsal of the overlapped multi-GPU stage schedule (bench.py --rehearse-world 8): ONE GPU computes rank r's share of an",
         "8-way split of the mesh; every neighbour exchange is a real RCCL send-to-self of the true size. Timing only. Final sources of the round,",
         "one box, one call (profiles/collect_round.sh): per order the rehearsal of ranks 0, 1, 4 and then the whole mesh on the same GPU.", ""]
i = 0
while i < len(reh):
    d = reh[i]
    if d.get("rehearsal") and i + 1 < len(reh):
        ms = reh[i + 1]["ms_per_step"]
        per = [(r["rank"], round(r["ms_per_stage"], 4), r["owned"] - r["interior"], r["peers"]) for r in d["ranks"]]
        worst, best = max(p[1] for p in per), min(p[1] for p in per)
        lines.append(f"N={d['order']} cells={d['cells']}: (rank, ms per stage, boundary elements, peers) {per}; whole mesh {ms:.4f} ms per stage "
                     f"=> {ms / worst:.2f}x .. {ms / best:.2f}x")
        i += 2
    else:
        i += 1
lines.append("")
ev = jsonl(os.path.join(O, "rehearsal_events.jsonl"))
if ev:
    lines.append("the same with the round-3 form of the dependencies (events on the queues, BDG_SW2D_EVENT_SYNC=1), same box, same call:")
    for d in ev:
        lines.append(f"N={d['order']} cells={d['cells']}: (rank, ms per stage) {[(r['rank'], round(r['ms_per_stage'], 4)) for r in d['ranks']]}")
    lines.append("")
for d in jsonl(os.path.join(O, "rehearsal_n4_w42.jsonl")):
    lines.append(f"N=4 world={d['world']}: {[(r['rank'], round(r['ms_per_stage'], 4)) for r in d['ranks']]}")
lines += ["", "== kernel timeline of the 8-way rehearsal, rank 4 (rocprofv3 --kernel-trace): start us, duration us, end us", "-- N=4"]
lines += open(os.path.join(O, "timeline_n4.txt")).read().rstrip().split("\n") + ["-- N=8"] + open(os.path.join(O, "timeline_n8.txt")).read().rstrip().split("\n")
exp = os.path.join(P, f"{TAG}_rehearsal_experiments.txt")
if os.path.exists(exp):
    lines += [""] + open(exp).read().rstrip().split("\n")
open(os.path.join(P, f"{TAG}_rehearsal.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[4:12]))
