#!/bin/bash
# Round 4's collections besides the headline / N=8 PMC runs (profiles/collect.sh r04, r04_n8), one gpurun call:
#   bash profiles/collect_round4.sh r04
# the 8-way rehearsal of every matrix-core order (ranks 0, 1, 4) with the whole-mesh time of the same box, 4- and 2-way at N=4,
# kernel timelines of the rehearsal at N=4 and N=8, the step-kernel timings of every variant at N = 3, 4, 6, 8.
# Raw output under gpurun_out/<tag>final/; profiles/round_files.py turns it into the committed files.
set -uo pipefail
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${TAG}final
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$R"
for n in 4 5 6 7 8; do
  cells=$([ $n = 4 ] && echo 1000x500 || ([ $n = 5 ] && echo 800x400 || ([ $n = 6 ] && echo 1000x250 || echo 500x250)))
  BDG_REHEARSE_RANKS=0,1,4 python3 bench.py --rehearse-world 8 --steps 200 --warmup 20 --order $n --cells $cells 2>/dev/null | grep '^{'
  python3 bench.py --order $n --cells $cells --steps 100 --warmup 20 --no-cpu-baseline --no-also 2>/dev/null | grep '^{'
done > "$OUT/rehearsal.jsonl"
for n in 4 8; do
  cells=$([ $n = 4 ] && echo 1000x500 || echo 500x250)
  BDG_SW2D_EVENT_SYNC=1 BDG_REHEARSE_RANKS=0,1,4 python3 bench.py --rehearse-world 8 --steps 200 --warmup 20 --order $n --cells $cells 2>/dev/null | grep '^{'
done > "$OUT/rehearsal_events.jsonl"
for w in 4 2; do BDG_REHEARSE_RANKS=0,1 python3 bench.py --rehearse-world $w --steps 200 --warmup 20 2>/dev/null | grep '^{'; done > "$OUT/rehearsal_n4_w42.jsonl"
( cd /tmp && export TMPDIR=/tmp && for n in 4 8; do
    cells=$([ $n = 4 ] && echo 1000x500 || echo 500x250)
    BDG_REHEARSE_RANKS=4 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/reh_trace_n$n" -- python3 "$R/bench.py" --rehearse-world 8 --steps 200 --warmup 20 --order $n --cells $cells > /dev/null 2>&1
    python3 "$R/profiles/timeline.py" "$OUT/reh_trace_n$n" > "$OUT/timeline_n$n.txt" 2>&1
    rm -rf "$OUT/reh_trace_n$n"
  done )
for n in 3 4 6 8; do
  cells=$([ $n -le 4 ] && echo 1000x500 || ([ $n = 6 ] && echo 1000x250 || echo 500x250))
  timeout -k 10 300 python3 profiles/time_rk2.py $n $cells 2>/dev/null | grep '^{'
done > "$OUT/rk2_timings.jsonl"
ls -la "$OUT"
