#!/bin/bash
# Everything else the round's profiles/ files are made of, in one gpurun call (bash profiles/collect_round.sh r03):
# curved-solver timings of every order on both kernel forms, its PMC collections at N=4 / N=8, the 8-way rehearsal of
# every matrix-core order with the whole-mesh time of the same box, a kernel timeline of the rehearsal, the step-kernel
# timings of every variant, the loop-back rehearsal of the partitioned curved solver. Raw output under gpurun_out/<tag>final/, PMC summaries under gpurun_out/summaries/.
set -uo pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${TAG}final
mkdir -p "$OUT" "$R/gpurun_out/summaries"
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$R"
cfg() { case $1 in 1|2|3|4) echo "500 250";; 5) echo "400 200";; 6) echo "300 200";; 7) echo "250 160";; 8) echo "250 120";; esac; }
for form in nodal-trace general; do
  for n in 2 3 4 5 6 7 8; do
    if [ $form = general ]; then export BDG_SW2D_CURVED_GENERAL=1; else unset BDG_SW2D_CURVED_GENERAL; fi
    timeout -k 10 200 python3 profiles/time_curved.py $n $(cfg $n) 20 | sed "s/^{/{\"form\": \"$form\", /"
  done
done > "$OUT/curved_timings.jsonl" 2>"$OUT/curved_timings.err"
unset BDG_SW2D_CURVED_GENERAL
bash profiles/collect_curved.sh ${TAG}_curved_n4 4 500 250 > "$OUT/collect_curved_n4.log" 2>&1
bash profiles/collect_curved.sh ${TAG}_curved_n8 8 250 120 > "$OUT/collect_curved_n8.log" 2>&1
for n in 4 5 6 7 8; do
  cells=$([ $n = 4 ] && echo 1000x500 || ([ $n = 5 ] && echo 800x400 || ([ $n = 6 ] && echo 1000x250 || echo 500x250)))
  BDG_REHEARSE_RANKS=0,1,4 python3 bench.py --rehearse-world 8 --steps 40 --warmup 10 --order $n --cells $cells 2>/dev/null | grep '^{'
  python3 bench.py --order $n --cells $cells --steps 100 --warmup 20 --no-cpu-baseline --no-also 2>/dev/null | grep '^{'
done > "$OUT/rehearsal.jsonl"
for n in 4 2; do w=$n; done
for w in 4 2; do BDG_REHEARSE_RANKS=0,1 python3 bench.py --rehearse-world $w --steps 40 --warmup 10 2>/dev/null | grep '^{'; done > "$OUT/rehearsal_n4_w42.jsonl"
( cd /tmp && export TMPDIR=/tmp && for n in 4 8; do
    cells=$([ $n = 4 ] && echo 1000x500 || echo 500x250)
    BDG_REHEARSE_RANKS=4 rocprofv3 --kernel-trace --output-format csv -d "$OUT/reh_trace_n$n" -- python3 "$R/bench.py" --rehearse-world 8 --steps 40 --warmup 10 --order $n --cells $cells > /dev/null 2>&1
    python3 "$R/profiles/timeline.py" "$OUT/reh_trace_n$n" > "$OUT/timeline_n$n.txt" 2>&1
  done )
for n in 3 4 6 8; do
  cells=$([ $n -le 4 ] && echo 1000x500 || ([ $n = 6 ] && echo 1000x250 || echo 500x250))
  timeout -k 10 300 python3 profiles/time_rk2.py $n $cells 2>/dev/null | grep '^{'
done > "$OUT/rk2_timings.jsonl"
for cfg in "4 500 250 2" "4 500 250 4" "4 500 250 8" "6 300 200 4" "8 250 120 2"; do
  timeout -k 10 300 python3 profiles/time_curved_rehearsal.py $cfg 1 50 2>/dev/null | grep '^{'
done > "$OUT/curved_rehearsal.jsonl"
ls -la "$OUT"
