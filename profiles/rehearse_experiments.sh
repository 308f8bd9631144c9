#!/bin/bash
# A/B runs of the 8-way loop-back rehearsal (bench.py --rehearse-world 8, rank 4) under environment switches, one line each:
#   bash profiles/rehearse_experiments.sh <out-file> "<N> <cells> <VAR=value ...>" ...
# Each experiment prints: order, switches, ms per stage of the rehearsed rank.
set -uo pipefail
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for spec in "$@"; do
  set -- $spec
  n=$1; cells=$2; shift 2
  line=$(env "$@" BDG_REHEARSE_RANKS=${RANKS:-4} timeout -k 10 240 python3 bench.py --rehearse-world 8 --steps ${STEPS:-40} --warmup 10 --order $n --cells $cells 2>/dev/null | grep '^{' | python3 -c 'import sys, json; d = json.loads(sys.stdin.read()); print(" ".join("%.4f" % r["ms_per_stage"] for r in d["ranks"]))')
  echo "N=$n $cells [$*] ms per stage: $line" | tee -a "$OUT"
done
