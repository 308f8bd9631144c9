#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun):
#   bash profiles/collect.sh r01
# Kernel trace/stats and each PMC group run as SEPARATE rocprofv3 invocations of the same
# bench.py command (PMC passes never combine with trace domains). Raw output goes to
# gpurun_out/prof_<tag>_*; profiles/summarize.py turns it into profiles/<tag>_*.{csv,json}.
set -euo pipefail
TAG=${1:-r01}
shift || true
EXTRA="$*"   # extra bench.py arguments, e.g. --order 8 --cells 500x250 (use a tag like r01_n8)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
rm -rf "$OUT"/prof_${TAG}_*   # stale CSVs from earlier collections would be averaged in
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --no-also $EXTRA"
$BENCH --steps 200 --warmup 20 > "$OUT/prof_${TAG}_warm.log" 2>&1   # a fresh box ramps its clocks during the first run
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${TAG}_trace" -- $BENCH --steps 200 --warmup 20 > "$OUT/prof_${TAG}_trace.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/prof_${TAG}_fetch" -- $BENCH --steps 20 --warmup 5 > "$OUT/prof_${TAG}_fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/prof_${TAG}_write" -- $BENCH --steps 20 --warmup 5 > "$OUT/prof_${TAG}_write.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d "$OUT/prof_${TAG}_tcc" -- $BENCH --steps 20 --warmup 5 > "$OUT/prof_${TAG}_tcc.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d "$OUT/prof_${TAG}_sq" -- $BENCH --steps 20 --warmup 5 > "$OUT/prof_${TAG}_sq.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d "$OUT/prof_${TAG}_grbm" -- $BENCH --steps 20 --warmup 5 > "$OUT/prof_${TAG}_grbm.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 --output-format csv -d "$OUT/prof_${TAG}_mfma" -- $BENCH --steps 20 --warmup 5 > "$OUT/prof_${TAG}_mfma.log" 2>&1 || true
python3 "$R/profiles/summarize.py" "$TAG" "$OUT" > "$OUT/prof_${TAG}_summary.txt" 2>&1 || true
cat "$OUT/prof_${TAG}_summary.txt"
