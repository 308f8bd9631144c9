#!/bin/bash
# rocprofv3 evidence for the curved / over-integrated RHS (profiles/time_curved.py) on the GPU box:
#   bash profiles/collect_curved.sh r03_curved_n4 4 500 250 [kernel-name substring, default sw2d_curved_nt_kernel]
# Kernel trace / stats and every PMC group are SEPARATE rocprofv3 runs of the same command (PMC passes never
# combine with trace domains). Raw output: gpurun_out/prof_<tag>_*; summary: profiles/<tag>_{pmc_summary.json,kernel_stats.csv}
set -euo pipefail
TAG=$1; ORDER=$2; NX=$3; NY=$4; KERNEL=${5:-sw2d_curved_nt_kernel}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
rm -rf "$OUT"/prof_${TAG}_*
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/profiles/time_curved.py $ORDER $NX $NY"
# BDG_COLLECT_CMD: another timing command with the same contract (last argument = number of evaluations / stages, one JSON line
# with order and elements), e.g. "python3 profiles/time_stage_variant.py B 4 1000x500" with the kernel substring as fifth argument
if [ -n "${BDG_COLLECT_CMD:-}" ]; then CMD="$BDG_COLLECT_CMD"; fi
$CMD 40 > "$OUT/prof_${TAG}_warm.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${TAG}_trace" -- $CMD 40 > "$OUT/prof_${TAG}_trace.log" 2>&1
for group in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
             "sq:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
             "grbm:GRBM_GUI_ACTIVE GRBM_COUNT" \
             "mfma:SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" \
             "mem:SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
  name=${group%%:*}; ctrs=${group#*:}
  timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/prof_${TAG}_${name}" -- $CMD 10 > "$OUT/prof_${TAG}_${name}.log" 2>&1 || true
done
python3 "$R/profiles/summarize.py" "$TAG" "$OUT" --into "$OUT/summaries" --kernel "$KERNEL" > "$OUT/prof_${TAG}_summary.txt" 2>&1 || true
tail -60 "$OUT/prof_${TAG}_summary.txt"
