#!/usr/bin/env bash
# Stall diagnosis of the halo conv kernel (run on the GPU box): three separate --pmc passes over benchmarks/probe_hconv.py,
# reduced per (kernel, grid size) by benchmarks/summarize_pmc.py.  Usage: benchmarks/pmc_hconv.sh <tag>
set -uo pipefail
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/benchmarks/probe_hconv.py"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES \
  --output-format csv -d "$OUT/sq" -o a -- $CMD > "$OUT/sq.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_TA_BUSY_sum TD_TD_BUSY_sum TCC_HIT_sum TCC_MISS_sum \
  --output-format csv -d "$OUT/tc" -o b -- $CMD > "$OUT/tc.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_LDS_DATA_FIFO_FULL SQ_INSTS_LDS SQ_WAIT_ANY \
  --output-format csv -d "$OUT/lds" -o c -- $CMD > "$OUT/lds.log" 2>&1
python "$ROOT/benchmarks/summarize_pmc.py" "$OUT" hconv2 > "$OUT/summary.txt" 2>&1
find "$OUT" -name '*.csv' -size +8M -delete
cat "$OUT/summary.txt"
