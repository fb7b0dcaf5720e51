#!/usr/bin/env bash
# Stall diagnosis of any probe script: three separate rocprofv3 --pmc passes (SQ issue / wait, LDS + instruction mix, L2 hit / miss), reduced per
# (kernel, grid size) by benchmarks/summarize_pmc.py.  Usage: benchmarks/pmc_probe.sh <tag> <kernel-name substring> <probe.py> [args]
set -uo pipefail
TAG=$1; SUB=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/benchmarks/$*"
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES \
  --output-format csv -d "$OUT/sq" -o a -- $CMD > "$OUT/sq.log" 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_SALU \
  --output-format csv -d "$OUT/lds" -o c -- $CMD > "$OUT/lds.log" 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum \
  --output-format csv -d "$OUT/l2" -o d -- $CMD > "$OUT/l2.log" 2>&1
python "$ROOT/benchmarks/summarize_pmc.py" "$OUT" "$SUB" > "$OUT/summary.txt" 2>&1
find "$OUT" -name '*.csv' -size +8M -delete
cat "$OUT/summary.txt"
