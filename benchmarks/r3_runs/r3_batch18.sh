#!/usr/bin/env bash
# PMC stall diagnosis of hgemm2_kernel<1, 64> on the ViT qkv shape (M 800, N 2304, K 768), via the timeline micro-benchmark's binary
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_r3hgemm
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="$ROOT/benchmarks/ubench/hgemm_stamps 800 2304 768 1 64 30 0"
timeout 40 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY \
  --output-format csv -d "$OUT/sq" -o a -- $CMD > "$OUT/sq.log" 2>&1
timeout 40 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU TCC_HIT_sum TCC_MISS_sum \
  --output-format csv -d "$OUT/lds" -o c -- $CMD > "$OUT/lds.log" 2>&1
python "$ROOT/benchmarks/summarize_pmc.py" "$OUT" hgemm2 > "$OUT/summary.txt" 2>&1
find "$OUT" -name '*.csv' -size +8M -delete
cat "$OUT/summary.txt"
