#!/usr/bin/env bash
# r6 GPU call 14: PMC passes (separate runs, counters only) of the lgemm experiment's micro-benchmark on the ViT qkv / out-proj shapes: MFMA busy, wait
# states, LDS conflicts, L2 hit / miss for lgemm_kernel next to hgemm2_kernel — the committed counter evidence behind profiles/r6_lgemm_microbench.txt
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
OUT=$ROOT/gpurun_out/pmc_r6lgemm
mkdir -p "$OUT"
B=$ROOT/benchmarks/ubench/lgemm_bench
[[ -x $B ]] || (cd benchmarks/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include lgemm_bench.hip -o lgemm_bench)
cd /tmp && export TMPDIR=/tmp
for shape in "800 2304 768 1 64 2 40 8" "800 768 3072 3 64 2 40 8"; do
  tag=$(echo $shape | tr ' ' '_')
  timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES \
    --output-format csv -d "$OUT/sq_$tag" -o a -- $B $shape > "$OUT/sq_$tag.log" 2>&1
  timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_SALU \
    --output-format csv -d "$OUT/lds_$tag" -o c -- $B $shape > "$OUT/lds_$tag.log" 2>&1
  timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum \
    --output-format csv -d "$OUT/l2_$tag" -o d -- $B $shape > "$OUT/l2_$tag.log" 2>&1
done
cd "$ROOT"
python benchmarks/summarize_pmc.py "$OUT" gemm > "$OUT/summary.txt" 2>&1
find "$OUT" -name '*.csv' -size +8M -delete
cat "$OUT/summary.txt"
