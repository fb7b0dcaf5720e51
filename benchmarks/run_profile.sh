#!/usr/bin/env bash
# rocprofv3 passes for the headline workload (run on the GPU box via gpurun):
#   1. --kernel-trace --stats          -> per-kernel time table (the summary committed under profiles/)
#   2. --pmc FETCH_SIZE                -> HBM read bytes per dispatch   (own pass; gfx950: x2 correction)
#   3. --pmc WRITE_SIZE                -> HBM write bytes per dispatch  (own pass)
#   4. --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE  -> MFMA pipe utilisation
# Usage: benchmarks/run_profile.sh <tag> [steps]
set -uo pipefail
TAG=${1:-run}
STEPS=${2:-4}
PT=${PASS_TIMEOUT:-240}   # per-pass limit (a counter pass has hung once: never let one pass eat the GPU budget)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-profile"
timeout $PT rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- $BENCH > "$OUT/trace.log" 2>&1
timeout $PT rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o f -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
timeout $PT rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o w -- $BENCH > "$OUT/pmc_write.log" 2>&1
[[ -n "${SKIP_MFMA:-}" ]] || timeout $PT rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma" -o m -- $BENCH > "$OUT/pmc_mfma.log" 2>&1
python "$ROOT/benchmarks/summarize_profile.py" "$OUT" "$OUT/pmc_traffic.json" > "$OUT/summary.txt" 2>&1
# keep the merged-back payload small: the raw per-dispatch CSVs of the PMC passes are reduced to the summary
find "$OUT" -name '*kernel_trace.csv' -size +20M -delete
tail -40 "$OUT/summary.txt"
