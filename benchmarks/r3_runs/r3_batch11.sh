#!/usr/bin/env bash
# Stall diagnosis of wconv_kernel on its two dominant shapes (input to the next round's work on the dominant kernel)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
bash benchmarks/pmc_probe.sh r3wconv wconv_kernel bench_wconv.py 6 "256,256,256;128,256,256;256,512,256" > gpurun_out/r3_pmc_wconv.txt 2>&1
tail -5 gpurun_out/pmc_r3wconv/l2.log
cat gpurun_out/r3_pmc_wconv.txt
