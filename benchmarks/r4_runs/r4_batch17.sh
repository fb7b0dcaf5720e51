#!/usr/bin/env bash
# r4 GPU call 17: cost of a grid-wide barrier (with / without cross-XCD payload visibility) against one kernel launch per phase
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT/benchmarks/ubench"
O=../../gpurun_out/r4b17
mkdir -p $O
(timeout 30 ./gridbarrier_probe 256 2000; timeout 30 ./gridbarrier_probe 512 2000; timeout 30 ./gridbarrier_probe 64 2000) > $O/gridbarrier.txt 2>&1
cat $O/gridbarrier.txt
