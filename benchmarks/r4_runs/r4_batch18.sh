#!/usr/bin/env bash
# r4 GPU call 18: does touching a weight block ahead of its consumer (MALL warm-up) make the consumer's stream faster?
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT/benchmarks/ubench"
O=../../gpurun_out/r4b18
mkdir -p $O
(for mb in 4 8 16 36 72; do for t in 8 32; do timeout 30 ./mall_probe $mb $t; done; done) > $O/mall_probe.txt 2>&1
cat $O/mall_probe.txt
