#!/usr/bin/env bash
# hgemm2_kernel<1, 64>: chunk-loop time against the number of workgroups that run at once / share an operand (is the 0.94 us per chunk a
# per-workgroup pipeline latency or contention?): one workgroup alone, 18 (one row tile: nobody shares a weight panel), 13 (one weight panel
# shared by all), the full ViT qkv launch
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT/benchmarks/ubench"
for args in "64 128 768 1 64" "64 2304 768 1 64" "800 128 768 1 64" "800 2304 768 1 64" "800 2304 3072 1 64"; do
  timeout 20 ./hgemm_stamps $args 20 0 0 | grep -E "^hgemm2|staged|chunk loop|stores out   "
done
