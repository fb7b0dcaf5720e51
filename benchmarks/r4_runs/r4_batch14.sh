#!/usr/bin/env bash
# r4 GPU call 14: wconv_kernel<GN,2,1,2> — 8-row tile with TWO workgroups per CU (two wavefronts per SIMD), against the 16-row tile (4) and the
# 8-row x 256-channel tile (22)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT/benchmarks/ubench"
O=../../gpurun_out/r4b14
mkdir -p $O
for shape in "256 256 256" "256 512 256" "256 256 512" "128 512 512" "128 1024 512"; do
  for gn in 0 1; do
    for nb in 4 22 222 2; do
      timeout 30 ./wconv_stamps $shape $gn $nb 10 | grep -E "wconv_kernel<|chunks 1..n-2|entry -> chunk 0|last chunk ->|workgroups per CU|CUs used" | tr '\n' ' ' | sed "s/  */ /g; s/^/nb $nb: /"
      echo
    done
  done
done > $O/wconv_occ2.txt 2>&1
cut -c1-330 $O/wconv_occ2.txt
