#!/usr/bin/env bash
# r4 GPU call 9: wconv_kernel chunk-loop ablation (which element of the loop costs what), 16-row tile and the 8-row x 256-channel tile, plain and fused GN
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT/benchmarks/ubench"
O=../../gpurun_out/r4b9
mkdir -p $O
for nb in 4 22; do
  for gn in 0 1; do
    for e in 0 1 2 4 8 16 3 11 15; do
      timeout 30 ./wconv_exp$e 256 256 256 $gn $nb 10 | grep -E "wconv_kernel<|chunks 1..n-2|entry -> chunk 0|last chunk ->" | tr '\n' ' ' | sed 's/  */ /g'
      echo
    done
  done
done > $O/wconv_ablation.txt 2>&1
# one workgroup per CU only (128x128 map on 16-row tiles: 64 tiles x 2 panels = 128 workgroups): the clock without the power cap
for e in 0 15; do
  timeout 30 ./wconv_exp$e 128 256 256 0 4 10 | grep -E "wconv_kernel<|chunks 1..n-2" | tr '\n' ' ' | sed 's/  */ /g'; echo
done >> $O/wconv_ablation.txt 2>&1
cut -c1-260 $O/wconv_ablation.txt
