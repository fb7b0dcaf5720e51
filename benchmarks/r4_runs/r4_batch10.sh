#!/usr/bin/env bash
# r4 GPU call 10: de-phasing the four wavefronts of wconv_kernel after every barrier (s_sleep w * n * 64 cycles)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT/benchmarks/ubench"
O=../../gpurun_out/r4b10
mkdir -p $O
for shape in "256 256 256" "256 512 256" "128 512 512"; do
for nb in 4 22 2; do
  for gn in 0 1; do
    for b in wconv_exp0 wconv_skew1 wconv_skew2 wconv_skew3 wconv_skew5; do
      [ "$shape" = "128 512 512" ] && [ $nb = 4 ] && continue
      timeout 30 ./$b $shape $gn $nb 10 | grep -E "wconv_kernel<|chunks 1..n-2|entry -> chunk 0|last chunk ->" | tr '\n' ' ' | sed "s/  */ /g; s/^/$b: /"
      echo
    done
  done
done
done > $O/wconv_skew.txt 2>&1
cut -c1-250 $O/wconv_skew.txt
