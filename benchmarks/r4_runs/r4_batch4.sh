#!/usr/bin/env bash
# r4 GPU call 4: last-arriver probe (sc1 relaxed atomics vs fences vs two launches) and the 128-row hgemm2 tile on the ViT shapes with more split-K
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT/benchmarks/ubench"
mkdir -p ../../gpurun_out/r4b4
(
timeout 120 ./lastblock_probe 1024 64 20000
timeout 60 ./lastblock_probe 512 64 4000
timeout 60 ./lastblock_probe 256 2048 4000
) > ../../gpurun_out/r4b4/lastblock.txt 2>&1
(
for args in "800 2304 768 1 64" "800 2304 768 2 128" "800 2304 768 1 128" "800 768 768 3 64" "800 768 768 6 128" "800 768 768 3 128" "800 3072 768 1 64" "800 3072 768 1 128" "800 3072 768 2 128" "800 768 3072 4 64" "800 768 3072 6 128" "800 768 3072 4 128"; do
  timeout 20 ./hgemm_stamps $args 20 0 0 | grep -E "^hgemm2|staged|chunk loop /|stores out   " | tr '\n' ' ' | sed 's/  */ /g'
  echo
done
) > ../../gpurun_out/r4b4/hgemm_tm128.txt 2>&1
cat ../../gpurun_out/r4b4/lastblock.txt
cut -c1-330 ../../gpurun_out/r4b4/hgemm_tm128.txt
