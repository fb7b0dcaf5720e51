#!/usr/bin/env bash
# hgemm2_kernel<1, 64> with the activation patch fetched four chunks ahead (four staging register sets): float64 check on small shapes that
# cover 1..9 chunks per slice (loop tails, split-K), then timeline + checksum on the ViT shapes of r3_batch17 / r3_batch20
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT/benchmarks/ubench"
for args in "70 128 64 1" "70 160 128 1" "130 128 192 1" "64 256 256 1" "200 128 320 1" "64 128 384 1" "100 128 448 1" "64 128 512 1" "64 128 576 1" "300 256 768 5" "200 128 1024 3" "800 768 768 4"; do
  timeout 20 ./hgemm_stamps $args 64 3 0 | grep -E "float64|^hgemm2" | cut -c1-150
done
for args in "64 128 768 1 64" "800 2304 768 1 64" "800 3072 768 1 64" "800 768 3072 4 64"; do
  timeout 20 ./hgemm_stamps $args 20 0 | grep -E "checksum|float64|^hgemm2|staged|chunk loop|stores out   "
done
