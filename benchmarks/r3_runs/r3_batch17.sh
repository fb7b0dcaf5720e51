#!/usr/bin/env bash
# Timeline of hgemm2_kernel's workgroups (benchmarks/ubench/hgemm_stamps.hip) on the ViT-B/32 GEMM shapes of the guidance step (M = 800 token rows):
# level 1 = stamps around the chunk loop only (undisturbed loop time), level 2 = a stamp after every chunk
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT/benchmarks/ubench"
[ -x hgemm_stamps ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include hgemm_stamps.hip -o hgemm_stamps
# qkv (N 2304), out_proj (768), fc1 (3072, K 768), fc2 (768, K 3072) with / without split-K; one 128-row case
for args in "800 2304 768 1 64" "800 768 768 4 64" "800 3072 768 1 64" "800 768 3072 4 64" "800 3072 768 3 64" "4096 1024 512 1 128"; do
  timeout 30 ./hgemm_stamps $args 20 0
  echo
done
