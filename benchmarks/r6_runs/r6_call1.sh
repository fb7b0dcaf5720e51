#!/usr/bin/env bash
# r6 GPU call 1: lgemm_kernel (LDS-DMA loader wavefronts + pre-split bf16 planes) against hgemm2_kernel on the ViT / UNet weight-GEMM shapes:
# bitwise agreement and time per launch, cache-resident (sets 1) and with rotating operand sets (sets 8: cold L2 like inside a step)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out/r6c1
mkdir -p $O
cd benchmarks/ubench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include lgemm_bench.hip -o lgemm_bench 2>/dev/null
B=./lgemm_bench
{
for sets in 1 8; do
  for nld in 2 1; do
    timeout 60 $B 800 2304 768 1 64 $nld 200 $sets
    timeout 60 $B 800 768 768 3 64 $nld 200 $sets
    timeout 60 $B 800 768 768 1 64 $nld 200 $sets
    timeout 60 $B 800 768 768 2 32 $nld 200 $sets
    timeout 60 $B 800 3072 768 1 96 $nld 200 $sets
    timeout 60 $B 800 3072 768 1 64 $nld 200 $sets
    timeout 60 $B 800 768 3072 3 64 $nld 200 $sets
    timeout 60 $B 800 768 2304 3 64 $nld 200 $sets
  done
done
# UNet shapes: 64x64 / 32x32 attention qkv + proj, 1x1 skips on the big maps
timeout 60 $B 4096 1536 512 1 128 2 100 4
timeout 60 $B 4096 512 512 1 64 2 100 4
timeout 60 $B 1024 3072 1024 1 64 2 100 4
timeout 60 $B 1024 1024 1024 2 64 2 100 4
timeout 60 $B 65536 256 512 1 128 2 50 2
timeout 60 $B 16384 512 768 1 128 2 50 2
} 2>&1 | tee ../../$O/lgemm_bench.txt
