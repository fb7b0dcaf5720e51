#!/usr/bin/env bash
# r6 GPU call 2: which element of lgemm_kernel's chunk loop is the exposed cost?  Ablation builds (-DCGD_LGEMM_EXP bits: 1 no weight DMA, 2 no activation
# DMA, 4 no MFMA, 8 no LDS fragment reads) + ring-depth probes (tm 33: 32-row tiles on 4 buffers, tm 65: 64-row tiles on 2 buffers)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out/r6c2
mkdir -p $O
cd benchmarks/ubench
for e in 0 1 2 3 4 8 12; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -DCGD_LGEMM_EXP=$e lgemm_bench.hip -o lgemm_exp$e 2>/dev/null &
done
wait
{
for e in 0 1 2 3 4 8 12; do
  B=./lgemm_exp$e
  timeout 60 $B 800 768 768 3 64 2 300 8 | grep time
  timeout 60 $B 800 768 3072 3 64 2 300 8 | grep time
  timeout 60 $B 800 2304 768 1 64 2 300 8 | grep time
  timeout 60 $B 800 768 3072 1 64 2 300 8 | grep time
done
B=./lgemm_exp0
for tm in 32 33 64 65; do
  timeout 60 $B 800 768 3072 3 $tm 2 300 8
  timeout 60 $B 800 768 768 3 $tm 2 300 8
  timeout 60 $B 800 2304 768 1 $tm 2 300 8
done
} 2>&1 | tee ../../$O/lgemm_ablation.txt
