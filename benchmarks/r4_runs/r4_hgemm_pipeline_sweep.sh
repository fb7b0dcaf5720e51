#!/usr/bin/env bash
# FIRST GPU call of the next round for the weight-GEMM block.  Model to test: a wavefront can have at most 63 vector-memory instructions in flight
# (vmcnt is 6 bits), and with the L2 cold at every launch both operand streams see ~1.7 us: the shipped pipeline (weight ring 7 k-steps ahead,
# activations 2 chunks ahead) is bounded at 4 x 1.7 / 7 = 0.97 us and 1.7 / 2 = 0.85 us per chunk — the measured 0.94.  Ring 16 + 4 staging sets
# (48 loads in flight, ~300 registers, one wavefront per SIMD) would allow 0.45.
# sweep hgemm2_kernel<1, 64>'s software-pipeline depth (weight ring 8 / 12 / 16 k-steps x
# activation staging sets 2 / 3 / 4) on the ViT GEMM shapes; every configuration must print the same checksum per shape and "ok" against float64 on
# the small shapes.  ~25 s of box time.  Then: make the best configuration the default of the library instantiation, same-box A/B of the step, full suite.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT/benchmarks/ubench"
[ -x hgemm_stamps ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include hgemm_stamps.hip -o hgemm_stamps
for cfg in 0 1 2 3 4 5 6 7 8; do
  for args in "130 128 192 1" "64 128 448 1" "200 128 1024 3"; do
    timeout 20 ./hgemm_stamps $args 64 3 0 $cfg | grep -E "float64" | sed "s/^/pipeline $cfg  $args: /"
  done
done
for args in "64 128 768 1" "800 2304 768 1" "800 768 768 4" "800 3072 768 1" "800 768 3072 4" "800 3072 768 3"; do
  for cfg in 0 1 2 3 4 5 6 7 8; do
    timeout 20 ./hgemm_stamps $args 64 20 0 $cfg | grep -E "checksum|^hgemm2|staged|chunk loop /|stores out   " | tr '\n' ' ' | sed 's/  */ /g'
    echo
  done
done

# Ablation (wrong results, timing only): the chunk loop with one element removed at a time — bit 0 weight-fragment loads, 1 activation patch path,
# 2 barrier, 3 A-fragment LDS reads; 15 = MFMAs + schedule only.  Shows which latency of the chain is exposed (per-k-step stamps are not usable).
for e in 1 2 4 8 3 15; do
  # build these HERE before the gpurun call (binaries travel with the snapshot; compiling on the GPU box costs ~25 s of box time each)
  [ -x hgemm_exp$e ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -DCGD_HGEMM_EXP=$e hgemm_stamps.hip -o hgemm_exp$e 2>/dev/null
  for args in "64 128 768 1" "800 2304 768 1"; do
    timeout 20 ./hgemm_exp$e $args 64 20 0 0 | grep -E "^hgemm2|chunk loop /" | tr '\n' ' ' | sed "s/  */ /g; s/^/ablation $e: /"
    echo
  done
done
