#!/usr/bin/env bash
# r4 GPU call 3: hgemm2 with two K-groups of wavefronts (KG = 2): micro-benchmark against the 4-wavefront kernel with float64 checks,
# the GEMM / ViT / UNet parity tests on the new default, step-level A/B
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
mkdir -p gpurun_out/r4b3
(
cd benchmarks/ubench
for cfg in 0 9; do
  for args in "130 128 256 1" "64 128 384 1" "200 128 1024 3" "70 96 640 2"; do
    timeout 20 ./hgemm_stamps $args 64 3 0 $cfg | grep -E "float64" | sed "s/^/pipeline $cfg  $args: /"
  done
done
for args in "64 128 768 1" "800 2304 768 1" "800 768 768 3" "800 3072 768 1" "800 768 3072 4" "1024 1536 512 1" "1024 512 512 2" "256 3072 1024 1" "256 1024 1024 4"; do
  for cfg in 0 9; do
    timeout 20 ./hgemm_stamps $args 64 20 0 $cfg | grep -E "checksum|^hgemm2|staged|chunk loop /|stores out   " | tr '\n' ' ' | sed 's/  */ /g'
    echo
  done
done
) > gpurun_out/r4b3/hgemm_kg.txt 2>&1
python -m pytest tests -m gpu -x -q -k "gemm or clip_vit or test_unet or attention or headline_shape_single" > gpurun_out/r4b3/pytest.log 2>&1
tail -5 gpurun_out/r4b3/pytest.log
for kg in 1 2 1 2; do
  CGD_HGEMM_KG=$kg python bench.py --steps 150 --warmup 5 --no-cpu-baseline > gpurun_out/r4b3/bench_kg${kg}.json 2>/dev/null
  python -c "
import json;r=json.load(open('gpurun_out/r4b3/bench_kg${kg}.json'));print('KG $kg', r['value'],r['ms_per_step'],r['roofline']['other_mfma_kernel'])"
done
