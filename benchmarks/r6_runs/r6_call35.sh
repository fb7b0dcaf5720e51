#!/usr/bin/env bash
# r6 GPU call 35: MFMA results in architectural VGPRs (-mllvm -amdgpu-mfma-vgpr-form): A = previous build (accumulators in AGPRs everywhere),
# B = this build (flag on attn_flash.hip: the softmax touches the accumulators every key block), C = B + the flag on hgemm.hip and hconv.hip (their
# epilogues read the accumulators once: 192-384 v_accvgpr moves per wavefront less).  Parity of B and C, then same box, alternating.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c35
mkdir -p $O
echo "parity B: $(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "attn or attention or vit or headline or mini" 2>&1 | tail -1)" | tee $O/parity.txt
echo "parity C: $(CGD_LIB_PATH=$ROOT/clip-guided-diffusion_amd/variants/libcgd_c.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm or conv or vit or headline or mini or cfg256" 2>&1 | tail -1)" | tee -a $O/parity.txt
run() { echo "$1: $(CGD_LIB_PATH=$2 timeout 300 python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step')")"; }
for i in 1 2 3 4; do
  run A $ROOT/clip-guided-diffusion_amd/libcgd_prev.so
  run B $ROOT/clip-guided-diffusion_amd/libcgd_mi355x.so
  run C $ROOT/clip-guided-diffusion_amd/variants/libcgd_c.so
done | tee $O/ab_mfma_vgpr_form.txt
