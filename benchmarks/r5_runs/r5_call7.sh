#!/usr/bin/env bash
# r5 GPU call 7: knob experiments (same-box pairs): kgemm_kernel also on the narrow-N linears of up to 1024 rows (the ViT's N = 768 GEMMs, the 32x32-level
# 1x1 convs), kconv 8x8 tiles / wconv also on the 64x64 level
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r5c7
mkdir -p $O
CGD_KGEMM=1,256,0,1024,1536 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "clip_vit_b32 or unet_small or unet_64" > $O/pytest_kgemm_big.log 2>&1
echo "pytest kgemm big rc $?"; tail -4 $O/pytest_kgemm_big.log
run() { echo "$1: $(env $2 timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', r['config']['launches_per_step'], 'launches', r['config']['splitk_reduce_per_step'], 'reduces')")"; }
for i in 1 2; do
  run "default                         " "A=1"
  run "kgemm M<=1024 N<=768            " "CGD_KGEMM=1,256,0,1024,768"
  run "kgemm M<=1024 N<=1536           " "CGD_KGEMM=1,256,0,1024,1536"
  run "kconv 8x8 also at 64x64         " "CGD_KCONV=1,4096,4,1"
  run "wconv also at 64x64             " "CGD_WINO=1,4096"
done | tee $O/ab.txt
