#!/usr/bin/env bash
# r6 GPU call 11: the UNet's embedding head as 3 GEMV launches (GemmParams::a_mode) instead of 8: bit-identity test, UNet / step tests, same-box A/B
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c11
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_step.py -m gpu -x -q -k "embedding_head or unet_small or unet_64 or headline_shape_single or p_sample_trajectory or test_gemm" > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -4 $O/pytest.log
run() { echo "$1: $(env $2 timeout 300 python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', r['config']['launches_per_step'], 'launches')")"; }
for i in 1 2 3; do
  run "8-launch embedding head (CGD_EMBED_FUSE=0)" "CGD_EMBED_FUSE=0"
  run "3 GEMVs forming their A rows (default)    " "A=1"
done | tee $O/ab_embed_fuse.txt
