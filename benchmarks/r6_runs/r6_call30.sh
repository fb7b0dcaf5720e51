#!/usr/bin/env bash
# r6 GPU call 30: wconv_kernel PAIR (two consecutive pixel tiles per workgroup on the 256 x 256 level; the second tile's chunk 0 is staged in the first
# tile's last chunk) — first look: parity of the Winograd / UNet tests, then CGD_WCONV_PAIR=0 / 1 on the same build and against the previous build
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c30
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "winograd or cfg256 or headline or conv" 2>&1 | tail -3 | tee $O/pytest.log
run() { echo "$1: $(env $2 timeout 300 python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step')")"; }
for i in 1 2 3; do
  run "previous build            " "CGD_LIB_PATH=$ROOT/clip-guided-diffusion_amd/libcgd_prev.so"
  run "this build, CGD_WCONV_PAIR=0" "CGD_WCONV_PAIR=0"
  run "this build, CGD_WCONV_PAIR=1" "CGD_WCONV_PAIR=1"
done | tee $O/ab_wconv_pair.txt
