#!/usr/bin/env bash
# r6 GPU call 36: hconv2_kernel's wavefront -> sub-tile mapping on the final kernels: 128 pixels x 32 channels per wavefront (default: every wavefront
# reads ALL A fragments of the tile from LDS, 8 ds_read_b128 per k-step) against 64 pixels x 64 channels (hconv_var bit 2: 4 reads, twice the weight loads)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c36
mkdir -p $O
run() { echo "$1: $(env $2 timeout 300 python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step')")"; }
for i in 1 2 3; do
  run "default (128 px x 32 ch per wavefront)" "A=1"
  run "64 px x 64 ch per wavefront           " "CGD_HCONV_VAR=4"
done | tee $O/ab_hconv_nj.txt
