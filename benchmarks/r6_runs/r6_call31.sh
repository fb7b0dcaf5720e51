#!/usr/bin/env bash
# r6 GPU call 31: the GroupNorm-fusion threshold again on the final kernels (CGD_FUSE_GN 3rd field = fuse only for convs of at least this many pixels)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c31
mkdir -p $O
run() { echo "$1: $(env $2 timeout 300 python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', r['config']['launches_per_step'], 'launches')")"; }
for i in 1 2 3; do
  run "default: fused for >= 4096 pixels (64^2 up)" "A=1"
  run "fused for >= 16384 pixels (128^2 up)        " "CGD_FUSE_GN=1,1073741824,16384"
  run "fused for >= 1024 pixels (32^2 up)          " "CGD_FUSE_GN=1,1073741824,1024"
  run "fused for >= 65536 pixels (256^2 only)      " "CGD_FUSE_GN=1,1073741824,65536"
done | tee $O/ab_fuse_gn_thresholds_final.txt
