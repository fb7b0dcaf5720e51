#!/usr/bin/env bash
# r5 GPU call 9: existing knobs re-swept on the round-5 build (same-box, 100 steps each)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r5c9
mkdir -p $O
run() { echo "$1: $(env $2 timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', r['config']['launches_per_step'], 'launches', r['config']['splitk_reduce_per_step'], 'reduces')")"; }
for i in 1 2; do
  run "default                              " "A=1"
  run "fused GN only at >= 65536 px         " "CGD_FUSE_GN=1,1073741824,65536"
  run "fused GN only at <= 16384 px         " "CGD_FUSE_GN=1,16384,0"
  run "hgemm min chunks 3                   " "CGD_HGEMM=1,64,3"
  run "hgemm min chunks 6                   " "CGD_HGEMM=1,64,6"
  run "kconv min chunks 2                   " "CGD_KCONV=1,1024,2,1"
  run "kconv min chunks 8                   " "CGD_KCONV=1,1024,8,1"
  run "small-map GroupNorm only <= 256 px   " "CGD_GN_SMALL_HW=256"
  run "tile order N-major                   " "CGD_TILE_ORDER=1"
  run "tile order M-major                   " "CGD_TILE_ORDER=2"
done | tee $O/ab.txt
