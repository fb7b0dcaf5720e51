#!/usr/bin/env bash
# r5 GPU call 17: fused GroupNorm staging only on the 256x256 level (the 128x128-level convs read a materialised normalised tensor)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r5c17
mkdir -p $O
run() { echo "$1: $(env $2 timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', r['config']['launches_per_step'], 'launches')")"; }
for i in 1 2 3; do
  run "default (fused everywhere)          " "A=1"
  
  run "fused everywhere except 16384 px    " "CGD_FUSE_GN=1,1073741824,0,16384"
done | tee $O/ab.txt
