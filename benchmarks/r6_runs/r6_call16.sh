#!/usr/bin/env bash
# r6 GPU call 16: hconv2 on the 64x64 level with TWO co-resident workgroups per CU (4 split-K slices of 4 chunks = 512 workgroups instead of 2 x 8 = 256;
# the slices are summed by the consuming GroupNorm sweep either way) — the round's observation that co-resident workgroups hide each other's waits
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c16
mkdir -p $O
run() { echo "$1: $(env $2 timeout 300 python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', r['config']['launches_per_step'], 'launches', r['config']['splitk_reduce_per_step'], 'reduces')")"; }
for i in 1 2 3; do
  run "default (256 slots, >= 4 chunks)     " "A=1"
  run "512 slots, >= 4 chunks               " "CGD_HCONV_SPLIT=512,4"
  run "512 slots, >= 2 chunks               " "CGD_HCONV_SPLIT=512,2"
done | tee $O/ab_hconv_split.txt
