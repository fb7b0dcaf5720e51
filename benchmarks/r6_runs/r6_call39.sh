#!/usr/bin/env bash
# r6 GPU call 39: more non-temporal accesses on top of the adopted ones (gn_bwd_apply x / dz / dx, wconv output stores): e0 = that build, e1 = + pool2x2 / upsample2x /
# pair kernel stores and 2 x 2-sum loads, e2 = + hgemm2 outputs beyond 32 MB, e3 = + gn_bwd_apply add / add2 loads, e4 = all three.  Same box, alternating.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c39
mkdir -p $O
run() { echo "$1: $(CGD_LIB_PATH=$ROOT/clip-guided-diffusion_amd/variants/libcgd_$1.so timeout 300 python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step')")"; }
for i in 1 2 3; do
  for v in e0 e1 e2 e3 e4; do run $v; done
done | tee $O/ab_nt_more.txt
