#!/usr/bin/env bash
# r6 GPU call 40: wconv_kernel patch loads of the chunk loop with the nt policy (p1) against the default policy (p0 = the build with the adopted non-temporal
# accesses).  Same box, alternating.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c40
mkdir -p $O
run() { echo "$1: $(CGD_LIB_PATH=$ROOT/clip-guided-diffusion_amd/variants/libcgd_$1.so timeout 300 python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step')")"; }
for i in 1 2 3 4; do
  for v in p0 p1; do run $v; done
done | tee $O/ab_wconv_patch_nt.txt
