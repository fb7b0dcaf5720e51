#!/usr/bin/env bash
# r6 GPU call 37: gn_bwd_apply_kernel (41 launches, 1.0 ms per step, 4.3-4.5 TB/s on the large maps) with non-temporal accesses, libraries linked by hand:
# nt0 = default policy, nt1 = the x / dz loads non-temporal, nt2 = the dx store, nt3 = both.  Same box, alternating.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c37
mkdir -p $O
run() { echo "$1: $(CGD_LIB_PATH=$ROOT/clip-guided-diffusion_amd/variants/libcgd_$1.so timeout 300 python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step')")"; }
for i in 1 2 3; do
  for v in nt0 nt1 nt2 nt3; do run $v; done
done | tee $O/ab_gn_bwd_apply_nt.txt
