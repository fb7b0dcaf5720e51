#!/usr/bin/env bash
# r6 GPU call 41: the non-temporal accesses only for tensors of at least 32 MB (the L2): g0 = always (the build), g1 = wconv output stores gated,
# g2 = gn_bwd_apply streams gated, g3 = both gated.  Same box, alternating.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c41
mkdir -p $O
run() { echo "$1: $(CGD_LIB_PATH=$ROOT/clip-guided-diffusion_amd/variants/libcgd_$1.so timeout 300 python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step')")"; }
for i in 1 2 3; do
  for v in g0 g1 g2 g3; do run $v; done
done | tee $O/ab_nt_size_gate.txt
