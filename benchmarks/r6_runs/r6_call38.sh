#!/usr/bin/env bash
# r6 GPU call 38: wconv_kernel epilogue with non-temporal accesses (gn_bwd_apply already non-temporal in all four): w0 = default policy, w1 = the output stores,
# w2 = the second operand (residual / backward-sum input) loads, w3 = both.  Same box, alternating.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c38
mkdir -p $O
run() { echo "$1: $(CGD_LIB_PATH=$ROOT/clip-guided-diffusion_amd/variants/libcgd_$1.so timeout 300 python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step')")"; }
for i in 1 2 3; do
  for v in w0 w1 w2 w3; do run $v; done
done | tee $O/ab_wconv_nt.txt
