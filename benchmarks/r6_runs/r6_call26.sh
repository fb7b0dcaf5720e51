#!/usr/bin/env bash
# r6 GPU call 26: wconv_kernel operand fetches as buffer loads, one source tree, three libraries linked by hand:
#   b0 = global loads (commit "kconv: operand prefetches as buffer loads"), b1 = weight fragments through a buffer resource (scalar step offsets),
#   b2 = patch pixels through a buffer resource (offsets computed once, padding = out-of-range = zeros without a select); b3 (both) spills in the
# second run (r6c26b): b3 = both, with 4 instead of 8 epilogue instructions in flight in the NC = 2 instantiations (no spills)
#   epilogue of the NC = 2 instantiations and is not measured.  Parity of b1 and b2 on the Winograd tests, then same box, alternating.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c26b
mkdir -p $O
for v in b3; do
  echo "== parity $v: $(CGD_LIB_PATH=$ROOT/clip-guided-diffusion_amd/variants/libcgd_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "winograd or cfg256 or headline" 2>&1 | tail -1)"
done | tee $O/parity.txt
run() { echo "$1: $(CGD_LIB_PATH=$ROOT/clip-guided-diffusion_amd/variants/libcgd_$1.so timeout 300 python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step')")"; }
for i in 1 2 3 4; do
  for v in b0 b2 b3; do run $v; done
done | tee $O/ab_wconv_bufload.txt
