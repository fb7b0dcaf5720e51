#!/usr/bin/env bash
# wconv_kernel's coalesced epilogue (CGD_WINO_EPI=1): timeline per layer shape, parity + bit-identity test, whole-step A/B
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT/benchmarks/ubench"
[ -x wconv_stamps ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include wconv_stamps.hip -o wconv_stamps
for args in "256 256 256 0 4" "256 256 256 1 4" "256 256 512 0 4" "128 256 256 0 2"; do
  for res in 0 1; do
    for lep in 0 1; do
      timeout 60 ./wconv_stamps $args 10 - $lep $res | grep -E "^wconv_kernel|staged|chunks 1|stores out  |entry -> stores"
    done
  done
  echo
done
cd "$ROOT"
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "winograd" 2>&1 | tail -3
run() {
  env "$@" timeout 100 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null |
    python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', r['value'], r['ms_per_step'])"
}
for _ in 1 2; do
  run CGD_WINO_EPI=0
  run CGD_WINO_EPI=1
done
