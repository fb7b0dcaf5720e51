#!/usr/bin/env bash
# Same-box A/B of two library builds (the pool's boxes differ by +-3 %, so only same-box comparisons are meaningful):
#   A = clip-guided-diffusion_amd/libcgd_prev.so (copy of the build to compare against), B = the current libcgd_mi355x.so.
# Alternates A B A B ...; prints steps/s and ms/step of each run.  Usage: benchmarks/ab.sh [rounds] [steps] [extra bench.py flags]
set -uo pipefail
ROUNDS=${1:-2}
STEPS=${2:-120}
shift 2 2>/dev/null || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
A=$ROOT/clip-guided-diffusion_amd/libcgd_prev.so
B=$ROOT/clip-guided-diffusion_amd/libcgd_mi355x.so
run() {
  CGD_LIB_PATH=$2 python "$ROOT/bench.py" --steps "$STEPS" --warmup 5 --no-cpu-baseline --no-profile "${@:3}" 2>/dev/null |
    python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', r['value'], 'steps/s', r['ms_per_step'], 'ms/step')"
}
for _ in $(seq "$ROUNDS"); do
  [[ -f $A ]] && run A "$A" "$@"
  run B "$B" "$@"
done
