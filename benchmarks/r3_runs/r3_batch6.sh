#!/usr/bin/env bash
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
run() {
  env "$@" timeout 100 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null |
    python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', r['value'], r['ms_per_step'])"
}
for _ in 1 2; do
  run CGD_NOP=1
  run CGD_DEFER=2
  run CGD_DEFER=0
  run CGD_HGEMM=1,64,2
  run CGD_HGEMM=1,64,8
done
