#!/usr/bin/env bash
# Final validation of round 6 (one GPU call): the whole GPU suite + smoke on the final build
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > $O/r6final_pytest_gpu.log 2>&1
echo "pytest rc $?"; grep -E "passed|failed|error" $O/r6final_pytest_gpu.log | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/r6final_smoke.log 2>&1
echo "smoke rc $?"; tail -3 $O/r6final_smoke.log
