#!/usr/bin/env bash
# r4: full GPU suite + smoke on the current build (one GPU call)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
O=gpurun_out/r4val
mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
tail -15 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -5 $O/smoke.log
