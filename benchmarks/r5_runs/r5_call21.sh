#!/usr/bin/env bash
# r5 GPU call 21: final build confirmation: attention selections (one predicate for forward / backward) + smoke (mini + headline step vs the oracle)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r5c21
mkdir -p $O
timeout 60 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "attention" > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -2 $O/pytest.log
timeout 75 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc $?"; tail -3 $O/smoke.log
