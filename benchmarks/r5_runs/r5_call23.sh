#!/usr/bin/env bash
# r5 GPU call 23: the attention tests after the launchers moved to one selector (attn_select): all four CGD_ATTN_FLASH selections + exact contexts
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r5c23
mkdir -p $O
timeout 35 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "attention" > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -2 $O/pytest.log
