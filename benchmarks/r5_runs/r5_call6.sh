#!/usr/bin/env bash
# r5 GPU call 6: the whole GPU suite on the mid-round build (flash attention, kgemm, 96-row hgemm2 tiles, kconv 8x8 tiles, CGD_DEFER=2 default)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r5c6
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
echo "pytest rc $?"; tail -15 $O/pytest_gpu.log
