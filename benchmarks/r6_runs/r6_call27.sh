#!/usr/bin/env bash
# r6 GPU call 27: hconv2_kernel on buffer loads (CGD_HCONV_BUFLOAD = 1, B) against global loads (A = build of commit "wconv: patch pixels and weight
# fragments through buffer loads"): same-box A/B first, then the WHOLE GPU suite on build B (every kernel family changed since the last full run)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c27
mkdir -p $O
timeout 1200 bash benchmarks/ab.sh 4 150 2>&1 | tee $O/ab_hconv_bufload.txt
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
