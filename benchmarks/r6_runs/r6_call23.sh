#!/usr/bin/env bash
# r6 GPU call 23: kconv_kernel without the loads / conversions of the last two chunks that nobody consumes (CGD_KCONV_PEEL = 1, B) against clamped
# chunk indices (A = build of commit 1b78e45).  Parity first, then a same-box A/B.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c23
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv or unet" 2>&1 | tail -5 | tee $O/pytest_conv.log
timeout 1200 bash benchmarks/ab.sh 4 150 2>&1 | tee $O/ab_kconv_peel.txt
