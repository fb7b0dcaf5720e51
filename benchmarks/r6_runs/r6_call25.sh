#!/usr/bin/env bash
# r6 GPU call 25: kconv_kernel's patch / weight-fragment prefetches as buffer loads (CGD_KCONV_BUFLOAD = 1, B: prefetches past the end of a slice are
# out of range and touch no memory) against clamped global loads (A = build of commit "hgemm2: operand fetches as buffer loads").  Parity, then A/B.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c25
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv or unet" 2>&1 | tail -5 | tee $O/pytest_conv.log
timeout 1200 bash benchmarks/ab.sh 4 150 2>&1 | tee $O/ab_kconv_bufload.txt
