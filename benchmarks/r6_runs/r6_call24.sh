#!/usr/bin/env bash
# r6 GPU call 24: hgemm2_kernel on buffer loads (CGD_HGEMM_BUFLOAD = 1, B: prefetches past the end of a slice are out-of-range loads that touch no
# memory; no per-lane 64-bit address arithmetic) against clamped global loads (A = build of commit 1b78e45).  Parity first, then a same-box A/B.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c24
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm or vit or unet or lgemm" 2>&1 | tail -5 | tee $O/pytest_gemm.log
timeout 1200 bash benchmarks/ab.sh 4 150 2>&1 | tee $O/ab_hgemm_bufload.txt
