#!/usr/bin/env bash
# r6 GPU call 19: kconv_kernel's patch-conversion passes one per k-step slot (CGD_KCONV_FINE = 1) against two in each of the first slots
# (libcgd_prev.so = the build of the previous commit): conv / UNet parity on the new build, then a same-box A/B
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c19
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv or unet" 2>&1 | tail -5 | tee $O/pytest_conv.log
timeout 1200 bash benchmarks/ab.sh 4 150 2>&1 | tee $O/ab_kconv_fine.txt
