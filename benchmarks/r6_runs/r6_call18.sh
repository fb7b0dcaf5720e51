#!/usr/bin/env bash
# r6 GPU call 18: wconv_kernel's staging transform in six pieces, one per step on the 8-row tile (CGD_WCONV_FINE = 1) against the three-piece schedule
# of rounds 2-5 (libcgd_prev.so = the same sources built with -DCGD_WCONV_FINE=0): parity tests of the conv paths on the new build, then a same-box A/B
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c18
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv or unet or groupnorm" 2>&1 | tail -5 | tee $O/pytest_conv.log
timeout 1200 bash benchmarks/ab.sh 3 150 2>&1 | tee $O/ab_wconv_fine.txt
