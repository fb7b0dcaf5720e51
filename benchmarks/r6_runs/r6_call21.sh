#!/usr/bin/env bash
# r6 GPU call 21: wconv_kernel's last chunk without staging / weight prefetch (CGD_WCONV_PEEL = 1, B) against the self-re-staging last chunk
# (A = same sources, -DCGD_WCONV_PEEL=0); both builds with the GroupNorm of the <= 32 x 32 maps materialised.  Parity first, then a same-box A/B.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c21
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv or unet" 2>&1 | tail -5 | tee $O/pytest_conv.log
timeout 1200 bash benchmarks/ab.sh 4 150 2>&1 | tee $O/ab_wconv_peel.txt
