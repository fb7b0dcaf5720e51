#!/usr/bin/env bash
# r5 GPU call 15: kgemm mode 2 as the default: UNets + the headline smoke
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r5c15
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "unet_small or unet_64 or unet_256 or unet_512 or test_gemm" > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -4 $O/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc $?"; tail -2 $O/smoke.log
