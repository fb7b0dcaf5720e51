#!/usr/bin/env bash
# r6 GPU call 28: rows beyond M (hgemm2) and padding pixels (kconv) as out-of-range buffer offsets — zeros without a select (B) against A = the build
# of commit "wconv: patch pixels and weight fragments through buffer loads" (i.e. B also carries hconv2's buffer loads, -0.02 ms in call 27).
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c28
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm or conv or unet or vit" 2>&1 | tail -3 | tee $O/pytest.log
timeout 1200 bash benchmarks/ab.sh 4 150 2>&1 | tee $O/ab_oob_rows.txt
