#!/usr/bin/env bash
# r6 GPU call 42: kgemm_kernel's activation runs as buffer loads (rows beyond M and k-steps beyond the wavefront's last out of range: no row selects, no
# clamped re-reads) — B — against A = the build of commit "wconv / attn_flash: drop variables ...".  Parity, then a same-box A/B.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c42
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm or unet or headline or mini" 2>&1 | tail -2 | tee $O/pytest.log
timeout 1200 bash benchmarks/ab.sh 4 150 2>&1 | tee $O/ab_kgemm_bufload.txt
