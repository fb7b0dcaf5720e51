#!/usr/bin/env bash
# r6 GPU call 29: the fp32 -> bf16 hi / lo split as cgd_split_quad / cgd_split_oct (common.h: one packed conversion per pair and plane, the hi values
# back through a shift and a mask of the packed word) in hgemm2, kgemm, kconv, hconv2, wconv and the flash-attention kernels (B) against the element-wise
# formulation (A = build of commit "hgemm2 / kconv: rows beyond M ...").  Parity, then a same-box A/B.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c29
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm or conv or unet or vit or attn or attention" 2>&1 | tail -3 | tee $O/pytest.log
timeout 1200 bash benchmarks/ab.sh 4 150 2>&1 | tee $O/ab_split_quad.txt
