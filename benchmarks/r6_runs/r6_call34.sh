#!/usr/bin/env bash
# r6 GPU call 34: the staging loads of the flash-attention kernels as buffer loads (rows beyond T out of range: zeros without an index select + four data
# selects per row; no 64-bit per-lane address arithmetic), the -inf key mask only in the ragged last block (B) against A = the build of commit 6f... (HEAD)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c34
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "attn or attention or vit or headline or cfg64 or mini" 2>&1 | tail -3 | tee $O/pytest.log
timeout 1200 bash benchmarks/ab.sh 4 150 2>&1 | tee $O/ab_attn_bufload.txt
