#!/usr/bin/env bash
# r5 GPU call 20: the bench line with the round's final defaults (CGD_ATTN_FLASH=3)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r5c20
mkdir -p $O
timeout 170 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc $?"; tail -c 1500 $O/bench.json
