#!/usr/bin/env bash
# r5 GPU call 18: the fused backward for T <= 64 attention (CGD_ATTN_FLASH=3: flash forward + one-workgroup backward): parity, micro-benchmark, step A/B
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r5c18
mkdir -p $O
CGD_ATTN_FLASH=3 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_attention or clip_vit_b32 or unet_small" > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -4 $O/pytest.log
CGD_ATTN_FLASH=3 timeout 200 python benchmarks/probe_attn.py 50 2>/dev/null | grep -E "T64|T50" | tee $O/attn_flash3.txt
run() { echo "$1: $(env $2 timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', r['config']['launches_per_step'], 'launches')")"; }
for i in 1 2; do
  run "default                        " "A=1"
  run "flash fwd + fused bwd at T<=64 " "CGD_ATTN_FLASH=3"
done | tee $O/ab.txt
