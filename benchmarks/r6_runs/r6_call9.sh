#!/usr/bin/env bash
# r6 GPU call 9: kconv_kernel on 16 x 16-pixel tiles (CGD_KCONV 7th field): parity (op level + UNets), same-box step A/B (16x16 + 32x32 levels,
# 32x32 only), per-kernel trace
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c9
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "16x16_tile or weight_streaming" > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -5 $O/pytest.log
run() { echo "$1: $(env $2 timeout 300 python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', r['config']['launches_per_step'], 'launches', r['config']['splitk_reduce_per_step'], 'reduces')")"; }
for i in 1 2 3; do
  run "default (8x8 tiles)                 " "A=1"
  run "16x16 tiles at 16x16 and 32x32      " "CGD_KCONV=1,1024,4,1,0,2,256"
  run "16x16 tiles at 32x32 only           " "CGD_KCONV=1,1024,4,1,0,2,1024"
done | tee $O/ab_kconv_th16.txt
(cd /tmp && export TMPDIR=/tmp && CGD_KCONV=1,1024,4,1,0,2,256 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile > $O/trace.log 2>&1)
T=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python benchmarks/trace_step.py "$T" 140 > $O/trace_step_th16.txt 2>&1
find $O/trace -name '*.csv' -size +5M -delete
grep -E "kconv|gn_small|splitk|step wall" $O/trace_step_th16.txt | head -40
