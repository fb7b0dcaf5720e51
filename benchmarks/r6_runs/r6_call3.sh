#!/usr/bin/env bash
# r6 GPU call 3: wconv_kernel on the 64 x 64 level (CGD_WINO=1,4096, planner fixed so that hconv2's split-K no longer disqualifies it): parity of the UNets
# on that routing, same-box step A/B against the default, per-kernel trace of both; lgemm experiment test
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c3
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lgemm" > $O/pytest_lgemm.log 2>&1
echo "pytest lgemm rc $?"; tail -3 $O/pytest_lgemm.log
CGD_WINO=1,4096 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "unet_256 or unet_512 or unet_small or unet_batch2 or conv_winograd or groupnorm_conv" > $O/pytest_wino64.log 2>&1
echo "pytest wino64 rc $?"; tail -5 $O/pytest_wino64.log
run() { echo "$1: $(env $2 timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', r['config']['launches_per_step'], 'launches', r['config']['splitk_reduce_per_step'], 'reduces', r['config']['groupnorm_record_merges_per_step'], 'merges')")"; }
for i in 1 2 3; do
  run "default              " "A=1"
  run "wconv also at 64x64  " "CGD_WINO=1,4096"
done | tee $O/ab_wino64.txt
for v in default wino64; do
  E="A=1"; [[ $v == wino64 ]] && E="CGD_WINO=1,4096"
  (cd /tmp && export TMPDIR=/tmp && env $E timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$v -o t -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile > $O/trace_$v.log 2>&1)
  T=$(find $O/trace_$v -name '*kernel_trace.csv' | head -1)
  python benchmarks/trace_step.py "$T" 120 > $O/trace_step_$v.txt 2>&1
  find $O/trace_$v -name '*.csv' -size +5M -delete
done
head -60 $O/trace_step_wino64.txt
