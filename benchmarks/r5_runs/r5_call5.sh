#!/usr/bin/env bash
# r5 GPU call 5: kconv_kernel on 8 x 8-pixel tiles (two workgroups per CU): parity of both tile widths, UNets; same-box A/B of the tile width and of the
# split-K target; per-kernel trace
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r5c5
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "weight_streaming or unet_small or unet_64 or previous_kernels" > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -5 $O/pytest.log
run() { echo "$1: $(env $2 timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', r['config']['launches_per_step'], 'launches', r['config']['splitk_reduce_per_step'], 'reduces')")"; }
for i in 1 2; do
  run "default (8x8 tiles, 256 slots)" "A=1"
  run "8x16 tiles                    " "CGD_KCONV=1,1024,4,0"
  run "8x8 tiles, 512 slots          " "CGD_KCONV=1,1024,4,1,512"
  run "8x8 tiles, 512 slots, 2 chunks" "CGD_KCONV=1,1024,2,1,512"
  run "8x8 tiles, defer 2            " "CGD_DEFER=2"
done | tee $O/ab.txt
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile > $O/trace.log 2>&1)
T=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python benchmarks/trace_step.py "$T" 90 > $O/trace_step.txt 2>&1; grep -E "kconv|splitk|step wall|idle" $O/trace_step.txt | head -30
find $O/trace -name '*.csv' -size +5M -delete
