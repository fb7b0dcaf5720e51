#!/usr/bin/env bash
# r6 GPU call 4: wconv_kernel<..., F32> (exact fp32 products on v_mfma_f32_32x32x2_f32): op-level parity, UNets at precision 0, the step tests that run
# in precision 0, then the exact-fp32 bench column against the previous routing (CGD_WINO=0: implicit GEMM)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c4
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "winograd or exact_fp32 or (unet_small and mini-0) or knob_change or groupnorm_conv" > $O/pytest_f32_wino.log 2>&1
echo "pytest rc $?"; tail -5 $O/pytest_f32_wino.log
timeout 1500 python -m pytest tests/test_gpu_step.py -m gpu -x -q -k "fp32_mfma or early_schedule" > $O/pytest_f32_steps.log 2>&1
echo "pytest steps rc $?"; tail -5 $O/pytest_f32_steps.log
run() { echo "$1: $(env $2 timeout 600 python bench.py --steps 40 --warmup 3 --precision f32 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', r['config']['launches_per_step'], 'launches')")"; }
for i in 1 2; do
  run "f32, implicit GEMM (CGD_WINO=0)   " "CGD_WINO=0"
  run "f32, wconv_kernel<F32> (default)  " "A=1"
done | tee $O/ab_f32.txt
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_f32 -o t -- python $ROOT/bench.py --steps 3 --warmup 2 --precision f32 --no-cpu-baseline --no-profile > $O/trace_f32.log 2>&1)
T=$(find $O/trace_f32 -name '*kernel_trace.csv' | head -1)
python benchmarks/trace_step.py "$T" 60 > $O/trace_step_f32.txt 2>&1
find $O/trace_f32 -name '*.csv' -size +5M -delete
head -50 $O/trace_step_f32.txt
