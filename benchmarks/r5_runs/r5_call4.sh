#!/usr/bin/env bash
# r5 GPU call 4: 96-row hgemm2 tiles parity; per-kernel trace of the step (kgemm in place); A/B of TM96 and the kgemm variants
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r5c4
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gemm or clip_vit_b32" > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -5 $O/pytest.log
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile > $O/trace.log 2>&1)
T=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python benchmarks/trace_step.py "$T" 90 > $O/trace_step.txt 2>&1; head -75 $O/trace_step.txt
find $O/trace -name '*.csv' -size +5M -delete
run() { echo "$1: $(env $2 timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', r['config']['launches_per_step'], 'launches', r['config']['splitk_reduce_per_step'], 'reduces')")"; }
for i in 1 2; do
  run "default            " "A=1"
  run "tm96 off           " "CGD_HGEMM_TM96=0"
  run "kgemm off          " "CGD_KGEMM=0"
  run "kgemm deep rings   " "CGD_KGEMM=1,256,4"
  run "kgemm 64-row tiles " "CGD_KGEMM=1,256,1"
  run "kgemm 32-row tiles " "CGD_KGEMM=1,256,2"
done | tee $O/ab.txt
