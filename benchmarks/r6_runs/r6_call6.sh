#!/usr/bin/env bash
# r6 GPU call 6: the (t, y)-only embedding head of the UNet one step ahead on a side stream (sampler.EmbedAhead): step tests that run through the
# sampler loops, same-box A/B through CGD_EMBED_AHEAD, per-kernel trace
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c6
mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_step.py -m gpu -x -q -k "headline or trajectory or batch2 or cosine or gating or dropin_generator_yields or dropin_generator_full or user_cond_fn or launcher_shards or two_ranks" > $O/pytest_steps.log 2>&1
echo "pytest rc $?"; tail -4 $O/pytest_steps.log
run() { echo "$1: $(env $2 timeout 300 python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', r['config']['launches_per_step'], 'launches')")"; }
for i in 1 2 3; do
  run "embedding head in line (CGD_EMBED_AHEAD=0)" "CGD_EMBED_AHEAD=0"
  run "embedding head one step ahead (default)   " "A=1"
done | tee $O/ab_embed_ahead.txt
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile > $O/trace.log 2>&1)
T=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python benchmarks/trace_step.py "$T" 30 > $O/trace_step.txt 2>&1
find $O/trace -name '*.csv' -size +5M -delete
head -24 $O/trace_step.txt
