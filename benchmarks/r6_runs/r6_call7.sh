#!/usr/bin/env bash
# r6 GPU call 7: EmbedAhead control experiment (mode 2: the same events, no second stream) + its parity tests + headline step test
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c7
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_step.py -m gpu -x -q -k "one_step_ahead or headline_shape_single or p_sample_trajectory_bf16x3" > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -4 $O/pytest.log
run() { echo "$1: $(env $2 timeout 300 python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', r['config']['launches_per_step'], 'launches')")"; }
for i in 1 2; do
  run "in line (default)                         " "A=1"
  run "one step ahead, side stream (=1)          " "CGD_EMBED_AHEAD=1"
  run "one step ahead, same stream + events (=2) " "CGD_EMBED_AHEAD=2"
done | tee $O/ab_embed_ahead_control.txt
