#!/usr/bin/env bash
# r6 GPU call 15: what does the side stream of sampler.EmbedAhead cost per side launch?  The head is 3 launches now (7 in calls 6-7): same A/B again,
# plus the unfused head (CGD_EMBED_FUSE=0: 7 launches) on the side stream
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c15
mkdir -p $O
run() { echo "$1: $(env $2 timeout 300 python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', r['config']['launches_per_step'], 'launches')")"; }
for i in 1 2; do
  run "in line (default, 3-launch head)          " "A=1"
  run "side stream, 3-launch head                " "CGD_EMBED_AHEAD=1"
  run "side stream, 7-launch head                " "CGD_EMBED_AHEAD=1 CGD_EMBED_FUSE=0"
  run "same stream + events, 3-launch head       " "CGD_EMBED_AHEAD=2"
done | tee $O/ab_embed_ahead_side_launches.txt
