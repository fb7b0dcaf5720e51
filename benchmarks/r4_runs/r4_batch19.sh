#!/usr/bin/env bash
# r4 GPU call 19: weight prefetch on a side stream (ViT tower first): crash / parity check of the tower, step-level A/B CGD_PREFETCH=0 / 1
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out/r4b19
mkdir -p $O
python -m pytest tests -m gpu -x -q -k "test_clip_vit_b32 or p_sample_trajectory_bf16x3" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for v in 0 1 0 1; do
  CGD_PREFETCH=$v python bench.py --steps 150 --warmup 5 --no-cpu-baseline > $O/bench.json 2>/dev/null
  python -c "
import json;r=json.load(open('$O/bench.json'));print('PREFETCH $v', r['value'],r['ms_per_step'],r['config']['launches_per_step'],r['roofline']['other_mfma_kernel']['ms_per_step'], r['config']['host_isolated_enqueue_ms_per_rank'])"
done
