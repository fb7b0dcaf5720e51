#!/usr/bin/env bash
# r4 GPU call 5: GroupNorm forward statistics from the wconv epilogue: UNet / headline parity, step-level A/B (CGD_GN_EPI=0 / 1)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
mkdir -p gpurun_out/r4b5
python -m pytest tests -m gpu -x -q -k "test_unet or headline_shape_single or groupnorm or winograd" > gpurun_out/r4b5/pytest.log 2>&1
tail -5 gpurun_out/r4b5/pytest.log
for v in 0 1 0 1; do
  CGD_GN_EPI=$v python bench.py --steps 150 --warmup 5 --no-cpu-baseline > gpurun_out/r4b5/bench_epi${v}.json 2>/dev/null
  python -c "
import json;r=json.load(open('gpurun_out/r4b5/bench_epi${v}.json'));print('GN_EPI $v', r['value'],r['ms_per_step'],r['config']['launches_per_step'], r['hbm']['ms_per_step'], r['hbm']['frac'], r['roofline']['frac'], r['roofline']['avg_launch_us'])"
done
