#!/usr/bin/env bash
# r4 GPU call 7: GroupNorm BACKWARD sums from the dgrad conv's epilogue (CGD_GN_EPI bit 1): UNet / headline parity, step-level A/B 1 vs 3
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out/r4b7
mkdir -p $O
python -m pytest tests -m gpu -x -q -k "test_unet or headline_shape_single" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
for v in 1 3 1 3; do
  CGD_GN_EPI=$v python bench.py --steps 150 --warmup 5 --no-cpu-baseline > $O/bench_epi${v}.json 2>/dev/null
  python -c "
import json;r=json.load(open('$O/bench_epi${v}.json'));print('GN_EPI $v', r['value'],r['ms_per_step'],r['config']['launches_per_step'], r['hbm']['ms_per_step'], r['hbm']['frac'], r['roofline']['frac'], r['roofline']['avg_launch_us'])"
done
