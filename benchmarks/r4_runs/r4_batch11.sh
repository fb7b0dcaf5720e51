#!/usr/bin/env bash
# r4 GPU call 11: wconv epilogue with its second operand (residual / norm input) fetched one row group ahead: parity of the touched paths, A/B GN_EPI 1 vs 3
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out/r4b11
mkdir -p $O
python -m pytest tests -m gpu -x -q -k "winograd or test_unet or headline_shape_single" > $O/pytest.log 2>&1
tail -4 $O/pytest.log
for v in 1 3 1 3; do
  CGD_GN_EPI=$v python bench.py --steps 150 --warmup 5 --no-cpu-baseline > $O/bench_epi${v}.json 2>/dev/null
  python -c "
import json;r=json.load(open('$O/bench_epi${v}.json'));print('GN_EPI $v', r['value'],r['ms_per_step'],r['config']['launches_per_step'], r['hbm']['ms_per_step'], r['hbm']['frac'], r['roofline']['frac'], r['roofline']['avg_launch_us'])"
done
