#!/usr/bin/env bash
# r4 GPU call 8: wconv_kernel with two channel blocks per wavefront (8 x 16 pixels x 256 channels): parity, step-level A/B
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out/r4b8
mkdir -p $O
python -m pytest tests -m gpu -x -q -k "winograd or test_unet or headline_shape_single" > $O/pytest.log 2>&1
tail -8 $O/pytest.log
for v in "1 1" "0 1" "3 1" "1 3" "0 3" "1 1" "0 1"; do
  set -- $v
  CGD_WINO_NC=$1 CGD_GN_EPI=$2 python bench.py --steps 150 --warmup 5 --no-cpu-baseline > $O/bench_nc$1_epi$2.json 2>/dev/null
  python -c "
import json;r=json.load(open('$O/bench_nc$1_epi$2.json'));print('WINO_NC $1 GN_EPI $2', r['value'],r['ms_per_step'],r['config']['launches_per_step'], r['hbm']['ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_us'], r['roofline']['launches_per_step'])"
done
