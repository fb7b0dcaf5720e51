#!/usr/bin/env bash
# r4 GPU call 12: hconv2 split-K target (two resident workgroups per CU on the 64x64 level), the two wconv launch classes in the bench line
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out/r4b12
mkdir -p $O
for v in "0,4" "512,4" "512,2" "384,4" "0,4" "512,4"; do
  CGD_HCONV_SPLIT=$v python bench.py --steps 150 --warmup 5 --no-cpu-baseline > $O/bench.json 2>/dev/null
  python -c "
import json;r=json.load(open('$O/bench.json'));o=r['roofline']['other_conv_kernel'];print('HCONV_SPLIT $v', r['value'],r['ms_per_step'],r['config']['launches_per_step'], 'hconv2', o['avg_launch_us'], o['frac'], r['roofline'].get('launch_classes'))"
done
