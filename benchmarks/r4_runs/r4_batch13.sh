#!/usr/bin/env bash
# r4 GPU call 13: hconv2 on the 64x64 level with a split-K target of two resident workgroups per CU (CGD_HCONV_SMALL = max pixels, slots, min chunks)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out/r4b13
mkdir -p $O
for v in "0,512,2" "4096,512,4" "4096,512,2" "4096,384,4" "0,512,2" "4096,512,4"; do
  CGD_HCONV_SMALL=$v python bench.py --steps 150 --warmup 5 --no-cpu-baseline > $O/bench.json 2>/dev/null
  python -c "
import json;r=json.load(open('$O/bench.json'));o=r['roofline']['other_conv_kernel'];print('HCONV_SMALL $v', r['value'],r['ms_per_step'],r['config']['launches_per_step'], 'hconv2', o['avg_launch_us'], o['frac'], o['launches_per_step'], r['roofline']['launches_per_step'])"
done
