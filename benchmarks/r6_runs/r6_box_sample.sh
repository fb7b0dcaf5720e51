#!/usr/bin/env bash
# one line per call: the final build on whatever box the call lands on (steps/s, ms per step, box calibration)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
python bench.py --steps 150 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1]); b = r['box_calibration']
print('BOX', r['value'], 'steps/s', r['ms_per_step'], 'ms/step', b['mfma_tflops'], 'TFLOP/s', b['wconv_ref_us_random'], 'us')"
