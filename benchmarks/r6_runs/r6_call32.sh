#!/usr/bin/env bash
# r6 GPU call 32: the two resamplings of a resampling ResBlock (forward: h and skip branch; backward: the two gradients that meet in its GroupNorm
# backward) in ONE launch (B: 855 launches per step) against two (A = build of commit f875810: 870).  UNet / step parity, then a same-box A/B.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c32
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_step.py -x -q -m gpu -k "unet or headline or step or mini" 2>&1 | tail -3 | tee $O/pytest.log
timeout 1200 bash benchmarks/ab.sh 4 150 2>&1 | tee $O/ab_resample_pair.txt
python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('launches per step', r['config']['launches_per_step'])" | tee -a $O/ab_resample_pair.txt
