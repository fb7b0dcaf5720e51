#!/usr/bin/env bash
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_step.py -q -x -k "cutouts or thin_input or headline_shape_single or p_sample_trajectory or cosine_nonsquare or config5_nonsquare or reference_recipe or use_augs_guided or dual_clip" 2>&1 | tail -4 | cut -c1-600
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/prof_r3c/trace -o t -- python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile > $ROOT/$O/prof_r3c.log 2>&1
cd "$ROOT"
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_r3c/trace/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if any(k in n for k in ('cutouts', 'gemv', 'thin_in', 'igemm_kernel<1, 256', 'embedding_add')) or ('hgemm2' in n):
        print(n[:70], r['Calls'], float(r['AverageNs'])/1e3)
PY
python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | cut -c1-200
