#!/usr/bin/env bash
# r4 GPU call 2: the new parity tests + the bench on the fan-in synthetic weights (finite trajectory check) with the launch counters
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
mkdir -p gpurun_out/r4b2
python -m pytest tests -m gpu -x -q -k "config1_full or full_scale_head or rccl or cutouts" -s > gpurun_out/r4b2/pytest_new.log 2>&1
tail -15 gpurun_out/r4b2/pytest_new.log
python bench.py --steps 250 --warmup 5 --no-cpu-baseline > gpurun_out/r4b2/bench_x3.json 2> gpurun_out/r4b2/bench_x3.err
tail -c 300 gpurun_out/r4b2/bench_x3.err
python -c "
import json;r=json.load(open('gpurun_out/r4b2/bench_x3.json'));print(r['value'],r['ms_per_step'],r['config']['launches_per_step'],r['config']['splitk_reduce_per_step'],r['config']['last_sample_peak'])"
