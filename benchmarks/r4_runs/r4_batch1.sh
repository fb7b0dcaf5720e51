#!/usr/bin/env bash
# r4 GPU call 1: hgemm2 pipeline sweep (ring depth x staging sets) + ablations, then a same-box baseline bench (bf16x3 and exact fp32).
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
mkdir -p gpurun_out/r4b1
bash benchmarks/r4_runs/r4_hgemm_pipeline_sweep.sh > gpurun_out/r4b1/hgemm_sweep.txt 2>&1
python bench.py --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/r4b1/bench_x3.json 2> gpurun_out/r4b1/bench_x3.err
python bench.py --steps 60 --warmup 5 --no-cpu-baseline --precision f32 > gpurun_out/r4b1/bench_f32.json 2> gpurun_out/r4b1/bench_f32.err
tail -c 600 gpurun_out/r4b1/bench_x3.json; tail -c 400 gpurun_out/r4b1/bench_f32.json
