#!/usr/bin/env bash
# last call of the round: the large parity tests that run through hgemm2's new default epilogue, then the bench line of the final build
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
timeout 170 python -m pytest tests/test_gpu_step.py tests/test_gpu_parity.py -q -m gpu -x -k "headline_shape_single or other_towers or unet_256" 2>&1 | tail -4
timeout 80 python bench.py > gpurun_out/r3final2_bench.json 2> gpurun_out/r3final2_bench.err
tail -1 gpurun_out/r3final2_bench.json | cut -c1-400
