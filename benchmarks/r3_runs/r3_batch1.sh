#!/usr/bin/env bash
# round-3 GPU batch: kconv parity + micro-benchmark + step A/B, early-schedule tests, config-4 full-shape failure detail
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "weight_streaming or previous_kernels or unet_small or unet_64 or test_conv" 2>&1 | tail -15 > $O/r3b1_kconv_parity.txt
timeout 200 env CGD_BENCH_TILES=0,516 python benchmarks/bench_ops.py r3_kconv > $O/r3b1_ops_kconv.txt 2>&1
timeout 200 env CGD_KCONV=0 CGD_BENCH_TILES=0 python benchmarks/bench_ops.py r3_kconv_off > $O/r3b1_ops_old.txt 2>&1
run() {
  env "$@" timeout 100 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null |
    python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', r['value'], r['ms_per_step'])"
}
{
  for _ in 1 2; do
    run CGD_KCONV=0
    run CGD_NOP=1
    run CGD_KCONV=1,256
    run CGD_KCONV=1,1024,2
    run CGD_KCONV=1,4096
  done
} > $O/r3b1_ab.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_step.py -q -k "early_schedule or headline_shape_single" 2>&1 | tail -25 > $O/r3b1_early.txt
timeout 400 python -m pytest tests/test_gpu_step.py -q -k "config4_full" 2>&1 | grep -E "AssertionError|passed|failed" | cut -c1-3000 > $O/r3b1_cfg4.txt
tail -12 $O/r3b1_kconv_parity.txt; cat $O/r3b1_ab.txt; tail -8 $O/r3b1_early.txt; tail -3 $O/r3b1_cfg4.txt | cut -c1-1500
