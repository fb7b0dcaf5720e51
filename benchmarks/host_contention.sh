#!/usr/bin/env bash
# Multi-GPU host readiness on a 1-GPU box: what one driver process costs the host per step, and what 2 / 4 / 8
# driver processes do to each other when they run side by side (all ranks on cuda:0 over gloo: the GPU is shared, so per-rank
# ms/step grows ~N-fold BY CONSTRUCTION; what is read off is the CPU time per step per rank and whether it inflates under
# N-way process contention).  Usage: bash benchmarks/host_contention.sh [steps] > gpurun_out/host_contention.txt
set -uo pipefail
STEPS=${1:-60}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
pick='import sys, json
r = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); c = r["config"]
print(json.dumps({"n_ranks": r["n_gpus"], "value_steps_per_s": r["value"], "ms_per_step_max_rank": r["ms_per_step"],
                  "ms_per_step_per_rank": c["ms_per_step_per_rank"], "host_cpu_ms_per_step_per_rank": c["host_cpu_ms_per_step_per_rank"],
                  "host_enqueue_ms_per_step_per_rank": c["host_enqueue_ms_per_step_per_rank"],
                  "host_isolated_enqueue_ms_per_rank": c["host_isolated_enqueue_ms_per_rank"],
                  "host_isolated_enqueue_cpu_ms_per_rank": c["host_isolated_enqueue_cpu_ms_per_rank"], "host_cores": c["host_cores"]}))'
nproc
timeout 300 python bench.py --gpus 1 --steps $((STEPS * 2)) --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "$pick"
for n in 2 4 8; do
  CGD_BENCH_DEVICE=0 CGD_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus $n --steps $((STEPS / (n / 2))) --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | python -c "$pick"
done
