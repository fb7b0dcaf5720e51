#!/usr/bin/env bash
# r6 GPU call 12: runtime environment knobs that act on dispatch latency (874 dependent launches per step): kernel arguments in device memory,
# hardware queue count, scratch reclaim — same-box A/B
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c12
mkdir -p $O
run() { echo "$1: $(env $2 timeout 300 python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', 'host isolated enqueue', r['config']['host_isolated_enqueue_ms_per_rank'])")"; }
for i in 1 2; do
  run "default                    " "A=1"
  run "HIP_FORCE_DEV_KERNARG=1    " "HIP_FORCE_DEV_KERNARG=1"
  run "HIP_FORCE_DEV_KERNARG=0    " "HIP_FORCE_DEV_KERNARG=0"
  run "GPU_MAX_HW_QUEUES=1        " "GPU_MAX_HW_QUEUES=1"
  run "HSA_NO_SCRATCH_RECLAIM=1   " "HSA_NO_SCRATCH_RECLAIM=1"
  run "AMD_SERIALIZE_KERNEL=0 DEBUG_HIP_BLOCK_SYNC? (noop control)" "HIP_DB=0"
done | tee $O/ab_env.txt
env | grep -E "^HIP_|^HSA_|^ROC|^AMD_|^GPU_" | tee $O/env.txt
