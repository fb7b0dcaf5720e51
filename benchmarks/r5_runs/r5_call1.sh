#!/usr/bin/env bash
# r5 GPU call 1: flash attention parity (op tests + smoke at the headline shape), attention micro-benchmark old vs new, same-box step A/B through
# the CGD_ATTN_FLASH knob, wconv power probe
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out/r5c1
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "attention" > $O/pytest_attn.log 2>&1
echo "pytest attention rc $?"; tail -5 $O/pytest_attn.log
timeout 300 python benchmarks/probe_attn.py 50 > $O/attn_flash.txt 2>&1; cat $O/attn_flash.txt
CGD_ATTN_FLASH=0 timeout 300 python benchmarks/probe_attn.py 50 > $O/attn_old.txt 2>&1; cat $O/attn_old.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc $?"; tail -30 $O/smoke.log
for i in 1 2; do
  for f in 0 1; do
    echo "flash=$f: $(CGD_ATTN_FLASH=$f timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', r['config']['launches_per_step'], 'launches', r['config']['splitk_reduce_per_step'], 'reduces')")"
  done
done | tee $O/ab_flash.txt
timeout 300 bash benchmarks/r5_power_probe.sh
