#!/usr/bin/env bash
# r5 GPU call 3: kgemm_kernel parity (GEMM cases, UNets, ViT towers with few rows) + same-box step A/B through CGD_KGEMM; the default bench.py line with
# the new precision_modes / clock / power fields
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out/r5c3
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gemm or unet_small or unet_64 or clip_vit or attention" > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -8 $O/pytest.log
for i in 1 2; do
  for f in 0 1; do
    echo "kgemm=$f: $(CGD_KGEMM=$f timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', r['config']['launches_per_step'], 'launches', r['config']['splitk_reduce_per_step'], 'reduces')")"
  done
done | tee $O/ab_kgemm.txt
echo "kgemm=1 defer=2: $(CGD_DEFER=2 timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step', r['config']['launches_per_step'], 'launches', r['config']['splitk_reduce_per_step'], 'reduces')")" | tee -a $O/ab_kgemm.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc $?"; python -c "
import json
r = json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r.get('precision_modes'))
ro = r['roofline']
print({k: ro.get(k) for k in ('frac', 'avg_launch_us', 'clock_ghz', 'power_w', 'power_clock_samples')})
print('gemm class', ro['other_mfma_kernel']); print('kconv', ro['small_map_conv_kernel']['avg_launch_us'], ro['small_map_conv_kernel']['frac_of_hbm_peak']); print('hbm', r['hbm'])
"
tail -3 $O/bench_default.err
