#!/usr/bin/env bash
# r6 GPU call 5: (i) A/B of the fp32 Winograd kernel's hand schedule + 4-step weight ring (B) against the first version (A = libcgd_prev.so);
# (ii) the whole GPU suite on the build with the ADVICE fixes; (iii) the default bench line with box_calibration
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c5
mkdir -p $O
bash benchmarks/ab.sh 2 40 --precision f32 2>&1 | tee $O/ab_f32_sched.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
echo "pytest rc $?"; tail -6 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc $?"; python - <<'PY'
import json
r = json.loads(open("gpurun_out/r6c5/bench.json").read().strip().splitlines()[-1])
print(r["value"], "steps/s", r["ms_per_step"], "ms/step; f32:", r.get("precision_modes"), "box:", r.get("box_calibration"))
print("roofline frac", r["roofline"]["frac"], "avg us", r["roofline"]["avg_launch_us"], "clock", r["roofline"].get("clock_ghz"), "power", r["roofline"].get("power_w"))
print("gemm class", r["roofline"]["other_mfma_kernel"], "hbm", r["hbm"]["frac"], r["hbm"]["ms_per_step"])
PY
