#!/usr/bin/env bash
# r6 GPU call 13: the round's last build — whole GPU suite + smoke, the measurement pass (bench lines, rocprofv3 passes, other configs), T5 drift report
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
bash benchmarks/r6_final_validate.sh 2>&1 | tail -8
bash benchmarks/r6_final_profile.sh > gpurun_out/r6final_profile_stdout.txt 2>&1
timeout 900 python benchmarks/drift_report.py > gpurun_out/r6final_t5_drift_config1.txt 2>&1
python - <<'PY'
import json
for f in ("gpurun_out/r6final_bench.json", "gpurun_out/r6final_bench_f32.json"):
    r = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, r["value"], r["ms_per_step"], r["config"]["launches_per_step"], r.get("box_calibration"), r["roofline"]["frac"], (r.get("precision_modes") or {}).get("f32", {}).get("steps_per_sec"))
PY
tail -5 gpurun_out/r6final_t5_drift_config1.txt
