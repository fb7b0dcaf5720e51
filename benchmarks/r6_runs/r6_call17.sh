#!/usr/bin/env bash
# r6 GPU call 17: the driver's exact bench command on the final build
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/driver_cmd.json 2> gpurun_out/driver_cmd.err
echo "rc $?"
python - <<'PY'
import json
r = json.loads(open("gpurun_out/driver_cmd.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], r["config"]["launches_per_step"], r["box_calibration"], r["precision_modes"]["f32"]["steps_per_sec"], r["cpu_baseline"]["value"], r["roofline"]["frac"], r["roofline"]["traffic"])
PY
