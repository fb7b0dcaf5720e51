#!/usr/bin/env bash
# r6 GPU call 10: (i) same-box A/B of the library before (A = libcgd_prev.so) and after (B) kconv_kernel's epilogue was generalised for the 16-row tile
# (the default 8 x 8 instantiations must not have moved); (ii) final validation: whole GPU suite + smoke; (iii) the final bench lines
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c10
mkdir -p $O
bash benchmarks/ab.sh 3 150 2>&1 | tee $O/ab_kconv_epilogue.txt
bash benchmarks/r6_final_validate.sh 2>&1 | tee $O/validate.txt
python bench.py > gpurun_out/r6final_bench.json 2> gpurun_out/r6final_bench.err
python bench.py --precision f32 --steps 60 --no-cpu-baseline > gpurun_out/r6final_bench_f32.json 2>/dev/null
python - <<'PY'
import json
for f in ("gpurun_out/r6final_bench.json", "gpurun_out/r6final_bench_f32.json"):
    r = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, r["value"], r["ms_per_step"], r.get("box_calibration"), r["roofline"]["frac"], r["roofline"].get("achieved_algorithmic"), (r.get("precision_modes") or {}).get("f32", {}).get("steps_per_sec"))
PY
