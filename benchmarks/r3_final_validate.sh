#!/usr/bin/env bash
# Final validation of the round on one box: smoke(), the whole GPU suite, the bench line.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
O=gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > $O/r3_smoke.log 2>&1; echo "smoke rc=$?" >> $O/r3_smoke.log
timeout 1700 python -m pytest tests -q -m gpu --durations=12 > $O/r3_pytest_gpu.log 2>&1
python bench.py > $O/r3final2_bench.json 2> $O/r3final2_bench.err
tail -4 $O/r3_smoke.log; tail -18 $O/r3_pytest_gpu.log | cut -c1-300; tail -1 $O/r3final2_bench.json | cut -c1-400
