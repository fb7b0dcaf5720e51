#!/usr/bin/env bash
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out
SKIP_MFMA=1 PASS_TIMEOUT=200 bash benchmarks/run_profile.sh r3a 4 > $O/r3b2_profile.log 2>&1
python benchmarks/trace_step.py "$(find $O/prof_r3a/trace -name "*kernel_trace.csv" | head -1)" 60 > $O/r3b2_trace_step.txt 2>&1 || true
bash benchmarks/host_contention.sh 48 > $O/r3_host_contention.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_step.py -q -k "early_schedule or config4_full" 2>&1 | tail -12 | cut -c1-1200 > $O/r3b2_tests.txt
python benchmarks/early_schedule_report.py cfg256 > $O/r3_early_cfg256_forced.txt 2>&1
python benchmarks/early_schedule_report.py mini > $O/r3_early_mini_forced.txt 2>&1
tail -5 $O/r3b2_tests.txt; cat $O/r3_host_contention.txt; head -60 $O/r3b2_trace_step.txt
