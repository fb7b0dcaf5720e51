#!/usr/bin/env bash
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out
bash benchmarks/host_contention.sh 48 > $O/r3_host_contention.txt 2>&1
python benchmarks/probe_attn.py 30 > $O/r3_probe_attn.txt 2>&1
bash benchmarks/pmc_probe.sh attn attn probe_attn.py 6 > $O/r3_pmc_attn.txt 2>&1
cat $O/r3_host_contention.txt; cat $O/r3_probe_attn.txt; head -150 $O/r3_pmc_attn.txt
