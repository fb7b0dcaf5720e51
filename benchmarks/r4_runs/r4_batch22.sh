#!/usr/bin/env bash
# r4 GPU call 22: kconv_kernel with warm vs cold packed weights, default split and finer split (more workgroups = more loads in flight)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out/r4b22
mkdir -p $O
(python benchmarks/probe_cold_weights.py 15; CGD_KCONV=1,1024,2 python benchmarks/probe_cold_weights.py 15; CGD_KCONV=1,1024,1 python benchmarks/probe_cold_weights.py 15) > $O/cold_weights.txt 2>&1
grep -v amdgpu.ids $O/cold_weights.txt
