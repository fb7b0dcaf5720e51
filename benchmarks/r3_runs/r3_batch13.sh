#!/usr/bin/env bash
# Timeline of wconv_kernel's workgroups (benchmarks/ubench/wconv_stamps.hip) on the UNet's large-map layer shapes
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT/benchmarks/ubench"
[ -x wconv_stamps ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include wconv_stamps.hip -o wconv_stamps
O=$ROOT/gpurun_out
for args in "256 256 256 0 4" "256 512 256 0 4" "256 256 512 0 4" "256 256 256 1 4" "128 256 256 0 2" "128 512 256 0 2" "128 256 256 0 4"; do
  timeout 60 ./wconv_stamps $args 10 "$O/wconv_stamps_$(echo $args | tr ' ' '_').csv"
  echo
done
