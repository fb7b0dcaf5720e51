#!/usr/bin/env bash
# r6 GPU call 22: wconv_kernel variants built from one source tree (clip-guided-diffusion_amd/variants/libcgd_<v>.so, linked by hand for this call):
#   p0r8 = last chunk re-stages itself, 8-step weight ring (rounds 2-5)      p1r8 = last chunk peeled in the NC = 1 instantiations only
#   p2r8 = peeled everywhere (the NC = 2 instantiations spill 9-20 registers -> scratch)   p2r6 / p0r6 = 6-step ring for NC = 2 (no spills) with / without the peel
# Same box, alternating.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r6c22
mkdir -p $O
run() { echo "$1: $(CGD_LIB_PATH=$ROOT/clip-guided-diffusion_amd/variants/libcgd_$1.so timeout 300 python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], 'steps/s', r['ms_per_step'], 'ms/step')")"; }
for i in 1 2 3; do
  for v in p0r8 p1r8 p2r6 p0r6 p2r8; do run $v; done
done | tee $O/ab_wconv_peel_variants.txt
