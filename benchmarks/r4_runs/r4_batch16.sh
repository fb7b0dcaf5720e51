#!/usr/bin/env bash
# r4 GPU call 16: attention kernels with the loop-invariant nt64 operand (Q / dO) split once into registers: parity, micro-benchmark, bench
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=gpurun_out/r4b16
mkdir -p $O
python -m pytest tests -m gpu -x -q -k "attention or test_unet_small" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
python benchmarks/probe_attn.py 50 > $O/probe_attn.txt 2>&1; cat $O/probe_attn.txt | grep attn
for i in 1 2; do
python bench.py --steps 150 --warmup 5 --no-cpu-baseline > $O/bench.json 2>/dev/null
python -c "
import json;r=json.load(open('$O/bench.json'));print(r['value'],r['ms_per_step'])"
done
