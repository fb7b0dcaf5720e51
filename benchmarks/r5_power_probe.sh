#!/usr/bin/env bash
# Round 5: is wconv_kernel at the chip's power budget?  Board power (W) and shader clock sampled with rocm-smi while
# >= 3 s loops of (i) the full kernel, (ii) the MFMA-only ablation (CGD_WCONV_EXP=15: no weight loads, no staging, no barrier, no A reads),
# (iii) the no-weight-load ablation (=1), (iv) the full kernel on ZERO inputs (the DVFS check of MI355X_MICROARCH.md) run back to back
# on the 256x256, 256 -> 256 layer (16-row tile `4` and the shipped 8-row x 256-channel tile `22`).
# If (ii) draws more than (i) at the same clock, the full kernel is NOT at the cap.
# Output: gpurun_out/r5_power/summary.txt (copied to profiles/r5_wconv_power.txt).
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT/benchmarks/ubench"
O=$ROOT/gpurun_out/r5_power
mkdir -p "$O"
SMI=/opt/rocm/bin/rocm-smi
REPS=${REPS:-20000}
probe() {  # name env-assignment binary args...
  local name=$1; shift
  local envs=$1; shift
  ( env $envs timeout 120 "$@" > "$O/$name.out" 2>&1 ) &
  local pid=$!
  sleep 1.2
  : > "$O/$name.smi"
  for _ in 1 2 3 4 5; do
    $SMI --showpower --showclocks 2>/dev/null | grep -E "sclk|Power" >> "$O/$name.smi"
    sleep 0.15
  done
  wait $pid
  local us
  us=$(grep -oE "[0-9.]+ us per launch" "$O/$name.out" | head -1)
  local pw clk
  pw=$(grep -E "Power" "$O/$name.smi" | grep -oE "[0-9]+\.[0-9]+" | sort -n | tr '\n' ' ')
  clk=$(grep -E "sclk" "$O/$name.smi" | grep -oE "\([0-9]+Mhz\)" | tr -d '()Mhz' | sort -n | tr '\n' ' ')
  echo "$name: $us | W: $pw| sclk MHz: $clk"
}
{
  echo "# benchmarks/r5_power_probe.sh: rocm-smi samples (5 per loop, 1.2 s after the loop starts), $REPS launches per loop"
  echo "idle: $($SMI --showpower --showclocks 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ')"
  for nb in 4 22; do
    probe "full_nb$nb" "A=1" ./wconv_exp0 256 256 256 0 $nb $REPS
    probe "mfma_only_nb$nb" "A=1" ./wconv_exp15 256 256 256 0 $nb $REPS
    probe "no_weight_loads_nb$nb" "A=1" ./wconv_exp1 256 256 256 0 $nb $REPS
    probe "full_zero_inputs_nb$nb" "CGD_UBENCH_ZERO=1" ./wconv_exp0 256 256 256 0 $nb $REPS
  done
  probe "full_gn_nb22" "A=1" ./wconv_exp0 256 256 256 1 22 $REPS
} 2>&1 | tee "$O/summary.txt"
