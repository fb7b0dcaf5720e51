#!/usr/bin/env bash
# Builds libcgd_mi355x.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libcgd_mi355x.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result"
mkdir -p build
objs=()
pids=()
for src in gemm hconv kconv wconv hgemm conv_thin norm elem attn attn_flash guidance unet vit resnet lpips capi; do
  obj=build/$src.o
  objs+=("$obj")
  if [[ ! -f $obj || $src.hip -nt $obj || common.h -nt $obj || kernels.h -nt $obj || net.h -nt $obj || guidance.h -nt $obj \
        || ../../include/cgd_mi355x.h -nt $obj ]]; then
    # MFMA results in architectural VGPRs for the kernels whose wavefronts have the registers (round 6): attn_flash — the softmax rescales / splits
    # the accumulators every key block, in AGPRs each touch is a v_accvgpr move (15 % of the forward kernel's vector-ALU instructions); hgemm / hconv —
    # 192-384 v_accvgpr moves per wavefront in zero-init and epilogue.  Same-box: neutral for attn_flash alone, -0.04 ms per step with hgemm + hconv
    # (profiles/r6_ab_mfma_vgpr_form.txt).  kconv already comes out that way; wconv's 512-register tiles cannot.
    extra=""
    [[ $src == attn_flash || $src == hgemm || $src == hconv ]] && extra="-mllvm -amdgpu-mfma-vgpr-form"
    $HIPCC $FLAGS $extra -c $src.hip -o $obj &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do
  [[ -n "$p" ]] && wait "$p"
done
$HIPCC --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o $OUT
echo "built $(realpath $OUT)"
