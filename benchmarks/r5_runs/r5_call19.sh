#!/usr/bin/env bash
# r5 GPU call 19: CGD_ATTN_FLASH=3 as the default: attention selections, ViT towers, UNets, quick whole-step tests, smoke
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r5c19
mkdir -p $O
timeout 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_step.py -m gpu -x -q -k "attention or clip_vit or unet_small or unet_64 or unet_256 or headline_shape or p_sample_trajectory or ddim_trajectory or batch2_prompts2 or cosine_nonsquare or user_cond_fn or reference_recipe" > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -4 $O/pytest.log
