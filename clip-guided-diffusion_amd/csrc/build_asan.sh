#!/usr/bin/env bash
# Host-sanitizer build of the C-ABI library (VERDICT r2 "missing" item 5): the HOST side of every translation unit
# is compiled with AddressSanitizer + UBSan (the gfx950 device side is compiled as usual: the sanitizer does not apply to it) into
# build/asan/libcgd_mi355x_asan.so, plus the driver tests/asan_host_driver.cpp that walks the host-only entry points (manifests,
# dispatch planner, schedule tables, NULL-handle and invalid-argument paths of every handle family).  No GPU needed.
# Usage: bash build_asan.sh   ->  prints the path of the driver binary
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=build/asan
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -Wno-option-ignored -fsanitize=address,undefined -fno-sanitize=vptr -fno-omit-frame-pointer"
objs=()
pids=()
for src in gemm hconv kconv wconv hgemm conv_thin norm elem attn attn_flash guidance unet vit resnet lpips capi; do
  obj=$OUT/$src.o
  objs+=("$obj")
  if [[ ! -f $obj || $src.hip -nt $obj || common.h -nt $obj || kernels.h -nt $obj || net.h -nt $obj || guidance.h -nt $obj \
        || ../../include/cgd_mi355x.h -nt $obj || build_asan.sh -nt $obj ]]; then
    $HIPCC $FLAGS -c $src.hip -o $obj 2> $OUT/$src.log &
    pids+=($!)
    if (( ${#pids[@]} >= 6 )); then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
  fi
done
for p in "${pids[@]:-}"; do
  [[ -n "$p" ]] && wait "$p"
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-sanitize=vptr "${objs[@]}" -o $OUT/libcgd_mi355x_asan.so
/opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I../../include \
  ../../tests/asan_host_driver.cpp -o $OUT/asan_host_driver -L$OUT -lcgd_mi355x_asan -Wl,-rpath,"$(realpath $OUT)" -Wl,-rpath,/opt/rocm/lib
echo "$(realpath $OUT/asan_host_driver)"
