#!/bin/bash
# builds the lab variants (gfx950 cross-compile, seconds each): tools/r03/lab/bin/<name>
cd "$(dirname "$0")" && mkdir -p bin
build() { name=$1; shift; /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I ../../../include -DLAB_NAME="\"$name\"" "$@" w128b_lab.hip -o bin/$name 2>&1 | grep -i "error" -A3; }
build base
build timing -DMILE_LAB_TIMING
build no_barrier -DMILE_LAB_NO_BARRIER
build no_mfma -DMILE_LAB_NO_MFMA
build no_epilogue -DMILE_LAB_NO_EPI
build no_mfma_no_barrier -DMILE_LAB_NO_MFMA -DMILE_LAB_NO_BARRIER
build no_tr -DMILE_LAB_NO_TR
build no_row -DMILE_LAB_NO_ROW
build no_store -DMILE_LAB_NO_STORE
build no_lds -DMILE_LAB_NO_TR -DMILE_LAB_NO_ROW -DMILE_LAB_NO_STORE
build no_lds_no_barrier -DMILE_LAB_NO_TR -DMILE_LAB_NO_ROW -DMILE_LAB_NO_STORE -DMILE_LAB_NO_BARRIER
build no_lds_no_mfma_no_barrier -DMILE_LAB_NO_TR -DMILE_LAB_NO_ROW -DMILE_LAB_NO_STORE -DMILE_LAB_NO_BARRIER -DMILE_LAB_NO_MFMA
ls bin
