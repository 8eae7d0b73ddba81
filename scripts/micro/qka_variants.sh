#!/bin/bash
# dev only: build pp_qkv_attn_split.hip variants into scripts/micro/build/libqka_<tag>.so:   qka_variants.sh tag1 "-DQKA_DBG=1" tag2 "" ...
# QKA_DBG bits (timing only, wrong results): 1 no attention phase, 2 no GEMM-phase MFMAs, 4 no DMA traffic
set -e
here="$(dirname "$(readlink -f "$0")")"
mkdir -p "$here/build"
cd "$here/../../probpose_code_amd/csrc"
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include $flags -shared pp_qkv_attn_split.hip pp_api.hip \
      -o "$here/build/libqka_$tag.so" -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 "qkv_attention_split_kernel" | grep -E "VGPRs:|ScratchSize" | tr '\n' ' '; echo " <- $tag ($flags)" ) &
done
wait
