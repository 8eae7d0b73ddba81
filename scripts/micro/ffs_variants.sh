#!/bin/bash
# dev only: build pp_ffn_split.hip variants into scripts/micro/build/libffs_<tag>.so:   ffs_variants.sh tag1 "-DFFS_DBG=4" tag2 "-DX=1 -DY=2" ...
# FFS_DBG bits: 2 no GELU, 4 no MFMA, 8 no DMA, 16 no fragment reads (timing only, wrong results)
set -e
here="$(dirname "$(readlink -f "$0")")"
mkdir -p "$here/build"
cd "$here/../../probpose_code_amd/csrc"
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include $flags -shared pp_ffn_split.hip pp_api.hip \
      -o "$here/build/libffs_$tag.so" -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 "ffn_split_kernel" | grep -E "VGPRs:|VGPRs Spill|ScratchSize" | tr '\n' ' '; echo " <- $tag ($flags)" ) &
done
wait
