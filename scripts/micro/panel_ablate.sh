#!/bin/bash
# dev only: build the library with pp_panel_gemm.hip ablation switches (-DPANEL_DBG=mask) into scripts/micro/build/libpanel_dbg<mask>.so
set -e
cd "$(dirname "$(readlink -f "$0")")/../../probpose_code_amd/csrc"
for d in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DPANEL_DBG=$d -shared pp_panel_gemm.hip pp_conv_halo.hip pp_panel_split.hip pp_gemm.hip pp_head.hip pp_api.hip \
      -o ../../scripts/micro/build/libpanel_dbg$d.so -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A3 "panel_gemm_kernel" | grep -E "VGPRs:" | tr '\n' ' '
  echo " <- dbg $d"
done
