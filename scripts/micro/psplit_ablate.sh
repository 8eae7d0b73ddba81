#!/bin/bash
# dev only: build the library with pp_panel_split.hip ablation switches (-DPSPLIT_DBG=mask) into scripts/micro/build/libpsplit_dbg<mask>.so
set -e
cd "$(dirname "$(readlink -f "$0")")/../../probpose_code_amd/csrc"
for d in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DPSPLIT_DBG=$d -shared pp_panel_split.hip pp_panel_gemm.hip pp_conv_halo.hip pp_gemm.hip pp_head.hip pp_api.hip \
      -o ../../scripts/micro/build/libpsplit_dbg$d.so
  echo "built dbg $d"
done
