#!/bin/bash
# dev only: build pp_mlp.hip with compile-time ablation switches (-DMLP_DBG=mask) into separate libraries.
# mask bits: 1 DMA out of bounds (no traffic), 2 no GELU, 4 no MFMA, 8 no DMA issue, 16 no ds_read, 32 no barrier, 64 no b1 load
set -e
cd "$(dirname "$(readlink -f "$0")")/../../probpose_code_amd/csrc"
for d in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -fno-slp-vectorize -DMLP_DBG=$d -shared pp_mlp.hip pp_api.hip \
      -o ../../scripts/micro/build/libmlp_dbg$d.so -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A3 "mlp_res_ln" | grep -E "VGPRs:" | tr '\n' ' '
  echo " <- dbg $d"
done
